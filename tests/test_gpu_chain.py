"""Chains (qmm_chain_begin / _end, csrc/qmm_chain.hiph): a run of dependent one-token MUL_MAT groups as one persistent launch.

The bar is bit-identity with the same calls made one launch each (which the parity tests pin against the oracle and the golden
vectors), on REAL dependency chains (step s+1 reads what step s wrote, through the norm / residual / SwiGLU folds of a llama layer)
so that a stale read of a hand-off shows up as wrong numbers: every hand-off is checked word for word, on buffers that are reused
from run to run with new inputs (the consumer's caches hold the previous run's values) and with uneven rows per workgroup.
The first step of each chain is additionally checked against the CPU oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle.pyoracle import Q4_0, Q4_K, Q5_K, Q6_K, Q8_0  # noqa: E402


@pytest.fixture(scope="module")
def qmm():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ggml_hexagon_amd.capi import Qmm
    q = Qmm(0)
    yield q
    q.close()


def W(t, m, k, seed):
    import ggml_hexagon_amd.synth as synth
    return torch.from_numpy(synth.synth_weights(t, m, k, seed=seed, sigma=0.05)).cuda()


class Layer:
    """wo(+res) -> gate/up (+norm, SwiGLU) -> down(+res) -> q/k/v (+norm) with llama shapes scaled by `d`, types per recipe"""

    def __init__(self, d, ff, kv, types, seed):
        to, tg, td, tq, tk, tv = types
        self.d, self.ff, self.kv = d, ff, kv
        self.wo, self.wg, self.wu = (to, W(to, d, d, seed)), (tg, W(tg, ff, d, seed + 1)), (tg, W(tg, ff, d, seed + 2))
        self.wd = (td, W(td, d, ff, seed + 3))
        self.wq, self.wk, self.wv = (tq, W(tq, d, d, seed + 4)), (tk, W(tk, kv, d, seed + 5)), (tv, W(tv, kv, d, seed + 6))
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.n1 = torch.rand(d, device="cuda", generator=g) + 0.5
        self.n2 = torch.rand(d, device="cuda", generator=g) + 0.5
        self.bufs = {k: torch.empty((1, n), device="cuda") for k, n in
                     (("h", d), ("act", ff), ("h2", d), ("q", d), ("k", kv), ("v", kv))}

    def issue(self, q, attn, resid):
        b = self.bufs
        q.mul_mat_group_ex([self.wo], self.d, attn, [b["h"]], residuals=[resid])
        q.mul_mat_group_ex([self.wg, self.wu], self.d, b["h"], [b["act"], b["act"]], norm_w=self.n2, eps=1e-5, swiglu=1)
        q.mul_mat_group_ex([self.wd], self.ff, b["act"], [b["h2"]], residuals=[b["h"]])
        q.mul_mat_group_ex([self.wq, self.wk, self.wv], self.d, b["h2"], [b["q"], b["k"], b["v"]], norm_w=self.n1, eps=1e-5)

    def snapshot(self):
        return {k: v.clone() for k, v in self.bufs.items()}


RECIPES = {
    "q4_k_m":  (Q4_K, Q4_K, Q6_K, Q4_K, Q4_K, Q6_K),
    "q4_0":    (Q4_0, Q4_0, Q4_0, Q4_0, Q4_0, Q4_0),
    "mixtral": (Q5_K, Q4_K, Q4_K, Q4_K, Q8_0, Q8_0),       # q/k/v of two activation formats: one call, two steps
    "q8_0":    (Q8_0, Q8_0, Q8_0, Q8_0, Q8_0, Q8_0),
    "q5/q6":   (Q5_K, Q5_K, Q6_K, Q6_K, Q5_K, Q6_K),
}


@pytest.mark.parametrize("recipe", list(RECIPES))
@pytest.mark.parametrize("shape", [(4096, 14336, 1024), (1024, 2816, 256), (256, 768, 48)], ids=["8b", "small", "ragged"])
def test_chain_is_bit_identical_to_launches(qmm, recipe, shape):
    d, ff, kv = shape
    L = Layer(d, ff, kv, RECIPES[recipe], seed=hash((recipe, d)) % 1000)
    g = torch.Generator(device="cuda").manual_seed(d)
    l0, s0 = qmm.chain_stats()
    for run in range(4):                                   # same buffers, new inputs: the previous run's values sit in the caches
        attn = torch.rand((1, d), device="cuda", generator=g) * 2 - 1
        resid = torch.rand((1, d), device="cuda", generator=g) * 2 - 1
        for v in L.bufs.values():
            v.fill_(float("nan"))
        L.issue(qmm, attn, resid)
        qmm.synchronize()
        want = L.snapshot()
        for v in L.bufs.values():
            v.fill_(float("nan"))
        qmm.chain_begin()
        L.issue(qmm, attn, resid)
        qmm.chain_end()
        qmm.synchronize()
        for k, v in L.bufs.items():
            assert torch.equal(v.view(torch.int32), want[k].view(torch.int32)), (recipe, shape, run, k)
    l1, s1 = qmm.chain_stats()
    n_steps = 5 if recipe == "mixtral" else 4
    assert (l1 - l0, s1 - s0) == (4, 4 * n_steps)


def test_first_step_matches_the_oracle(qmm, oracle):
    """a chain's arithmetic against the CPU restatement (the per-launch kernels are pinned the same way in test_gpu_parity.py)"""
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(5)
    for t in (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K):
        k, m = 1024, 320
        w = synth.synth_weights(t, m, k, seed=t, sigma=0.1)
        w2 = synth.synth_weights(t, k, m - 64, seed=t + 1, sigma=0.1) if (m - 64) % 256 == 0 else None
        x = rng.uniform(-1, 1, (1, k)).astype(np.float32)
        wd, xd = torch.from_numpy(w).cuda(), torch.from_numpy(x).cuda()
        out, out2 = torch.empty((1, m), device="cuda"), torch.empty((1, m), device="cuda")
        qmm.chain_begin()
        qmm.mul_mat_group([(t, wd)], k, xd, [out])
        qmm.mul_mat_group([(t, wd)], k, xd, [out2])          # a second step so that the persistent kernel runs
        qmm.chain_end()
        qmm.synchronize()
        want = oracle.mul_mat(t, w, k, x)
        err = np.max(np.abs(out.cpu().numpy() - want)) / np.sqrt(np.mean(want.astype(np.float64) ** 2))
        assert err <= 2e-5, (t, err)
        assert torch.equal(out, out2)
        del w2


def test_long_chain_is_cut_into_launches_and_keeps_order(qmm):
    """30 dependent square steps (x -> W x -> W (W x) ...): more than one persistent launch, each step reading its predecessor"""
    d = 1024
    ws = [(Q4_K, W(Q4_K, d, d, 100 + i)) for i in range(3)]
    bufs = [torch.empty((1, d), device="cuda") for _ in range(31)]
    g = torch.Generator(device="cuda").manual_seed(3)

    def issue():
        for i in range(30):
            qmm.mul_mat_group([ws[i % 3]], d, bufs[i], [bufs[i + 1]])

    for run in range(3):
        bufs[0].copy_(torch.rand((1, d), device="cuda", generator=g) * 2 - 1)
        issue()
        qmm.synchronize()
        want = [b.clone() for b in bufs]
        for b in bufs[1:]:
            b.fill_(float("nan"))
        l0, s0 = qmm.chain_stats()
        qmm.chain_begin()
        issue()
        qmm.chain_end()
        qmm.synchronize()
        l1, s1 = qmm.chain_stats()
        assert (l1 - l0, s1 - s0) == (3, 30)
        for i, (b, w) in enumerate(zip(bufs, want)):
            assert torch.equal(b.view(torch.int32), w.view(torch.int32)), (run, i)


def test_other_calls_flush_the_recording(qmm):
    """anything that is not a one-token group goes out BEHIND what was recorded: stream order is call order"""
    d = 512
    w = (Q4_K, W(Q4_K, d, d, 7))
    x = torch.rand((1, d), device="cuda") * 2 - 1
    a, b2 = torch.empty((1, d), device="cuda"), torch.empty((1, d), device="cuda")
    x4 = torch.empty((4, d), device="cuda")
    o4 = torch.empty((4, d), device="cuda")
    qmm.mul_mat_group([w], d, x, [a])
    x4[:] = a
    qmm.mul_mat_group([w], d, x4, [o4])
    qmm.synchronize()
    want = o4.clone()
    a.fill_(0)
    o4.fill_(0)
    qmm.chain_begin()
    qmm.mul_mat_group([w], d, x, [a])                    # recorded
    qmm.mul_mat_group([w], d, x, [b2])                   # recorded
    qmm.chain_flush()
    x4[:] = a                                             # torch work on the same stream, behind the flush
    qmm.mul_mat_group([w], d, x4, [o4])                  # four tokens: not recordable, launched in order
    qmm.chain_end()
    qmm.synchronize()
    assert torch.equal(o4, want) and torch.equal(a, b2)


def test_graph_capture_of_a_chain(qmm):
    """kernel arguments are passed by value: a captured chain replays without any table upload"""
    d = 1024
    L = Layer(d, 2816, 256, RECIPES["q4_k_m"], seed=11)
    attn = torch.rand((1, d), device="cuda") * 2 - 1
    resid = torch.rand((1, d), device="cuda") * 2 - 1
    L.issue(qmm, attn, resid)
    qmm.synchronize()
    want = L.snapshot()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        qmm.chain_begin(); L.issue(qmm, attn, resid); qmm.chain_end()       # warm-up on the capture stream
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            qmm.chain_begin(); L.issue(qmm, attn, resid); qmm.chain_end()
    for _ in range(3):
        for v in L.bufs.values():
            v.fill_(float("nan"))
        gr.replay()
        torch.cuda.synchronize()
        for k, v in L.bufs.items():
            assert torch.equal(v.view(torch.int32), want[k].view(torch.int32)), k


@pytest.mark.parametrize("t", [Q4_K, Q8_0, Q6_K], ids=["q4_K", "q8_0", "q6_K"])
def test_fused_norm_does_not_depend_on_the_launch_geometry(qmm, t):
    """the RMS_NORM folded into a mat-vec launch sums its squares in a fixed order (1024 virtual threads, stage_rms_norm): the same rows
    through a 16-wave, an 8-wave and a 4-wave launch (4096, 2048 and 512 rows on 256 CUs) give the same bits.  Until round 2 the
    partition followed blockDim, and about one input in 150 quantized one activation differently between an 8-wave launch and the
    16-wave chain kernel."""
    d = 4096
    w = W(t, d, d, 3)
    nw = torch.rand(d, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) + 0.5
    g = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(40):
        x = torch.rand((1, d), device="cuda", generator=g) * 2 - 1
        outs = []
        for rows in (4096, 2048, 512):
            o = torch.empty((1, rows), device="cuda")
            qmm.mul_mat_group_ex([(t, w[:rows])], d, x, [o], norm_w=nw, eps=1e-5)
            outs.append(o)
        qmm.synchronize()
        assert torch.equal(outs[0][:, :2048].view(torch.int32), outs[1].view(torch.int32))
        assert torch.equal(outs[0][:, :512].view(torch.int32), outs[2].view(torch.int32))
