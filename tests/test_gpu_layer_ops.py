"""GPU parity of the entry points that fold a layer's glue into fewer launches (include/ggml_mi355x_ops.h and
qmm_mul_mat_group_ex), called through the C-ABI and checked against numpy restatements of the ggml CPU semantics
(ggml-cpu.c:6254-6300 rms_norm, :8261-8352 soft_max, :8708-8893 rope; build_attn_mha src/llama-graph.cpp:1166-1203) and, for
the quantized products, against the CPU oracle.  The per-op kernels behind qmm_op_compute are covered by the reference's own
test-backend-ops (tests/test_gpu_backend_ops.py); these tests cover what that harness cannot express: the multi-node launches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle.pyoracle import ACT_REF, Q4_0, Q4_K, Q6_K, TYPE_NAMES  # noqa: E402

F32, F16, I32 = 0, 1, 26


@pytest.fixture(scope="module")
def qmm():
    from ggml_hexagon_amd.capi import Qmm
    q = Qmm(0)
    yield q
    q.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def T(t, type_, ne, nb=None, op_params=()):
    from ggml_hexagon_amd.capi import QmmTensor
    return QmmTensor.make(type_, ne, nb=nb, data=t.data_ptr(), op_params=op_params)


def f2i(x):
    return int(np.float32(x).view(np.int32))


def rms_norm(x, w, eps):
    x64 = x.astype(np.float64)
    return (x * (1.0 / np.sqrt((x64 * x64).mean(axis=-1, keepdims=True) + eps)).astype(np.float32) * w).astype(np.float32)


def rel_rms(got, want):
    want = want.astype(np.float64)
    return float(np.max(np.abs(got - want)) / max(np.sqrt(np.mean(want ** 2)), 1e-30))


@pytest.mark.parametrize("n", [1, 3, 8])
def test_mat_vec_with_norm_and_residual(qmm, oracle, n):
    """qmm_mul_mat_group_ex: x -> rms_norm(x) * w formed in the kernel's staging phase, dst = W x + residual in the epilogue.
    Against the oracle fed with the numpy-normed row (mat-vec bar 2e-5), for a same-type group, a mixed K-quant group and Q4_0."""
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(100 + n)
    for k, spec in ((4096, ((Q4_K, 320), (Q4_K, 64))), (4096, ((Q4_K, 256), (Q6_K, 96), (Q4_K, 33))), (1024, ((Q4_0, 200),))):
        x = rng.normal(0, 1.5, (n, k)).astype(np.float32)
        w = rng.normal(1, 0.1, k).astype(np.float32)
        eps = 1e-5
        xn = rms_norm(x, w, eps)
        ws_np = [(t, synth.synth_weights(t, m, k, seed=m + t, sigma=0.25)) for t, m in spec]
        ws = [(t, dev(a)) for t, a in ws_np]
        res = [rng.normal(0, 1, (n, a.shape[0])).astype(np.float32) for _, a in ws_np]
        outs = [torch.zeros((n, a.shape[0]), device="cuda") for _, a in ws_np]
        if n * k * 4 + n * k * 11 // 8 + 4096 > 150 * 1024:      # the f32 rows do not fit LDS beside the quantized ones: refused, not emulated
            from ggml_hexagon_amd.capi import QmmError
            with pytest.raises(QmmError):
                qmm.mul_mat_group_ex(ws, k, dev(x), outs, norm_w=dev(w), eps=eps)
        else:
            qmm.mul_mat_group_ex(ws, k, dev(x), outs, norm_w=dev(w), eps=eps, residuals=[dev(r) for r in res])
            for (t, a), o, r in zip(ws_np, outs, res):
                want = oracle.mul_mat(t, a, k, xn, ACT_REF) + r
                assert rel_rms(o.cpu().numpy(), want) < 5e-5, (TYPE_NAMES[t], k, n)
        # residual only, in place (dst is the residual buffer): the wo / ffn_down form
        acc = [dev(r) for r in res]
        qmm.mul_mat_group_ex(ws[:1], k, dev(x), acc[:1], residuals=acc[:1])
        assert rel_rms(acc[0].cpu().numpy(), oracle.mul_mat(ws_np[0][0], ws_np[0][1], k, x, ACT_REF) + res[0]) < 2e-5


@pytest.mark.parametrize("n", [33, 200, 512])
def test_prefill_group_with_residual_add_and_norm(qmm, oracle, n):
    """qmm_mul_mat_group_ex at prompt batch sizes (round 3): the activation prep forms rms_norm(x + add) * w itself and stores x + add; against
    the launches it replaces (the add, the norm, then the plain group): the sum bit for bit, the products within the prefill bar of each
    other and of the oracle fed with the numpy-normed rows.  A same-type group, q / k / v with a Q6_K matrix (one prep for both operand
    orders' shared bytes), Q5_K + Q3_K, without the add, the sum stored over x (ggml-alloc's in-place add)"""
    import ggml_hexagon_amd.synth as synth
    from oracle.pyoracle import Q3_K, Q5_K
    rng = np.random.default_rng(300 + n)
    for k, spec, with_add, in_place in ((4096, ((Q4_K, 320), (Q4_K, 64)), True, False), (4096, ((Q4_K, 256), (Q4_K, 96), (Q6_K, 128)), True, True),
                                        (2048, ((Q5_K, 130), (Q3_K, 70)), True, False), (1024, ((Q4_K, 200),), False, False)):
        x0 = rng.normal(0, 1.5, (n, k)).astype(np.float32)
        b0 = rng.normal(0, 1.0, (n, k)).astype(np.float32)
        w = rng.normal(1, 0.1, k).astype(np.float32)
        eps = 1e-5
        ws_np = [(t, synth.synth_weights(t, m, k, seed=m + t, sigma=0.25)) for t, m in spec]
        ws = [(t, dev(a)) for t, a in ws_np]
        x, b, dw = dev(x0), dev(b0), dev(w)
        s_fused = x if in_place else torch.full((n, k), float("nan"), device="cuda")
        outs = [torch.zeros((n, a.shape[0]), device="cuda") for _, a in ws_np]
        labels = qmm.trace(lambda: qmm.mul_mat_group_ex(ws, k, x, outs, norm_w=dw, eps=eps, norm_add=b if with_add else None, norm_sum=s_fused if with_add else None))
        assert any("norm" in l for l in labels) and not any(l.startswith("prep_act_q8k_kernel<") and "norm" not in l for l in labels), labels
        # the launches it replaces
        x2 = dev(x0)
        xs = x2 + b if with_add else x2
        xn = torch.from_numpy(rms_norm(xs.cpu().numpy(), w, eps)).cuda()
        want = [torch.zeros((n, a.shape[0]), device="cuda") for _, a in ws_np]
        qmm.mul_mat_group(ws, k, xn, want)
        if with_add:
            assert torch.equal(s_fused.view(torch.int32), xs.view(torch.int32))
        for (t, a), o, wv in zip(ws_np, outs, want):
            ref = oracle.mul_mat(t, a, k, xn.cpu().numpy(), ACT_REF)
            rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
            assert float(np.linalg.norm(o.cpu().numpy() - ref) / np.linalg.norm(ref)) < 1e-3, (TYPE_NAMES[t], k, n)
            # (the two norms differ in the last bit here and there; such a row quantizes one activation differently: 1 / 127 of one of K terms)
            assert float((o - wv).abs().max()) / rms < 2.5e-3 * (4096 / k) ** 0.5, (TYPE_NAMES[t], k, n)


@pytest.mark.parametrize("n", [1, 4, 8])
def test_mat_vec_swiglu_pairs(qmm, oracle, n):
    """qmm_mul_mat_group_ex with swiglu: ffn_gate and ffn_up as row pairs, dst = silu(Wg x) * (Wu x), either order, with and
    without the norm in front"""
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(300 + n)
    for t, k, m in ((Q4_K, 1024, 704), (Q4_0, 512, 130), (Q6_K, 2048, 96)):
        x = rng.normal(0, 1.0, (n, k)).astype(np.float32)
        w = rng.normal(1, 0.1, k).astype(np.float32)
        wg, wu = synth.synth_weights(t, m, k, seed=11, sigma=0.25), synth.synth_weights(t, m, k, seed=12, sigma=0.25)
        for which, norm in ((1, False), (2, True)):
            xin = rms_norm(x, w, 1e-5) if norm else x
            g, u = oracle.mul_mat(t, wg, k, xin, ACT_REF).astype(np.float64), oracle.mul_mat(t, wu, k, xin, ACT_REF).astype(np.float64)
            want = g / (1.0 + np.exp(-g)) * u
            ws = [(t, dev(wg)), (t, dev(wu))] if which == 1 else [(t, dev(wu)), (t, dev(wg))]
            outs = [torch.full((n, m), 7.0, device="cuda"), torch.full((n, m), 7.0, device="cuda")]
            qmm.mul_mat_group_ex(ws, k, dev(x), outs, norm_w=dev(w) if norm else None, eps=1e-5, swiglu=which)
            assert rel_rms(outs[0].cpu().numpy(), want) < 1e-4, (TYPE_NAMES[t], n, which)
            assert torch.all(outs[1] == 7.0)                      # the second destination is not written


@pytest.mark.parametrize("n", [9, 40, 300])
def test_prefill_with_swiglu_input(qmm, oracle, n):
    """qmm_mul_mat_swiglu_in: dst = W (silu(gate) * up) with the product formed by the MFMA path's activation prep; against the
    oracle fed with the numpy product (prefill bar: rel-L2 <= 1e-3)"""
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(500 + n)
    for t, k, m in ((Q4_K, 2816, 1024), (Q6_K, 1024, 300), (Q4_0, 512, 128)):
        gate = rng.normal(0, 1.0, (n, k)).astype(np.float32)
        up = rng.normal(0, 1.0, (n, k)).astype(np.float32)
        x = (gate / (1.0 + np.exp(-gate)) * up).astype(np.float32)
        w = synth.synth_weights(t, m, k, seed=21, sigma=0.25)
        wd, dg, du = dev(w), dev(gate), dev(up)
        out = torch.zeros((n, m), device="cuda")
        qmm._chk(qmm.lib.qmm_mul_mat_swiglu_in(qmm.ctx, t, wd.data_ptr(), wd.stride(0), k, m, dg.data_ptr(), k, du.data_ptr(), k, n,
                                               out.data_ptr(), m, qmm._stream()))
        want = oracle.mul_mat(t, w, k, x, ACT_REF).astype(np.float64)
        got = out.cpu().numpy()
        assert np.sqrt(np.sum((got - want) ** 2) / np.sum(want ** 2)) < 1e-3, (TYPE_NAMES[t], n)


def test_add_rms_norm_two_results(qmm):
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(5)
    for rows, k in ((1, 4096), (7, 4096), (300, 1024), (2, 8192)):
        a, b = rng.normal(0, 1, (rows, k)).astype(np.float32), rng.normal(0, 2, (rows, k)).astype(np.float32)
        w = rng.normal(1, 0.1, k).astype(np.float32)
        da, db, dw = dev(a), dev(b), dev(w)
        s, y = torch.empty_like(da), torch.empty_like(da)
        r = lambda t: capi.C.byref(t)
        qmm._chk(qmm.lib.qmm_op_add_rms_norm(qmm.ctx, r(T(da, F32, [k, rows])), r(T(db, F32, [k, rows])), r(T(dw, F32, [k])),
                                             r(T(s, F32, [k, rows])), r(T(y, F32, [k, rows])), 1e-6, qmm._stream()))
        assert np.array_equal(s.cpu().numpy(), a + b)
        assert rel_rms(y.cpu().numpy(), rms_norm(a + b, w, 1e-6)) < 2e-6


def rope_ref(x, pos, n_dims, theta_scale):
    """normal-mode rope of x [n_tok, n_head, d] with the CPU's repeated f32 multiply for theta (ggml-cpu.c:8634-8648)"""
    out = x.copy()
    for t in range(x.shape[0]):
        theta = np.float32(pos[t])
        for p in range(n_dims // 2):
            c, s = np.float32(np.cos(np.float64(theta))), np.float32(np.sin(np.float64(theta)))
            x0, x1 = x[t, :, 2 * p].copy(), x[t, :, 2 * p + 1].copy()
            out[t, :, 2 * p] = x0 * c - x1 * s
            out[t, :, 2 * p + 1] = x0 * s + x1 * c
            theta = np.float32(theta * theta_scale)
    return out


@pytest.mark.parametrize("n_tok", [1, 5, 70])
def test_rope_kv_store_one_launch(qmm, n_tok):
    """rope(q) -> f32, rope(k) -> f16 K cache rows, v -> transposed f16 V cache, as build_attn lays them out"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(7 + n_tok)
    d, h, hk, n_ctx, kv_head = 128, 8, 2, 96, 17
    q = rng.normal(0, 1, (n_tok, h, d)).astype(np.float32)
    k = rng.normal(0, 1, (n_tok, hk, d)).astype(np.float32)
    v = rng.normal(0, 1, (n_tok, hk * d)).astype(np.float32)
    pos = np.arange(kv_head, kv_head + n_tok, dtype=np.int32)
    freq_base = 10000.0
    theta_scale = np.float32(np.float32(freq_base) ** np.float32(-2.0 / d))
    dq, dk, dv, dpos = dev(q), dev(k), dev(v), dev(pos)
    q_out = torch.empty_like(dq)
    k_cache = torch.zeros((n_ctx, hk * d), dtype=torch.float16, device="cuda")
    v_cache = torch.zeros((hk * d, n_ctx), dtype=torch.float16, device="cuda")
    params = [0, d, 0, 0, 8192, f2i(freq_base), f2i(1.0), f2i(0.0), f2i(1.0), f2i(32.0), f2i(1.0)]
    tq = T(dq, F32, [d, h, n_tok])
    tqo = T(q_out, F32, [d, h, n_tok], op_params=params)
    tk = T(dk, F32, [d, hk, n_tok])
    tkd = capi.QmmTensor.make(F16, [d, hk, n_tok], data=k_cache.data_ptr() + kv_head * hk * d * 2)
    tv = capi.QmmTensor.make(F32, [n_tok, hk * d], nb=[hk * d * 4, 4, 4 * n_tok * hk * d, 4 * n_tok * hk * d], data=dv.data_ptr())   # v_cur^T
    tvd = capi.QmmTensor.make(F16, [n_tok, hk * d], nb=[2, n_ctx * 2, n_ctx * 2 * hk * d, n_ctx * 2 * hk * d], data=v_cache.data_ptr() + kv_head * 2)
    r = lambda t: capi.C.byref(t)
    qmm._chk(qmm.lib.qmm_rope_kv_store(qmm.ctx, r(tq), r(T(dpos, I32, [n_tok])), None, r(tqo), r(tk), r(tkd), r(tv), r(tvd), qmm._stream()))
    want_q = rope_ref(q, pos, d, theta_scale)
    want_k = rope_ref(k, pos, d, theta_scale).reshape(n_tok, hk * d).astype(np.float16)
    assert rel_rms(q_out.cpu().numpy(), want_q) < 2e-5
    kc = k_cache.cpu().numpy()
    assert np.max(np.abs(kc[kv_head:kv_head + n_tok].astype(np.float32) - want_k.astype(np.float32))) < 4e-3
    assert not kc[:kv_head].any() and not kc[kv_head + n_tok:].any()
    vc = v_cache.cpu().numpy()
    assert np.array_equal(vc[:, kv_head:kv_head + n_tok], v.T.astype(np.float16))
    assert not vc[:, :kv_head].any() and not vc[:, kv_head + n_tok:].any()


@pytest.mark.parametrize("n_tok,n_kv,d", [(1, 640, 128), (3, 96, 128), (1, 32, 64), (8, 1024, 128), (2, 4160, 128), (1, 1056, 64), (3, 16384, 128)])
def test_attn_decode_one_launch(qmm, n_tok, n_kv, d):
    """KQ -> soft_max(scale, mask) -> KQV -> merged heads, grouped-query, against f64 numpy with the CPU's f16 roundings of q and p"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_kv + n_tok)
    h, hk, n_ctx = 8, 2, n_kv + 64
    q = rng.normal(0, 1, (h, n_tok, d)).astype(np.float32)
    kc = rng.normal(0, 1, (n_ctx, hk, d)).astype(np.float16)
    vc = rng.normal(0, 1, (hk, d, n_ctx)).astype(np.float16)
    mask = np.zeros((64, n_kv), np.float32)
    for t in range(n_tok):
        mask[t, n_kv - (n_tok - 1 - t) * 3:] = -np.inf          # each token sees a different prefix
    mask[:, 5] = -np.inf
    scale = 1.0 / np.sqrt(d)
    dq, dk, dv, dm = dev(q.transpose(1, 0, 2)), dev(kc), dev(vc), dev(mask)          # q_cur layout [n_tok, h, d], permuted view below
    out = torch.empty((n_tok, h * d), device="cuda")
    tq = capi.QmmTensor.make(F32, [d, n_tok, h], nb=[4, h * d * 4, d * 4, n_tok * h * d * 4], data=dq.data_ptr())
    tk = capi.QmmTensor.make(F16, [d, n_kv, hk], nb=[2, hk * d * 2, d * 2, n_ctx * hk * d * 2], data=dk.data_ptr())
    tv = capi.QmmTensor.make(F16, [n_kv, d, hk], nb=[2, n_ctx * 2, n_ctx * d * 2, n_ctx * d * hk * 2], data=dv.data_ptr())
    tm = capi.QmmTensor.make(F32, [n_kv, 64], data=dm.data_ptr())
    td = capi.QmmTensor.make(F32, [h * d, n_tok], data=out.data_ptr())
    r = lambda t: capi.C.byref(t)
    qmm._chk(qmm.lib.qmm_attn_decode(qmm.ctx, r(tq), r(tk), r(tv), r(tm), r(td), scale, qmm._stream()))
    got = out.cpu().numpy().reshape(n_tok, h, d)
    qh = q.astype(np.float16).astype(np.float64)
    for hh in range(h):
        g = hh // (h // hk)
        s = qh[hh] @ kc[:n_kv, g].astype(np.float64).T * np.float32(scale) + mask[:n_tok]
        p = np.exp(s - s.max(axis=1, keepdims=True))
        p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32).astype(np.float16).astype(np.float64)
        want = p @ vc[g, :, :n_kv].astype(np.float64).T
        # p is rounded to f16 (as the CPU's F16 vec_dot does): an f32-vs-f64 difference in exp / sum can flip such a rounding,
        # one flip is 2^-11 of one probability.  From n_kv = 1024 the kv range is split over workgroups and p stays f32.
        assert rel_rms(got[:, hh], want) < (5e-4 if n_kv < 1024 else 1.5e-3), (hh, n_tok, n_kv)


@pytest.mark.parametrize("n_tok,n_kv,d", [(512, 512, 128), (70, 96, 128), (33, 64, 64), (200, 480, 128), (300, 1024, 128), (130, 2080, 128),
                                           (64, 576, 64)])
def test_attn_prefill_one_launch(qmm, n_tok, n_kv, d):
    """the same chain for a prompt batch with the scores held in LDS, 512 kv columns at a time (running max / sum beyond that):
    ragged token tiles, n_kv not a multiple of the 64-row K tile or of the chunk, D = 64 and 128, grouped-query, causal-style mask"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_kv * 7 + n_tok)
    h, hk, n_ctx = 8, 2, n_kv + 32
    q = rng.normal(0, 1, (h, n_tok, d)).astype(np.float32)
    kc = rng.normal(0, 1, (n_ctx, hk, d)).astype(np.float16)
    vc = rng.normal(0, 1, (hk, d, n_ctx)).astype(np.float16)
    n_pad = (n_tok + 63) // 64 * 64
    mask = np.zeros((n_pad, n_kv), np.float32)
    for t in range(n_tok):
        mask[t, min(n_kv, n_kv - n_tok + t + 1):] = -np.inf           # token t sees the prefix that ends at its own position
    scale = 1.0 / np.sqrt(d)
    dq, dk, dv, dm = dev(q.transpose(1, 0, 2)), dev(kc), dev(vc), dev(mask)
    out = torch.empty((n_tok, h * d), device="cuda")
    tq = capi.QmmTensor.make(F32, [d, n_tok, h], nb=[4, h * d * 4, d * 4, n_tok * h * d * 4], data=dq.data_ptr())
    tk = capi.QmmTensor.make(F16, [d, n_kv, hk], nb=[2, hk * d * 2, d * 2, n_ctx * hk * d * 2], data=dk.data_ptr())
    tv = capi.QmmTensor.make(F16, [n_kv, d, hk], nb=[2, n_ctx * 2, n_ctx * d * 2, n_ctx * d * hk * 2], data=dv.data_ptr())
    tm = capi.QmmTensor.make(F32, [n_kv, n_pad], data=dm.data_ptr())
    td = capi.QmmTensor.make(F32, [h * d, n_tok], data=out.data_ptr())
    r = lambda t: capi.C.byref(t)
    qmm._chk(qmm.lib.qmm_attn_prefill(qmm.ctx, r(tq), r(tk), r(tv), r(tm), r(td), scale, qmm._stream()))
    got = out.cpu().numpy().reshape(n_tok, h, d)
    qh = q.astype(np.float16).astype(np.float64)
    for hh in range(h):
        g = hh // (h // hk)
        s = qh[hh] @ kc[:n_kv, g].astype(np.float64).T * np.float32(scale) + mask[:n_tok]
        p = np.exp(s - s.max(axis=1, keepdims=True))
        p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32).astype(np.float16).astype(np.float64)
        want = p @ vc[g, :, :n_kv].astype(np.float64).T
        # tokens early in the prompt have few, large probabilities: one flipped f16 rounding of p ~ 0.5 moves an output by 2.4e-4
        # (beyond one 512-column chunk p is rounded before it is normalised: f16 rounding of the largest p instead of the normalised one)
        assert rel_rms(got[:, hh], want) < (2e-3 if n_kv <= 512 else 6e-3), (hh, n_tok, n_kv)
        assert np.sqrt(np.mean((got[:, hh] - want) ** 2) / np.mean(want ** 2)) < (1e-4 if n_kv <= 512 else 4e-4)


@pytest.mark.parametrize("n_tok,j0,d", [(1, 100, 128), (3, 45, 128), (8, 0, 64), (2, 254, 128)])
def test_attn_decode_rope_one_launch(qmm, n_tok, j0, d):
    """rope(q), rope(k) -> K cache, v -> V cache and the attention over the updated cache as ONE launch: the result and both
    caches against the two-step numpy reference (the new rows are used from LDS, never read back from the cache)"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(900 + n_tok + j0)
    h, hk, n_ctx = 8, 2, 320
    n_kv = (j0 + n_tok + 31) // 32 * 32
    q = rng.normal(0, 1, (n_tok, h, d)).astype(np.float32)
    k = rng.normal(0, 1, (n_tok, hk, d)).astype(np.float32)
    v = rng.normal(0, 1, (n_tok, hk * d)).astype(np.float32)
    pos = np.arange(j0, j0 + n_tok, dtype=np.int32)
    kc = np.zeros((n_ctx, hk, d), np.float16)
    vc = np.zeros((hk * d, n_ctx), np.float16)
    kc[:j0] = rng.normal(0, 1, (j0, hk, d)).astype(np.float16)
    vc[:, :j0] = rng.normal(0, 1, (hk * d, j0)).astype(np.float16)
    kc[j0:j0 + n_tok] = 77.0                     # stale garbage where the new rows go: must never be read
    vc[:, j0:j0 + n_tok] = -55.0
    mask = np.full((64, n_kv), -np.inf, np.float32)
    for t in range(n_tok):
        mask[t, :j0 + t + 1] = 0.0
    freq_base = 10000.0
    theta_scale = np.float32(np.float32(freq_base) ** np.float32(-2.0 / d))
    scale = 1.0 / np.sqrt(d)
    dq, dk, dv, dpos, dkc, dvc, dm = dev(q), dev(k), dev(v), dev(pos), dev(kc), dev(vc), dev(mask)
    out = torch.empty((n_tok, h * d), device="cuda")
    params = [0, d, 0, 0, 8192, f2i(freq_base), f2i(1.0), f2i(0.0), f2i(1.0), f2i(32.0), f2i(1.0)]
    M = capi.QmmTensor.make
    tq = M(F32, [d, h, n_tok], data=dq.data_ptr())
    tqr = M(F32, [d, h, n_tok], data=dq.data_ptr(), op_params=params)
    tkn = M(F32, [d, hk, n_tok], data=dk.data_ptr())
    tks = M(F16, [d, hk, n_tok], data=dkc.data_ptr() + j0 * hk * d * 2)
    tvn = M(F32, [n_tok, hk * d], nb=[hk * d * 4, 4, 4 * n_tok * hk * d, 4 * n_tok * hk * d], data=dv.data_ptr())
    tvs = M(F16, [n_tok, hk * d], nb=[2, n_ctx * 2, n_ctx * 2 * hk * d, n_ctx * 2 * hk * d], data=dvc.data_ptr() + j0 * 2)
    tk = M(F16, [d, n_kv, hk], nb=[2, hk * d * 2, d * 2, n_ctx * hk * d * 2], data=dkc.data_ptr())
    tv = M(F16, [n_kv, d, hk], nb=[2, n_ctx * 2, n_ctx * d * 2, n_ctx * d * hk * 2], data=dvc.data_ptr())
    tm = M(F32, [n_kv, 64], data=dm.data_ptr())
    td = M(F32, [h * d, n_tok], data=out.data_ptr())
    r = lambda t: capi.C.byref(t)
    qmm._chk(qmm.lib.qmm_attn_decode_rope(qmm.ctx, r(tq), r(M(I32, [n_tok], data=dpos.data_ptr())), None, r(tqr), r(tkn), r(tks), r(tvn), r(tvs),
                                          r(tk), r(tv), r(tm), r(td), scale, j0, qmm._stream()))
    want_k = rope_ref(k, pos, d, theta_scale).astype(np.float16)
    kc2, vc2 = dkc.cpu().numpy(), dvc.cpu().numpy()
    assert np.max(np.abs(kc2[j0:j0 + n_tok].astype(np.float32) - want_k.astype(np.float32))) < 4e-3
    assert np.array_equal(kc2[:j0], kc[:j0]) and not kc2[j0 + n_tok:].any()
    assert np.array_equal(vc2[:, j0:j0 + n_tok], v.T.astype(np.float16)) and np.array_equal(vc2[:, :j0], vc[:, :j0])
    # reference attention over the reference caches
    kr, vr = kc.copy(), vc.copy()
    kr[j0:j0 + n_tok] = want_k
    vr[:, j0:j0 + n_tok] = v.T.astype(np.float16)
    qh = rope_ref(q, pos, d, theta_scale).astype(np.float16).astype(np.float64)            # [n_tok, h, d]
    got = out.cpu().numpy().reshape(n_tok, h, d)
    for hh in range(h):
        g = hh // (h // hk)
        s = qh[:, hh] @ kr[:n_kv, g].astype(np.float64).T * np.float32(scale) + mask[:n_tok]
        p = np.exp(s - s.max(axis=1, keepdims=True))
        p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32).astype(np.float16).astype(np.float64)
        want = p @ vr[g * d:(g + 1) * d, :n_kv].astype(np.float64).T
        assert rel_rms(got[:, hh], want) < 3e-3, (hh, n_tok, j0)                        # f16 roundings of the roped q / k can flip
        assert np.sqrt(np.mean((got[:, hh] - want) ** 2) / np.mean(want ** 2)) < 5e-4


@pytest.mark.parametrize("n_tok,n_expert,n_used", [(1, 8, 2), (77, 8, 2), (5, 64, 6), (512, 16, 4)])
def test_moe_router_one_launch(qmm, n_tok, n_expert, n_used):
    """soft_max -> argsort(desc) -> top-k weights / their sum (build_moe_ffn's router) in one launch, against numpy"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_tok * 3 + n_expert)
    logits = rng.normal(0, 2, (n_tok, n_expert)).astype(np.float32)
    dl = dev(logits)
    ids = torch.full((n_tok, n_expert), -1, dtype=torch.int32, device="cuda")
    w = torch.zeros((n_tok, n_used), device="cuda")
    M = capi.QmmTensor.make
    r = lambda t: capi.C.byref(t)
    qmm._chk(qmm.lib.qmm_moe_router(qmm.ctx, r(M(F32, [n_expert, n_tok], data=dl.data_ptr())), r(M(I32, [n_expert, n_tok], data=ids.data_ptr())),
                                    r(M(F32, [n_used, n_tok], data=w.data_ptr())), n_used, 1, qmm._stream()))
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    p = e / e.sum(axis=1, keepdims=True)
    order = np.argsort(-p, axis=1, kind="stable")
    assert np.array_equal(ids.cpu().numpy(), order.astype(np.int32))
    sel = np.take_along_axis(p, order[:, :n_used], axis=1)
    assert np.allclose(w.cpu().numpy(), sel / sel.sum(axis=1, keepdims=True), rtol=2e-6, atol=1e-7)


QMM_OP_MUL_MAT_F = 19          # include/ggml_mi355x_ops.h: QMM_OP_ADD = 1 ... QMM_OP_GET_ROWS = 18, QMM_OP_MUL_MAT_F


@pytest.mark.parametrize("n_tok,n_expert,n_used,k", [(1, 8, 2, 4096), (3, 8, 2, 4096), (8, 64, 6, 2048), (2, 6, 3, 1030), (1, 16, 4, 5120)])
def test_moe_router_with_its_logits_one_launch(qmm, n_tok, n_expert, n_used, k):
    """qmm_moe_router_logits (round 3): logits = gate_inp x, soft_max, argsort, top-k weights in ONE launch for a few tokens, against the
    two launches it replaces (qmm_op MUL_MAT on F32 + qmm_moe_router): the same bits in all three outputs.  Shapes: experts that do
    not fill the four groups of 256 threads (6), the 64-expert limit, K that is not a multiple of 4 (scalar tail)"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_tok * 7 + n_expert + k)
    gi = dev(rng.normal(0, 0.05, (n_expert, k)).astype(np.float32))
    x = dev(rng.normal(0, 1, (n_tok, k)).astype(np.float32))
    M = capi.QmmTensor.make
    r = lambda t: capi.C.byref(t)
    outs = []
    for fused in (False, True):
        lg = torch.full((n_tok, n_expert), float("nan"), device="cuda")
        ids = torch.full((n_tok, n_expert), -1, dtype=torch.int32, device="cuda")
        w = torch.zeros((n_tok, n_used), device="cuda")
        tg, tx = M(F32, [k, n_expert], data=gi.data_ptr()), M(F32, [k, n_tok], data=x.data_ptr())
        tl, ti, tw = M(F32, [n_expert, n_tok], data=lg.data_ptr()), M(I32, [n_expert, n_tok], data=ids.data_ptr()), M(F32, [n_used, n_tok], data=w.data_ptr())
        if fused:
            assert qmm.lib.qmm_moe_router_logits_supported(r(tg), r(tx), r(tl), r(ti), r(tw), n_used)
            qmm._chk(qmm.lib.qmm_moe_router_logits(qmm.ctx, r(tg), r(tx), r(tl), r(ti), r(tw), n_used, 1, qmm._stream()))
        else:
            qmm._chk(qmm.lib.qmm_op_compute(qmm.ctx, QMM_OP_MUL_MAT_F, r(tg), r(tx), None, r(tl), qmm._stream()))
            qmm._chk(qmm.lib.qmm_moe_router(qmm.ctx, r(tl), r(ti), r(tw), n_used, 1, qmm._stream()))
        qmm.synchronize()
        outs.append((lg, ids, w))
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2].view(torch.int32), outs[1][2].view(torch.int32))
    want = x.cpu().numpy().astype(np.float64) @ gi.cpu().numpy().astype(np.float64).T
    assert np.allclose(outs[1][0].cpu().numpy(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n_tok,n_expert,n_used,k,in_place", [(1, 8, 2, 4096, False), (1, 8, 2, 4096, True), (4, 8, 2, 4096, False), (2, 16, 4, 5120, True), (8, 64, 6, 16384, False)])
def test_moe_router_with_norm_and_logits_one_launch(qmm, n_tok, n_expert, n_used, k, in_place):
    """qmm_moe_router_logits_norm (round 3): ffn_norm = rms_norm(x) * w, the router's logits against it, soft_max, argsort and the top-k
    weights in ONE launch, against the three launches it replaces (qmm_op RMS_NORM_MUL, qmm_op MUL_MAT on F32, qmm_moe_router): the
    same bits in the normed row and in all three router outputs; also with the normed row written over x (ggml-alloc's in-place mul)"""
    import struct
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_tok * 11 + n_expert + k)
    gi = dev(rng.normal(0, 0.05, (n_expert, k)).astype(np.float32))
    x0 = rng.normal(0, 1.5, (n_tok, k)).astype(np.float32)
    nw = dev(rng.uniform(0.5, 1.5, (k,)).astype(np.float32))
    eps = 1e-5
    eps_bits = struct.unpack("<i", struct.pack("<f", eps))[0]
    M = capi.QmmTensor.make
    r = lambda t: capi.C.byref(t)
    outs = []
    for fused in (False, True):
        x = dev(x0)
        y = x if in_place else torch.full((n_tok, k), float("nan"), device="cuda")
        lg = torch.full((n_tok, n_expert), float("nan"), device="cuda")
        ids = torch.full((n_tok, n_expert), -1, dtype=torch.int32, device="cuda")
        w = torch.zeros((n_tok, n_used), device="cuda")
        tg, tx, tn = M(F32, [k, n_expert], data=gi.data_ptr()), M(F32, [k, n_tok], data=x.data_ptr()), M(F32, [k], data=nw.data_ptr())
        ty = M(F32, [k, n_tok], data=y.data_ptr(), op_params=(eps_bits,))
        tl, ti, tw = M(F32, [n_expert, n_tok], data=lg.data_ptr()), M(I32, [n_expert, n_tok], data=ids.data_ptr()), M(F32, [n_used, n_tok], data=w.data_ptr())
        if fused:
            assert qmm.lib.qmm_moe_router_logits_norm_supported(r(tg), r(tx), r(tn), r(ty), r(tl), r(ti), r(tw), n_used)
            qmm._chk(qmm.lib.qmm_moe_router_logits_norm(qmm.ctx, r(tg), r(tx), r(tn), eps, r(ty), r(tl), r(ti), r(tw), n_used, 1, qmm._stream()))
        else:
            qmm._chk(qmm.lib.qmm_op_compute(qmm.ctx, capi.OP_RMS_NORM_MUL, r(tx), r(tn), None, r(ty), qmm._stream()))
            qmm._chk(qmm.lib.qmm_op_compute(qmm.ctx, QMM_OP_MUL_MAT_F, r(tg), r(ty), None, r(tl), qmm._stream()))
            qmm._chk(qmm.lib.qmm_moe_router(qmm.ctx, r(tl), r(ti), r(tw), n_used, 1, qmm._stream()))
        qmm.synchronize()
        outs.append((y.clone(), lg, ids, w))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert np.allclose(outs[1][0].cpu().numpy(), rms_norm(x0, nw.cpu().numpy(), eps), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("n_tok,n_used,e", [(1, 2, 4096), (70, 2, 1024), (5, 6, 512)])
def test_moe_combine_one_launch(qmm, n_tok, n_used, e):
    """experts * weights and the sum over the used experts (build_moe_ffn's tail) in one launch, in place over slice 0 as ggml-alloc
    lays it out, against numpy in the graph's order of operations"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_tok + e)
    x = rng.normal(0, 1, (n_tok, n_used, e)).astype(np.float32)
    w = rng.uniform(0.1, 0.9, (n_tok, n_used, 1)).astype(np.float32)
    dx, dw = dev(x), dev(w)
    M = capi.QmmTensor.make
    r = lambda t: capi.C.byref(t)
    tx = M(F32, [e, n_used, n_tok], data=dx.data_ptr())
    tw = M(F32, [1, n_used, n_tok], data=dw.data_ptr())
    to = M(F32, [e, n_tok], nb=[4, n_used * e * 4, n_used * e * 4 * n_tok, n_used * e * 4 * n_tok], data=dx.data_ptr())   # view_2d of slice 0
    qmm._chk(qmm.lib.qmm_moe_combine(qmm.ctx, r(tx), r(tw), r(to), qmm._stream()))
    want = x[:, 0] * w[:, 0]
    for u in range(1, n_used):
        want = want + x[:, u] * w[:, u]
    got = dx.cpu().numpy()[:, 0]
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n_tok,n_used,e", [(1, 2, 4096), (8, 2, 4096), (3, 6, 1024), (300, 2, 2048), (2, 2, 8192)])
def test_moe_combine_with_residual_add_and_rms_norm_one_launch(qmm, n_tok, n_used, e):
    """qmm_moe_combine_add_rms_norm (round 3) against qmm_moe_combine followed by qmm_op_add_rms_norm: l_out and the normed row, bit for bit
    (few rows: the 1024-thread partition; 300 rows: the 256-thread one; 8192 columns: two float4 per thread)"""
    from ggml_hexagon_amd import capi
    rng = np.random.default_rng(n_tok * 5 + e)
    dx = dev(rng.normal(0, 1, (n_tok, n_used, e)).astype(np.float32))
    dw = dev(rng.uniform(0.1, 0.9, (n_tok, n_used, 1)).astype(np.float32))
    db = dev(rng.normal(0, 1, (n_tok, e)).astype(np.float32))
    dn = dev(rng.uniform(0.5, 1.5, (e,)).astype(np.float32))
    M = capi.QmmTensor.make
    r = lambda t: capi.C.byref(t)
    tx, tw = M(F32, [e, n_used, n_tok], data=dx.data_ptr()), M(F32, [1, n_used, n_tok], data=dw.data_ptr())
    tb, tn = M(F32, [e, n_tok], data=db.data_ptr()), M(F32, [e], data=dn.data_ptr())
    outs = []
    for fused in (False, True):
        tmp = torch.full((n_tok, e), float("nan"), device="cuda")
        s_, y_ = torch.full((n_tok, e), float("nan"), device="cuda"), torch.full((n_tok, e), float("nan"), device="cuda")
        ts, ty, tt = M(F32, [e, n_tok], data=s_.data_ptr()), M(F32, [e, n_tok], data=y_.data_ptr()), M(F32, [e, n_tok], data=tmp.data_ptr())
        if fused:
            assert qmm.lib.qmm_moe_combine_add_rms_norm_supported(r(tx), r(tw), r(tb), r(tn), r(ts), r(ty))
            qmm._chk(qmm.lib.qmm_moe_combine_add_rms_norm(qmm.ctx, r(tx), r(tw), r(tb), r(tn), r(ts), r(ty), 1e-5, qmm._stream()))
        else:
            qmm._chk(qmm.lib.qmm_moe_combine(qmm.ctx, r(tx), r(tw), r(tt), qmm._stream()))
            qmm._chk(qmm.lib.qmm_op_add_rms_norm(qmm.ctx, r(tt), r(tb), r(tn), r(ts), r(ty), 1e-5, qmm._stream()))
        qmm.synchronize()
        outs.append((s_, y_))
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32))
    want = (dx * dw).sum(dim=1) + db
    assert torch.allclose(outs[1][0], want, rtol=1e-5, atol=1e-6)
