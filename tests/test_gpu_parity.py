"""GPU parity tests proper: every call goes through the C-ABI (include/ggml_mi355x_qmm.h) into the HIP
kernels and is checked against the CPU oracle / the golden vectors of the real reference.

Bars (SURVEY.md §8a "parity modes"):
  * block unpack and activation quantization: BIT-EXACT;
  * mat-vec (N <= 8, Q8-exact mode): only f32 summation order differs from the CPU backend -> max |err| / rms <= 2e-5
    (north-star tolerance: 1e-3 rel on the f32 accumulator);
  * MFMA path, QMM_PREC_F16_Q8 (default): same quantized activations as the CPU, f16 operand rounding only
    -> rel-L2 <= 1e-3;  QMM_PREC_BF16: the reference's own bar NMSE <= 5e-4 (tests/test-backend-ops.cpp:1982-1984).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle.pyoracle import ACT_REF, ACT_X86, Q4_0, Q4_K, Q5_K, Q6_K, Q8_0, Q8_K, TYPE_NAMES, WEIGHT_TYPES, vec_dot_type  # noqa: E402

ALL = WEIGHT_TYPES          # the north-star's five + SURVEY 8f-4's six (Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ4_NL)
IDS = [TYPE_NAMES[t] for t in ALL]
LEGACY = tuple(t for t in ALL if vec_dot_type(t) != Q8_K)      # 32-weight blocks


@pytest.fixture(scope="module")
def qmm():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ggml_hexagon_amd.capi import Qmm
    q = Qmm(0)          # raises if the HIP library is missing or the device is not gfx950: no fallback
    yield q
    q.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_rms(got, want):
    want = want.astype(np.float64)
    return float(np.max(np.abs(got - want)) / max(np.sqrt(np.mean(want ** 2)), 1e-30))


def rel_l2(got, want):
    want = want.astype(np.float64)
    return float(np.sqrt(np.sum((got - want) ** 2) / max(np.sum(want ** 2), 1e-30)))


def nmse(got, want):
    want = want.astype(np.float64)
    return float(np.sum((got - want) ** 2) / max(np.sum(want ** 2), 1e-30))


def weights(ref_or_none, t, m, k, seed):
    """realistic weights when the real quantizer is available (oracle/_ref), synthetic valid blocks otherwise"""
    import ggml_hexagon_amd.synth as synth
    if ref_or_none is not None:
        rng = np.random.default_rng(seed)
        return ref_or_none.quantize_weights(t, rng.uniform(-1, 1, (m, k)).astype(np.float32))
    return synth.synth_weights(t, m, k, seed=seed, sigma=0.3)


@pytest.fixture(scope="module")
def maybe_ref():
    from oracle.pyoracle import RefGgml, ref_available
    return RefGgml("avx2") if ref_available() else None


# ----------------------------------------------------------------------------- bit-exact stages

def test_dequant_golden_bit_exact(qmm, golden):
    t, k = int(golden["type"]), int(golden["K"])
    got = qmm.dequantize(t, dev(golden["w"]), k).cpu().numpy().view(np.uint32)
    want = golden["deq_bits"]
    nan = np.isnan(got.view(np.float32)) & np.isnan(want.view(np.float32))
    assert ((got == want) | nan).all()


@pytest.mark.parametrize("t", ALL, ids=IDS)
def test_dequant_random_blocks_bit_exact(qmm, oracle, t):
    import ggml_hexagon_amd.synth as synth
    k = 2048 if t in LEGACY else 4096
    w = synth.synth_weights(t, 37, k, seed=11)
    got = qmm.dequantize(t, dev(w), k).cpu().numpy()
    want = oracle.dequantize(t, w, k)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("mode", [ACT_REF, ACT_X86])
def test_activation_quantizer_bit_exact(qmm, oracle, golden, mode):
    t = int(golden["type"])
    vt = vec_dot_type(t)
    rng = np.random.default_rng(3)
    x = np.concatenate([golden["act"], rng.standard_normal((21, 512)).astype(np.float32) * 5])
    qmm.set_act_mode(mode)
    q, d, bs = qmm.quantize_act(vt, dev(x))
    qmm.set_act_mode(ACT_REF)
    wire = oracle.quantize_act(t, x, mode)                       # ggml wire blocks
    rows = x.shape[0]
    if vt == Q8_0:
        blk = wire.reshape(rows, -1, 34)
        want_d = blk[:, :, :2].copy().view(np.float16).astype(np.float32).reshape(rows, -1)
        want_q = blk[:, :, 2:].copy().view(np.int8).reshape(rows, -1)
    elif vt == 9:                                                # Q8_1: d, s = f16(d * sum), 32 int8 (ggml-common.h:216-227)
        blk = wire.reshape(rows, -1, 36)
        want_d = blk[:, :, :2].copy().view(np.float16).astype(np.float32).reshape(rows, -1)
        want_s = blk[:, :, 2:4].copy().view(np.float16).astype(np.float32).reshape(rows, -1)
        want_q = blk[:, :, 4:].copy().view(np.int8).reshape(rows, -1)
        assert np.array_equal(bs.cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    else:
        blk = wire.reshape(rows, -1, 292)
        want_d = blk[:, :, :4].copy().view(np.float32).reshape(rows, -1)
        want_q = blk[:, :, 4:260].copy().view(np.int8).reshape(rows, -1)
        want_bs = blk[:, :, 260:].copy().view(np.int16).reshape(rows, -1)
        assert np.array_equal(bs.cpu().numpy(), want_bs)
    assert np.array_equal(q.cpu().numpy(), want_q)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), want_d.view(np.uint32))


def test_q8_K_double_rounding_ties_bit_exact(qmm, oracle):
    """regression: hipcc's default -ffp-contract=fast fused iscale*x into the magic add of nearest_int"""
    from helpers import q8_K_tie_rows
    x = q8_K_tie_rows(3, 512, seed=4)
    q, d, bs = qmm.quantize_act(Q8_K, dev(x))
    blk = oracle.quantize_act(Q4_K, x).reshape(3, -1, 292)
    assert np.array_equal(q.cpu().numpy(), blk[:, :, 4:260].copy().view(np.int8).reshape(3, -1))
    assert np.array_equal(bs.cpu().numpy(), blk[:, :, 260:].copy().view(np.int16).reshape(3, -1))
    # and through the fused in-kernel quantizer of the mat-vec path
    import ggml_hexagon_amd.synth as synth
    w = synth.synth_weights(Q6_K, 64, 512, seed=1, sigma=0.2)
    got = qmm.mul_mat(Q6_K, dev(w), 512, dev(x)).cpu().numpy()
    assert rel_rms(got, oracle.mul_mat(Q6_K, w, 512, x, ACT_REF)) < 2e-5


# ----------------------------------------------------------------------------- mat-vec (N <= 8)

@pytest.mark.parametrize("n", [1, 5])
def test_matvec_golden(qmm, golden, n):
    t, k = int(golden["type"]), int(golden["K"])
    qmm.set_act_mode(ACT_X86)                                    # the golden graph ran the AVX2 quantizer
    got = qmm.mul_mat(t, dev(golden["w"][:-1]), k, dev(golden["act"][:n])).cpu().numpy()
    qmm.set_act_mode(ACT_REF)
    assert rel_rms(got, golden[f"dst_n{n}"]) < 2e-5


@pytest.mark.parametrize("t", ALL, ids=IDS)
@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8])
def test_matvec_vs_oracle(qmm, oracle, maybe_ref, t, n):
    k, m = 4096, 333
    w = weights(maybe_ref, t, m, k, seed=t)
    x = np.random.default_rng(n).uniform(-1, 1, (n, k)).astype(np.float32)
    got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
    want = oracle.mul_mat(t, w, k, x, ACT_REF)
    assert rel_rms(got, want) < 2e-5


@pytest.mark.parametrize("t", ALL, ids=IDS)
def test_matvec_small_and_ragged_shapes(qmm, oracle, t):
    """test-backend-ops shapes: m=16,k=256 (tests/test-backend-ops.cpp:4132-4136); k = one block; m = 1"""
    import ggml_hexagon_amd.synth as synth
    for (m, k, n) in ((16, 256, 1), (16, 256, 8), (1, 256, 3), (5, 512 if t in LEGACY else 768, 2)):
        w = synth.synth_weights(t, m, k, seed=m + k, sigma=0.2)
        x = np.random.default_rng(k).uniform(-1, 1, (n, k)).astype(np.float32)
        got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
        assert rel_rms(got, oracle.mul_mat(t, w, k, x, ACT_REF)) < 2e-5
    if t in LEGACY:                                             # k = blck (32)
        w = synth.synth_weights(t, 7, 32, seed=1, sigma=0.2)
        x = np.random.default_rng(0).uniform(-1, 1, (2, 32)).astype(np.float32)
        got = qmm.mul_mat(t, dev(w), 32, dev(x)).cpu().numpy()
        assert rel_rms(got, oracle.mul_mat(t, w, 32, x, ACT_REF)) < 2e-5


def test_matvec_zero_activations_and_strided_dst(qmm, oracle):
    import ggml_hexagon_amd.synth as synth
    k, m = 1024, 40
    w = synth.synth_weights(Q4_K, m, k, seed=2)
    x = np.zeros((2, k), np.float32)
    x[1, 300:] = np.random.default_rng(0).uniform(-1, 1, k - 300)
    out = torch.full((2, m + 24), 7.0, device="cuda")            # ldd > M: the padding must stay untouched
    qmm.mul_mat(Q4_K, dev(w), k, dev(x), out=out[:, :m])
    o = out.cpu().numpy()
    assert (o[:, m:] == 7.0).all() and (o[0, :m] == 0).all()
    assert rel_rms(o[1:, :m], oracle.mul_mat(Q4_K, w, k, x, ACT_REF)[1:]) < 2e-5


def test_group_launch_matches_single(qmm, maybe_ref):
    k = 4096
    ws = [(Q4_K, dev(weights(maybe_ref, Q4_K, 512, k, 1))), (Q4_K, dev(weights(maybe_ref, Q4_K, 128, k, 2))),
          (Q6_K, dev(weights(maybe_ref, Q6_K, 128, k, 3)))]
    for n in (1, 4, 64):
        x = dev(np.random.default_rng(n).uniform(-1, 1, (n, k)).astype(np.float32))
        outs = [torch.empty((n, w.shape[0]), device="cuda") for _, w in ws]
        qmm.mul_mat_group(ws, k, x, outs)
        for (t, w), o in zip(ws, outs):
            assert torch.equal(o, qmm.mul_mat(t, w, k, x))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
def test_mixed_type_group_one_launch(qmm, oracle, n):
    """K-quant matrices of different types sharing src1 go out as ONE mixed-type mat-vec launch (matvec_kmix_kernel):
    every matrix against the oracle, bit-identical to its own single launch, ragged row counts, K = 14336 included"""
    import ggml_hexagon_amd.synth as synth
    for k, spec in ((4096, ((Q4_K, 300), (Q4_K, 70), (Q6_K, 130))), (1024, ((Q5_K, 64), (Q6_K, 33), (Q4_K, 1), (Q5_K, 257))),
                    (14336, ((Q6_K, 96), (Q4_K, 160)))):
        ws_np = [(t, synth.synth_weights(t, m, k, seed=m + t, sigma=0.25)) for t, m in spec]
        ws = [(t, dev(w)) for t, w in ws_np]
        x = np.random.default_rng(k + n).uniform(-1, 1, (n, k)).astype(np.float32)
        outs = [torch.full((n, w.shape[0]), 3.0, device="cuda") for _, w in ws]
        qmm.mul_mat_group(ws, k, dev(x), outs)
        for (t, w), (_, wd), o in zip(ws_np, ws, outs):
            assert rel_rms(o.cpu().numpy(), oracle.mul_mat(t, w, k, x, ACT_REF)) < 2e-5, (TYPE_NAMES[t], k, n)
            assert torch.equal(o, qmm.mul_mat(t, wd, k, dev(x)))


@pytest.mark.parametrize("n", [1, 2, 4, 5])
def test_mixed_format_group_one_launch(qmm, oracle, n):
    """round 3: K-quant matrices and Q8_0 matrices sharing src1 (Mixtral's attn_q in Q4_K beside attn_k / attn_v in Q8_0) go out as ONE
    launch that stages the row as Q8_K and as Q8_0 (up to 4 tokens; 5: the per-type launches): every matrix against the oracle and
    bit-identical to its own single launch; wire and planar Q8_0 rows, ragged row counts, K = 14336"""
    import ggml_hexagon_amd.synth as synth
    for k, spec in ((4096, ((Q4_K, 300), (Q8_0, 70), (Q8_0, 130))), (1024, ((Q8_0, 64), (Q6_K, 33), (Q4_K, 1), (Q8_0, 257))), (14336, ((Q8_0, 96), (Q5_K, 160)))):
        ws_np = [(t, synth.synth_weights(t, m, k, seed=m + t, sigma=0.25)) for t, m in spec]
        x = np.random.default_rng(k + n).uniform(-1, 1, (n, k)).astype(np.float32)
        for planar in (False, True):
            ws = []
            for t, w in ws_np:
                wd = dev(w)
                tt = qmm.repack_rows(t, wd, k, True) if planar and t == Q8_0 and qmm.planar_type(t, k, wd.stride(-2)) else t
                ws.append((tt, wd))
            outs = [torch.full((n, w.shape[0]), 3.0, device="cuda") for _, w in ws]
            labels = qmm.trace(lambda: qmm.mul_mat_group(ws, k, dev(x), outs))
            assert (len(labels) == 1 and "q8_0" in labels[0]) == (n <= 4 and n * k * 21 // 8 + 4096 <= 150 * 1024), labels      # (both staged rows must fit LDS)
            for (t, w), (tt, wd), o in zip(ws_np, ws, outs):
                assert rel_rms(o.cpu().numpy(), oracle.mul_mat(t, w, k, x, ACT_REF)) < 2e-5, (TYPE_NAMES[t], k, n)
                assert torch.equal(o, qmm.mul_mat(tt, wd, k, dev(x)))


@pytest.mark.parametrize("t", ALL, ids=IDS)
def test_prefill_group_one_tiled_launch(qmm, oracle, t):
    """same-type matrices sharing src1 at prefill batch sizes go out as ONE tiled launch (+ one split-K reduce): every
    matrix against the oracle and identical to its own single launch; ragged row counts, strided dst, 2 to 4 matrices"""
    import ggml_hexagon_amd.synth as synth
    kblk = 32 if t in LEGACY else 256
    for k, ms, n in ((kblk * (2048 // kblk), (512, 128, 128), 300), (kblk * (1024 // kblk), (300, 70), 200), (kblk * (1536 // kblk), (256, 256, 1, 33), 512)):
        ws_np = [synth.synth_weights(t, m, k, seed=m + i, sigma=0.25) for i, m in enumerate(ms)]
        ws = [(t, dev(w)) for w in ws_np]
        x = np.random.default_rng(k + n).uniform(-1, 1, (n, k)).astype(np.float32)
        big = [torch.full((n, m + 8), 3.0, device="cuda") for m in ms]
        outs = [b[:, :m] for b, m in zip(big, ms)]
        qmm.mul_mat_group(ws, k, dev(x), outs)
        for w, (_, wd), o, b, m in zip(ws_np, ws, outs, big, ms):
            assert (b[:, m:] == 3.0).all()
            assert rel_l2(o.cpu().numpy(), oracle.mul_mat(t, w, k, x, ACT_REF)) < 1e-3, (TYPE_NAMES[t], k, m, n)
            assert torch.equal(o, qmm.mul_mat(t, wd, k, dev(x)))


# ----------------------------------------------------------------------------- MFMA path (N > 8)

@pytest.mark.parametrize("t", ALL, ids=IDS)
@pytest.mark.parametrize("n", [9, 32, 33, 64, 129, 512])
def test_mfma_f16q8_vs_oracle(qmm, oracle, maybe_ref, t, n):
    k, m = 2048, 200
    w = weights(maybe_ref, t, m, k, seed=20 + t)
    x = np.random.default_rng(n).uniform(-1, 1, (n, k)).astype(np.float32)
    got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
    want = oracle.mul_mat(t, w, k, x, ACT_REF)
    assert rel_l2(got, want) < 1e-3, rel_l2(got, want)


@pytest.mark.parametrize("t", ALL, ids=IDS)
def test_mfma_bf16_nmse(qmm, oracle, maybe_ref, t):
    from ggml_hexagon_amd.capi import PREC_BF16, PREC_F16_Q8
    k, m, n = 2048, 200, 64
    w = weights(maybe_ref, t, m, k, seed=40 + t)
    x = np.random.default_rng(1).uniform(-1, 1, (n, k)).astype(np.float32)
    qmm.set_precision(PREC_BF16)
    got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
    qmm.set_precision(PREC_F16_Q8)
    assert nmse(got, oracle.mul_mat(t, w, k, x, ACT_REF)) < 5e-4


@pytest.mark.parametrize("t", (Q4_0, Q4_K), ids=["q4_0", "q4_K"])
def test_mfma_ragged(qmm, oracle, t):
    import ggml_hexagon_amd.synth as synth
    for (m, k, n) in ((16, 256, 9), (512, 256, 32), (77, 1024, 130), (130, 512, 257)):
        w = synth.synth_weights(t, m, k, seed=m, sigma=0.2)
        x = np.random.default_rng(n).uniform(-1, 1, (n, k)).astype(np.float32)
        got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
        assert rel_l2(got, oracle.mul_mat(t, w, k, x, ACT_REF)) < 1e-3
    if t == Q4_0:
        w = synth.synth_weights(t, 20, 32, seed=3, sigma=0.2)    # k = blck, padded K-step
        x = np.random.default_rng(5).uniform(-1, 1, (40, 32)).astype(np.float32)
        got = qmm.mul_mat(t, dev(w), 32, dev(x)).cpu().numpy()
        assert rel_l2(got, oracle.mul_mat(t, w, 32, x, ACT_REF)) < 1e-3


# ----------------------------------------------------------------------------- MUL_MAT_ID

@pytest.mark.parametrize("ne11", [1, 2])
def test_mul_mat_id_golden(qmm, golden, ne11):
    t, k = int(golden["type"]), int(golden["K"])
    n_used = int(golden["id_n_used"])
    ids_full = dev(golden["id_ids_full"])
    ids = ids_full[:, :n_used]                                   # strided view (test-backend-ops.cpp:2097-2102)
    qmm.set_act_mode(ACT_X86)
    got = qmm.mul_mat_id(t, dev(golden["id_w"]), k, dev(golden[f"id_b_ne11_{ne11}"]), ids).cpu().numpy()
    qmm.set_act_mode(ACT_REF)
    assert rel_rms(got, golden[f"id_dst_ne11_{ne11}"]) < 2e-5


@pytest.mark.parametrize("t", ALL, ids=IDS)
@pytest.mark.parametrize("n_tokens,n_expert,n_used", [(1, 8, 2), (4, 4, 4), (32, 8, 2), (129, 4, 1), (129, 8, 4)])
def test_mul_mat_id_vs_oracle(qmm, oracle, t, n_tokens, n_expert, n_used):
    """shapes of test-backend-ops' MUL_MAT_ID family (tests/test-backend-ops.cpp:4229-4258): m=512, k=256"""
    import ggml_hexagon_amd.synth as synth
    m, k = 512, 256
    w = np.stack([synth.synth_weights(t, m, k, seed=e, sigma=0.2) for e in range(n_expert)])
    rng = np.random.default_rng(n_tokens)
    ids_full = np.stack([rng.permutation(n_expert) for _ in range(n_tokens)]).astype(np.int32)
    for ne11 in (1, n_used):
        b = rng.uniform(-1, 1, (n_tokens, ne11, k)).astype(np.float32)
        got = qmm.mul_mat_id(t, dev(w), k, dev(b), dev(ids_full)[:, :n_used]).cpu().numpy()
        want = oracle.mul_mat_id(t, w, k, m, b, ids_full[:, :n_used], ACT_REF)
        if n_tokens * n_used <= 16:
            assert rel_rms(got, want) < 2e-5
        else:
            assert rel_l2(got, want) < 1e-3
    qmm.synchronize()


@pytest.mark.parametrize("t", (Q4_K, Q8_0), ids=["q4_K", "q8_0"])
@pytest.mark.parametrize("n_tokens", [1, 4, 40, 300])
def test_mul_mat_id_pair_matches_two_calls(qmm, oracle, t, n_tokens):
    """ffn_gate_exps + ffn_up_exps in one call (one mat-vec launch / one sort + prep) == the two single calls, and the oracle"""
    import ggml_hexagon_amd.synth as synth
    k, m, n_expert, n_used = 256 if t == Q4_K else 128, 96, 8, 2
    rng = np.random.default_rng(n_tokens)
    w0 = synth.synth_weights(t, n_expert * m, k, seed=1, sigma=0.25).reshape(n_expert, m, -1)
    w1 = synth.synth_weights(t, n_expert * m, k, seed=2, sigma=0.25).reshape(n_expert, m, -1)
    ids = np.ascontiguousarray(np.stack([rng.permutation(n_expert) for _ in range(n_tokens)]).astype(np.int32)[:, :n_used])
    b = rng.uniform(-1, 1, (n_tokens, 1, k)).astype(np.float32)
    o0 = torch.full((n_tokens, n_used, m), 2.0, device="cuda")
    o1 = torch.full((n_tokens, n_used, m), 2.0, device="cuda")
    qmm.mul_mat_id_pair(t, dev(w0), dev(w1), k, dev(b), dev(ids), o0, o1)
    for w, o in ((w0, o0), (w1, o1)):
        assert torch.equal(o, qmm.mul_mat_id(t, dev(w), k, dev(b), dev(ids)))
        want = oracle.mul_mat_id(t, w, k, m, b, ids, ACT_REF)
        err = rel_rms(o.cpu().numpy(), want) if n_tokens * n_used <= 16 else rel_l2(o.cpu().numpy(), want)
        assert err < (2e-5 if n_tokens * n_used <= 16 else 1e-3)


def test_mul_mat_id_bad_expert_is_reported(qmm):
    import ggml_hexagon_amd.synth as synth
    from ggml_hexagon_amd.capi import QmmError
    w = dev(np.stack([synth.synth_weights(Q4_K, 16, 256, seed=e) for e in range(2)]))
    b = torch.ones((1, 1, 256), device="cuda")
    ids = torch.tensor([[1, 5]], dtype=torch.int32, device="cuda")
    out = torch.full((1, 2, 16), 3.0, device="cuda")
    qmm.mul_mat_id(Q4_K, w, 256, b, ids, out=out)
    with pytest.raises(QmmError):
        qmm.synchronize()
    assert (out[0, 1] == 3.0).all()          # the bad slot's row is left untouched
    qmm.synchronize()                        # flag is cleared


# ----------------------------------------------------------------------------- full-size properties

@pytest.mark.parametrize("t", (Q4_0, Q4_K, Q6_K), ids=["q4_0", "q4_K", "q6_K"])
def test_full_size_properties(qmm, oracle, t):
    """BASELINE sizes (4096 x 4096 and the 14336-wide FFN): sampled rows against the oracle, and
    size-independent properties: homogeneity in x (exact for a power-of-two scale: Q8 scales carry it),
    batch independence (row n of a batch == the same row alone), determinism."""
    import ggml_hexagon_amd.synth as synth
    for (m, k) in ((4096, 4096), (4096, 14336)):
        w = synth.synth_weights(t, m, k, seed=9)
        wd = dev(w)
        rng = np.random.default_rng(m + k)
        x = rng.uniform(-1, 1, (4, k)).astype(np.float32)
        xd = dev(x)
        y = qmm.mul_mat(t, wd, k, xd)
        rows = rng.choice(m, 64, replace=False)
        want = oracle.mul_mat(t, w[rows], k, x, ACT_REF)
        assert rel_rms(y.cpu().numpy()[:, rows], want) < 2e-5
        assert torch.equal(y, qmm.mul_mat(t, wd, k, xd))                       # deterministic
        assert torch.equal(qmm.mul_mat(t, wd, k, xd * 4.0), y * 4.0)           # exact homogeneity (2^2)
        y1 = qmm.mul_mat(t, wd, k, xd[2:3].contiguous())
        assert torch.equal(y1[0], y[2])                                        # batch independence
        # prefill kernel at full size against the mat-vec kernel (different code path, same math up to f16 rounding)
        xb = dev(rng.uniform(-1, 1, (64, k)).astype(np.float32))
        yb = qmm.mul_mat(t, wd, k, xb)
        ys = torch.cat([qmm.mul_mat(t, wd, k, xb[i:i + 8].contiguous()) for i in range(0, 64, 8)])
        err = torch.linalg.norm(yb - ys) / torch.linalg.norm(ys)
        assert float(err) < 1e-3


# ----------------------------------------------------------------------------- strides (views, as ggml hands them over)

@pytest.mark.parametrize("n", [3, 40])
def test_strided_operands(qmm, oracle, n):
    """src1 rows ldx > K apart, weight rows nb01 > row size apart, dst rows ldd > M apart — both kernels"""
    import ggml_hexagon_amd.synth as synth
    for t, k in ((Q4_K, 512), (Q6_K, 256), (Q4_0, 160), (Q8_0, 96)):
        m = 70
        w = synth.synth_weights(t, m, k, seed=k, sigma=0.2)
        rb = w.shape[1]
        wbig = torch.zeros((m, rb + 32), dtype=torch.uint8, device="cuda")
        wbig[:, :rb] = dev(w)
        x = np.random.default_rng(k + n).uniform(-1, 1, (n, k)).astype(np.float32)
        xbig = torch.full((n, k + 64), 9.0, device="cuda")
        xbig[:, :k] = dev(x)
        out = torch.full((n, m + 10), 5.0, device="cuda")
        qmm.mul_mat(t, wbig[:, :rb], k, xbig[:, :k], out=out[:, :m])
        o = out.cpu().numpy()
        assert (o[:, m:] == 5.0).all()
        want = oracle.mul_mat(t, w, k, x, ACT_REF)
        assert (rel_rms(o[:, :m], want) < 2e-5) if n <= 8 else (rel_l2(o[:, :m], want) < 1e-3)


@pytest.mark.parametrize("t", (Q4_K, Q6_K, Q8_0), ids=["q4_K", "q6_K", "q8_0"])
def test_splitk_prefill_strided_dst_and_ragged_rows(qmm, oracle, t):
    """few tiles + long K: the tiled kernel splits K over workgroups and splitk_reduce_kernel writes dst — here with a
    row count that is not a multiple of the tile or of 4 (scalar reduce path) and dst rows ldd > M apart (M % 4 == 0:
    the float4 path), against the oracle; the result must also be the same on a second run (fixed summation order)"""
    import ggml_hexagon_amd.synth as synth
    for m, k, n in ((300, 2048, 300), (253, 1536, 270)):      # N > 128: past the few-token kernel
        w = synth.synth_weights(t, m, k, seed=m + k, sigma=0.2)
        x = np.random.default_rng(m).uniform(-1, 1, (n, k)).astype(np.float32)
        out = torch.full((n, m + 12), 7.0, device="cuda")
        qmm.mul_mat(t, dev(w), k, dev(x), out=out[:, :m])
        first = out.cpu().numpy().copy()
        out.fill_(7.0)
        qmm.mul_mat(t, dev(w), k, dev(x), out=out[:, :m])
        o = out.cpu().numpy()
        assert (o[:, m:] == 7.0).all()
        assert np.array_equal(o, first)
        assert rel_l2(o[:, :m], oracle.mul_mat(t, w, k, x, ACT_REF)) < 1e-3


@pytest.mark.parametrize("t", ALL, ids=IDS)
def test_random_shapes_sweep_every_kernel(qmm, oracle, t):
    """seeded random (M, K, N) draws: ragged row counts, K any multiple of the block size, and N on both sides of every
    dispatch boundary (mat-vec <= 8 < few-token kernel <= 64/128 < tiled kernel, with and without split-K)"""
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(1000 + t)
    kblk = 32 if t in LEGACY else 256
    n_choices = [1, 2, 7, 8, 9, 17, 31, 32, 33, 63, 64, 65, 100, 128, 129, 200, 257, 300]
    for case in range(14):
        m = int(rng.integers(1, 700))
        k = kblk * int(rng.integers(1, 2304 // kblk + 1))
        n = int(n_choices[int(rng.integers(0, len(n_choices)))])
        w = synth.synth_weights(t, m, k, seed=case * 7 + t, sigma=0.25)
        x = rng.uniform(-1, 1, (n, k)).astype(np.float32)
        got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
        want = oracle.mul_mat(t, w, k, x, ACT_REF)
        assert got.shape == want.shape
        err = rel_rms(got, want) if n <= 8 else rel_l2(got, want)
        assert err < (2e-5 if n <= 8 else 1e-3), (TYPE_NAMES[t], m, k, n, err)


def test_bad_arguments_fail_loudly(qmm):
    """the C-ABI rejects what the kernels do not cover with an error code and a message; nothing falls back, nothing is
    silently computed: unsupported weight type, K that is not a whole number of blocks, a weight row stride smaller
    than a row, misaligned or too-short src1 rows, dst rows shorter than M"""
    from ggml_hexagon_amd.capi import QmmError
    import ggml_hexagon_amd.synth as synth
    k, m = 512, 64
    w = dev(synth.synth_weights(Q4_K, m, k, seed=1))
    x = torch.zeros((2, k), device="cuda")
    out = torch.empty((2, m), device="cuda")
    lib, ctx, st = qmm.lib, qmm.ctx, qmm._stream()

    def call(t=Q4_K, wp=w.data_ptr(), rb=w.stride(0), kk=k, mm=m, xp=x.data_ptr(), n=2, ldx=k, dp=out.data_ptr(), ldd=m):
        return lib.qmm_mul_mat(ctx, t, wp, rb, kk, mm, xp, n, ldx, dp, ldd, st)

    assert call() == 0
    assert call(n=0) == 0                                     # empty batch: nothing to do
    for kwargs, needle in ((dict(t=16), "type"), (dict(t=0), "type"), (dict(kk=k - 32), "multiple"), (dict(rb=w.stride(0) - 2), "stride"),
                           (dict(xp=x.data_ptr() + 4), "aligned"), (dict(ldx=k - 4), "ldx"), (dict(ldd=m - 1), "ldd")):
        rc = call(**kwargs)
        assert rc < 0, kwargs
        assert needle in lib.qmm_last_error().decode(), (kwargs, lib.qmm_last_error())
    with pytest.raises(QmmError):
        qmm.mul_mat(Q8_K, w, k, x)                            # Q8_K is an activation format, not a weight type
    torch.cuda.synchronize()
    assert call() == 0                                        # the context is still usable after errors


def test_f16_prefill_overflow_is_a_loud_error_and_bf16_handles_it(qmm, oracle):
    """valid GGUF bits the default prefill mode cannot represent: Q4_0 / Q4_K blocks whose d is the largest fp16 (65504; the golden
    edge row tests/golden/make_golden.py keeps out of the matmul fixtures).  (q - 8) * d overflows the f16 the weights are rounded
    to: the launch must not return silently wrong numbers: qmm_synchronize reports it, and QMM_PREC_BF16 computes the product
    (NMSE <= 5e-4, the reference's bar).  The mat-vec path (N <= 8) is exact arithmetic in f32 and is not affected."""
    import ggml_hexagon_amd.synth as synth
    from ggml_hexagon_amd.capi import PREC_BF16, PREC_F16_Q8, QmmError
    k, m, n = 512, 64, 40
    for t, doff in ((Q4_0, 0), (Q4_K, 0)):
        w = synth.synth_weights(t, m, k, seed=3, sigma=0.05).reshape(m, -1, synth.TYPE_SIZE[t])
        w[5, 1, doff:doff + 2] = np.array([0x7BFF], np.uint16).view(np.uint8)         # one block of row 5: d = 65504
        w = w.reshape(m, -1)
        x = np.random.default_rng(9).uniform(-1, 1, (n, k)).astype(np.float32)
        want = oracle.mul_mat(t, w, k, x, ACT_REF)
        assert np.isfinite(want).all()
        got1 = qmm.mul_mat(t, dev(w), k, dev(x[:4])).cpu().numpy()                     # mat-vec kernels: fine
        assert rel_rms(got1, want[:4]) < 2e-5
        qmm.synchronize()
        qmm.mul_mat(t, dev(w), k, dev(x))
        with pytest.raises(QmmError, match="non-finite"):
            qmm.synchronize()
        qmm.synchronize()                                                              # the report clears the condition
        qmm.set_precision(PREC_BF16)
        got = qmm.mul_mat(t, dev(w), k, dev(x)).cpu().numpy()
        qmm.synchronize()
        qmm.set_precision(PREC_F16_Q8)
        assert nmse(got, want) < 5e-4, (TYPE_NAMES[t], nmse(got, want))
