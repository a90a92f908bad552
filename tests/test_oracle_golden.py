"""Pin the CPU oracle (oracle/qmm_oracle.c) against golden vectors produced by the real reference
(tests/golden/make_golden.py) and, when oracle/_ref is present, against the reference live."""
import numpy as np
import pytest

from oracle.pyoracle import ACT_REF, ACT_X86, Q4_0, Q4_K, Q5_K, Q6_K, Q8_0, Q8_K, TYPE_NAMES, WEIGHT_TYPES, vec_dot_type

ALL = WEIGHT_TYPES


def rel_rms(a, b):
    return float(np.max(np.abs(a - b)) / np.sqrt(np.mean(b.astype(np.float64) ** 2)))


def test_type_table(oracle):
    # ggml-common.h block sizes; bytes/weight of SURVEY.md §8d
    assert [oracle.lib.qmo_type_size(t) for t in ALL] == [18, 34, 144, 176, 210, 20, 22, 24, 84, 110, 18, 136]
    assert [oracle.lib.qmo_blck_size(t) for t in ALL] == [32, 32, 256, 256, 256, 32, 32, 32, 256, 256, 32, 256]
    assert oracle.lib.qmo_type_size(9) == 36                        # Q8_1: d, s, 32 int8
    assert oracle.row_size(Q4_K, 4096) == 4096 // 256 * 144
    assert oracle.lib.qmo_type_size(15) == 292


def test_fp16_exhaustive(oracle):
    h = np.arange(65536, dtype=np.uint16)
    mine = np.array([oracle.lib.qmo_fp16_to_fp32(int(v)) for v in h], np.float32)
    want = h.view(np.float16).astype(np.float32)
    same = (mine.view(np.uint32) == want.view(np.uint32)) | (np.isnan(mine) & np.isnan(want))
    assert same.all()
    fin = want[np.isfinite(want)]
    back = np.array([oracle.lib.qmo_fp32_to_fp16(float(v)) for v in fin], np.uint16)
    assert np.array_equal(back, fin.astype(np.float16).view(np.uint16))


def test_dequant_bit_exact(oracle, golden):
    t, k = int(golden["type"]), int(golden["K"])
    got = oracle.dequantize(t, golden["w"], k).view(np.uint32)
    want = golden["deq_bits"]
    nan = np.isnan(got.view(np.float32)) & np.isnan(want.view(np.float32))
    assert ((got == want) | nan).all()


def test_activation_bytes(oracle, golden):
    t = int(golden["type"])
    x = golden["act"]
    assert np.array_equal(oracle.quantize_act(t, x, ACT_REF), golden["act_q_ref"])
    got = oracle.quantize_act(t, x, ACT_X86)
    if vec_dot_type(t) != Q8_K:                      # Q8_0 / Q8_1: the scalar and the AVX2 quantizer
        assert np.array_equal(got, golden["act_q_cpu"])
    else:  # Q8_K has one implementation on x86 (the _ref one)
        assert np.array_equal(golden["act_q_cpu"], golden["act_q_ref"])


def test_act_variants_differ_only_on_ties(golden):
    """the scalar and the AVX2 Q8_0 quantizers of the reference: count, don't hide, differing bytes"""
    a, b = golden["act_q_ref"], golden["act_q_cpu"]
    assert (a != b).mean() < 0.02


@pytest.mark.parametrize("n", [1, 5])
def test_mul_mat_vs_graph(oracle, golden, n):
    t, k = int(golden["type"]), int(golden["K"])
    w = golden["w"][:-1]
    got = oracle.mul_mat(t, w, k, golden["act"][:n], ACT_X86)
    want = golden[f"dst_n{n}"]
    assert got.shape == want.shape
    # same integer stages; only the f32 summation order differs (scalar vs AVX2/llamafile)
    assert rel_rms(got, want) < 2e-6


@pytest.mark.parametrize("ne11", [1, 2])
def test_mul_mat_id_vs_graph(oracle, golden, ne11):
    t, k = int(golden["type"]), int(golden["K"])
    n_used = int(golden["id_n_used"])
    ids = golden["id_ids_full"][:, :n_used]          # strided view, row stride 4
    assert ids.strides[0] == 16
    got = oracle.mul_mat_id(t, golden["id_w"], k, 16, golden[f"id_b_ne11_{ne11}"], ids, ACT_X86)
    assert rel_rms(got, golden[f"id_dst_ne11_{ne11}"]) < 2e-6


def test_mul_mat_is_rowwise_vec_dot(oracle, golden):
    t, k = int(golden["type"]), int(golden["K"])
    w = golden["w"][:8]
    x = golden["act"][:3]
    acts = oracle.quantize_act(t, x, ACT_REF)
    want = np.array([[oracle.vec_dot(t, k, w[m], acts[n]) for m in range(8)] for n in range(3)], np.float32)
    assert np.array_equal(oracle.mul_mat(t, w, k, x, ACT_REF), want)


# ---- live against the real reference (skips when oracle/_ref is absent)

@pytest.mark.parametrize("t", ALL, ids=[TYPE_NAMES[t] for t in ALL])
def test_live_reference(oracle, ref, t):
    import ggml_hexagon_amd.synth as synth
    rng = np.random.default_rng(7 + t)
    k = 1024
    w = np.concatenate([ref.quantize_weights(t, rng.uniform(-1, 1, (12, k)).astype(np.float32)),
                        synth.synth_weights(t, 12, k, seed=5)])
    a, b = oracle.dequantize(t, w, k), ref.dequantize(t, w, k)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x = rng.uniform(-1, 1, (4, k)).astype(np.float32)
    assert np.array_equal(oracle.quantize_act(t, x, ACT_REF), ref.quantize_act(t, x, "ref"))
    assert np.array_equal(oracle.quantize_act(t, x, ACT_X86), ref.quantize_act(t, x, "cpu"))
    want, _ = ref.graph_mul_mat(t, w, k, x, n_threads=2)
    assert rel_rms(oracle.mul_mat(t, w, k, x, ACT_X86), want) < 2e-6


def test_synth_weights_are_sane(oracle):
    import ggml_hexagon_amd.synth as synth
    for t in ALL:
        w = synth.synth_weights(t, 8, 1024, seed=1)
        assert w.shape == (8, synth.row_size(t, 1024)) and synth.row_size(t, 1024) == oracle.row_size(t, 1024)
        d = oracle.dequantize(t, w, 1024)
        assert np.isfinite(d).all() and 0.002 < d.std() < 0.2


def test_q8_K_double_rounding_ties_live(oracle, ref):
    """the reference rounds iscale*x to f32 BEFORE the magic add (no FMA in its build): pin that on adversarial rows"""
    from helpers import q8_K_tie_rows
    x = q8_K_tie_rows(2, 512, seed=1)
    assert np.array_equal(oracle.quantize_act(Q4_K, x), ref.quantize_act(Q4_K, x, "cpu"))
