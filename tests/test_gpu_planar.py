"""planar rows (qmm_repack_rows, type codes 102 / 108 / 114).  The repack is a permutation of each row's bytes, so
the bars are exact: wire -> planar -> wire is the identity on bytes, the unpack of a planar row has the bits of the wire row's
(which tests/golden pins against the reference), and every kernel gives bit-identical results on both layouts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle.pyoracle import Q4_0, Q4_K, Q6_K, Q8_0, TYPE_NAMES  # noqa: E402

TYPES = [(Q4_0, 4096), (Q8_0, 1024), (Q6_K, 2048), (Q6_K, 14336), (Q4_0, 11008)]
IDS = [f"{TYPE_NAMES[t]}-K{k}" for t, k in TYPES]


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


@pytest.mark.parametrize("t,k", TYPES, ids=IDS)
def test_roundtrip_and_unpack_bits(qmm, t, k):
    m = 77
    w = W(t, m, k, seed=k)
    assert qmm.planar_type(t, k, w.stride(0)) == t + 100
    wire = w.clone()
    deq = qmm.dequantize(t, w, k)
    tp = qmm.repack_rows(t, w, k, True)
    assert tp == t + 100 and not torch.equal(w, wire)
    assert torch.equal(qmm.dequantize(tp, w, k).view(torch.int32), deq.view(torch.int32))
    qmm.repack_rows(t, w, k, False)
    assert torch.equal(w, wire)


@pytest.mark.parametrize("t,k", TYPES, ids=IDS)
@pytest.mark.parametrize("n", [1, 3, 8, 40, 130, 512])
def test_every_kernel_is_bit_identical_on_planar_rows(qmm, t, k, n):
    m = 320 if n > 8 else 1000
    w = W(t, m, k, seed=k + n)
    x = torch.rand((n, k), device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)) * 2 - 1
    want = qmm.mul_mat(t, w, k, x)
    tp = qmm.repack_rows(t, w, k, True)
    got = qmm.mul_mat(tp, w, k, x)
    if t == Q4_0 and n <= 8:
        # round 3: the planar Q4_0 mat-vec unit holds two blocks per lane (MvUnit<T_Q4_0P>): the same exact block products, summed over
        # the lanes in another f32 order than the wire rows' one-block unit
        assert float((got - want).norm() / want.norm()) < 1e-6
    else:
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_mixed_group_with_planar_q6k(qmm):
    """q / k / v of a Q4_K_M more-bits layer with the Q6_K matrix planar: still ONE mixed-type launch, same bits"""
    k = 4096
    ws = [(Q4_K, W(Q4_K, 512, k, 1)), (Q4_K, W(Q4_K, 128, k, 2)), (Q6_K, W(Q6_K, 128, k, 3))]
    x = torch.rand((1, k), device="cuda") * 2 - 1
    outs = [torch.empty((1, w.shape[0]), device="cuda") for _, w in ws]
    qmm.mul_mat_group(ws, k, x, outs)
    want = [o.clone() for o in outs]
    tp = qmm.repack_rows(Q6_K, ws[2][1], k, True)
    qmm.mul_mat_group([ws[0], ws[1], (tp, ws[2][1])], k, x, outs)
    for o, w in zip(outs, want):
        assert torch.equal(o, w)


def test_mul_mat_id_on_planar_experts(qmm):
    k, m, n_expert, n_used = 2048, 256, 4, 2
    for t in (Q4_0, Q6_K):
        w = W(t, n_expert * m, k, seed=t).reshape(n_expert, m, -1)
        g = torch.Generator(device="cuda").manual_seed(t)
        for n_tokens in (1, 40):
            ids = torch.stack([torch.randperm(n_expert, device="cuda", generator=g) for _ in range(n_tokens)]).to(torch.int32)
            b = torch.rand((n_tokens, 1, k), device="cuda", generator=g) * 2 - 1
            want = qmm.mul_mat_id(t, w, k, b, ids[:, :n_used])
            wp = w.clone()
            tp = qmm.repack_rows(t, wp, k, True)
            got = qmm.mul_mat_id(tp, wp, k, b, ids[:, :n_used])
            if t == Q4_0 and n_tokens == 1:                     # (the mat-vec form: two blocks per lane on planar rows, see above)
                assert float((got - want).norm() / want.norm()) < 1e-6
            else:
                assert torch.equal(got, want)
    qmm.synchronize()


def test_shapes_without_a_planar_form(qmm):
    from ggml_hexagon_amd.capi import QmmError
    assert qmm.planar_type(Q6_K, 4096 + 256, 0) == 0 and qmm.planar_type(Q4_0, 4096 + 32, 0) == 0 and qmm.planar_type(Q4_K, 4096, 2304) == 0
    w = W(Q6_K, 8, 512, 1)
    with pytest.raises(QmmError):
        qmm.repack_rows(Q6_K, w, 512, True)
