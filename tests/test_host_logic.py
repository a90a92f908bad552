"""CPU-side tests: the C-ABI library loads and exports every declared symbol (no compute without a GPU), the product
path fails loudly without a device, the workload tables match SURVEY.md's byte/flop accounting, and the row-split
partition + concat (the N > 1 path) is correct under a world_size-2 gloo run."""
import ctypes
import os
import re
import socket
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_capi_exports_every_declared_symbol():
    from ggml_hexagon_amd import build, capi
    so = build.build_qmm()
    lib = ctypes.CDLL(str(so))
    for hdr, exports in (("ggml_mi355x_qmm.h", capi.EXPORTS), ("ggml_mi355x_ops.h", capi.OPS_EXPORTS)):
        header = (ROOT / "include" / hdr).read_text()
        declared = re.findall(r"QMM_API\s+[\w\s\*]+?\b(qmm_\w+)\s*\(", header)
        assert len(declared) >= (20 if hdr.endswith("qmm.h") else 6)
        for name in declared:
            assert hasattr(lib, name), f"{name} declared in include/{hdr} but not exported"
        assert set(declared) == set(exports)
    assert lib.qmm_abi_version() == 2            # 2: chains, timing events, six more weight formats, Q8_1 activations
    lib.qmm_row_size.restype = ctypes.c_size_t
    lib.qmm_row_size.argtypes = [ctypes.c_int, ctypes.c_int64]
    assert [lib.qmm_row_size(t, 4096) for t in (2, 8, 12, 13, 14)] == [2304, 4352, 2304, 2816, 3360]
    assert [lib.qmm_row_size(t, 4096) for t in (3, 6, 7, 10, 11, 20)] == [2560, 2816, 3072, 1344, 1760, 2304]     # SURVEY 8f-4 formats
    assert lib.qmm_row_size(12, 100) == 0 and lib.qmm_row_size(0, 4096) == 0


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    from ggml_hexagon_amd.capi import Qmm, QmmError
    with pytest.raises(QmmError):
        Qmm(0)


def test_plugin_module_exports_the_ggml_entry_points():
    so = ROOT / "ggml-hexagon_amd" / "libggml-mi355x.so"
    if not so.exists():
        pytest.skip("plugin not built (needs the ggml headers of the host tree)")
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    for name in ("ggml_backend_init", "ggml_backend_score", "ggml_backend_mi355x_reg", "ggml_backend_mi355x_init",
                 "ggml_backend_is_mi355x", "ggml_backend_mi355x_get_device_count", "ggml_backend_mi355x_get_devname",
                 "ggml_backend_mi355x_buffer_type"):
        assert re.search(rf"\bT {name}\b", syms), name


def test_workload_accounting_matches_survey():
    from ggml_hexagon_amd import workload
    w = workload.get("synth-7b-q4_k")
    # 6,607,077,376 matmul weights x 0.5625 B = 3,716,481,024 B/token; pp512 = 6.766e12 flop
    assert w.weight_bytes() == 3_716_481_024
    assert w.flops(512) == 2 * 6_607_077_376 * 512
    assert len(w.all_mats()) == 32 * 7 + 1
    l3 = workload.get("llama3-8b-q4_k_m")
    types = {m.name: m.type for m in l3.all_mats()}
    assert types["blk.0.attn_v"] == 14 and types["blk.0.ffn_down"] == 14          # use_more_bits layers -> Q6_K
    assert types["blk.4.attn_v"] == 12 and types["blk.4.ffn_down"] == 12
    assert types["output"] == 14 and types["blk.7.attn_q"] == 12
    mx = workload.get("mixtral-8x7b-q4_k_m")
    ids = [m for m in mx.all_mats() if m.n_expert]
    assert len(ids) == 32 * 3 and ids[0].n_expert == 8 and ids[0].n_used == 2
    assert {m.type for m in mx.all_mats() if m.name.endswith("attn_k")} == {8}     # n_expert == 8: attn_k/v -> Q8_0
    l70 = workload.get("llama3-70b-q4_k_m")
    assert 13 in {m.type for m in l70.all_mats() if m.name.endswith("attn_v")}      # 70B rule: Q5_K


def test_row_ranges_follow_ggml_row_split():
    from ggml_hexagon_amd.rowsplit import ROW_ROUNDING, all_ranges, rounding_for, row_range
    assert ROW_ROUNDING == 256                               # the prefill kernels' row tile = the plugin's SPLIT_ROW_ROUNDING
    for m in (4096, 14336, 28672, 128256, 1024, 100):
        for world in (1, 2, 4, 8):
            rs = all_ranges(m, world)
            rnd = rounding_for(m, world)
            assert rs[0][0] == 0 and rs[-1][1] == m
            for (lo, hi), (lo2, _) in zip(rs, rs[1:]):
                assert hi == lo2 and lo % rnd == 0
            if m // world >= 32:                             # no rank left without rows where 32-row slices exist (1024 rows on 8 ranks)
                assert all(hi > lo for lo, hi in rs), (m, world, rs)
    assert rounding_for(1024, 8) == 128 and rounding_for(8192, 8) == 256 and rounding_for(1024, 2) == 256
    assert all_ranges(1024, 8) == [(128 * i, 128 * i + 128) for i in range(8)]
    # cumulative fractions rounded DOWN to the rounding, the last device takes the remainder (ggml-cuda.cu:740-753)
    assert row_range(128256, 7, 8) == (112128, 128256)
    assert row_range(128256, 7, 8, rounding=64) == (112192, 128256)
    assert all_ranges(100, 4, rounding=64) == [(0, 0), (0, 0), (0, 64), (64, 100)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rowsplit_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ggml_hexagon_amd import rowsplit, synth
        from oracle.pyoracle import Oracle
        o = Oracle()
        cc = rowsplit.RowConcat()
        ok = True
        for (t, m, k, n) in ((12, 256, 512, 3), (14, 200, 256, 1), (2, 192, 64, 5)):
            w = synth.synth_weights(t, m, k, seed=3, sigma=0.2)                 # every rank builds the full matrix ...
            x = np.random.default_rng(1).uniform(-1, 1, (n, k)).astype(np.float32)
            ranges = rowsplit.all_ranges(m, world)
            lo, hi = ranges[rank]
            local = o.mul_mat(t, w[lo:hi], k, x) if hi > lo else np.zeros((n, 0), np.float32)   # ... computes only its rows
            full = cc.concat(torch.from_numpy(np.ascontiguousarray(local)), ranges).numpy()
            ok &= np.array_equal(full, o.mul_mat(t, w, k, x))                   # concat == single-device result, bit for bit
        q.put((rank, bool(ok)))
    except Exception as e:                                                      # report instead of hanging the parent
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


def test_rowsplit_concat_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rowsplit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


# ---- the bench driver's multi-GPU path (HotPath + RowConcat) on CPU tensors, world_size 2, gloo, with the kernels replaced
#      by the oracle (allowed here: tests may use the oracle as the checker/stand-in; the product path never does)

class _OracleQmm:
    """same method surface as capi.Qmm, computing with oracle/qmm_oracle.c on CPU tensors"""

    def __init__(self):
        from oracle.pyoracle import Oracle
        self.o = Oracle()

    def planar_type(self, t, k, row_bytes):      # the CPU stand-in keeps GGUF wire layout (the planar repack is a device matter)
        return 0

    def mul_mat_group(self, weights, k, x, outs):
        import torch
        for (t, w), out in zip(weights, outs):
            out.copy_(torch.from_numpy(self.o.mul_mat(t, w.numpy(), k, x.numpy())))
        return outs


def _hotpath_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ggml_hexagon_amd import rowsplit, synth, workload
        from ggml_hexagon_amd.hotpath import HotPath
        # a 2-layer toy model with the real per-layer structure (grouped q/k/v with mixed types, gate/up, down, output)
        wl = workload._llama("toy", 2, 256, 512, 4, 2, 1000, "q4_k_m")
        orig = synth.synth_weights_torch
        synth.synth_weights_torch = lambda t, rows, k, device, seed=0, sigma=0.02: torch.from_numpy(synth.synth_weights(t, rows, k, seed=seed, sigma=0.2))
        try:
            qm = _OracleQmm()
            hp = HotPath(qm, wl, torch.device("cpu"), rank, world, rowsplit.RowConcat(), seed=5)
            for n, n_out in ((1, None), (3, None), (3, 1)):      # (3, 1): a prompt batch that wants one row of logits
                if n_out is not None:
                    for t in hp.prepare(n_out)[2].values():      # stale results of the n = 1 pass must not satisfy the check
                        t.zero_()
                hp.run(n, n_out)
                # every full dst must equal the single-device product of the concatenated shards
                assert any(isinstance(k, tuple) and k and k[0] == "group" for k in hp.prepare(n)[1]), "grouped exchange not in use"
                for grp in wl.groups[-5:]:                       # the last layer (q/k/v and gate/up travel as ONE all-gather each) + the output projection (ragged split)
                    nn = n_out if (n_out is not None and grp.outputs_only) else n     # the last layer's FFN and the output projection ran at n_outputs
                    x, dst_local, dst_full, _ = hp.prepare(nn)
                    for m in grp.mats:
                        w_local, ranges = hp.weights[m.name]
                        gathered = [None] * world
                        dist.all_gather_object(gathered, w_local.numpy())
                        w_full = np.concatenate([g for g in gathered if g.shape[0] > 0])
                        want = qm.o.mul_mat(m.type, w_full, m.K, x[m.K].numpy())
                        got = dst_full[(m.name.split(".")[-1], w_local.shape[0])].numpy()
                        assert got.shape == want.shape and np.array_equal(got, want), (m.name, n, n_out)
            q.put((rank, True))
        finally:
            synth.synth_weights_torch = orig
    except Exception as e:
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()[-800:]}"))
    finally:
        dist.destroy_process_group()


def test_hotpath_rowsplit_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hotpath_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res


def test_prompt_pass_counts_output_rows_only():
    """llama-bench's prompt test wants one row of logits: the last layer's FFN and the output projection run at
    n_outputs = 1 (workload.Group.outputs_only), everything else at n_prompt"""
    from ggml_hexagon_amd import workload
    wl = workload.get("llama3-8b-q4_k_m")
    only = [g for g in wl.groups if g.outputs_only]
    assert [m.name for g in only for m in g.mats] == ["blk.31.ffn_gate", "blk.31.ffn_up", "blk.31.ffn_down", "output"]
    full, bench = wl.flops(512), wl.flops(512, 1)
    saved = sum(m.flops(512) - m.flops(1) for g in only for m in g.mats)
    assert full - bench == saved and 0.05 < saved / full < 0.12
    assert wl.flops(1, 1) == wl.flops(1) and wl.algo_bytes(1, 1) == wl.algo_bytes(1)          # token generation is unchanged


def test_glue_op_surface_predicates():
    """qmm_op_supported is pure host logic (no device needed): the surface the plugin's supports_op reports.  Shapes follow
    llama.cpp's layer graph (src/llama-model.cpp:4191-4350, src/llama-graph.cpp:1126-1213)."""
    from ggml_hexagon_amd import build, capi
    from ggml_hexagon_amd.capi import QmmTensor as T
    lib = ctypes.CDLL(str(build.build_qmm()))
    P = ctypes.POINTER(capi.QmmTensor)
    lib.qmm_op_supported.argtypes = [ctypes.c_int, P, P, P, P]
    lib.qmm_attn_decode_supported.argtypes = [P, P, P, P, P]
    lib.qmm_op_add_rms_norm_supported.argtypes = [P, P, P, P, P]

    def sup(op, a, b, c, d):
        r = lambda t: ctypes.byref(t) if t is not None else None
        return lib.qmm_op_supported(op, r(a), r(b), r(c), r(d))

    F32, F16, I32, BF16, Q4K = 0, 1, 26, 30, 12
    x = T.make(F32, [4096, 512])
    w = T.make(F32, [4096])
    assert sup(capi.OP_RMS_NORM, x, None, None, x) == 1
    assert sup(capi.OP_RMS_NORM_MUL, x, w, None, x) == 1
    assert sup(capi.OP_RMS_NORM, T.make(F16, [4096, 512]), None, None, T.make(F16, [4096, 512])) == 0
    assert sup(capi.OP_ADD, x, x, None, x) == 1 and sup(capi.OP_MUL, x, w, None, x) == 1          # broadcast row
    assert sup(capi.OP_ADD, x, T.make(F32, [4095]), None, x) == 0                                   # not repeatable
    assert sup(capi.OP_SILU_MUL, x, x, None, x) == 1
    # rope: normal and neox yes, m-rope / vision no; positions must be i32 of ne2 entries
    q = T.make(F32, [128, 32, 512])
    pos = T.make(I32, [512])
    rope = lambda mode: T.make(F32, [128, 32, 512], op_params=[0, 128, mode, 0, 8192])
    assert sup(capi.OP_ROPE, q, pos, None, rope(0)) == 1 and sup(capi.OP_ROPE, q, pos, None, rope(2)) == 1
    assert sup(capi.OP_ROPE, q, pos, None, rope(8)) == 0 and sup(capi.OP_ROPE, q, pos, None, rope(24)) == 0
    assert sup(capi.OP_ROPE, q, T.make(I32, [511]), None, rope(0)) == 0
    # soft_max with an f32 or f16 mask of at least ne01 rows
    kq = T.make(F32, [640, 512, 32])
    assert sup(capi.OP_SOFT_MAX, kq, T.make(F32, [640, 512]), None, kq) == 1
    assert sup(capi.OP_SOFT_MAX, kq, T.make(F16, [640, 512]), None, kq) == 1
    assert sup(capi.OP_SOFT_MAX, kq, T.make(F32, [640, 256]), None, kq) == 0
    # cpy: f32 <-> f16 any strides, no bf16 / quantized destinations
    assert sup(capi.OP_CPY, x, None, None, T.make(F16, [4096, 512])) == 1
    assert sup(capi.OP_CPY, x, None, None, T.make(BF16, [4096, 512])) == 0
    assert sup(capi.OP_CPY, x, None, None, T.make(Q4K, [4096, 512])) == 0
    # get_rows from a quantized embedding table
    emb = T.make(Q4K, [4096, 32000], nb=[144, 2304, 2304 * 32000, 2304 * 32000])
    assert sup(capi.OP_GET_ROWS, emb, T.make(I32, [512]), None, x) == 1
    # KQ: K is a strided view of the f16 cache, Q a permuted f32 tensor; grouped-query broadcast 32 / 8
    k = T.make(F16, [128, 640, 8], nb=[2, 2048, 256, 2048 * 640])
    qp = T.make(F32, [128, 512, 32], nb=[4, 16384, 512, 16384 * 512])
    assert sup(capi.OP_MUL_MAT_F, k, qp, None, T.make(F32, [640, 512, 32])) == 1
    assert sup(capi.OP_MUL_MAT_F, T.make(F16, [128, 640, 8], nb=[2048, 2, 256, 1]), qp, None, T.make(F32, [640, 512, 32])) == 0   # K not dense along D
    # the fused forms
    r = ctypes.byref
    assert lib.qmm_op_add_rms_norm_supported(r(x), r(x), r(w), r(x), r(x)) == 1
    assert lib.qmm_op_add_rms_norm_supported(r(x), r(w), r(w), r(x), r(x)) == 0                   # broadcast add is not the residual add
    q1 = T.make(F32, [128, 1, 32], nb=[4, 16384, 512, 16384])
    v = T.make(F16, [640, 128, 8], nb=[2, 1280, 1280 * 128, 1280 * 1024])
    m1 = T.make(F32, [640, 64])
    out = T.make(F32, [4096, 1])
    assert lib.qmm_attn_decode_supported(r(q1), r(k), r(v), r(m1), r(out)) == 1
    assert lib.qmm_attn_decode_supported(r(qp), r(k), r(v), r(T.make(F32, [640, 512])), r(T.make(F32, [4096, 512]))) == 0      # prefill batch
