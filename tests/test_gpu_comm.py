"""qmm_comm_*: the RCCL exchange of a row split driven from one process (include/ggml_mi355x_qmm.h).  RCCL wants one rank per
physical device, so the data-path test needs two GPUs and SKIPS on the one-GPU box; what runs everywhere: the argument checks, the
refusal of two ranks on one device, and the plugin falling back to peer copies when GGML_MI355X_RCCL=1 meets logical devices."""
import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = Path(__file__).resolve().parents[1]


def _lib():
    from ggml_hexagon_amd.capi import load_library as load
    lib = load()
    v = C.c_void_p
    lib.qmm_comm_create.restype = C.c_int
    lib.qmm_comm_create.argtypes = [C.POINTER(v), C.c_int, C.POINTER(v)]
    lib.qmm_comm_destroy.restype = None
    lib.qmm_comm_destroy.argtypes = [v]
    lib.qmm_comm_size.restype = C.c_int
    lib.qmm_comm_size.argtypes = [v]
    for f in (lib.qmm_comm_broadcast, lib.qmm_comm_gather, lib.qmm_comm_all_gather):
        f.restype = C.c_int
    lib.qmm_comm_broadcast.argtypes = [v, C.c_int, C.POINTER(v), C.c_size_t, C.POINTER(v)]
    lib.qmm_comm_gather.argtypes = [v, C.c_int, C.POINTER(v), C.POINTER(v), C.POINTER(C.c_size_t), C.POINTER(v)]
    lib.qmm_comm_all_gather.argtypes = [v, C.POINTER(v), C.POINTER(v), C.c_size_t, C.POINTER(v)]
    return lib


def test_two_ranks_on_one_device_are_refused():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ggml_hexagon_amd.capi import Qmm
    lib = _lib()
    a, b = Qmm(0), Qmm(0)
    try:
        ctxs = (C.c_void_p * 2)(a.ctx, b.ctx)
        out = C.c_void_p()
        assert lib.qmm_comm_create(ctxs, 2, C.byref(out)) == -1 and b"twice" in lib.qmm_last_error()
        assert lib.qmm_comm_create(ctxs, 1, C.byref(out)) == -1
        assert lib.qmm_comm_size(None) == 0
    finally:
        a.close()
        b.close()


def test_plugin_falls_back_to_peer_copies_on_logical_devices():
    exe = ROOT / "oracle" / "_ref" / "test-split-buffer"
    plugin = ROOT / "ggml-hexagon_amd" / "libggml-mi355x.so"
    if not exe.exists() or not plugin.exists():
        pytest.skip("oracle/_ref/test-split-buffer or the plugin module is not built (needs the reference tree at build time)")
    env = dict(os.environ, GGML_BACKEND_PATH=str(plugin), GGML_MI355X_VIRTUAL_DEVICES="2", GGML_MI355X_RCCL="1")
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=900, cwd=str(exe.parent))
    out = p.stdout + p.stderr
    assert "RCCL exchange unavailable" in out, out[-2000:]
    m = re.search(r"(\d+) OK, (\d+) FAILED", out)
    assert p.returncode == 0 and m and int(m.group(2)) == 0 and int(m.group(1)) >= 200, out[-3000:]


def test_broadcast_gather_all_gather_over_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two physical GPUs (RCCL: one rank per device)")
    from ggml_hexagon_amd.capi import Qmm
    lib = _lib()
    qs = [Qmm(0), Qmm(1)]
    comm = C.c_void_p()
    try:
        ctxs = (C.c_void_p * 2)(*[q.ctx for q in qs])
        assert lib.qmm_comm_create(ctxs, 2, C.byref(comm)) == 0, lib.qmm_last_error()
        assert lib.qmm_comm_size(comm) == 2
        n = 1 << 16
        x = [torch.arange(n, dtype=torch.float32, device=f"cuda:{i}") * (1 if i == 0 else 0) for i in range(2)]
        for t in x:
            torch.cuda.synchronize(t.device)
        bufs = (C.c_void_p * 2)(*[t.data_ptr() for t in x])
        assert lib.qmm_comm_broadcast(comm, 0, bufs, n * 4, None) == 0, lib.qmm_last_error()
        for q in qs:
            q.synchronize()
        assert torch.equal(x[1].cpu(), x[0].cpu())
        # gather: rank 1's slice lands in rank 0's buffer
        src = torch.full((n,), 7.0, device="cuda:1")
        dst = torch.zeros(n, device="cuda:0")
        torch.cuda.synchronize("cuda:0"); torch.cuda.synchronize("cuda:1")
        send = (C.c_void_p * 2)(None, src.data_ptr())
        recv = (C.c_void_p * 2)(None, dst.data_ptr())
        nb = (C.c_size_t * 2)(0, n * 4)
        assert lib.qmm_comm_gather(comm, 0, send, recv, nb, None) == 0, lib.qmm_last_error()
        for q in qs:
            q.synchronize()
        assert torch.equal(dst.cpu(), src.cpu())
        # all-gather
        parts = [torch.full((n,), float(i + 1), device=f"cuda:{i}") for i in range(2)]
        full = [torch.zeros(2 * n, device=f"cuda:{i}") for i in range(2)]
        torch.cuda.synchronize("cuda:0"); torch.cuda.synchronize("cuda:1")
        assert lib.qmm_comm_all_gather(comm, (C.c_void_p * 2)(*[t.data_ptr() for t in parts]), (C.c_void_p * 2)(*[t.data_ptr() for t in full]),
                                       n * 4, None) == 0, lib.qmm_last_error()
        for q in qs:
            q.synchronize()
        want = torch.cat([torch.full((n,), 1.0), torch.full((n,), 2.0)])
        assert all(torch.equal(f.cpu(), want) for f in full)
    finally:
        if comm:
            lib.qmm_comm_destroy(comm)
        for q in qs:
            q.close()
