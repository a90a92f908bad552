"""RCCL on the GPU box: the row split's exchange (rowsplit.RowConcat over torch.distributed's nccl backend = RCCL) has one GPU to
run on here, so the process group is a world of ONE and RowConcat is told to go through the backend anyway.  What this pins:
RCCL loads and runs in this image, the grouped all-gather + placement copies produce the single-device result, and the whole
token-generation pass (launches + collectives) can be captured in a hipGraph and replayed, which is how bench.py runs tg at N > 1.
The N = 2 arithmetic of the same code is covered on CPU tensors with gloo (tests/test_host_logic.py)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def nccl_world_of_one():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_grouped_all_gather_and_graph_capture(nccl_world_of_one):
    from ggml_hexagon_amd import rowsplit, workload
    from ggml_hexagon_amd.capi import Qmm
    from ggml_hexagon_amd.hotpath import HotPath
    dev = torch.device("cuda", 0)
    q = Qmm(0)
    try:
        wl = workload._llama("toy", 2, 512, 1024, 8, 2, 2048, "q4_k_m")
        plain = HotPath(q, wl, dev, 0, 1, None, seed=5)
        split = HotPath(q, wl, dev, 0, 1, rowsplit.RowConcat(always_collective=True), seed=5)
        assert split.split and not plain.split
        for n in (1, 64):
            plain.run(n)
            split.run(n)
            torch.cuda.synchronize()
            _, loc_p, _, _ = plain.prepare(n)
            _, loc_s, full_s, _ = split.prepare(n)
            assert any(isinstance(k, tuple) and k[0] == "group" for k in loc_s)
            checked = 0
            for key, t in full_s.items():                 # every gathered dst equals the plain pass's local dst (same seeds, same kernels)
                assert torch.equal(t, loc_p[key]), key
                checked += 1
            assert checked >= 6
        # the token-generation pass with its collectives as one hipGraph
        g = split.capture(1)
        _, _, full_s, _ = split.prepare(1)
        want = {k: v.clone() for k, v in full_s.items()}
        for v in full_s.values():
            v.zero_()
        g.replay()
        torch.cuda.synchronize()
        for k, v in full_s.items():
            assert torch.equal(v, want[k]), k
    finally:
        q.close()
