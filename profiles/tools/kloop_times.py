"""Fixed cost vs K-loop cost of the Q4_K prefill kernels: 28672 rows x 512 tokens at K = 4096, 8192, 16384 (prep included).
T(K) = a + b K: b is the K loop (us per 64-deep K-step), a the prologue + epilogue + launch.  Run under GGML_MI355X_R64 / _WIDE."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd import synth
from ggml_hexagon_amd.capi import Qmm

q = Qmm(0)
dev = torch.device("cuda", 0)
M, n, reps = 28672, 512, 20
res = []
for k in (4096, 8192, 16384):
    w = synth.synth_weights_torch(12, M, k, dev, seed=k)
    x = torch.rand((n, k), device=dev) * 2 - 1
    out = torch.empty((n, M), device=dev)
    for _ in range(3):
        q.mul_mat(12, w, k, x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        q.mul_mat(12, w, k, x, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    res.append((k, us))
    print(f"K={k:6d} {us:8.1f} us  {2.0 * n * k * M / us / 1e6:7.1f} TF/s")
    del w
(k0, t0), (k2, t2) = res[0], res[-1]
b = (t2 - t0) / ((k2 - k0) / 64)
print(f"per K-step {b:.3f} us ({2.0 * n * 64 * M / b / 1e6:.0f} TF/s in the loop), fixed {t0 - b * k0 / 64:.1f} us")
q.close()
