"""Event-timed launches of the Q4_K prefill shapes of llama3-8b (512 tokens): python profiles/tools/shape_times.py [reps]
Prints us per call (prep included) and TFLOP/s for each shape; run under GGML_MI355X_R64 / _WIDE settings to compare kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd import synth
from ggml_hexagon_amd.capi import Qmm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
q = Qmm(0)
dev = torch.device("cuda", 0)
K, n = 4096, 512
shapes = [("wo 4096x4096", 4096, (4096,)), ("qkv 4096+1024+1024", 4096, (4096, 1024, 1024)), ("gate+up 2x14336", 4096, (14336, 14336)),
          ("gate 14336", 4096, (14336,)), ("down 4096x14336", 14336, (4096,))]
for name, k, ms in shapes:
    ws = [synth.synth_weights_torch(12, m, k, dev, seed=i) for i, m in enumerate(ms)]
    x = torch.rand((n, k), device=dev) * 2 - 1
    outs = [torch.empty((n, m), device=dev) for m in ms]
    def call():
        if len(ms) == 1:
            q.mul_mat(12, ws[0], k, x, out=outs[0])
        else:
            q.mul_mat_group([(12, w) for w in ws], k, x, outs)
    for _ in range(3):
        call()
    q.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * n * k * sum(ms)
    print(f"{name:24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
q.close()
