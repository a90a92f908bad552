"""A few launches of one prefill shape (default: 14336 x 4096 Q4_K at 512 tokens) for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd import synth
from ggml_hexagon_amd.capi import Qmm

t, M, K, n = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (12, 14336, 4096, 512)))
q = Qmm(0)
dev = torch.device("cuda", 0)
ws = [synth.synth_weights_torch(t, M, K, dev, seed=i) for i in range(3)]
x = torch.rand((n, K), device=dev) * 2 - 1
out = torch.empty((n, M), device=dev)
for _ in range(2):
    for w in ws:
        q.mul_mat(t, w, K, x, out=out)
torch.cuda.synchronize()
