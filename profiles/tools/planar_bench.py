"""wire vs planar rows on single launches over rotating weights (defeats the Infinity Cache); MI355X box
   python profiles/tools/planar_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd.capi import Qmm
from ggml_hexagon_amd import synth
q = Qmm(0)
dev = torch.device("cuda", 0)
shapes = [("wo q4_0", 2, 4096, 4096), ("gate/up q4_0", 2, 22016, 4096), ("down q4_0", 2, 4096, 11008), ("down q6_K", 14, 4096, 14336),
          ("out q6_K", 14, 128256, 4096), ("k/v q8_0", 8, 2048, 4096), ("out q6_K 32000", 14, 32000, 4096)]
print(f"{'shape':16s} {'N':>4s} {'MB':>7s} | {'wire us':>8s} {'GB/s':>6s} | {'planar us':>9s} {'GB/s':>6s} | planar/wire time")
for name, t, m, k in shapes:
    per = synth.row_size(t, k) * m
    copies = max(2, min(48, int(700e6 // per) + 1))
    for n in (1, 512):
        if n == 512 and m > 40000:
            continue
        res = []
        for planar in (False, True):
            ws = [synth.synth_weights_torch(t, m, k, dev, seed=i) for i in range(copies)]
            tt = t
            if planar:
                for w in ws:
                    tt = q.repack_rows(t, w, k, True)
            x = torch.rand((n, k), device=dev) * 2 - 1
            out = torch.empty((n, m), device=dev)
            def run():
                for w in ws:
                    q.mul_mat(tt, w, k, x, out=out)
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / (5 * copies))
            del ws, g
            torch.cuda.empty_cache()
        print(f"{name:16s} {n:4d} {per / 1e6:7.1f} | {res[0]:8.2f} {per / res[0] / 1e3:6.0f} | {res[1]:9.2f} {per / res[1] / 1e3:6.0f} | {res[1] / res[0]:.3f}", flush=True)
