"""Event-timed token-generation launches of the llama3-8b shapes, plain against the forms a graph folds in (qmm_mul_mat_group_ex: RMS norm of
the input, residual add, SwiGLU of gate / up), on rotating weight sets:  python profiles/tools/ex_times.py   (MI355X box)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd import synth
from ggml_hexagon_amd.capi import Qmm

q = Qmm(0)
dev = torch.device("cuda", 0)
SETS, REPS = 8, 40


def timed(fn):
    """the launches replayed from a hipGraph (an eager Python call costs ~10 us: more than the small launches themselves)"""
    for s in range(SETS):
        fn(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(REPS * SETS):
                fn(i % SETS)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (REPS * SETS)


def case(name, k, ms):
    ws = [[synth.synth_weights_torch(12, m, k, dev, seed=17 * s + i) for i, m in enumerate(ms)] for s in range(SETS)]
    x = torch.rand((1, k), device=dev) * 2 - 1
    nw = torch.rand((k,), device=dev) + 0.5
    outs = [torch.empty((1, m), device=dev) for m in ms]
    res = [torch.rand((1, m), device=dev) for m in ms]
    grp = lambda s: [(12, w) for w in ws[s]]
    row = [name, timed(lambda s: q.mul_mat_group(grp(s), k, x, outs))]
    row.append(timed(lambda s: q.mul_mat_group_ex(grp(s), k, x, outs, norm_w=nw, eps=1e-5)))
    row.append(timed(lambda s: q.mul_mat_group_ex(grp(s), k, x, outs, residuals=res)))
    if len(ms) == 2 and ms[0] == ms[1]:
        row.append(timed(lambda s: q.mul_mat_group_ex(grp(s), k, x, outs[:1] + outs[1:], swiglu=1)))
        row.append(timed(lambda s: q.mul_mat_group_ex(grp(s), k, x, outs, norm_w=nw, eps=1e-5, swiglu=1)))
    print("%-22s plain %6.2f | + norm %6.2f | + residual %6.2f" % tuple(row[:4]) + (" | swiglu %6.2f | norm + swiglu %6.2f" % tuple(row[4:]) if len(row) > 4 else ""))


case("qkv 4096+1024+1024", 4096, (4096, 1024, 1024))
case("wo 4096", 4096, (4096,))
case("gate+up 2x14336", 4096, (14336, 14336))
case("down 14336->4096", 14336, (4096,))
q.close()
