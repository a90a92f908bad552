"""One dispatch population for the PMC passes of pmc_traffic.sh: `passes` eager token-generation passes of the bench workload and
nothing else (no prompt pass, no roofline leg), every group issued by the closure bench.py times (HotPath.group_call).  Writes, per
kernel label (qmm_trace_begin / _end), how many launches the population holds and their algorithmic bytes, so that
pmc_traffic.py divides the counters by the bytes of exactly the dispatches it counted (the old passes mixed the timed
pass with a differently bucketed roofline leg).
    python3 profiles/tools/tg_population.py <out.json> [passes] [workload]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ggml_hexagon_amd import workload  # noqa: E402
from ggml_hexagon_amd.capi import Qmm  # noqa: E402
from ggml_hexagon_amd.hotpath import HotPath  # noqa: E402

out, passes = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
name = sys.argv[3] if len(sys.argv) > 3 else "llama3-8b-q4_k_m"
torch.cuda.set_device(0)
q = Qmm(0)
wl = workload.get(name)
hp = HotPath(q, wl, torch.device("cuda", 0))
io = hp.prepare(1)
pop = {}
calls = []
for grp in wl.groups:
    fn = hp.group_call(grp, io)
    labels = q.trace(fn)
    m0 = grp.mats[0]
    nbytes = sum(m.algo_bytes(1) for m in grp.mats) - (len(grp.mats) - 1) * m0.K * 4
    key = " + ".join(labels)
    p = pop.setdefault(key, {"launches": 0, "algo_bytes": 0})
    calls.append((fn, key, len(labels), nbytes))
torch.cuda.synchronize()
for _ in range(passes):
    for fn, key, nl, nbytes in calls:
        fn()
        pop[key]["launches"] += nl
        pop[key]["algo_bytes"] += nbytes
torch.cuda.synchronize()
# the tracing call above issued every group once more: it belongs to the population the profiler sees
for fn, key, nl, nbytes in calls:
    pop[key]["launches"] += nl
    pop[key]["algo_bytes"] += nbytes
json.dump({"workload": name, "passes": passes, "kernels": pop}, open(out, "w"), indent=1)
print(json.dumps(pop))
