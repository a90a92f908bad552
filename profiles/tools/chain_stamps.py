"""In-kernel phase stamps of one token-generation pass issued as persistent chains (qmm_chain_debug): where a step's time goes.
   python profiles/tools/chain_stamps.py [workload]      (MI355X box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggml_hexagon_amd import workload
from ggml_hexagon_amd.capi import Qmm
from ggml_hexagon_amd.hotpath import HotPath

q = Qmm(0)
dev = torch.device("cuda", 0)
wl = workload.get(sys.argv[1] if len(sys.argv) > 1 else "llama3-8b-q4_k_m")
hp = HotPath(q, wl, dev)
hp.prepare(1)
for _ in range(3):
    hp.run(1)
torch.cuda.synchronize()
nsteps = len(wl.groups)
cus = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.zeros((nsteps + 16, cus, 8), dtype=torch.int64, device=dev)
q.chain_debug(st)
hp.run(1)
torch.cuda.synchronize()
q.chain_debug(None)
s = st.cpu().numpy()[:nsteps].astype(np.float64) / 100.0          # us
t0 = s[:, :, 0].min()
names = ["start", "wait_done", "staged", "rows_done(all waves)", "published"]
print(f"{'step':>4} {'group':28s} {'MB':>7} | {'start(min)':>10} {'wait':>6} {'stage':>6} {'rows':>6} {'publish':>7} | {'step us':>7}  GB/s   (medians over workgroups; rows = staged -> last wave done, max over WGs in [])")
tot = 0.0
for i, g in enumerate(wl.groups[:nsteps]):
    a = s[i]
    mb = sum(m.weight_bytes for m in g.mats) / 1e6
    start = a[:, 0]
    wait = np.median(a[:, 1] - a[:, 0]); stage = np.median(a[:, 2] - a[:, 1]); rows = np.median(a[:, 3] - a[:, 2]); pub = np.median(a[:, 4] - a[:, 3])
    nxt = s[i + 1][:, 0].min() if i + 1 < nsteps and s[i + 1][:, 0].min() > 0 else a[:, 4].max()
    dur = nxt - start.min()
    if i < 9 or i >= nsteps - 2:
        print(f"{i:4d} {g.mats[0].name.split('.')[-1] + ' x' + str(len(g.mats)):28s} {mb:7.1f} | {start.min() - t0:10.2f} {wait:6.2f} {stage:6.2f} {rows:6.2f} [{(a[:, 3] - a[:, 2]).max():5.2f}] {pub:7.2f} | {dur:7.2f} {mb / dur * 1e3:6.0f}")
    tot += dur
print(f"sum of step spans {tot:.1f} us")
