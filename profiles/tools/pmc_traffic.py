"""summarise the FETCH_SIZE / WRITE_SIZE passes of pmc_traffic.sh: per kernel (name as rocprofv3 prints it, template arguments
included), average corrected HBM bytes per launch; the dominant mat-vec kernel goes to profiles/pmc_traffic.json for bench.py"""
import collections, csv, json, re, sys
fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def load(path, counter):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        d[(name, row.get("Grid_Size", ""), row.get("Workgroup_Size", ""))].append(float(row["Counter_Value"]))
    return d


f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
rows = []
for key, fv in f.items():
    if "qmm::" not in key[0]:
        continue
    wv = w.get(key, [0.0])
    fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
    rows.append((key[0], key[1], key[2], len(fv), round(fa, 1), round(wa, 1), int((2 * fa + wa) * 1024)))
rows.sort(key=lambda r: -r[3] * r[6])
out = f"profiles/{tag}_pmc_hbm_traffic.csv"
with open(out, "w") as fh:
    fh.write("kernel,grid_size,workgroup_size,dispatches,FETCH_SIZE_avg_KiB_raw,WRITE_SIZE_avg_KiB,hbm_bytes_per_launch_corrected\n")
    for r in rows:
        fh.write('"%s",%s,%s,%d,%.1f,%.1f,%d\n' % r)
mv = [r for r in rows if "matvec_kernel<12, 1, false>" in r[0] and r[1] == "262144"]
if mv:
    r = mv[0]
    json.dump({"matvec_bytes_per_launch": r[6], "kernel": r[0], "workload": "llama3-8b-q4_k_m", "dispatches": r[3],
               "source": f"{out}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 --n-gen 8 "
                         "--no-e2e --no-graph`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 being the gfx950 FETCH_SIZE correction for "
                         "16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section); averaged over the kernel's full-grid dispatches"},
              open("profiles/pmc_traffic.json", "w"), indent=1)
print(open(out).read())
