"""summarise the FETCH_SIZE / WRITE_SIZE passes of pmc_traffic.sh: per kernel (name as rocprofv3 prints it, template arguments
included), average corrected HBM bytes per launch beside the algorithmic bytes of THE SAME dispatches (tg_population.py wrote them);
the kernel with the most bytes goes to profiles/pmc_traffic.json, which bench.py copies into `roofline.traffic` when its own dominant
kernel and workload are the ones named there"""
import collections, csv, json, re, sys
fetch_csv, write_csv, pop_json, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]


def norm(name):                     # "void qmm::matvec_kernel<12, 1, false>(qmm::MatvecGroup, ...)" -> "matvec_kernel<12,1,false>"
    n = re.sub(r"\(.*", "", name).replace("void ", "").replace("qmm::", "").replace(" ", "")
    # matvec_kmix_kernel's third template argument (the Q8_0 fields) is ",q8_0" in the library's trace labels, absent when false
    return re.sub(r"(matvec_kmix_kernel<\d+,(?:true|false)),(true|false)>", lambda m: m.group(1) + (",q8_0>" if m.group(2) == "true" else ">"), n)


def load(path, counter):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and "qmm::" in row["Kernel_Name"]:
            d[norm(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return d


pop = json.load(open(pop_json))
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
rows = []
for key, fv in f.items():
    wv = w.get(key, [0.0])
    fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
    p = pop["kernels"].get(key)
    algo = p["algo_bytes"] / p["launches"] if p and p["launches"] == len(fv) else None     # only when the counts agree: same population
    hbm = int((2 * fa + wa) * 1024)
    rows.append((key, len(fv), round(fa, 1), round(wa, 1), hbm, int(algo) if algo else "", round(hbm / algo, 4) if algo else ""))
rows.sort(key=lambda r: -r[1] * r[4])
out = f"profiles/{tag}_pmc_hbm_traffic.csv"
with open(out, "w") as fh:
    fh.write("kernel,dispatches,FETCH_SIZE_avg_KiB_raw,WRITE_SIZE_avg_KiB,hbm_bytes_per_launch_corrected,algo_bytes_per_launch_same_dispatches,ratio\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.1f,%d,%s,%s\n' % r)
top = [r for r in rows if r[5] != ""]
if top:
    r = top[0]
    json.dump({"label": r[0], "workload": pop["workload"], "bytes_per_launch": r[4], "algo_bytes_per_launch": r[5], "ratio": r[6], "dispatches": r[1],
               "source": f"{out}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing flags) over `python3 profiles/tools/tg_population.py` "
                         f"({pop['passes']} eager token-generation passes + the tracing pass, nothing else); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 being the "
                         "gfx950 FETCH_SIZE correction for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section); algorithmic bytes summed over exactly these dispatches"},
              open("profiles/pmc_traffic.json", "w"), indent=1)
print(open(out).read())
