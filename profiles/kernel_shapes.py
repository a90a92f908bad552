"""Per-(kernel, launch shape) duration summary of a rocprofv3 kernel trace:  python profiles/kernel_shapes.py <kernel_trace.csv> > shapes.csv"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if not name.startswith("qmm::"):
        continue
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
    rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_x_threads", "grid_y", "grid_z", "workgroup", "dispatches", "median_us", "min_us", "mean_us"])
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    w.writerow(list(k) + [len(v), round(v[len(v) // 2], 3), round(v[0], 3), round(sum(v) / len(v), 3)])
