import csv, sys, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', row['Kernel_Name']).replace('void qmm::', '')[:50]
    if len(sys.argv) > 2 and sys.argv[2] not in name: continue
    d[name][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in d.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()})
