"""per-kernel averages of the counters in a rocprofv3 counter_collection.csv: python tools/pmc_kernels.py file.csv [name filter]"""
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if flt and flt not in k:
        continue
    a = acc[k][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.1f}  (n={n})")
