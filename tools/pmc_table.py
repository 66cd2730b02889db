"""Per-kernel averages of rocprofv3 --pmc passes (csv counter_collection files).  usage: python tools/pmc_table.py <dir> [name-substring ...]"""
import collections
import csv
import glob
import sys

root, keys = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if keys and not any(s in k for s in keys):
            continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("##", k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]
        print("  %-28s %12.4g   (n=%d)" % (c, sum(v) / len(v), len(v)))
