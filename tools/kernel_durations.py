"""Average GPU duration per (kernel name, grid) from a rocprofv3 rocpd database (dev tool):
python tools/kernel_durations.py <results.db> [substring]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = "select name, start, end%s from kernels order by start" % ((", " + gx) if gx else "")
acc = collections.OrderedDict()
for row in c.execute(q):
    name, s, e = row[0], row[1], row[2]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    key = (name[:90], row[3] if gx else 0)
    acc.setdefault(key, []).append((e - s) / 1e3)
for (name, grid), v in acc.items():
    v = v[3:] if len(v) > 6 else v                       # drop the warm-up launches
    print("%8.1f us  n=%3d  grid %-8s %s" % (sum(v) / len(v), len(v), grid, name))
