"""Turn a rocprofv3 (rocpd sqlite) kernel trace of bench.py into a markdown per-step summary.
usage: python tools/prof_summary.py <results.db> <out.md> "<title>" [<once-per-step kernel substring> "<workload line>"] """
import collections
import sqlite3
import sys

db, out, title = sys.argv[1:4]
marker = sys.argv[4] if len(sys.argv) > 4 else "photo_tile_kernel<1"
workload = sys.argv[5] if len(sys.argv) > 5 else "config B (ResNet-50, B=12, 192x640, fp32, 1 x MI355X)"
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
pf = [i for i, r in enumerate(rows) if marker in r[0]]
nsteps_total = next((k for k in range(1, len(pf)) if pf[k] - pf[k - 1] < 5), len(pf))   # train steps precede the roofline loop
lo, hi = min(3, nsteps_total - 2), nsteps_total - 1
s0, s1 = rows[pf[lo]][1], rows[pf[hi]][1]
n = hi - lo
sel = [r for r in rows if s0 <= r[1] < s1]
tot, cnt = collections.Counter(), collections.Counter()
for name, s, e in sel:
    tot[name] += e - s
    cnt[name] += 1
T = sum(tot.values())
with open(out, "w") as f:
    f.write("# %s\n\n" % title)
    f.write("Steady-state window = %d train steps (after warm-up), %s.\n\n" % (n, workload))
    f.write("wall %.2f ms/step under the profiler, GPU busy %.2f ms/step, %d kernel launches/step\n\n" % ((s1 - s0) / n / 1e6, T / n / 1e6, len(sel) // n))
    f.write("| us/step | calls/step | avg us | kernel |\n|---:|---:|---:|---|\n")
    for name, t in tot.most_common(160):
        f.write("| %.1f | %.1f | %.1f | `%s` |\n" % (t / n / 1e3, cnt[name] / n, t / cnt[name] / 1e3, name[:120].replace("|", "/")))
    for key in ("photo_tile_kernel<1", "photo_tile_kernel<0", "photo_tile_kernel<2", "photo_bwd_tile_kernel", "sql_fwd_kernel", "sql_bwd"):
        pk = [(e - s) / 1e3 for name, s, e in rows if key in name]
        if pk:
            f.write("\n`%s`: %d dispatches in the whole run, avg %.2f us, min %.2f us\n" % (key, len(pk), sum(pk) / len(pk), min(pk)))
print(open(out).read()[:600])
