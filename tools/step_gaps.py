"""From a rocprofv3 kernel trace (rocpd sqlite): per training step, the device-idle time and the kernels that run between the
end of the fused photometric backward's step and the next step's first kernel.  usage: python tools/step_gaps.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
marks = marks[len(marks) // 2:]
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    busy = sum(e - s for _, s, e in seg)
    wall = seg[-1][2] - seg[0][1] if len(seg) > 1 else 0
    gaps = sorted(((seg[k + 1][1] - seg[k][2]) / 1e3, seg[k][0][:50], seg[k + 1][0][:50]) for k in range(len(seg) - 1))[-4:]
    print("step: wall %.2f ms busy %.2f ms kernels %d; largest gaps (us): %s" % ((rows[b][1] - rows[a][1]) / 1e6, busy / 1e6, len(seg),
                                                                            "; ".join("%.0f after %s" % (g, n) for g, n, _ in gaps)))
names = {}
for n, s, e in rows[marks[-2]:marks[-1]]:
    if "nccl" in n.lower() or "rccl" in n.lower() or "copy" in n.lower() or "Memcpy" in n:
        names[n[:70]] = names.get(n[:70], 0) + (e - s) / 1e3
print({k: round(v, 1) for k, v in names.items()})
