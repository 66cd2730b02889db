"""Time span and composition of the launches between two kernels of one replayed step.
usage: python tools/span.py <results.db> <first-kernel-substring> <last-kernel-substring>"""
import collections
import sqlite3
import sys

rows = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels order by start").fetchall()
a, b = sys.argv[2], sys.argv[3]
ia = [i for i, r in enumerate(rows) if a in r[0]]
i0 = ia[len(ia) * 3 // 4]
i1 = next(i for i in range(i0 + 1, len(rows)) if b in rows[i][0])
seg = rows[i0:i1 + 1]
busy = sum(e - s for _, s, e in seg) / 1e3
print("%d launches, span %.1f us, kernel time %.1f us" % (len(seg), (seg[-1][2] - seg[0][1]) / 1e3, busy))
tot, cnt = collections.Counter(), collections.Counter()
for n, s, e in seg:
    tot[n[:70]] += (e - s) / 1e3
    cnt[n[:70]] += 1
for k, v in tot.most_common(12):
    print("  %7.1f us x%-3d %s" % (v, cnt[k], k))
