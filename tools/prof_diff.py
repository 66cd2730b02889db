"""Per-kernel time difference between two rocprofv3 kernel traces of bench.py (steady-state steps).
usage: python tools/prof_diff.py <a.db> <b.db>"""
import collections
import sqlite3
import sys


def load(db):
    rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    lo, hi = marks[len(marks) // 2], marks[-1]
    n = len(marks) - 1 - len(marks) // 2
    tot, cnt = collections.Counter(), collections.Counter()
    for name, s, e in rows[lo:hi]:
        tot[name[:90]] += (e - s) / 1e3 / n
        cnt[name[:90]] += 1.0 / n
    return tot, cnt, (rows[hi][1] - rows[lo][1]) / 1e6 / n


ta, ca, wa = load(sys.argv[1])
tb, cb, wb = load(sys.argv[2])
print("wall ms/step: a %.3f b %.3f; busy a %.3f b %.3f" % (wa, wb, sum(ta.values()) / 1e3, sum(tb.values()) / 1e3))
d = sorted(((ta[k] - tb[k], k) for k in set(ta) | set(tb)), key=lambda t: -abs(t[0]))
for diff, k in d[:25]:
    print("%+8.1f us  a %7.1f (x%.1f)  b %7.1f (x%.1f)  %s" % (diff, ta[k], ca[k], tb[k], cb[k], k))
