#!/usr/bin/env python
"""GPU busy time of a rocprofv3 kernel trace: python tools/busy.py <dir with *.db>
Prints per-kernel totals and the idle gaps between consecutive kernels (one stream)."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
busy = sum(e - s for _, s, e in rows)
gaps = [(rows[i + 1][1] - rows[i][2], rows[i][0], rows[i + 1][0]) for i in range(len(rows) - 1)]
print("kernels %d  span %.1f ms  busy %.1f ms (%.1f%%)" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
tot = {}
for n, s, e in rows:
    k = n.split("(")[0].replace("void kh::", "").replace("kh::", "")
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-46s calls=%6d  total=%9.2f ms  avg=%8.2f us" % (k[:46], c, ms, ms / c * 1e3))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us): " + ", ".join("%.0f" % (g[0] / 1e3) for g in big))
gt = {}
for g, a, b in gaps:
    if g > 0:
        k = a.split("(")[0].replace("void kh::", "")[:28] + " -> " + b.split("(")[0].replace("void kh::", "")[:28]
        x = gt.setdefault(k, [0, 0.0])
        x[0] += 1
        x[1] += g / 1e6
for k, (c, ms) in sorted(gt.items(), key=lambda kv: -kv[1][1])[:10]:
    print("  gap %-62s n=%5d total=%8.2f ms avg=%7.1f us" % (k, c, ms, ms / c * 1e3))
