#!/usr/bin/env python
"""Where the wall time of a bench leg goes, from a rocprofv3 kernel trace (the *.db of `rocprofv3 --kernel-trace`):
python tools/timeline.py <trace.db> [window_ms_from_the_end]
Prints, for the last `window` ms of the trace (default 300): busy time per kernel, the idle time in front of each kernel
name, and every idle gap above 100 us with the kernels on both sides - the tool behind DESIGN's account of config 5's
slab (set-up 23 ms, Arnoldi loop 241 ms, finalisation 3 ms of a 266 ms solve)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0


def short(n):
    return n.split("(")[0].replace("void kh::", "").replace("kh::", "")[:40]


t0, tend = rows[0][1], rows[-1][2]
sel = [(short(n), s, e) for n, s, e in rows if s > tend - win * 1e6]
print("trace span %.1f ms, window: the last %.1f ms, %d kernels" % ((tend - t0) / 1e6, win, len(sel)))
tot, prev, gaps = {}, None, []
for k, s, e in sel:
    g = (s - prev) / 1e3 if prev else 0.0
    a = tot.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += max(g, 0.0)
    if g > 100.0:
        gaps.append(((s - t0) / 1e6, g, k))
    prev = e
busy = sum(a[1] for a in tot.values())
print("busy %.2f ms of %.2f ms (%.1f %%)" % (busy / 1e3, (sel[-1][2] - sel[0][1]) / 1e6, 100.0 * busy * 1e3 / (sel[-1][2] - sel[0][1])))
for k, (c, b, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s n=%5d busy %10.1f us (avg %8.1f)  idle in front %9.1f us" % (k, c, b, b / c, g))
print("idle gaps above 100 us:")
for t, g, k in gaps:
    print("  at %9.2f ms: %8.1f us before %s" % (t, g, k))
