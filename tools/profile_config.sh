#!/bin/bash
# Kernel trace + fabric-traffic counters of one secondary configuration:
#   tools/profile_config.sh <3|4|5|5s|band|ragged>  -> gpurun_out/prof_cfg<N>/summary.md
# (rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes with --kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled - its gfx950 correction for wide streams)
set -u
CFG=${1:-3}
export TMPDIR=/tmp
OUT=gpurun_out/prof_cfg$CFG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_configs.py $CFG > $OUT/bench_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc -- python tools/bench_configs.py $CFG > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
done
python - <<PY
import glob, json, sqlite3
line = [l for l in open("$OUT/bench_trace.json").read().strip().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
db = sorted(glob.glob("$OUT/trace/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average from top_kernels order by total_duration desc limit 14").fetchall()
tot = sum(r[2] for r in con.execute("select name, total_calls, total_duration, average from top_kernels").fetchall())
out = ["# rocprofv3 kernel trace + fabric traffic: config $CFG", "",
       "Command: \`rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py $CFG\` on one MI355X, then \`--pmc FETCH_SIZE\` and "
       "\`--pmc WRITE_SIZE\` passes of the same command with \`--kernel-trace\` only (tools/profile_config.sh).", "",
       "Result under the profiler: " + ", ".join("%s = %s" % (k, (round(v, 3) if isinstance(v, float) else v)) for k, v in d.items() if k != "solve_ms"), "",
       "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---:|---:|---:|---:|"]
for name, calls, total, avg in rows:
    out.append("| \`%s\` | %d | %.2f | %.2f | %.1f |" % (name[:70], calls, total / 1e3, avg, 100.0 * total / tot))     # (the stats view is in us)
pm = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = sorted(glob.glob("$OUT/pmc_%s/**/*.db" % cname, recursive=True))
    if not dbs:
        continue
    c2 = sqlite3.connect(dbs[-1])
    try:
        for name, n, avg in c2.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name='%s' group by name" % cname):
            pm.setdefault(name, {})[cname] = (n, avg)
    except Exception as exc:
        out += ["", "(%s pass unreadable: %r)" % (cname, exc)]
if pm:
    avg_us = dict((r[0], r[3]) for r in con.execute("select name, total_calls, total_duration, average from top_kernels").fetchall())
    out += ["", "## Fabric traffic per launch (PMC passes)", "",
            "FETCH_SIZE / WRITE_SIZE are KiB per dispatch, averaged over the kernel's dispatches; read = 2 x FETCH_SIZE (the guide's gfx950 "
            "correction), write = WRITE_SIZE.  These counters sit in front of the Infinity Cache: fabric traffic, an upper bound on HBM traffic.  "
            "TB/s = (read + write) / the kernel-trace average duration above.", "",
            "| kernel | launches | read MB | write MB | total MB | TB/s | of 8 TB/s |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, dd in sorted(pm.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 0))[1]))[:12]:
        f = dd.get("FETCH_SIZE", (0, 0.0))
        w = dd.get("WRITE_SIZE", (0, 0.0))
        rd, wr = 2 * f[1] * 1024 / 1e6, w[1] * 1024 / 1e6
        us = avg_us.get(name)
        tbs = (rd + wr) / us if us else None          # MB / us = TB/s
        out.append("| \`%s\` | %d | %.1f | %.1f | %.1f | %s | %s |" % (name[:70], f[0], rd, wr, rd + wr,
                                                                   "%.2f" % tbs if tbs else "", "%.2f" % (tbs / 8.0) if tbs else ""))
# the config's line with `traffic_over_bytes` filled in: fabric traffic of everything but the set-up kernels (copies, fills, the
# diagonal-major copy of the operator) over the algorithmic bytes of the iterations the process made
try:
    skip = ("copyBuffer", "fillBuffer", "k_dia_fill", "k_zdia_fill")
    total = 0.0
    for name, dd in pm.items():
        if any(t in name for t in skip):
            continue
        f = dd.get("FETCH_SIZE", (0, 0.0))
        w = dd.get("WRITE_SIZE", (0, 0.0))
        total += 2 * f[0] * f[1] * 1024 + w[0] * w[1] * 1024
    if pm and d.get("bytes_per_iteration") and d.get("iterations_in_process"):
        d["traffic_over_bytes"] = total / (d["bytes_per_iteration"] * d["iterations_in_process"])
        out += ["", "Fabric traffic of the solver's kernels over the algorithmic bytes of the %d iterations of the process: **%.3f** "
                    "(`traffic_over_bytes`; bytes_per_iteration = %.3f GB, frac = %.3f)." % (d["iterations_in_process"], d["traffic_over_bytes"],
                                                                                          d["bytes_per_iteration"] / 1e9, d.get("frac", 0.0))]
except Exception as exc:
    out += ["", "(traffic_over_bytes unavailable: %r)" % exc]
open("$OUT/line.json", "w").write(json.dumps(d) + "\n")
open("$OUT/summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
