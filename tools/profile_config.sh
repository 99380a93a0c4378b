#!/bin/bash
# Kernel trace of one secondary configuration: tools/profile_config.sh <3|4|5|5s>  -> gpurun_out/prof_cfg<N>/summary.md
set -u
CFG=${1:-3}
export TMPDIR=/tmp
OUT=gpurun_out/prof_cfg$CFG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_configs.py $CFG > $OUT/bench_trace.json 2> $OUT/trace.err
python - <<PY
import glob, json, sqlite3
line = open("$OUT/bench_trace.json").read().strip().splitlines()[-1]
d = json.loads(line)
db = sorted(glob.glob("$OUT/trace/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average from top_kernels order by total_duration desc limit 14").fetchall()
tot = sum(r[2] for r in con.execute("select name, total_calls, total_duration, average from top_kernels").fetchall())
out = ["# rocprofv3 kernel trace: config $CFG", "",
       "Command: \`rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py $CFG\` on one MI355X (tools/profile_config.sh).", "",
       "Result under the profiler: " + ", ".join("%s = %s" % (k, (round(v, 3) if isinstance(v, float) else v)) for k, v in d.items() if k != "solve_ms"), "",
       "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---:|---:|---:|---:|"]
for name, calls, total, avg in rows:
    out.append("| \`%s\` | %d | %.2f | %.2f | %.1f |" % (name[:70], calls, total / 1e3, avg, 100.0 * total / tot))     # (the stats view is in us)
open("$OUT/summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
