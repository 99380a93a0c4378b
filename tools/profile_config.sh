#!/bin/bash
# Kernel trace of one secondary configuration: tools/profile_config.sh <3|4|5>  -> gpurun_out/prof_cfg<N>/
set -u
CFG=${1:-3}
export TMPDIR=/tmp
OUT=gpurun_out/prof_cfg$CFG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_configs.py $CFG > $OUT/bench_trace.json 2> $OUT/trace.err
tail -c 400 $OUT/bench_trace.json; echo
python - <<PY
import sqlite3, glob
db = sorted(glob.glob("$OUT/trace/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average from top_kernels order by total_duration desc limit 20").fetchall() if True else []
for r in rows:
    print("%-60s calls=%6d total_ms=%9.2f avg_us=%9.2f" % (r[0][:60], r[1], r[2]/1e6, r[3]/1e3))
PY
