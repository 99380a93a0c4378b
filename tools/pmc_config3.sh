#!/bin/bash
# HBM-side traffic of the MINRES + Jacobi iteration's one launch (config 3): separate --pmc passes, kernel trace only
#   bash tools/pmc_config3.sh  ->  gpurun_out/pmc_cfg3/summary.txt
export TMPDIR=/tmp
OUT=gpurun_out/pmc_cfg3
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d $OUT/$T -o pmc -- python tools/bench_configs.py 3 > $OUT/$T.json 2> $OUT/$T.err
done
python - <<'PY'
import glob, sqlite3
out = []
for d in sorted(glob.glob("gpurun_out/pmc_cfg3/*/")):
    for db in glob.glob(d + "**/*.db", recursive=True):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name order by avg(counter_value) desc").fetchall()
        except Exception as e:
            out.append("%s: %s" % (db, e)); continue
        for name, c, n, avg in rows[:8]:
            out.append("%-22s %-70s launches %5d  avg %.1f" % (c, name[:70], n, avg))
open("gpurun_out/pmc_cfg3/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $OUT/*/  
