cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/_dbg_cfg5.py 2>&1 | tail -45
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p5 -o t -- python $GRAFT_REPO_ROOT/tools/_dbg_cfg5.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/p5/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
for r in con.execute("select name,total_calls,total_duration,average from top_kernels limit 14"):
    print(r[0][:70], r[1], round(r[2]/1e6,1), 'ms', round(r[3]/1e3,1), 'us')
PY
