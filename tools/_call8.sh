set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --other-modes none > gpurun_out/bench_r$i.json 2> gpurun_out/bench_r$i.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_r$i.json'))
print('run $i', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms']*1000/64, 'us/link', d['config']['cycle_ms'])
PY
done
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_c8.log 2>&1; tail -8 gpurun_out/pytest_c8.log
