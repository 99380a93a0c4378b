cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/pytest_c13.log 2>&1; tail -15 gpurun_out/pytest_c13.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c13.json 2> gpurun_out/bench_c13.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_c13.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
print(d['config'])
print(d.get('other_modes'))
PY
