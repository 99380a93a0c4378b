set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in 10000000 8000000 4000000; do
  timeout 300 python tools/chain_trace.py $n > gpurun_out/trace_$n.log 2>&1
done
timeout 900 python bench.py --no-cpu-baseline --other-modes none > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c2.json'))
print(d['value'], d['roofline'])
for k,v in d['attainable']['probes'].items(): print(k, v)
PY
cat gpurun_out/trace_10000000.log gpurun_out/trace_8000000.log gpurun_out/trace_4000000.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_c2.log 2>&1; tail -5 gpurun_out/pytest_c2.log
