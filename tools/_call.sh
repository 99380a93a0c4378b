set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or interleaved or panel_apply or fused or native or timeout or complex" > gpurun_out/pytest_c4.log 2>&1; tail -8 gpurun_out/pytest_c4.log
for pf in 1 0; do
  KRYPY_AMD_CHAIN_PF=$pf timeout 600 python bench.py --no-cpu-baseline --other-modes none > gpurun_out/bench_pf$pf.json 2> gpurun_out/bench_pf$pf.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_pf$pf.json'))
print('PF=$pf', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms']*1000/64, 'us/link', d['config']['final_relres'])
PY
done
for n in 8000000 6000000 4000000 2000000 500000; do for pf in 1 0; do echo "n=$n pf=$pf"; KRYPY_AMD_CHAIN_PF=$pf python tools/chain_bench.py $n 2>&1 | grep "chain"; done; done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "config2" > gpurun_out/pytest_c4b.log 2>&1; tail -8 gpurun_out/pytest_c4b.log
