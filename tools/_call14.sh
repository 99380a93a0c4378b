cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
set -x
timeout 900 python -m pytest tests -m gpu -q -k "complex or register_shape or timeout" > gpurun_out/pytest_c14.log 2>&1; tail -4 gpurun_out/pytest_c14.log
./tools/probe/gsum_probe 2>&1 | grep "G=245 bg_rows=0" | head -3
python tools/complex_bench.py mgs cgs > gpurun_out/complex_r02.jsonl 2>&1; cat gpurun_out/complex_r02.jsonl
timeout 1500 python tools/bench_configs.py 3 5 5s 5sc 4 > gpurun_out/configs_r02.jsonl 2> gpurun_out/configs_r02.err; cat gpurun_out/configs_r02.jsonl
for n in 100000 1000000; do
  python - <<PY
import sys, time, json
sys.path.insert(0,'.')
import numpy as np, bench
from krypy_amd import _hip, linsys, utils
ctx=_hip.get_context()
n=$n
nx=int(round(n**0.5)); A=bench.laplace2d(nx,nx); b=np.random.default_rng(0).standard_normal(A.shape[0])
ls=linsys.LinearSystem(A,b)
for ortho in ('mgs','cgs'):
    for it in range(2):
        ctx.sync(); t0=time.perf_counter()
        try: s=linsys.RestartedGmres(ls,maxiter=100,max_restarts=9,tol=1e-14,ortho=ortho)
        except utils.ConvergenceError as e: s=e.solver
        ctx.sync(); dt=time.perf_counter()-t0
    print(json.dumps({"config":"GMRES(100) 2-D Laplacian N=%d ortho=%s"%(A.shape[0],ortho),"iterations_per_s":(len(s.resnorms)-1)/dt}))
PY
done
