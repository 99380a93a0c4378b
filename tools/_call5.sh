set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
./tools/probe/gsum_probe > gpurun_out/gsum_probe.log 2>&1; cat gpurun_out/gsum_probe.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "register_shape or timeout or interleaved" > gpurun_out/pytest_c5.log 2>&1; tail -8 gpurun_out/pytest_c5.log
for n in 12000000 14000000; do KRYPY_AMD_CHAIN_PF=0 python tools/chain_bench.py $n 2>&1 | grep -E "chain|link"; done
