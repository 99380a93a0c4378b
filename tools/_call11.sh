set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_c11.log 2>&1; tail -12 gpurun_out/pytest_c11.log
bash tools/profile.sh r02_mgs_chain --ortho mgs --other-modes none > gpurun_out/profile_r02.log 2>&1; tail -5 gpurun_out/profile_r02.log
