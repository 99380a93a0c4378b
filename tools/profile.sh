#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
echo "== kernel trace + stats: bench.py $ARGS"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
tail -c 600 $OUT/bench_trace.json; echo
# HBM traffic counters: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $* > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
  tail -c 200 $OUT/bench_pmc_$C.json; echo
done
find $OUT -type f | head -50
du -sh $OUT
