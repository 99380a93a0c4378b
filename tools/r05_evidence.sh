#!/bin/bash
# Round-5 evidence on one MI355X box (through gpurun): tools/r05_evidence.sh <tests|bench|final|configs|small|blk2|ranks|fallback|fuzz>
# Everything lands under gpurun_out/ev/; the summaries that are judged are copied to profiles/ by hand afterwards.
set -u
WHAT=${1:-bench}
export TMPDIR=/tmp
mkdir -p gpurun_out/ev
case $WHAT in
tests)
  rm -f gpurun_out/ev/fullsize_parity.log
  KRYPY_AMD_PARITY_LOG=$PWD/gpurun_out/ev/fullsize_parity.log python -m pytest tests -m gpu -q --durations=40 > gpurun_out/ev/gputest.log 2>&1 < /dev/null
  tail -4 gpurun_out/ev/gputest.log
  python __graft_entry__.py smoke > gpurun_out/ev/smoke.log 2>&1; tail -1 gpurun_out/ev/smoke.log
  cat gpurun_out/ev/fullsize_parity.log
  ;;
bench)
  # kernel trace + PMC passes of the bench command, for the reference order (the default) and the panel form
  bash tools/profile.sh r05_mgs --ortho mgs --other-modes none > gpurun_out/ev/profile_mgs.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r05_mgs profiles/r05_bench_mgs_chain.md
  bash tools/profile.sh r05_cgs --ortho cgs --other-modes none > gpurun_out/ev/profile_cgs.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r05_cgs profiles/r05_bench_cgs.md
  cp profiles/r05_bench_mgs_chain.md profiles/r05_bench_mgs_chain_traffic.json profiles/r05_bench_cgs.md profiles/r05_bench_cgs_traffic.json gpurun_out/ev/ 2>/dev/null
  rm -rf gpurun_out/prof_r05_mgs/trace gpurun_out/prof_r05_mgs/pmc_* gpurun_out/prof_r05_cgs/trace gpurun_out/prof_r05_cgs/pmc_*
  # the line itself (the traffic files just written carry the stamp of these sources)
  python bench.py > gpurun_out/ev/r05_bench.json 2> gpurun_out/ev/r05_bench.err
  tail -c 400 gpurun_out/ev/r05_bench.json
  ;;
final)
  # the short form for the last minutes of a round's GPU budget: kernel trace + PMC passes of the default (reference-order)
  # bench command so that the traffic file carries the stamp of the final sources, then the line itself
  bash tools/profile.sh r05_mgs --ortho mgs --other-modes none > gpurun_out/ev/profile_mgs.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_r05_mgs profiles/r05_bench_mgs_chain.md
  cp profiles/r05_bench_mgs_chain.md profiles/r05_bench_mgs_chain_traffic.json gpurun_out/ev/ 2>/dev/null
  rm -rf gpurun_out/prof_r05_mgs/trace gpurun_out/prof_r05_mgs/pmc_*
  python bench.py > gpurun_out/ev/r05_bench.json 2> gpurun_out/ev/r05_bench.err
  tail -c 400 gpurun_out/ev/r05_bench.json
  ;;
blk2)
  # the A / B lines of the eight-wave blocked kernel (profiles/r05_blk2_cw.log, r05_blk2_cw7.log, r05_blk2_one.log,
  # r05_blk2_one_1gpu.log, r05_shard_loop.log were made by scratch scripts of this shape; this mode makes them again)
  line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('$1: %.0f it/s (sums %s, halo %s)' % (d['value'], c.get('cross_rank_sums'), c.get('halo')))"; }
  : > gpurun_out/ev/blk2_ab.log
  for rep in 1 2; do
    # one GPU: communication wave on / off / up to 6 rows, 1.12 ... 1.6 M rows
    for cw in 1 0 2; do for ny in 280 313 330 375 400; do
      KRYPY_AMD_BLK2_CW=$cw python bench.py --nx 4000 --ny $ny --ortho mgs --no-roofline --no-cpu-baseline --steps 10 --other-modes none 2>/dev/null \
        | line "KRYPY_AMD_BLK2_CW=$cw: one GPU, 4000 x $ny, mgs" >> gpurun_out/ev/blk2_ab.log; done; done
    # one GPU: the one-block shapes on (2) / on a communicator only (1), 1.68 ... 2.5 M rows
    for one in 1 2; do for ny in 420 500 625; do
      KRYPY_AMD_BLK2_ONE=$one python bench.py --nx 4000 --ny $ny --ortho mgs --no-roofline --no-cpu-baseline --steps 10 --other-modes none 2>/dev/null \
        | line "one GPU, 4000 x $ny, KRYPY_AMD_BLK2_ONE=$one, mgs" >> gpurun_out/ev/blk2_ab.log; done; done
    # one MIDDLE rank (sums through the own mailbox inside the blocked kernel, halo inside the SpMV's launch): N/8 and N/4 slabs
    for ny in 313 450 500 625; do
      for o in mgs cgs; do
        python bench.py --force-sharded --loop-halo --nx 4000 --ny $ny --ortho $o --no-roofline --no-cpu-baseline --steps 10 --other-modes none 2>/dev/null \
          | line "one middle rank, 4000 x $ny, $o" >> gpurun_out/ev/blk2_ab.log; done
      KRYPY_AMD_BLK2_ONE=0 python bench.py --force-sharded --loop-halo --nx 4000 --ny $ny --ortho mgs --no-roofline --no-cpu-baseline --steps 10 --other-modes none 2>/dev/null \
        | line "one middle rank, 4000 x $ny, KRYPY_AMD_BLK2_ONE=0, mgs" >> gpurun_out/ev/blk2_ab.log
    done
  done
  cat gpurun_out/ev/blk2_ab.log
  ;;
ranks)
  # N rank PROCESSES on the one device, no RCCL communicator (profiles/r05_ranks_on_one_gpu.log): the residual of every run against
  # the one-rank run's
  : > gpurun_out/ev/ranks_on_one_gpu.log
  for n in 1 2 3 4 8; do
    if [ $n = 1 ]; then X=""; else X="--gpus $n --share-devices --transport xr"; fi
    python bench.py $X --nx 800 --ny 600 --ortho mgs --no-roofline --no-cpu-baseline --steps 2 --warmup 1 --other-modes none 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('ranks %s on %s device(s): final relres %.15e, %.0f it/s (sums %s, halo %s)' % (c.get('ranks'), d['n_gpus'], c['final_relres'], d['value'], c.get('cross_rank_sums'), c.get('halo')))" >> gpurun_out/ev/ranks_on_one_gpu.log
  done
  cat gpurun_out/ev/ranks_on_one_gpu.log
  ;;
configs)
  # kernel trace + PMC passes of each secondary configuration; every line of r05_configs.jsonl carries bytes_per_iteration, frac and
  # traffic_over_bytes (the last from the PMC passes: tools/profile_config.sh)
  : > gpurun_out/ev/r05_configs.jsonl
  for c in 3 4 5 5s; do bash tools/profile_config.sh $c > gpurun_out/ev/profile_cfg$c.log 2>&1; cp gpurun_out/prof_cfg$c/summary.md gpurun_out/ev/r05_config$c.md; cat gpurun_out/prof_cfg$c/line.json >> gpurun_out/ev/r05_configs.jsonl; done
  for c in band ragged; do python tools/bench_configs.py $c 2>/dev/null | tail -1 >> gpurun_out/ev/r05_configs.jsonl; done
  # (the rates of an unprofiled run beside them)
  for c in 3 4 5 5s; do python tools/bench_configs.py $c 2>/dev/null | tail -1; done > gpurun_out/ev/r05_configs_unprofiled.jsonl
  cat gpurun_out/ev/r05_configs.jsonl | cut -c1-400
  ;;
small)
  python tools/blk_bench.py 100 200 316 500 1000 2>/dev/null | grep "N =" > gpurun_out/ev/blk_bench.log; cat gpurun_out/ev/blk_bench.log
  python tools/small_bench.py 64 100 200 316 500 1000 2>/dev/null | grep "N =" > gpurun_out/ev/small_bench.log; cat gpurun_out/ev/small_bench.log
  python tools/proj_bench.py 2>/dev/null | grep "N =" > gpurun_out/ev/proj_bench.log; cat gpurun_out/ev/proj_bench.log
  for ny in 1250 625 313; do for o in cgs mgs; do python bench.py --force-sharded --nx 4000 --ny $ny --ortho $o --no-roofline --steps 10 --other-modes none 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shard 4000 x $ny ($o): %.0f it/s' % d['value'])"; done; done | tee gpurun_out/ev/shards.log
  KRYPY_AMD_MGS_LOWSYNC=0 python bench.py --force-sharded --nx 4000 --ny 313 --ortho mgs --no-roofline --steps 3 --other-modes none 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shard 4000 x 313 (mgs, one all-reduce per link): %.0f it/s' % d['value'])" | tee -a gpurun_out/ev/shards.log
  python tools/complex_bench.py minres mgs cgs 2>/dev/null | tail -3 | tee gpurun_out/ev/complex.log
  ;;
fallback)
  # the fallback paths are tested paths: the parity / complex / blocked / loopback files under the switches that take the
  # rounds' kernels away.  Every "which kernel ran" statement of those tests is an expect_kernel (tests/support/kernel_expect.py):
  # recorded, and reported at teardown only after the test body - all numeric comparisons - has passed.  So in this log a
  # line "ERROR ... KERNEL-PATH EXPECTATION (all numeric comparisons of this test passed)" is a counter that is false by
  # construction under the switch; a line "FAILED ..." would be a numeric comparison that failed on the fallback path.
  F="tests/test_gpu_parity.py tests/test_gpu_complex.py tests/test_gpu_blocked.py tests/test_gpu_halo_loopback.py tests/test_gpu_xr.py tests/test_gpu_blk2.py tests/test_gpu_chain_xr.py tests/test_gpu_gram.py"
  : > gpurun_out/ev/fallback.log
  run() { echo "## $1" >> gpurun_out/ev/fallback.log; env $1 python -m pytest $F -q -rfE -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|error" | cut -c1-420 >> gpurun_out/ev/fallback.log; echo >> gpurun_out/ev/fallback.log; }
  if [ -n "${2:-}" ]; then          # tools/r05_evidence.sh fallback "<switch set>": that one set only
    run "$2"; cat gpurun_out/ev/fallback.log; exit 0
  fi
  run "KRYPY_AMD_TEST_FORCE_MULTI=1"
  run "KRYPY_AMD_MGS_CHAIN=0"
  run "KRYPY_AMD_CHAIN_BLK=0 KRYPY_AMD_MGS_LOWSYNC=0 KRYPY_AMD_PROJ_REG=0 KRYPY_AMD_MINRES_CYCLE=0 KRYPY_AMD_CG_CYCLE=0 KRYPY_AMD_GMRES_CYCLE=0"
  run "KRYPY_AMD_CHAIN_SPMV=0 KRYPY_AMD_SPMV_DIA=0 KRYPY_AMD_CHAIN_LDS=0"
  run "KRYPY_AMD_CHAIN_PF=0 KRYPY_AMD_CHAIN_ONEX=0 KRYPY_AMD_CHAIN_SMALL=0 KRYPY_AMD_TAG_WAIT=0 KRYPY_AMD_LANCZOS_FUSED=0"
  run "KRYPY_AMD_CG_STEP=0 KRYPY_AMD_SPMV_SPLIT=0 KRYPY_AMD_PROJ_PANEL=0 KRYPY_AMD_CGS_REVERSE=0 KRYPY_AMD_BLK_NX=0 KRYPY_AMD_XR=0 KRYPY_AMD_CHAIN_BLK2=0 KRYPY_AMD_XH=0 KRYPY_AMD_SPMV_WIN=0 KRYPY_AMD_BLK2_ONE=0"
  run "KRYPY_AMD_BLK2_CW=0"
  run "KRYPY_AMD_BLK2_CW=2"
  run "KRYPY_AMD_CHAIN_XR=0 KRYPY_AMD_CHAIN_LONG=0 KRYPY_AMD_GRAM_MFMA=0"
  grep -c "^FAILED" gpurun_out/ev/fallback.log | sed 's/^/numeric FAILED lines in all switch sets: /' >> gpurun_out/ev/fallback.log
  cat gpurun_out/ev/fallback.log
  ;;
fuzz)
  # randomised layers on the final tree: every C entry against NumPy (kh_minres_cycle included), whole solves against the oracle,
  # a soak of solves of every kind, the one-XCD launches
  python tools/abi_fuzz.py 500 > gpurun_out/ev/abi_fuzz.log 2>&1; tail -1 gpurun_out/ev/abi_fuzz.log
  python tools/solve_fuzz.py 150 > gpurun_out/ev/solve_fuzz.log 2>&1; tail -1 gpurun_out/ev/solve_fuzz.log
  python tools/soak.py 30 > gpurun_out/ev/soak.log 2>&1; tail -3 gpurun_out/ev/soak.log
  timeout 400 python tools/onex_soak.py > gpurun_out/ev/onex_soak.log 2>&1; tail -2 gpurun_out/ev/onex_soak.log
  ;;
esac
