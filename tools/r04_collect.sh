#!/bin/bash
# Copy the round-4 evidence that tools/r04_evidence.sh left under gpurun_out/ev/ into profiles/ (run here, after the gpurun calls).
set -u
E=gpurun_out/ev
P=profiles
HEAD=$(git rev-parse --short HEAD)
hdr() { echo "# $1 (round 4, one MI355X through gpurun, tree $HEAD + working copy; tools/r04_evidence.sh)"; }
[ -f $E/r04_bench.json ] && python - <<PY
l=[x for x in open("$E/r04_bench.json") if x.startswith("{")][-1]
open("$P/r04_bench.json","w").write(l if l.endswith("\n") else l+"\n")
PY
for f in r04_bench_mgs_chain.md r04_bench_mgs_chain_traffic.json r04_bench_cgs.md r04_bench_cgs_traffic.json r04_config3.md r04_config4.md r04_config5.md r04_config5s.md r04_configs.jsonl; do [ -f $E/$f ] && cp $E/$f $P/$f; done
[ -f $E/fullsize_parity.log ] && { hdr "full-size comparisons of the GPU suite (tests/test_gpu_fullsize.py): measured deviations next to their bars"; cat $E/fullsize_parity.log; echo "# suite: $(grep -E ' passed' $E/gputest.log | tail -1); $(tail -1 $E/smoke.log)"; } > $P/r04_fullsize_parity.log
[ -f $E/blk_bench.log ] && { hdr "tools/blk_bench.py: GMRES(100) mgs, the blocked kernel (chain_blk.h) against the per-column kernel in one process"; cat $E/blk_bench.log; echo "# experiment, not kept (an earlier box of this round): BC = 8 columns per sum, ONE block in registers (-DKH_BLK_BC_CFG=8 -DKH_BLK_NSLOT_CFG=1): N = 1e4 16.2k, 1e5 11.5k, 2.5e5 12.5k, 1e6 6.96k it/s against 16.5k / 12.6k / 11.4k / 7.5k on that box - half as many sums, but no block in flight across a sum"; } > $P/r04_blk_bench.log
[ -f $E/small_bench.log ] && { hdr "tools/small_bench.py: GMRES(100) mgs / cgs and MINRES + Jacobi (150-step solves, set-up included) at short vectors; onex = 0: KRYPY_AMD_CHAIN_ONEX=0"; cat $E/small_bench.log; } > $P/r04_small_vectors.log
[ -f $E/proj_bench.log ] && { hdr "tools/proj_bench.py: the deflation projector (d = 16, two sweeps), one launch with the vector in registers (proj_reg.h) against four launches per sweep"; cat $E/proj_bench.log; } > $P/r04_proj_bench.log
[ -f $E/shards.log ] && { hdr "bench.py --force-sharded: one rank's shard of the benchmark problem (N/2, N/4, N/8 rows) through the multi-rank code path of a 1-rank communicator, GMRES(100); mgs = all coefficients from one pass and ONE all-reduce (Gram-table correction)"; cat $E/shards.log; } > $P/r04_shards.log
[ -f $E/complex.log ] && { hdr "tools/complex_bench.py: complex (c128) GMRES(100), N = 5e6 (the real bench's bytes per vector)"; cat $E/complex.log; } > $P/r04_complex.log
for f in abi_fuzz solve_fuzz soak onex_soak; do [ -f $E/$f.log ] && { hdr "tools/$f.py on the final tree"; tail -n 12 $E/$f.log; } > $P/r04_$f.log; done
[ -f $E/fallback.log ] && { hdr "the GPU test files under the switches that take the round's kernels away (tools/r04_evidence.sh fallback)"; cat $E/fallback.log; } > $P/r04_fallback_suites.log
rm -f $P/r04_blk_bench_final.log
ls $P | grep r04
