#!/bin/bash
# Round-6 evidence on one MI355X box (through gpurun): tools/r06_evidence.sh <bench|legs|slabs|soak|tests>
# Everything lands under gpurun_out/ev6/; profiles/r06_* are written in place (copy them back from gpurun_out/ev6/profiles).
set -u
WHAT=${1:-bench}
export TMPDIR=/tmp
EV=gpurun_out/ev6
mkdir -p $EV/profiles
prof() {      # prof <tag> <bench args...>: kernel trace + PMC passes of `python bench.py <args>`, summary + stamped traffic file
  local tag=$1; shift
  bash tools/profile.sh $tag "$@" > $EV/profile_$tag.log 2>&1
  python tools/summarize_prof.py gpurun_out/prof_$tag profiles/$tag.md "$@" > /dev/null 2> $EV/summarize_$tag.err
  cp profiles/$tag.md profiles/${tag}_traffic.json $EV/profiles/ 2>/dev/null
  rm -rf gpurun_out/prof_$tag/trace gpurun_out/prof_$tag/pmc_*
}
case $WHAT in
bench)
  # the headline: kernel trace + PMC passes of the default (reference-order) bench command, then the line itself (the
  # traffic file just written carries the stamp of these sources)
  prof r06_bench_mgs_chain --ortho mgs --other-modes none
  python bench.py > $EV/profiles/r06_bench.json 2> $EV/r06_bench.err
  tail -c 300 $EV/profiles/r06_bench.json
  ;;
legs)
  # the secondary configurations' legs of bench.py, each with its kernel trace + PMC passes first
  prof r06_config3 --config 3
  python bench.py --config 3 > $EV/profiles/r06_config3.json 2> $EV/r06_config3.err
  prof r06_config4 --config 4
  python bench.py --config 4 > $EV/profiles/r06_config4.json 2> $EV/r06_config4.err
  prof r06_config5_slab --config 5 --nz 50 --ortho mgs
  python bench.py --config 5 --nz 50 > $EV/profiles/r06_config5_slab.json 2> $EV/r06_config5_slab.err
  prof r06_config5_full --config 5
  python bench.py --config 5 --steps 3 --warmup 1 > $EV/profiles/r06_config5_full.json 2> $EV/r06_config5_full.err
  for f in $EV/profiles/r06_config*.json; do echo $f; head -c 260 $f; echo; done
  ;;
slabs)
  # one MIDDLE rank of N (every sum through the own mailbox inside the launch, the halo inside the SpMV's): the N/2 slab of the
  # benchmark problem through the chain kernel with the cross-rank stage against the one-reduction form and the panel form, and
  # config 5's 12.5 M-row slab (deflated) the same way
  line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('$1: %.1f it/s (sums %s, halo %s, per iteration %s)' % (d['value'], c.get('cross_rank_sums'), c.get('halo'), (c.get('sharded_diagnostics') or {}).get('per_iteration')))"; }
  : > $EV/profiles/r06_slabs.log
  for rep in 1 2; do
    for ny in 1250 2500; do
      for o in mgs cgs; do
        python bench.py --force-sharded --loop-halo --nx 4000 --ny $ny --ortho $o --no-roofline --no-cpu-baseline --steps 6 --other-modes none 2>/dev/null \
          | line "one middle rank, 4000 x $ny, $o" >> $EV/profiles/r06_slabs.log; done
      KRYPY_AMD_CHAIN_XR=0 python bench.py --force-sharded --loop-halo --nx 4000 --ny $ny --ortho mgs --no-roofline --no-cpu-baseline --steps 6 --other-modes none 2>/dev/null \
        | line "one middle rank, 4000 x $ny, mgs with KRYPY_AMD_CHAIN_XR=0" >> $EV/profiles/r06_slabs.log
    done
    for o in mgs cgs; do
      python bench.py --config 5 --nz 50 --force-sharded --loop-halo --ortho $o --no-roofline --no-cpu-baseline --steps 3 2>/dev/null \
        | line "config 5, one middle rank of eight (500 x 500 x 50), $o" >> $EV/profiles/r06_slabs.log; done
    KRYPY_AMD_CHAIN_XR=0 python bench.py --config 5 --nz 50 --force-sharded --loop-halo --ortho mgs --no-roofline --no-cpu-baseline --steps 3 2>/dev/null \
      | line "config 5, one middle rank of eight, mgs with KRYPY_AMD_CHAIN_XR=0" >> $EV/profiles/r06_slabs.log
    KRYPY_AMD_CHAIN_LONG=0 python bench.py --config 5 --nz 50 --force-sharded --loop-halo --ortho mgs --no-roofline --no-cpu-baseline --steps 3 2>/dev/null \
      | line "config 5, one middle rank of eight, mgs with KRYPY_AMD_CHAIN_LONG=0" >> $EV/profiles/r06_slabs.log
  done
  cat $EV/profiles/r06_slabs.log
  ;;
soak)
  # the round's two kernels for long enough to mean something: 4,000 launches of the chain with the cross-rank stage on the N/2
  # slab (200 k sums through the mailbox) and 2,000 of the 48-row kernel with it on config 5's slab, every cycle's residual
  # history compared with the first one's by the line itself (final_relres / deflated_relres must repeat to the last bit)
  : > $EV/profiles/r06_soak.log
  for rep in 1 2; do
    python bench.py --force-sharded --loop-halo --nx 4000 --ny 1250 --ortho mgs --no-roofline --no-cpu-baseline --steps 20 --warmup 0 --other-modes none 2>/dev/null \
      | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('N/2 slab, 2000 iterations: %.1f it/s, final_relres %.17g, cycles %s' % (d['value'], c['final_relres'], c['cycle_ms']))" >> $EV/profiles/r06_soak.log
    python bench.py --config 5 --nz 50 --force-sharded --loop-halo --ortho mgs --no-roofline --no-cpu-baseline --steps 10 --warmup 0 2>/dev/null \
      | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('config 5 slab, 1000 deflated iterations: %.1f it/s, deflated_relres %.17g, solves %s' % (d['value'], c['deflated_relres'], c['cycle_ms']))" >> $EV/profiles/r06_soak.log
  done
  cat $EV/profiles/r06_soak.log
  ;;
tests)
  rm -f $EV/fullsize_parity.log
  KRYPY_AMD_PARITY_LOG=$PWD/$EV/fullsize_parity.log python -m pytest tests -m gpu -q --durations=25 > $EV/gputest.log 2>&1 < /dev/null
  tail -4 $EV/gputest.log
  python __graft_entry__.py smoke > $EV/smoke.log 2>&1; tail -1 $EV/smoke.log
  cat $EV/fullsize_parity.log
  ;;
esac
