#!/bin/bash
# Kernel trace + fabric-traffic counters of ONE RANK'S SHARE of the benchmark problem on N = 8 ranks (a 4000 x 313 slab, 1.252 M rows)
# through the forced multi-rank path on one MI355X: bench.py --force-sharded --ortho mgs = sharded SpMV (interior / boundary launches)
# + k_mgs_chain_blk2 with the cross-rank sums inside the launch (every sum through the rank's own mailbox).
#   tools/profile_shard.sh [--loop-halo]  ->  gpurun_out/prof_shard/summary.md
# --loop-halo: the slab as a MIDDLE rank has it - itself as previous and next neighbour, boundary rows out and ghost rows in inside the
# banded SpMV's one launch (xh), instead of the two launches without a halo
# (rocprofv3 --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes with --kernel-trace only;
# FETCH_SIZE doubled - the guide's gfx950 correction for wide streams)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_shard
mkdir -p $OUT
EXTRA=${1:-}
CMD="python bench.py --force-sharded $EXTRA --nx 4000 --ny 313 --ortho mgs --no-roofline --steps 6 --warmup 1 --other-modes none"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err < /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc -- $CMD > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err < /dev/null
done
python - <<PY
import glob, json, sqlite3
d = json.loads([l for l in open("$OUT/bench_trace.json").read().strip().splitlines() if l.startswith("{")][-1])
con = sqlite3.connect(sorted(glob.glob("$OUT/trace/**/*.db", recursive=True))[-1])
rows = con.execute("select name, total_calls, total_duration, average from top_kernels order by total_duration desc limit 12").fetchall()
tot = sum(r[2] for r in con.execute("select name, total_calls, total_duration, average from top_kernels").fetchall())
out = ["# rocprofv3: one rank's slab of the benchmark problem on 8 ranks (4000 x 313, 1.252 M rows), forced multi-rank path on one MI355X", "",
       "Command: \`rocprofv3 --kernel-trace --stats -- $CMD\`, then \`--pmc FETCH_SIZE\` / \`--pmc WRITE_SIZE\` passes of the same command "
       "with \`--kernel-trace\` only (tools/profile_shard.sh).", "",
       "Under the profiler: **%.0f iterations/s**, ortho = %s, sums across the ranks: %s (every sum of the blocked kernel goes through the rank's own "
       "mailbox inside the launch), halo: %s, %d iterations timed." % (d["value"], d["config"]["ortho"], d["config"]["cross_rank_sums"], d["config"]["halo"], d["config"]["iterations_timed"]), "",
       "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---:|---:|---:|---:|"]
for name, calls, total, avg in rows:
    out.append("| \`%s\` | %d | %.2f | %.2f | %.1f |" % (name[:90], calls, total / 1e3, avg, 100.0 * total / tot))
k = con.execute("select start, end from kernels where name like '%k_mgs_chain_blk2%' order by start").fetchall()
if len(k) >= 200:
    cyc = k[100:200]
    out += ["", "Launch duration of \`k_mgs_chain_blk2\` against the step number k (one cycle, k + 1 links = ceil((k + 1) / 4) blocks + the norm):", "",
            "| k | " + " | ".join(str(x) for x in (0, 3, 7, 15, 31, 50, 75, 99)) + " |", "|---|" + "---:|" * 8,
            "| us | " + " | ".join("%.1f" % ((cyc[x][1] - cyc[x][0]) / 1e3) for x in (0, 3, 7, 15, 31, 50, 75, 99)) + " |"]
    gaps = [(cyc[i + 1][0] - cyc[i][1]) / 1e3 for i in range(99)]
    slope = ((cyc[99][1] - cyc[99][0]) - (cyc[3][1] - cyc[3][0])) / 1e3 / 24.0
    out += ["", "Per block of four columns: **%.2f us** ((k = 99) - (k = 3)) / 24 blocks; the block's 4 x 10.0 MB at 6.5 TB/s would be 6.2 us.  "
            "Between the end of one blocked launch and the start of the next: %.1f us on average (the sharded SpMV - two launches without a halo, or one with the halo inside it - and the "
            "dependent-launch gaps)." % (slope, sum(gaps) / len(gaps))]
pm = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = sorted(glob.glob("$OUT/pmc_%s/**/*.db" % cname, recursive=True))
    if dbs:
        c2 = sqlite3.connect(dbs[-1])
        for name, n, avg in c2.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name='%s' group by name" % cname):
            pm.setdefault(name, {})[cname] = (n, avg)
if pm:
    avg_us = dict((r[0], r[3]) for r in con.execute("select name, total_calls, total_duration, average from top_kernels").fetchall())
    out += ["", "## Fabric traffic per launch (PMC passes)", "",
            "read = 2 x FETCH_SIZE (the guide's gfx950 correction), write = WRITE_SIZE, KiB per dispatch averaged over the kernel's dispatches; "
            "TB/s = (read + write) / the kernel-trace average duration.  The blocked kernel's algorithmic bytes per launch, averaged over a cycle: "
            "50.5 columns x 10.016 MB + w in + v out = 526 MB.", "",
            "| kernel | launches | read MB | write MB | total MB | TB/s | of 8 TB/s |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, dd in sorted(pm.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 0))[1]))[:8]:
        f, w = dd.get("FETCH_SIZE", (0, 0.0)), dd.get("WRITE_SIZE", (0, 0.0))
        rd, wr = 2 * f[1] * 1024 / 1e6, w[1] * 1024 / 1e6
        us = avg_us.get(name)
        tbs = (rd + wr) / us if us else None
        out.append("| \`%s\` | %d | %.1f | %.1f | %.1f | %s | %s |" % (name[:90], f[0], rd, wr, rd + wr, "%.2f" % tbs if tbs else "", "%.2f" % (tbs / 8.0) if tbs else ""))
open("$OUT/summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
