#!/bin/bash
# A / B of the CSR-stream SpMV's index / value loads on one box: the product library (KH_SPMV_VEC = 1: one entry per lane and load)
# against a build with aligned runs of four entries per lane (-DKH_SPMV_VEC=4 -> krypy_amd/lib/libkrylov_hip_vec4.so), HIP-event
# timing (tools/spmv_bench.py, 50 back-to-back products) and the kernel-trace average of the same launches.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/spmv_ab
mkdir -p $OUT
: > $OUT/spmv_ab.log
for rep in 1 2; do
  for lib in libkrylov_hip.so libkrylov_hip_vec4.so; do
    [ -f krypy_amd/lib/$lib ] || continue
    echo "## $lib (rep $rep)" >> $OUT/spmv_ab.log
    KRYPY_AMD_LIB=$PWD/krypy_amd/lib/$lib KRYPY_AMD_SPMV_DIA=0 python tools/spmv_bench.py 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' >> $OUT/spmv_ab.log
  done
done
for lib in libkrylov_hip.so libkrylov_hip_vec4.so; do
  [ -f krypy_amd/lib/$lib ] || continue
  rm -rf $OUT/trace
  KRYPY_AMD_LIB=$PWD/krypy_amd/lib/$lib KRYPY_AMD_SPMV_DIA=0 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/spmv_bench.py > /dev/null 2> $OUT/trace.err
  python - >> $OUT/spmv_ab.log <<PY
import glob, sqlite3
db = sorted(glob.glob("$OUT/trace/**/*.db", recursive=True))[-1]
con = sqlite3.connect(db)
print("## kernel trace, $lib")
for name, calls, total, avg in con.execute("select name, total_calls, total_duration, average from top_kernels where name like '%k_spmv_stream%'"):
    print("   %s: %d launches, avg %.1f us" % (name[:90], calls, avg / 1e3 if avg > 1e4 else avg))
PY
done
rm -rf $OUT/trace
cat $OUT/spmv_ab.log
