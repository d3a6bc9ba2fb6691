#!/bin/bash
# k_rebin_lds of two builds on one box: rocprofv3 kernel stats of scripts/f4_ragged.py.  usage: bash scripts/f4_kernel_ab.sh [base.so] [events] [config]
BASE=${1:-build/libm3hip_base.so}; EV=${2:-10000000}; CFG=${3:-c3}
ROOT=$(pwd); export TMPDIR=/tmp
for L in $BASE mach3_amd/lib/libm3hip.so; do
  T=$(basename $L .so); D=$ROOT/gpurun_out/prof_f4ab_$T; rm -rf $D
  (cd /tmp && M3HIP_LIB=$ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o f4 -- python3 $ROOT/scripts/f4_ragged.py $EV $CFG 2>/dev/null | tail -1)
  grep -E 'k_rebin|k_reweight_brick' $(find $D -name '*kernel_stats.csv') | awk -F'","' '{printf "   %-60s calls %s avg %.1f us min %.1f us\n", substr($1,2,60), $2, $4/1e3, $6/1e3}'
done
