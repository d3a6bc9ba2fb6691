#!/bin/bash
# A/B of the re-binning pass on BASELINE configs[2] with one energy-scale parameter: the build in build/libm3hip_base.so against the
# build in mach3_amd/lib, alternating; then the general LDS pass of the current build (M3HIP_REBIN=lds).  usage: bash scripts/f4_ab.sh [reps]
REPS=${1:-2}
for ((r = 0; r < REPS; ++r)); do
  M3HIP_LIB=build/libm3hip_base.so python3 scripts/f4_ragged.py 10000000 c3 2>&1 | tail -1
  python3 scripts/f4_ragged.py 10000000 c3 2>&1 | tail -1
done
M3HIP_REBIN=lds python3 scripts/f4_ragged.py 10000000 c3 2>&1 | tail -1
