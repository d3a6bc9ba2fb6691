#!/bin/bash
# A/B of two builds on one GPU's share of BASELINE configs[4] (1.25e7 events) and, optionally, other shapes: alternating runs of
# scripts/gexp.py with M3HIP_LIB = the baseline build and the build in mach3_amd/lib.  usage: bash scripts/ab_c5.sh [base.so] [reps] [config events]...
BASE=${1:-build/libm3hip_base.so}; REPS=${2:-2}; shift 2 || true
SHAPES=("$@"); [ ${#SHAPES[@]} -eq 0 ] && SHAPES=(c5 12500000)
for ((i = 0; i < ${#SHAPES[@]}; i += 2)); do
  for ((r = 0; r < REPS; ++r)); do
    M3HIP_LIB=$BASE python3 scripts/gexp.py ${SHAPES[i]} ${SHAPES[i+1]} 2>&1 | tail -1
    python3 scripts/gexp.py ${SHAPES[i]} ${SHAPES[i+1]} 2>&1 | tail -1
  done
done
