#!/bin/bash
# A/B/C of builds on one shape, interleaved: bash scripts/ab3.sh reps config events lib...
REPS=$1; CFG=$2; EV=$3; shift 3
for ((r = 0; r < REPS; ++r)); do
  for L in "$@"; do M3HIP_LIB=$L python3 scripts/gexp.py $CFG $EV 2>&1 | tail -1; done
done
