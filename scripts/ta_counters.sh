#!/bin/bash
# Which unit of the vector-memory path is busy under the brick kernel: TA / TD / TCP (vector L1) counters of a short bench run,
# a few counters per pass (whatever of the candidate list this rocprofv3 knows).  Run on the GPU box from the repo root:
#   bash scripts/ta_counters.sh <tag> <bench args...>     -> gpurun_out/ta_<tag>/summary.txt   (TA_CAND="..." : a shorter counter list)
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ta_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
CAND="${TA_CAND:-TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum}"
HAVE=""
for C in $CAND; do grep -qw "$C" $OUT/avail.txt && HAVE="$HAVE $C"; done
echo "counters found:$HAVE" > $OUT/summary.txt
BENCH="${TA_CMD:-python3 $ROOT/bench.py $* --no-cpu-baseline --no-extra --steps 8 --warmup 2 --spinup-seconds 0}"  # TA_CMD: another driver (absolute paths), TA_KERNEL: the kernel to report
set -- $HAVE
P=0
while [ $# -gt 0 ]; do
  SET="$1"; BLK=${1%%_*}; shift
  N=1
  # up to 3 counters of one block per pass
  while [ $# -gt 0 ] && [ "${1%%_*}" = "$BLK" ] && [ $N -lt 3 ]; do SET="$SET $1"; shift; N=$((N+1)); done
  P=$((P+1))
  echo "== pass $P: $SET" >> $OUT/summary.txt
  timeout -k 10 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$P -o ta -- $BENCH > /dev/null 2> $OUT/p$P.log || { echo "pass $P failed" >> $OUT/summary.txt; tail -3 $OUT/p$P.log >> $OUT/summary.txt; }
done
cd $ROOT
python3 - <<PY >> $OUT/summary.txt
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "${TA_KERNEL:-k_reweight}" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:45s} {sum(v)/len(v):18.1f}  ({len(v)} launches)")
PY
cat $OUT/summary.txt
