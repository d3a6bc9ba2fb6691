#!/bin/bash
# rocprofv3 evidence of one configuration with the binary as it stands: kernel stats of a bench command, then FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes, optionally SQ counters.  Run on the GPU box from the repo root:
#   bash scripts/profile_round.sh <tag> <bench args...>       e.g.  bash scripts/profile_round.sh c3 --config c3
# writes gpurun_out/prof_<tag>/{stats,fetch,write,sq}/ and gpurun_out/prof_<tag>_bench.json
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py $* --no-cpu-baseline --no-extra"
echo "== kernel stats ($TAG)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $BENCH --steps 100 --warmup 10 > $ROOT/gpurun_out/prof_${TAG}_bench.json 2> $OUT/stats.log || tail -5 $OUT/stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C ($TAG)"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o $TAG -- $BENCH --steps 10 --warmup 2 --spinup-seconds 0 > /dev/null 2> $OUT/$C.log || tail -5 $OUT/$C.log
done
if [ "${SQ:-0}" = "1" ]; then
  for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
    N=$(echo $SET | cut -d' ' -f1)
    echo "== pmc $SET ($TAG)"
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/sq_$N -o $TAG -- $BENCH --steps 8 --warmup 2 --spinup-seconds 0 > /dev/null 2> $OUT/sq_$N.log || tail -5 $OUT/sq_$N.log
  done
fi
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, os
out = "$OUT"
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
res = {}
for d in sorted(os.listdir(out)):
    if os.path.isdir(os.path.join(out, d)) and d != "stats":
        for k, cs in counters(d).items():
            if "k_reweight" in k:
                for c, v in cs.items():
                    res.setdefault(k, {})[c] = {"avg": sum(v) / len(v), "launches": len(v)}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_reweight" in r["Name"]:
            res.setdefault(r["Name"], {})["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = 2 * v["FETCH_SIZE"]["avg"] * 1024 + v["WRITE_SIZE"]["avg"] * 1024
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
