#!/bin/bash
# One FETCH_SIZE pass of a bench command; prints the reweight kernel's read traffic per launch (2 x FETCH_SIZE x 1024, the gfx950
# correction of MI355X_MICROARCH.md).  Run on the GPU box from the repo root:  bash scripts/fetch_only.sh <tag> <bench args>
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/fo_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o $TAG -- python3 $ROOT/bench.py $* --no-cpu-baseline --no-extra --steps 10 --warmup 2 --spinup-seconds 0 > /dev/null 2> $OUT/log.txt || tail -5 $OUT/log.txt
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_reweight" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("$TAG", k[:60], "FETCH x2 KB->GB", 2 * sum(v) / len(v) * 1024 / 1e9, len(v))
PY
