#!/bin/bash
# rocprofv3 kernel stats only (no PMC passes) of one bench configuration: bash scripts/profile_stats_only.sh <tag> <bench args...>
TAG=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp; D=$ROOT/gpurun_out/stats_$TAG; rm -rf $D
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $TAG -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --no-extra --steps 400 --warmup 10 > $ROOT/gpurun_out/${TAG}_bench.json 2>/dev/null)
python3 - <<PY
import csv, glob, json
line = json.loads(open("$ROOT/gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
alg = line["roofline"]["alg_bytes_per_launch"]
for f in glob.glob("$D/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_reweight" in r["Name"]:
            print(f"$TAG: {r['Name'][:58]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e3:.1f} us -> {alg / float(r['AverageNs']) / 8000:.4f}; in-run {line['roofline']['frac']:.4f} over {line['roofline']['launches_timed']}; {line['value']:.0f} steps/s")
PY
