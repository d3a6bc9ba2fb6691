#!/bin/bash
# One configuration's rocprofv3 evidence in the form profiles/ keeps it: kernel stats + FETCH_SIZE / WRITE_SIZE (separate --pmc
# passes, scripts/profile_round.sh) -> gpurun_out/<name>_{kernel_stats.csv,pmc_<kind>.json,bench_under_rocprof.json}
#   usage: bash scripts/profile_to_profiles.sh <name> <kind: full|shard> <version text> <bench args...>
#   e.g.   bash scripts/profile_to_profiles.sh r05_c5 shard "round 5 final" --config c5 --events 12500000
set -u
NAME=$1; KIND=$2; VER=$3; shift 3
TAG=$(echo $NAME | sed 's/^r[0-9]*_//')
bash scripts/profile_round.sh $TAG "$@" > gpurun_out/${NAME}_profile_round.log 2>&1
D=gpurun_out/prof_$TAG
cp $(find $D/stats -name '*kernel_stats.csv' | head -1) gpurun_out/${NAME}_${KIND}_kernel_stats.csv
cp gpurun_out/prof_${TAG}_bench.json gpurun_out/${NAME}_${KIND}_bench_under_rocprof.json
M3_ROUND=${M3_ROUND:-6} python3 scripts/make_pmc_json.py $(dirname $(find $D/FETCH_SIZE -name '*counter_collection.csv' | head -1)) $(dirname $(find $D/WRITE_SIZE -name '*counter_collection.csv' | head -1)) \
  gpurun_out/${NAME}_${KIND}_kernel_stats.csv gpurun_out/${NAME}_pmc_${KIND}.json "$VER" gpurun_out/${NAME}_${KIND}_bench_under_rocprof.json > /dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/${NAME}_pmc_${KIND}.json"))
for k, v in d["kernels"].items():
    if "k_reweight" in k and "hbm_bytes_per_launch" in v:
        print(f"${NAME} ${KIND}: {k[:64]} calls {v.get('rocprof_calls')} avg {v.get('rocprof_avg_kernel_ns', 0) / 1e3:.1f} us -> {v.get('achieved_GBps_from_rocprof', 0) / 8000:.4f} of 8 TB/s; "
              f"HBM {v['hbm_bytes_per_launch'] / 1e9:.3f} GB / launch = {v['hbm_bytes_per_launch'] / v['alg_bytes_per_launch']:.3f} x algorithmic")
PY
