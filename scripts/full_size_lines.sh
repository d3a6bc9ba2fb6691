#!/bin/bash
# The bench line of every BASELINE configuration at full size on one GPU (the parity-test configurations beside the bench's own):
#   bash scripts/full_size_lines.sh <round tag, e.g. r05>   ->  gpurun_out/<tag>_c{1,2,4,5}_full_bench.json
TAG=${1:-r05}
for C in c1 c2 c4 c5; do
  echo "== $C $(date -u +%T)"
  timeout -k 10 900 python3 bench.py --config $C --no-cpu-baseline --no-extra --steps 60 --warmup 5 > gpurun_out/${TAG}_${C}_full_bench.json 2> gpurun_out/${TAG}_${C}_full_bench.err || { tail -3 gpurun_out/${TAG}_${C}_full_bench.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_${C}_full_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$C", round(d["value"], 1), "steps/s", round(d["ms_per_step"], 4), "ms/step; kernel", round(r["avg_kernel_ms"] * 1e3, 1), "us ->", round(r["frac"], 4))
PY
done
