#!/bin/bash
# The bench line's configuration (BASELINE configs[2]) under rocprofv3 on THIS box, with the in-run `roofline.frac` of the same
# process and the box's clocks beside it: is a figure below or above 0.70 the box or the sample of launches?
#   bash scripts/profile_c3_box.sh <tag> [steps=2000]     writes gpurun_out/<tag>_{stats.csv,bench.json,box.txt}
set -u
TAG=$1; STEPS=${2:-2000}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
{ echo "== $(date -u) $(hostname)"; rocm-smi --showclocks --showperflevel --showpower 2>&1 | grep -v '^=*$' | head -40; } > $OUT/${TAG}_box.txt
# the clocks UNDER LOAD: sampled beside the run (an idle card reports its low-power levels)
( for i in $(seq 1 80); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | sed 's/GPU\[0\]\s*: //' | tr '\n' ';')"; sleep 1; done ) > $OUT/${TAG}_clocks.txt 2>&1 &
SAMPLER=$!
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o $TAG -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra --steps $STEPS --warmup 20 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_prof.log || tail -5 $OUT/${TAG}_prof.log
cd $ROOT
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
{ echo "== under load (1 s samples; the bench's timed region is its last second or two)"; sed 's/^t=[0-9]* //' $OUT/${TAG}_clocks.txt | grep -oE 'sclk[^;]*;|mclk[^;]*;|Power \(W\): [0-9.]*' | sort | uniq -c | sort -rn | head -12; } >> $OUT/${TAG}_box.txt
{ echo "== after"; rocm-smi --showclocks --showperflevel --showpower 2>&1 | grep -v '^=*$' | head -40; } >> $OUT/${TAG}_box.txt
F=$(find $OUT/${TAG}_prof -name '*kernel_stats.csv' | head -1)
cp "$F" $OUT/${TAG}_stats.csv 2>/dev/null
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/${TAG}_stats.csv")))
line = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
alg = line["roofline"]["alg_bytes_per_launch"]
for r in rows:
    if "k_reweight" in r["Name"]:
        avg = float(r["AverageNs"])
        print(f"rocprofv3: {r['Name'][:60]} calls {r['Calls']} avg {avg / 1e3:.1f} us -> {alg / avg / 8000:.4f} of 8 TB/s; in-run events of the same process: "
              f"{line['roofline']['avg_kernel_ms'] * 1e3:.1f} us over {line['roofline']['launches_timed']} launches -> {line['roofline']['frac']:.4f}; {line['value']:.0f} steps/s")
PY
