#!/bin/bash
# rocprofv3 evidence for the kernels of SURVEY 8(f): kernel stats + FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of
#   f1  binned splines      scripts/bench_rows.py binned      (k_binned_weights(_f64), the binned event kernel in its list forms)
#   f2  parameter handler   scripts/par_rows_loop.py 1040     (k_par_stage / k_par_rows / k_par_finish)
#   f4  dynamic kinematics  scripts/f4_ragged.py 10000000 c3  (k_rebin_lds beside the brick kernel)
# usage (on the GPU box, repo root): bash scripts/profile_f_rows.sh [f1 f2 f4]   -> gpurun_out/rNN_<row>_ (NN = M3_ROUND, default 06){kernel_stats.csv,pmc.json,run.log}
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
ROWS=${@:-f1 f2 f4}
R=r$(printf %02d ${M3_ROUND:-6})
for ROW in $ROWS; do
  case $ROW in
    f1) CMD="python3 $ROOT/scripts/bench_rows.py binned";;
    f2) CMD="python3 $ROOT/scripts/par_rows_loop.py 1040 2000";;
    f4) CMD="python3 $ROOT/scripts/f4_ragged.py ${F4_EVENTS:-10000000} ${F4_CONFIG:-c3}";;
    *) echo "unknown row $ROW"; continue;;
  esac
  D=$OUT/prof_$ROW; rm -rf $D; mkdir -p $D
  cd /tmp
  echo "== $ROW kernel stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o $ROW -- $CMD > $OUT/${R}_${ROW}_run.log 2> $D/stats.log || tail -3 $D/stats.log
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $ROW pmc $C"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D/$C -o $ROW -- $CMD > /dev/null 2> $D/$C.log || tail -3 $D/$C.log
  done
  cd $ROOT
  cp "$(find $D/stats -name '*kernel_stats.csv' | head -1)" $OUT/${R}_${ROW}_kernel_stats.csv 2>/dev/null
  python3 - <<PY
import csv, glob, collections, json, os
d = "$D"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
for k, cs in acc.items():
    e = res.setdefault(k, {})
    for c, v in cs.items():
        e[c + "_KB_avg"] = sum(v) / len(v)
    if "FETCH_SIZE_KB_avg" in e and "WRITE_SIZE_KB_avg" in e:
        # gfx950: FETCH_SIZE counts 64-B units as KB of 32-B (MI355X_MICROARCH.md): bytes = 2 x FETCH_SIZE x 1024
        e["hbm_bytes_per_launch"] = 2 * e["FETCH_SIZE_KB_avg"] * 1024 + e["WRITE_SIZE_KB_avg"] * 1024
        if e.get("avg_us"):
            e["hbm_TBps"] = e["hbm_bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e12
json.dump({"row": "$ROW", "command": "$CMD".replace("$ROOT/", ""), "kernels": res}, open("$OUT/${R}_${ROW}_pmc.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0))[:8]:
    print(f"  {k[:70]:70s} calls {v.get('calls', 0):6d} avg {v.get('avg_us', 0):8.1f} us  hbm {v.get('hbm_bytes_per_launch', 0) / 1e6:8.2f} MB/launch  {v.get('hbm_TBps', 0):.2f} TB/s")
PY
  tail -3 $OUT/${R}_${ROW}_run.log
done
