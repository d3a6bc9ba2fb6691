"""Turn the rocprofv3 PMC passes (one --pmc counter per run) + the kernel stats of the bench command into
profiles/rNN_c2_pmc_vK.json.  usage: make_pmc_json.py <fetch_dir> <write_dir> <kernel_stats.csv> <out.json> <version text>"""
import csv, glob, json, os, sys, collections

fetch_dir, write_dir, stats_csv, out, version = sys.argv[1:6]
# optional: the bench JSON line of the profiled configuration (algorithmic bytes, events, workload text)
bench = json.load(open(sys.argv[6])) if len(sys.argv) > 6 else None
K = collections.defaultdict(dict)
for d, name in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        K[k][f"{name}_KB_avg"] = sum(v) / len(v)
        K[k][f"launches_{name}"] = len(v)
stats = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
ALG = bench["roofline"]["alg_bytes_per_launch"] if bench else 16 * 128_000_000 + 8 * 1_000_000
for k, v in K.items():
    if "k_reweight" in k and "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v:
        v["hbm_bytes_per_launch"] = 2 * v["FETCH_SIZE_KB_avg"] * 1024 + v["WRITE_SIZE_KB_avg"] * 1024
        v["correction"] = ("gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM "
                           "section) -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for stores/atomics")
        v["alg_bytes_per_launch"] = ALG
        if k in stats:
            ns = float(stats[k]["AverageNs"])
            v["rocprof_avg_kernel_ns"] = ns
            v["rocprof_calls"] = int(stats[k]["Calls"])
            v["achieved_GBps_from_rocprof"] = ALG / ns
json.dump({"round": int(os.environ.get("M3_ROUND", "2")), "version": version, "events_per_gpu": bench["config"]["events_per_gpu"] if bench else 1000000,
           "config": bench["config"]["workload"] if bench else "c2 (1e6 events, 128 TSpline3 x 7 knots, dense), 1x MI355X",
           "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 2 --spinup-seconds 0 "
                      "--no-cpu-baseline (one pass per counter); kernel stats: rocprofv3 --kernel-trace --stats -- python3 bench.py",
           "kernels": K}, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in K.items() if "k_reweight" in k}, indent=1))
