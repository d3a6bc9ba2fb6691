#!/usr/bin/env python3
"""bench.py -- MCMC-step (LLH evaluation) throughput of the MI355X likelihood engine.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms run N ranks: started WITHOUT a launcher (WORLD_SIZE unset) and N > 1, this process only starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a
child -- decided before torch or anything else that could touch a GPU is imported -- and relays rank 0's JSON line and the
exit code.  Under a launcher whose WORLD_SIZE differs from --gpus it exits with code 2: a line must never report another N
than the one it was asked for.  `config.ranks_seen` is the sum of ones over the ranks that took part.

A step = one pass of the hot path over one batch of synthetic MC:
    new parameter vector -> SampleHandlerFD::Reweight() -> GetLikelihood() -> LLH on the host
(the pair MR2T2::ProposeStep runs per proposal, ref: Fitters/MR2T2.cpp:66-76), timed
DragRace-style (ref: Fitters/FitterBase.cpp:461-489).

Workload.  EVERY N runs BASELINE.json configs[2] ("c3": 1e7 events, 512 mixed TSpline3/TF1 systematics at
occupancy 1/16, 200 bins, Barlow-Beeston) -- the largest configuration BASELINE sizes for one GPU and
the one north_star states its 1-GPU and its 8-GPU target on.  N > 1: that ONE sample sharded over the ranks
by events (strong scaling: `value` at N = 1 is the one-GPU line, and value(N) / value(1) is the curve);
shard boundaries balance the responses per rank (mach3_amd/dist.py).  Every step ends in ONE in-library RCCL
all-reduce of the bin vector.  value = K / t: LLH evaluations (MCMC steps) per second of the whole sample.
N >= 4 additionally measures configs[3] ("c4": 4 samples x 1e7 events, 800 bins -- BASELINE sizes it for eight GPUs)
the same way, AFTER the headline numbers are in hand and behind a watchdog, under `extra.configs[3]` (its own
ms_per_step, roofline and event_steps_per_s).  `--config` overrides the headline workload.

Prints ONE JSON line on rank 0 (contract in the task description), carrying `roofline` (fused
reweight kernel: algorithmic bytes / dispatch-stamped launch time vs 8 TB/s; traffic from the committed
PMC passes), at N = 1 `cpu_baseline` (the OpenMP oracle port timed on this host on a bounded sample of
the same configuration, arrays first-touched inside OpenMP loops, best of a thread sweep) and `extra`
(the configs[1] line of round 1, measured the same way).  Nothing under `extra` can cost a run its line: a leg that does not
return within its limit is reported as text in its place; `--strict` turns such a leg (or a failed one) into exit code 6.  `config.infinity_cache` says what rank 0's brick kernel keeps in the
256 MB Infinity Cache from step to step (m3hip_info::cached_stream_bytes, tf1_codes_around).
"""
import argparse
import json
import os
import sys
import time


def _gpus_arg(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: run the N ranks as fresh children of THIS process
    (which never imports torch, never touches a GPU and never execs), pass rank 0's line through and leave with the
    children's exit code.  Under a launcher, refuse a WORLD_SIZE that is not --gpus."""
    try:
        want = _gpus_arg(sys.argv[1:])
    except ValueError:
        return  # argparse reports it
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != want and os.environ.get("M3_BENCH_FORCE_COMM") != "1":
            print(f"bench.py: --gpus {want} but the launcher started WORLD_SIZE={world} ranks; refusing to report a line for another N", file=sys.stderr)
            raise SystemExit(2)
        return
    if want <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(want), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {want} without a launcher: starting {want} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # stderr is inherited: progress shows as it happens.  Rank 0's JSON lines are passed on AS THEY COME (the headline line before
    # the legs under `extra`, the complete one at the end), so that a caller who stops this process early still has the first
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    seen = 0
    for ln in p.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            print(ln.rstrip("\n"), flush=True)
            seen += 1
    rc = p.wait()
    if not seen and rc == 0:
        print("bench.py: the ranks ended without a JSON line", file=sys.stderr)
        raise SystemExit(5)
    raise SystemExit(rc)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
    pass  # (handled below, once the helpers are defined: no launcher, no torch)
elif __name__ == "__main__":
    _self_launch()

# host thread pools sized to one GPU's CPU share (the reference recommends 8, allows 16 threads:
# Manager/Monitor.cpp:122-138); must be set before any OpenMP runtime initialises
_local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
if _local_world > 1:
    # one rank per GPU under torch.distributed.run: it exports OMP_NUM_THREADS=1 unless the caller chose a value, which would
    # generate and lay out every rank's shard on one thread; take one GPU's share of the host instead and leave the threads
    # unbound (bound "close", every rank's team would sit on the same first cores)
    if os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or 1) // _local_world)))
else:
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
CFG_INDEX = {"c1": 0, "c2": 1, "c3": 2, "c4": 3, "c5": 4}
STAT_NAME = {0: "Poisson", 1: "Barlow-Beeston", 2: "IceCube", 3: "Pearson", 4: "Dembinski-Abdelmotteleb"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(config_name, per_gpu):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_<config>_pmc_*.json: FETCH_SIZE and WRITE_SIZE collected in separate runs and
    corrected as MI355X_MICROARCH.md prescribes).  The newest profile of this configuration at this number of events per
    GPU wins; failing that, the newest profile of the configuration at another size, scaled by events (the store and its
    traffic are linear in the events of one configuration; the line says so: `traffic_scaled_from_events`).  None when no
    profile of the configuration exists."""
    import glob
    exact, other = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config_name}_pmc_*.json"))):
        try:
            d = json.load(open(f))
            for k, v in d["kernels"].items():
                if "k_reweight" in k and "hbm_bytes_per_launch" in v:
                    ev = d.get("events_per_gpu", 1_000_000)
                    rec = {"bytes": v["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT), "round": d.get("round")}
                    if ev == per_gpu:
                        exact = rec
                    elif ev > 0:
                        other = dict(rec, bytes=v["hbm_bytes_per_launch"] * per_gpu / ev, scaled_from_events=ev)
        except Exception:
            pass
    return exact or other


EVENT_EVERY = int(os.environ.get("M3_BENCH_EVENT_EVERY", "2"))  # a 20-step driver run times 10 launches


def cpu_baseline_child(cfg_name, n_events_total, sample_events, threads_csv, seconds_per_point):
    """`python bench.py --cpu-baseline-child ...`: one binding's thread sweep of the OpenMP port in a process of its own -- the binding
    (OMP_PROC_BIND / OMP_PLACES) is fixed when the OpenMP runtime starts, so every binding needs its process; no torch, no GPU here.
    For EVERY team size the oracle is built afresh with that team: its arrays are first touched by the threads that will stream them
    (m3o_monolith_localise inside OpenMP loops of the compute loops' static schedule) -- a team timed on pages a wider team placed
    reads most of them from the wrong memory node.  Prints one JSON list."""
    # the parent's main thread may be bound to one core (its own OpenMP runtime, started under OMP_PROC_BIND, pins it), and a child
    # inherits that mask: every team of this process would then share that core.  Back to every CPU the host allows, BEFORE an
    # OpenMP runtime is loaded here (libgomp reads the mask when it is loaded)
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from mach3_amd import synth
    base = synth.CONFIGS[cfg_name]
    spec = base.scaled(n_events_total) if n_events_total != base.n_events else base
    n = min(sample_events, spec.n_events)
    lib = ob.load("omp")
    runs = []
    for threads in [int(t) for t in threads_csv.split(",")]:
        lib.m3o_set_num_threads(threads)
        mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, 0, n, variant="omp", localise=True)
        samp.reweight()
        samp.add_data(samp.mc)
        for lap in range(1, 3):  # warm-up
            par[:] = synth.params(spec, lap)
            samp.reweight()
            samp.get_likelihood()
        laps, t_spl, t_fill, t_llh = 0, 0.0, 0.0, 0.0
        t0 = time.perf_counter()
        while True:
            par[:] = synth.params(spec, 100 + laps)
            samp.reweight()
            t1 = time.perf_counter()
            samp.get_likelihood()
            t_llh += time.perf_counter() - t1
            a, b = samp.last_timing()
            t_spl += a
            t_fill += b
            laps += 1
            el = time.perf_counter() - t0
            if el > seconds_per_point or laps >= 4000:
                break
        runs.append({"threads": int(lib.m3o_max_threads()), "laps": laps, "seconds": el, "steps_per_s": laps / el * n / spec.n_events,
                     "ms_per_lap": 1e3 * el / laps, "spline_eval_ms": 1e3 * t_spl / laps, "fill_ms": 1e3 * t_fill / laps, "llh_ms": 1e3 * t_llh / laps})
        samp.close()
        mono.close()
    print(json.dumps(runs), flush=True)


def cpu_baseline(spec, budget_s=18.0, sample_events=300_000, full_lap=False, confirm_events=2_400_000):
    """Reference-layout OpenMP port (oracle/libm3oracle_omp.so) on the first `sample_events` events of the same configuration;
    steps/s scaled to the full event count.  Two bindings, each a child process of its own (cpu_baseline_child): threads packed on
    neighbouring cores (OMP_PROC_BIND=close) and spread over the sockets (spread: all memory links), OMP_PLACES=cores; within a
    binding a sweep over team sizes -- 8, 16 (the most the reference uses unless OMP_NUM_THREADS is exported,
    Manager/Monitor.cpp:122-138), 32, 64 and the physical cores as far as the host has them -- every team on arrays it first-touched
    itself.  `value` is the best point of all, with its binding stated; the reference's 16-thread point is beside it.  Everything in
    `sample` was measured by this run.  `full_lap` (--cpu-full-lap, off by default: minutes of host time and ~25 GB): one lap of ALL
    events on this host (scripts/cpu_full_lap.py) beside the event-scaled figure."""
    import subprocess
    n = min(sample_events, spec.n_events)
    cores = os.cpu_count() or 1
    physical = max(1, cores // 2) if cores >= 32 else cores  # SMT hosts of the pool: two hardware threads per core
    sweep = sorted({t for t in (8, 16, 32, 64, min(physical, 128)) if t <= min(cores, physical)} or {cores})
    plans = [("close", sweep)]
    if cores >= 32:
        plans.append(("spread", [t for t in sweep if t >= 16]))
    per_point = budget_s / sum(len(p[1]) for p in plans)
    base_name = spec.name.split("@")[0]
    runs = []
    for bind, threads in plans:
        env = dict(os.environ, OMP_PROC_BIND=bind, OMP_PLACES="cores", OMP_NUM_THREADS=str(max(threads)))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", base_name, str(spec.n_events), str(n),
                            ",".join(str(t) for t in threads), f"{per_point:.2f}"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"cpu baseline child ({bind}): {r.stderr[-400:]}")
        for x in json.loads(r.stdout.strip().splitlines()[-1]):
            runs.append(dict(x, bind=bind))
    sweep_best = max(runs, key=lambda r: r["steps_per_s"])
    at16 = max((r for r in runs if r["threads"] == 16), key=lambda r: r["steps_per_s"], default=None)
    # The sweep's sample is small (configs[2]: 0.6 GB in the reference's layout) and, with the threads spread over the sockets, sits
    # in the host's L3 slices: its steps/s scaled by events would flatter the CPU (the full sample is 18 GB and streams from DRAM).
    # So the sweep only RANKS the (binding, team) points; the two best are then timed on a sample eight times as large, too large
    # for any cache, and `value` is the better of those, scaled by events.
    big_n = min(spec.n_events, max(n, confirm_events))
    confirmed = []
    if big_n > n:
        ranked = sorted(runs, key=lambda r: -r["steps_per_s"])
        cands = [ranked[0]] + [r for r in ranked[1:] if r["bind"] != ranked[0]["bind"]][:1]
        for c in cands:
            env = dict(os.environ, OMP_PROC_BIND=c["bind"], OMP_PLACES="cores", OMP_NUM_THREADS=str(c["threads"]))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", base_name, str(spec.n_events), str(big_n),
                                str(c["threads"]), "2.0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            if r.returncode == 0:
                confirmed.append(dict(json.loads(r.stdout.strip().splitlines()[-1])[0], bind=c["bind"]))
    best = max(confirmed, key=lambda r: r["steps_per_s"]) if confirmed else sweep_best
    best_n = big_n if confirmed else n
    out = {
        "value": best["steps_per_s"],
        "unit": "steps/s",
        "cores": best["threads"],
        "kind": "port",
        "binding": f"OMP_PROC_BIND={best['bind']} OMP_PLACES=cores",
        "sample": f"first {best_n} of {spec.n_events} events of the same configuration, {best['laps']} laps in {best['seconds']:.1f} s at "
                  f"{best['threads']} threads bound {best['bind']} ({best['ms_per_lap']:.2f} ms/lap: spline eval {best['spline_eval_ms']:.2f} ms, fill "
                  f"{best['fill_ms']:.2f} ms, LLH {best['llh_ms']:.3f} ms); steps/s scaled by {best_n}/{spec.n_events}; arrays first-touched by the "
                  f"timed team in OpenMP loops; the (binding, team) point chosen by a sweep on the first {n} events (" +
                  "; ".join(f"{b} {', '.join(str(t) for t in ts)}" for b, ts in plans) + f", {per_point:.1f} s per point, every team on arrays it first-touched), "
                  f"whose own event-scaled figures are not `value`: that sample fits the host's caches",
        "sweep_events": n,
        "steps_per_s_by_threads": {f"{r['bind']}:{r['threads']}": r["steps_per_s"] for r in runs},
        "confirmed_on_events": best_n,
        "confirmed": {f"{r['bind']}:{r['threads']}": r["steps_per_s"] for r in confirmed},
        "host_cores": cores,
    }
    if at16 is not None:  # the reference's own thread cap (Manager/Monitor.cpp:122-138)
        out["at_16_threads"] = {"value": at16["steps_per_s"], "unit": "steps/s", "ms_per_lap": at16["ms_per_lap"], "laps": at16["laps"], "bind": at16["bind"],
                                "on_events": n}  # (the sweep's sample)
    if full_lap:
        env = dict(os.environ, OMP_PROC_BIND=best["bind"], OMP_PLACES="cores", OMP_NUM_THREADS=str(best["threads"]))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu_full_lap.py"), base_name, str(best["threads"])], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("full size") or ln.startswith("ratio")]
        out["full_lap_this_host"] = {"threads": best["threads"], "bind": best["bind"], "output": lines or r.stdout[-400:]}
    elif spec.name == "c3" and spec.n_events == 10_000_000:
        # NOT measured by this run: one full-size lap on record, which backs the scaling by events (another host, round 3)
        out["full_lap_on_record"] = {"file": "profiles/r03_cpu_full_c3.log", "ms": 161.6, "threads": 16, "ratio_to_event_scaled_sample": 0.970,
                                     "host": "the box of that log (round 3), not this run's; --cpu-full-lap measures one on this host"}
    return out


def shard_of(spec, rank, n_ranks, tdist=None):
    """[e0, e1) of this rank: contiguous blocks on 1024-event boundaries, balanced by responses.  The
    weights come from the generator's own counts in blocks of 65536 events.  With a process group (`tdist`) rank 0
    counts once and broadcasts the cuts (eight ranks counting 4e7 events each is eight times the set-up work on one
    host); without one every rank computes the same cuts by itself (event data are a pure function of the event number)."""
    from mach3_amd import dist, synth
    if n_ranks == 1:
        return 0, spec.n_events
    if tdist is not None:
        cuts = [None]
        if rank == 0:
            cuts = [[shard_of(spec, r, n_ranks) for r in range(n_ranks)]]
        tdist.broadcast_object_list(cuts, src=0)
        return tuple(cuts[0][rank])
    blk = 65536
    edges = list(range(0, spec.n_events, blk)) + [spec.n_events]
    w = np.empty(spec.n_events, np.float32)

    def fill(ab):
        a, b = ab
        ns, nt = synth.count(spec, a, b)  # a ctypes call: runs without the interpreter lock
        w[a:b] = (ns + nt) / max(b - a, 1)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 8) // max(n_ranks, 1)))) as pool:
        list(pool.map(fill, zip(edges[:-1], edges[1:])))
    return dist.shard_ranges(spec.n_events, n_ranks, weights=w)[rank]


def measure(spec, rank, n_ranks, local_rank, args, dist, torch, allow_fallback, oneshot=True):
    """Build this rank's shard, run warm-up + the timed steps, return the numbers of the JSON line."""
    from mach3_amd import synth
    from mach3_amd.engine import build_from_synth

    e0, e1 = shard_of(spec, rank, n_ranks, dist if n_ranks > 1 else None)
    t_setup = time.perf_counter()
    eng, mono, sh, par = build_from_synth(spec, device_id=local_rank, chunk_events=args.chunk, event_range=(e0, e1), order_by_osc=getattr(args, "order_key", False))
    info = eng.info()
    log(f"[bench] rank {rank}: {spec.name} events [{e0}, {e1}) of {spec.n_events}, P={spec.n_params}, S={info.n_splines}, "
        f"T={info.n_tf1}, coeff store {info.coeff_bytes / 1e9:.2f} GB, tiles {info.n_tiles_uniform}u/{info.n_tiles_brick}b, "
        f"setup {time.perf_counter() - t_setup:.1f} s")

    comm_mode = "none"
    force_comm = os.environ.get("M3_BENCH_FORCE_COMM") == "1"  # rehearsal: run the N > 1 plumbing with one rank
    nccl = os.environ.get("M3_BENCH_BACKEND", "nccl") == "nccl"
    fallback = False
    if (n_ranks > 1 or force_comm) and nccl:
        # one in-library RCCL all-reduce per step; torch.distributed only carries the unique id.  ncclCommInitRank is
        # collective: the ranks first agree that every one of them can load RCCL, or none of them enters it
        ok = torch.tensor([1.0 if eng.comm_available() else 0.0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            uid = [None]
            if rank == 0:
                try:
                    uid = [eng.comm_unique_id()]
                except Exception as e:
                    log(f"rank 0: ncclGetUniqueId failed ({e})")
            dist.broadcast_object_list(uid, src=0)  # every rank takes part, whatever rank 0 got
            ok = torch.zeros(1, device="cuda")
            if uid[0] is not None:
                try:
                    eng.comm_init(uid[0], rank, n_ranks)
                    ok = torch.ones(1, device="cuda")
                except Exception as e:
                    log(f"rank {rank}: in-library RCCL init failed ({e})")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            comm_mode = "rccl-allreduce(f64 bin vector) in-library, one per step"
        else:
            # a communicator that only some ranks joined would dead-lock the first reweight: every rank drops its own
            eng.comm_destroy()
            if not allow_fallback:
                log("in-library RCCL is not available on every rank and --allow-host-fallback was not given: no number is "
                    "better than the number of a path that round-trips the bins through the host")
                raise SystemExit(3)
            fallback = True
            comm_mode = "torch.distributed all_reduce of the bin vector through the host (FALLBACK, --allow-host-fallback)"
    elif n_ranks > 1 and os.environ.get("M3_BENCH_EXCHANGE") == "oneshot":
        # rehearsal on one card: RCCL refuses two ranks on one device, the one-shot exchange does not need it.  The handles
        # travel through the rehearsal backend (gloo); timing means nothing with two ranks on one GPU, --check does.
        handles = [None] * n_ranks
        dist.all_gather_object(handles, eng.xch_create(rank, n_ranks))
        eng.xch_open(handles)
        comm_mode = "one-shot peer stores through IPC-mapped buffers (m3hip_xch_*), rehearsal: ranks share one card"
    elif n_ranks > 1:
        comm_mode = "none (rehearsal backend: ranks do not exchange bins)"

    w2_tot = [None]

    def reduce_bins():
        """fallback only: sum this rank's partial bins over ranks through torch.distributed"""
        mc, w2 = sh._bins()
        if w2_tot[0] is None:  # w2 is frozen after the first Reweight (SampleHandlerFD.cpp:342): reduce it once
            t2 = torch.from_numpy(w2).cuda()
            dist.all_reduce(t2)
            w2_tot[0] = t2.cpu().numpy()
        t = torch.from_numpy(mc).cuda()
        dist.all_reduce(t)
        tot = np.ascontiguousarray(t.cpu().numpy())
        eng._chk(eng.lib.m3hip_debug_set_bins(eng.h, tot.ctypes.data, w2_tot[0].ctypes.data, len(tot)))
        return tot

    # Asimov data at nominal (ref: Fitters/MaCh3Factory.h:142-153); freezes W2 as the reference does
    par[:] = synth.params(spec, 0)
    sh.Reweight()
    sh.AddData(reduce_bins() if fallback else sh.GetMCArray())
    assert abs(sh.GetLikelihood()) < 1e-9

    laps = [synth.params(spec, 1 + i) for i in range(64)]

    # The step of a fit that MOVES oscillation parameters (--table host | device): the weight table is replaced before every
    # reweight, as OscillationHandler::Evaluate refills it inside every Reweight of the reference (SampleHandlerFD.cpp:323).  Eight
    # tables of the same values with every entry different from lap to lap; "device": they lie in device memory the engine
    # does not own (where a GPU calculator leaves them) and are read in place, "host": they come through the pinned staging
    # copy.  The oscillation calculation itself is not part of the path (BASELINE: probabilities pre-staged).
    table = getattr(args, "table", "resting")
    move = None
    if table != "resting":
        base_tbl = synth.osc_table(spec)
        tables = [np.ascontiguousarray(np.roll(base_tbl, 977 * k)) for k in range(8)]
        if table == "device":
            d_tables = [torch.from_numpy(t).cuda() for t in tables]
            torch.cuda.synchronize()
            ptrs = [t.data_ptr() for t in d_tables]
            move = lambda i: sh.SetOscillationWeightsDevice(ptrs[i % 8], spec.osc_n)  # noqa: E731
        else:
            move = lambda i: sh.SetOscillationWeights(tables[i % 8])  # noqa: E731

    def step(i):
        par[:] = laps[i % len(laps)]
        if move is not None:
            move(i)
        if not fallback:
            return sh.Step()
        sh.Reweight()  # partial bins of this rank's shard
        reduce_bins()
        return sh.GetLikelihood()

    # The exchange validates itself on the untimed steps: every rank's partial bin vector is all-gathered and summed on
    # the host in rank order, and the all-reduced vector must agree per bin (m3hip_comm_set_validate); a mismatch fails
    # the step.  Off again for the timed region (it costs an all-gather and a device-to-host copy per step).
    in_library = comm_mode.startswith("rccl")
    if in_library:
        eng.comm_set_validate(True)
    # torch's own device context is created by its first cuda call; with one rank that would be the synchronize() that opens the
    # timed region, and what the runtime does in the background after it (observed on config 5's shard: one stall of ~30 ms inside
    # the first timed steps) belongs in front of the untimed spin-up steps instead
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    n_spin = 0

    def keep_spinning():
        # every step is a collective: the ranks must agree on how many there are, so rank 0's clock decides for all of them
        go = time.perf_counter() - t_spin < args.spinup_seconds
        if dist is not None and n_ranks > 1:
            flag = torch.tensor([1.0 if go else 0.0], device="cuda")
            dist.broadcast(flag, 0)
            go = float(flag.item()) > 0
        return go

    # the kernel's time over the UNTIMED steps as well (thousands of launches: what a profiler averages over), beside the figure of
    # the timed region's few: `roofline.untimed_steps`
    eng.timing_enable(EVENT_EVERY)
    eng.timing_reset()
    while keep_spinning():
        step(n_spin)
        n_spin += 1
    for i in range(max(args.warmup, 2 if in_library else 0)):
        step(i)
    k_ms_untimed, k_n_untimed = eng.timing_read(0)
    comm_val = None
    if in_library:
        rel, n_val = eng.comm_last_validation()
        eng.comm_set_validate(False)
        v = torch.tensor([rel, float(n_val)], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(v[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(v[1:], op=dist.ReduceOp.MIN)
        comm_val = {"validated_steps": int(v[1].item()), "max_rel": float(v[0].item())}
    eng.timing_enable(EVENT_EVERY)  # hipEvents around every EVENT_EVERY-th launch of the timed region
    eng.timing_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    llh = 0.0
    for i in range(args.steps):
        llh = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    k_ms, k_n = eng.timing_read(0)
    info = eng.info()  # (again: with what the reweights planned for the Infinity Cache)
    gap_ms, gap_n = eng.timing_read(2) if in_library else (0.0, 0)
    eng.timing_enable(False)
    k_all = [k_ms]
    if dist is not None and n_ranks > 1:
        t = torch.zeros(n_ranks, dtype=torch.float64, device="cuda")
        t[rank] = k_ms
        dist.all_reduce(t)
        k_all = [float(x) for x in t.cpu()]
    if os.environ.get("M3_BENCH_PHASES") == "1":  # diagnostics: where the host side of a step spends its time (untimed extra steps)
        import ctypes as C
        ph, acc = (C.c_double * 6)(), np.zeros(6)
        t_ph = time.perf_counter()
        for i in range(40):
            step(2000 + i)
            eng.lib.m3hip_debug_host_times(eng.h, ph)
            acc += np.array(list(ph))
        log(f"[bench] rank {rank}: 40 extra steps {(time.perf_counter() - t_ph) / 40 * 1e6:.1f} us each; host phases per step (us): "
            "stage params, launch prep, launch reweight, launch llh, wait, collect = " + ", ".join(f"{x / 40 * 1e6:.1f}" for x in acc))
    check = None
    if args.check:  # LLH of a few laps, for the comparison with a one-rank run of the same sample
        check = []
        for i in range(3):
            check.append(step(1000 + i))
    # the one-shot leg runs AFTER the caller has the line's numbers in hand (main() guards it with a watchdog: nothing that
    # happens in an `extra` may cost the run its line)
    after = None
    if oneshot and in_library and os.environ.get("M3_BENCH_ONESHOT", "1") != "0":
        after = lambda: oneshot_leg(eng, step, rank, n_ranks, dist, torch, args)  # noqa: E731
    out = dict(el=el, llh=llh, k_ms=k_ms, k_n=int(k_n), k_ms_untimed=k_ms_untimed, k_n_untimed=int(k_n_untimed), info=info, comm=comm_mode, e0=e0, e1=e1, check=check, comm_val=comm_val, oneshot=None,
               comm_us=1e3 * gap_ms if gap_n else None, k_all=k_all, after=after, close=eng.close)
    if after is None:
        eng.close()
    return out


def oneshot_leg(eng, step, rank, n_ranks, dist, torch, args):
    """After the timed region of an N > 1 (or rehearsal) run: the same steps once more through the ONE-SHOT exchange
    (m3hip_xch_*: every rank stores its bin vector into every rank's IPC-mapped buffer, rank-ordered sum) -- its
    likelihoods against the RCCL path's on the same laps, its step time and the time it adds between the reweight kernel
    and the bin likelihood.  Reported under `extra`, never as `value`.  Any failure is reported as text: a peer that does
    not push times out after 2 s on the device (nothing hangs), and the ranks agree after every phase whether all of them
    are still in it, so that no rank enters a collective the others have left."""
    res = {"exchange": "one-shot peer stores into IPC-mapped fine-grained buffers + rank-ordered sum (m3hip_xch_*)"}

    def agree(ok, what):  # every rank calls this the same number of times, whatever happened to it
        flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
        if dist is not None and n_ranks > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) <= 0:
            res.setdefault("error", f"{what} failed on another rank")
            return False
        return True

    def attempt(what, fn):
        try:
            return True, fn()
        except Exception as e:
            res["error"] = f"rank {rank}: {what}: {e}"
            return False, None

    def peers_reachable():  # one process per card: every other card must be able to map this one's buffer
        n_dev, me = torch.cuda.device_count(), torch.cuda.current_device()
        if n_ranks > 1 and n_dev >= n_ranks:
            blind = [d for d in range(n_ranks) if d != me and not torch.cuda.can_device_access_peer(me, d)]
            if blind:
                raise RuntimeError(f"device {me} has no peer access to devices {blind}")
        return True

    laps = [5000 + i for i in range(3)]
    ok, _ = attempt("peer access", peers_reachable)
    if not agree(ok, "the peer access check"):  # before the first collective step below: all ranks go on, or none
        return res
    ok, ref = attempt("RCCL reference laps", lambda: [step(i) for i in laps])
    ok, hdl = attempt("m3hip_xch_create", lambda: eng.xch_create(rank, n_ranks)) if ok else (False, None)
    handles = [hdl if hdl is not None else b""]
    if dist is not None and n_ranks > 1:
        handles = [None] * n_ranks
        dist.all_gather_object(handles, hdl if hdl is not None else b"")
    alive = agree(ok, "m3hip_xch_create")
    if alive:
        ok, _ = attempt("m3hip_xch_open", lambda: eng.xch_open(handles))
        alive = agree(ok, "m3hip_xch_open")
    if alive:
        ok, got = attempt("first steps through the one-shot exchange", lambda: [step(i) for i in laps])
        alive = agree(ok, "the first steps through the one-shot exchange")
    if alive:
        res["llh_max_rel_diff_vs_rccl"] = max(abs(a - b) / max(abs(b), 1.0) for a, b in zip(got, ref))
        n = max(10, min(args.steps, 100))

        def timed():
            eng.timing_enable(EVENT_EVERY)
            eng.timing_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                step(6000 + i)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        ok, el = attempt("timed steps through the one-shot exchange", timed)
        if agree(ok, "the timed steps through the one-shot exchange"):
            if dist is not None and n_ranks > 1:
                t = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            gap_ms, gap_n = eng.timing_read(2)
            res.update({"steps": n, "ms_per_step": 1e3 * el / n, "steps_per_s": n / el, "comm_us": 1e3 * gap_ms if gap_n else None,
                        "ok": bool(res["llh_max_rel_diff_vs_rccl"] <= 1e-9)})
    try:
        eng.timing_enable(False)
        eng.xch_destroy()
    except Exception:
        pass
    return res


def kernel_name(info):
    if info.n_tiles_brick and info.n_tiles_uniform:
        return "k_reweight_uniform<fill> + k_reweight_brick<fill> (spline eval + weight product + bin fill, fused; both launches timed together)"
    if info.n_tiles_brick:
        return "k_reweight_brick<fill> (ragged groups: spline eval + weight product + bin fill, fused)"
    return "k_reweight_uniform<fill> (spline eval + weight product + bin fill, fused)"


def workload_text(base, spec, n_ranks):
    idx = CFG_INDEX.get(base.name, "?")
    shard = "" if n_ranks == 1 else f", sharded by events over {n_ranks} GPUs (response-balanced contiguous blocks)"
    return (f"BASELINE configs[{idx}] ({base.name}): {spec.n_events} events{' in ' + str(spec.n_samples) + ' samples' if spec.n_samples > 1 else ''}, "
            f"{spec.n_params} {'TSpline3/TF1 (alternating)' if spec.mixed else 'TSpline3'} systematics x {spec.n_knots} knots "
            f"(occupancy {spec.occupancy:g}), {spec.n_norm} norm systematics, pre-staged osc table, {spec.total_bins} bins, "
            f"{STAT_NAME.get(int(spec.test_statistic), '?')} LLH{shard}")


def table_text(args, spec):
    t = getattr(args, "table", "resting")
    out = "" if t == "resting" else f"; oscillation table REPLACED EVERY STEP from {t} memory ({spec.osc_n} entries)"
    out += "; MC in oscillation-bin order" if spec.osc_sorted else ""
    return out + ("; events stored by the engine in oscillation-bin order (m3hip_monolith_order_key)" if getattr(args, "order_key", False) else "")


def osc_moves_leg(args, local_rank, torch):
    """`--osc-moves`: the step of a fit that moves oscillation parameters, beside the pre-staged one.  One GPU's share of configs[4]
    (1.25e7 events, 1 024 systematics, beam + atmospheric tables = 204 000 entries) and configs[2] (4 000 entries), each with the
    table resting, replaced every step from host memory and from device memory; configs[4]'s share also with the MC registered in
    oscillation-bin order (M3::OscillationBinOrder), where a wave's 64 events read one or two lines of the table instead of 64."""
    from mach3_amd import synth
    res = {"what": "kernel = fused reweight kernel by dispatch-stamped events; frac = algorithmic bytes / kernel / 8 TB/s; table replaced before every step"}
    shapes = [("configs[4] one GPU's share of 8", synth.CONFIGS["c5"].scaled(12_500_000), "c5", False),
              ("configs[4] share, stored by the engine in oscillation-bin order (m3hip_monolith_order_key)",
               synth.CONFIGS["c5"].scaled(12_500_000), "c5", True),
              ("configs[4] share, MC in oscillation-bin order", synth.CONFIGS["c5"].scaled(12_500_000, osc_sorted=True), "c5", False),
              ("configs[2]", synth.CONFIGS["c3"], "c3", False)]
    for name, spec, base_name, keyed in shapes:
        row = {"events": spec.n_events, "osc_entries": spec.osc_n}
        lines = None
        for tbl in ("resting", "device", "host"):
            a = argparse.Namespace(**vars(args))
            a.order_key = keyed  # (the engine holds keyed chunks back until 2^20 events are together: the upload's slicing does not matter)
            a.check, a.table, a.spinup_seconds, a.warmup = False, tbl, min(args.spinup_seconds, 0.5), max(args.warmup, 8)
            m = measure(spec, 0, 1, local_rank, a, None, torch, False)
            row[tbl] = {"ms_per_step": 1e3 * m["el"] / a.steps, "avg_kernel_ms": m["k_ms"], "launches_timed": m["k_n"],
                        "frac": m["info"].alg_bytes_fused / (m["k_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if m["k_ms"] > 0 else None}
            lines = round(float(m["info"].osc_lines_per_kevent), 1)
        row["osc_lines_per_kevent"] = lines
        res[name] = row
    return res


def roofline(base_name, m, per_gpu):
    info, k_ms = m["info"], m["k_ms"]
    tr = pmc_traffic(base_name, per_gpu)
    ach = info.alg_bytes_fused / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
    return {
        "bound": "hbm",
        "kernel": kernel_name(info),
        "achieved": ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS if ach else None,
        "traffic": tr["bytes"] if tr else None,
        "traffic_source": tr["source"] if tr else None,
        "traffic_measured_in_run": False,  # a constant from the committed rocprofv3 --pmc passes of this workload (counters cannot be read in-process)
        "traffic_profile_round": tr.get("round") if tr else None,
        "traffic_scaled_from_events": tr.get("scaled_from_events") if tr else None,  # not None: the profile is of this configuration at another size
        "alg_bytes_per_launch": int(info.alg_bytes_fused),
        "avg_kernel_ms": k_ms,
        "launches_timed": m["k_n"],
        # the same events over the untimed spin-up and warm-up steps of this process: many more launches, the average a profiler reports
        "untimed_steps": ({"avg_kernel_ms": m["k_ms_untimed"], "launches": m["k_n_untimed"],
                           "frac": info.alg_bytes_fused / (m["k_ms_untimed"] * 1e-3) / 1e9 / HBM_PEAK_GBS} if m.get("k_n_untimed") else None),
        "event_sampling": f"every {EVENT_EVERY}th launch of the timed region carries a hipEvent pair stamped with the dispatch's own start/end (hipExtLaunchKernelGGL); rank 0's shard",
    }


def main():
    # stdout carries exactly ONE line, the JSON: libraries that print banners there (RCCL's version block at communicator
    # init, for one) are sent to stderr for the whole run, and the line is written to the saved descriptor at the end
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="auto", help="c1..c5; auto = c3 (BASELINE configs[2]) for every N, with configs[3] under `extra` from 4 GPUs on")
    ap.add_argument("--no-c4-extra", action="store_true", help="N >= 4: skip the configs[3] leg under `extra`")
    ap.add_argument("--strict", action="store_true", help="exit with code 6 when a leg under `extra` failed or was abandoned (the line is printed either way)")
    ap.add_argument("--events", type=int, default=0, help="override the TOTAL number of events (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full-lap", action="store_true", help="cpu_baseline: also time ONE lap of all events on this host (scripts/cpu_full_lap.py; minutes, ~25 GB of host memory)")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] line under `extra`")
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--table", choices=["resting", "host", "device"], default="resting",
                    help="the oscillation table of the HEADLINE workload: resting = pre-staged once (BASELINE's configs; the default), host / device = "
                         "replaced before every step through m3hip_set_osc_table / m3hip_set_osc_table_device (profiling the moving-table step; the workload text says so)")
    ap.add_argument("--osc-entries", type=int, default=0, help="override the number of entries of the oscillation table (debug; the workload text says so)")
    ap.add_argument("--order-key", action="store_true", help="every upload chunk goes up with its events' oscillation indices as the order key (m3hip_monolith_order_key)")
    ap.add_argument("--osc-sorted", action="store_true", help="the MC registered in oscillation-bin order (SynthSpec.osc_sorted; the workload text says so)")
    ap.add_argument("--osc-moves", action="store_true",
                    help="N = 1: also measure the step of a fit that moves oscillation parameters -- one GPU's share of configs[4] (1.25e7 events, 204 000-entry table) and "
                         "configs[2], table resting / replaced every step from host memory / from device memory, MC in random and in oscillation-bin order -- under "
                         "`extra.osc_moves`, never as `value`")
    ap.add_argument("--check", action="store_true",
                    help="N > 1: rank 0 also runs the whole sample on its own GPU and compares the LLH of three laps (<= 1e-9 relative)")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="N > 1 without in-library RCCL: time the torch.distributed host round trip instead of failing")
    ap.add_argument("--spinup-seconds", type=float, default=2.0,
                    help="untimed steps before the warm-up: the first ~1 s of GPU work on a fresh process shows one-off "
                         "50-100 ms stalls (clock / runtime ramp-up) that would otherwise land in a short timed region")
    args = ap.parse_args()

    import torch  # device sync + torch.distributed (RCCL) plumbing only

    from mach3_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and os.environ.get("M3_BENCH_FORCE_COMM") != "1":  # (_self_launch has seen to it; kept for imports of main())
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        raise SystemExit(2)
    n_gpus = world
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if os.environ.get("M3_BENCH_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
        local_rank = local_rank % max(n_dev, 1)  # rehearsal: ranks share the card
    torch.cuda.set_device(local_rank)
    dist = None
    if world == 1 and os.environ.get("M3_BENCH_FORCE_COMM") == "1":  # one-rank rehearsal started without a launcher
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29571")):
            os.environ.setdefault(k, v)
    if world > 1 or os.environ.get("M3_BENCH_FORCE_COMM") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("M3_BENCH_BACKEND", "nccl")  # "gloo": plumbing rehearsal on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = args.config if args.config != "auto" else "c3"  # ONE workload for the whole --gpus series
    base = synth.CONFIGS[cfg]
    spec = base.scaled(args.events) if args.events else base
    if args.osc_sorted:
        spec = spec.scaled(spec.n_events, osc_sorted=True)
    if args.osc_entries:
        spec = spec.scaled(spec.n_events, osc_n=args.osc_entries, osc_sorted=spec.osc_sorted)

    m = measure(spec, rank, n_gpus, local_rank, args, dist, torch, args.allow_host_fallback)
    ranks_seen = 1
    if dist is not None:  # every rank that measured adds one
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        ranks_seen = int(round(float(one.item())))

    check = None
    if args.check and n_gpus > 1:
        # the same sample, unsharded, on rank 0's GPU alone (the other ranks wait at the barrier below)
        if rank == 0:
            a1 = argparse.Namespace(**vars(args))
            a1.steps, a1.warmup, a1.spinup_seconds = 1, 0, 0.0
            os.environ["M3_BENCH_FORCE_COMM"] = "0"
            one = measure(spec, 0, 1, local_rank, a1, None, torch, False)
            rel = max(abs(x - y) / max(abs(y), 1.0) for x, y in zip(m["check"], one["check"]))
            check = {"llh_sharded": m["check"], "llh_one_rank": one["check"], "max_rel_diff": rel, "ok": bool(rel <= 1e-9)}
        dist.barrier()

    out = None
    failed_extras = []  # legs under `extra` that failed or were abandoned (the line says so: `extra_failed`)
    if rank == 0:
        per_gpu = m["e1"] - m["e0"]
        out = {
            "metric": "LLH evaluations/sec (MCMC steps/sec)",
            "value": args.steps / m["el"],
            "unit": "steps/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * m["el"] / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload_text(base, spec, n_gpus) + table_text(args, spec),
                "total_events": spec.n_events,
                # which configuration `value` is of (rounds <= 3 switched to configs[3] at N >= 4; since round 4 every N runs configs[2])
                "headline": f"configs[{CFG_INDEX.get(base.name, '?')}]",
                "ranks_seen": ranks_seen,
                # the job's work rate in events x steps per second (comparable with `extra.configs[3]`, four times the events)
                "event_steps_per_s": spec.n_events * args.steps / m["el"],
                "events_per_gpu": per_gpu,
                "responses_rank0": int(m["info"].n_splines + m["info"].n_tf1),
                "comm": m["comm"],
                "last_llh": m["llh"],
                "brick_events": 1024 * int(m["info"].brick_groups or 1),
                # distinct 128-B lines of the oscillation table per 1024 consecutive events (m3hip_info; ~1: MC in oscillation-bin order)
                "osc_lines_per_kevent": round(float(m["info"].osc_lines_per_kevent), 1),
                # rank 0's brick kernel: MB of the small streams a step re-reads that the engine keeps cached, to find them in the
                # 256 MB Infinity Cache again next step, and whether the TF1 codes had to go around it
                "infinity_cache": {"cached_small_streams_MB": round(int(m["info"].cached_stream_bytes) / 1e6, 1),
                                   "tf1_codes_around": bool(m["info"].tf1_codes_around)},
            },
            "roofline": roofline(base.name, m, per_gpu),
        }
        if m["comm_val"] is not None:
            # self-validation of the exchange on the untimed steps (every rank; the worst bin of the worst rank), the time the
            # exchange adds between the reweight kernel's end and the bin likelihood's start, and every rank's kernel time
            out["config"]["comm_validated"] = bool(m["comm_val"]["validated_steps"] > 0)
            out["config"]["comm_validated_steps"] = m["comm_val"]["validated_steps"]
            out["config"]["comm_max_rel"] = m["comm_val"]["max_rel"]
            out["config"]["comm_us"] = m["comm_us"]
        if len(m["k_all"]) > 1:
            out["roofline"]["avg_kernel_ms_by_rank"] = {"min": min(m["k_all"]), "max": max(m["k_all"])}
        if check is not None:
            out["check"] = check
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(spec, full_lap=args.cpu_full_lap)
            except Exception as e:  # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if n_gpus == 1 and not args.no_extra and cfg != "c2":
            # round 1's line: BASELINE configs[1] (1e6 events x 128 TSpline3, dense: the uniform-tile kernel), same protocol
            try:
                a2 = argparse.Namespace(**vars(args))
                a2.check = False
                a2.spinup_seconds = min(args.spinup_seconds, 0.5)
                c2 = synth.CONFIGS["c2"]
                m2 = measure(c2, 0, 1, local_rank, a2, None, torch, False)
                out.setdefault("extra", {})["configs[1]"] = {"workload": workload_text(c2, c2, 1), "value": args.steps / m2["el"], "unit": "steps/s",
                                               "ms_per_step": 1e3 * m2["el"] / args.steps, "roofline": roofline("c2", m2, c2.n_events)}
            except Exception as e:
                out.setdefault("extra", {})["configs[1]"] = {"failed": str(e)}
                failed_extras.append("configs[1]")
        if n_gpus == 1 and args.osc_moves:
            try:
                out.setdefault("extra", {})["osc_moves"] = osc_moves_leg(args, local_rank, torch)
            except Exception as e:
                out.setdefault("extra", {})["osc_moves"] = {"failed": f"{type(e).__name__}: {e}"}
                failed_extras.append("osc_moves")
    # ---- legs under `extra` that are collectives (every rank runs them): the one-shot exchange beside RCCL, and from four
    # GPUs on BASELINE configs[3].  The headline numbers are in hand; each leg runs behind a watchdog, and a leg that does not
    # return within its limit is reported as text in its place -- rank 0 prints the line as it stands and every rank leaves.
    import threading

    def run_leg(name, limit, fn):
        # the line as it stands when the leg starts, serialised HERE: the watchdog's thread must not walk a dictionary the main
        # thread may be filling at that moment
        snap = None
        if rank == 0:
            o = json.loads(json.dumps(out))
            o.setdefault("extra", {})[name] = {"error": f"did not finish within {limit:.0f} s; abandoned (the line above it stands)"}
            o["extra_failed"] = failed_extras + [name]
            snap = json.dumps(o)

        def give_up():
            if rank == 0:
                print(snap, file=json_out, flush=True)
            log(f"[bench] rank {rank}: `extra.{name}` did not return within {limit:.0f} s; leaving without it")
            os._exit(4 if check is not None and not check["ok"] else (6 if args.strict else 0))
        wd = threading.Timer(limit, give_up)
        wd.daemon = True
        wd.start()
        try:
            res = fn()
        except Exception as e:  # (the legs report their own failures as text; this is the belt to those braces.  SystemExit and KeyboardInterrupt pass)
            res = {"error": f"rank {rank}: {type(e).__name__}: {e}"}
        finally:
            wd.cancel()
        if isinstance(res, dict) and ("error" in res or "failed" in res or res.get("ok") is False):
            failed_extras.append(name)
        if rank == 0:
            out.setdefault("extra", {})[name] = res
        return res

    # N > 1: the headline line goes out BEFORE the collective legs under `extra` (flushed; `"final": false`), and the complete line
    # once more at the end: whoever stops this process inside a leg -- a driver's time-out on a cold eight-GPU node -- still finds
    # a parsable line with the headline numbers; a reader that takes the LAST line gets everything.  N = 1 prints one line.
    will_run_legs = m.get("after") is not None or (n_gpus >= 4 and args.config == "auto" and not args.no_c4_extra)
    if rank == 0 and n_gpus > 1 and will_run_legs:
        print(json.dumps(dict(out, final=False)), file=json_out, flush=True)

    if m.get("after") is not None:
        run_leg("oneshot_exchange", float(os.environ.get("M3_BENCH_ONESHOT_LIMIT_S", "180")), m["after"])
        try:
            m["close"]()
        except Exception:
            pass
    if n_gpus >= 4 and args.config == "auto" and not args.no_c4_extra:
        def c4_leg():
            if os.environ.get("M3_BENCH_TEST_STALL_LEG_S"):  # tests: a leg that is slow but alive (the caller stops the run inside it)
                time.sleep(float(os.environ["M3_BENCH_TEST_STALL_LEG_S"]))
            a4 = argparse.Namespace(**vars(args))
            a4.check = False
            a4.spinup_seconds = min(args.spinup_seconds, 0.5)
            c4 = synth.CONFIGS["c4"]
            spec4 = c4.scaled(4 * args.events) if args.events else c4  # (a reduced --events keeps the 1 : 4 ratio of the two)
            m4 = measure(spec4, rank, n_gpus, local_rank, a4, dist, torch, args.allow_host_fallback, oneshot=False)
            per4 = m4["e1"] - m4["e0"]
            res = {"workload": workload_text(c4, spec4, n_gpus), "value": args.steps / m4["el"], "unit": "steps/s", "ms_per_step": 1e3 * m4["el"] / args.steps,
                   "event_steps_per_s": spec4.n_events * args.steps / m4["el"], "events_per_gpu": per4, "comm": m4["comm"], "comm_us": m4["comm_us"],
                   "roofline": roofline("c4", m4, per4)}
            if m4["comm_val"] is not None:
                res["comm_validated_steps"], res["comm_max_rel"] = m4["comm_val"]["validated_steps"], m4["comm_val"]["max_rel"]
            if len(m4["k_all"]) > 1:
                res["roofline"]["avg_kernel_ms_by_rank"] = {"min": min(m4["k_all"]), "max": max(m4["k_all"])}
            return res
        run_leg("configs[3]", float(os.environ.get("M3_BENCH_C4_LIMIT_S", "420")), c4_leg)
    if rank == 0:
        if failed_extras:
            out["extra_failed"] = failed_extras
        print(json.dumps(out), file=json_out, flush=True)
        if check is not None and not check["ok"]:
            raise SystemExit(4)
    if dist is not None:
        dist.destroy_process_group()
    if failed_extras and args.strict:
        raise SystemExit(6)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        cpu_baseline_child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], float(sys.argv[6]))
    else:
        main()
