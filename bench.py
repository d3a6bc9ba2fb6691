#!/usr/bin/env python3
"""bench.py -- MCMC-step (LLH evaluation) throughput of the MI355X likelihood engine.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic MC:
    new parameter vector -> SampleHandlerFD::Reweight() -> GetLikelihood() -> LLH on the host
(the pair MR2T2::ProposeStep runs per proposal, ref: Fitters/MR2T2.cpp:66-76), timed
DragRace-style (ref: Fitters/FitterBase.cpp:461-489).

Workload: BASELINE.json configs[1] ("c2": 1e6 events, 128 TSpline3 systematics x 7 knots,
single-sample Poisson LLH) PER GPU; with N > 1 every rank holds its own 1e6-event shard of an
N x 1e6-event sample and each step ends in ONE RCCL all-reduce of the bin-rate vector (weak
scaling).  value = N * K / t: LLH evaluations of a 1e6-event shard per second, which at
N = 1 is exactly MCMC steps/s on configs[1].

Prints ONE JSON line on rank 0 (contract in the task description), carrying `roofline`
(fused reweight kernel: algorithmic bytes / hipEvent-measured launch time vs 8 TB/s) and,
at N = 1, `cpu_baseline` (the OpenMP oracle port timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# host thread pools sized to one GPU's CPU share (the reference recommends 8, allows 16 threads:
# Manager/Monitor.cpp:122-138); must be set before any OpenMP runtime initialises
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
os.environ.setdefault("OMP_PROC_BIND", "close")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pmc_traffic(config_name, per_gpu):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_<config>_pmc_*.json: FETCH_SIZE and WRITE_SIZE collected in separate runs and
    corrected as MI355X_MICROARCH.md prescribes).  None when no profile matches this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config_name}_pmc_*.json"))):
        try:
            d = json.load(open(f))
            for k, v in d["kernels"].items():
                if "k_reweight" in k and "hbm_bytes_per_launch" in v and d.get("events_per_gpu", 1_000_000) == per_gpu:
                    best = {"bytes": v["hbm_bytes_per_launch"], "source": os.path.relpath(f, ROOT)}
        except Exception:
            pass
    return best


EVENT_EVERY = int(os.environ.get("M3_BENCH_EVENT_EVERY", "4"))


def cpu_baseline(spec, budget_s=10.0, sample_events=200_000):
    """Reference-layout OpenMP port (oracle/libm3oracle_omp.so) on the first `sample_events`
    events of the same workload; steps/s scaled to the full event count.  Timed DragRace style at
    min(cores, 16) threads (the reference refuses more unless OMP_NUM_THREADS is exported,
    Manager/Monitor.cpp:122-138) and, briefly, at 8 threads; phases as SURVEY section 8d asks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from mach3_amd import synth

    n = min(sample_events, spec.n_events)
    os.environ.setdefault("OMP_PROC_BIND", "close")  # SURVEY section 8d; read by libgomp when the library is loaded
    lib = ob.load("omp")
    cores = os.cpu_count() or 1
    lib.m3o_set_num_threads(min(cores, 16))
    mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, 0, n, variant="omp")
    samp.reweight()
    samp.add_data(samp.mc)

    def run(threads, budget, max_laps):
        lib.m3o_set_num_threads(threads)
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
            if el > budget or laps >= max_laps:
                break
        return {"threads": int(lib.m3o_max_threads()), "laps": laps, "seconds": el, "steps_per_s": laps / el * n / spec.n_events,
                "ms_per_lap": 1e3 * el / laps, "spline_eval_ms": 1e3 * t_spl / laps, "fill_ms": 1e3 * t_fill / laps,
                "llh_ms": 1e3 * t_llh / laps}

    main_run = run(min(cores, 16), budget_s, 2000)
    sweep = {str(main_run["threads"]): main_run["steps_per_s"]}
    if cores >= 8 and main_run["threads"] != 8:
        r8 = run(8, 0.35 * budget_s, 1000)
        sweep["8"] = r8["steps_per_s"]
    if cores >= 64:  # what exporting OMP_NUM_THREADS buys the reference on this host: half the logical CPUs, at most 128
        rmany = run(min(cores // 2, 128), 0.35 * budget_s, 1000)
        sweep[str(rmany["threads"])] = rmany["steps_per_s"]
    samp.close()
    mono.close()
    return {
        "value": main_run["steps_per_s"],
        "unit": "steps/s",
        "cores": main_run["threads"],
        "kind": "port",
        "sample": f"first {n} of {spec.n_events} events, {main_run['laps']} laps in {main_run['seconds']:.1f} s "
                  f"({main_run['ms_per_lap']:.2f} ms/lap: spline eval {main_run['spline_eval_ms']:.2f} ms, fill {main_run['fill_ms']:.2f} ms, "
                  f"LLH {main_run['llh_ms']:.3f} ms); steps/s scaled by {n}/{spec.n_events}",
        "steps_per_s_by_threads": sweep,
        "host_cores": cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--events", type=int, default=0, help="override events per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--spinup-seconds", type=float, default=2.0,
                    help="untimed steps before the warm-up: the first ~1 s of GPU work on a fresh process shows one-off "
                         "50-100 ms stalls (clock / runtime ramp-up) that would otherwise land in a short timed region")
    args = ap.parse_args()

    import torch  # device sync + torch.distributed (RCCL) plumbing only

    from mach3_amd import synth
    from mach3_amd.engine import build_from_synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        if os.environ.get("M3_BENCH_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
        local_rank = local_rank % max(n_dev, 1)  # rehearsal: ranks share the card
    torch.cuda.set_device(local_rank)
    dist = None
    force_comm = os.environ.get("M3_BENCH_FORCE_COMM") == "1"  # rehearsal: run the N > 1 plumbing with one rank
    if world > 1 or force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("M3_BENCH_BACKEND", "nccl")  # "gloo": plumbing rehearsal on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    base = synth.CONFIGS[args.config]
    per_gpu = args.events or base.n_events
    spec = base.scaled(per_gpu * n_gpus)
    e0, e1 = rank * per_gpu, (rank + 1) * per_gpu

    t_setup = time.perf_counter()
    eng, mono, sh, par = build_from_synth(spec, device_id=local_rank, chunk_events=args.chunk, event_range=(e0, e1))
    info = eng.info()
    if rank == 0:
        log(f"[bench] {spec.name}: {per_gpu} events/GPU x {n_gpus} GPU, P={spec.n_params}, S={info.n_splines}, "
            f"T={info.n_tf1}, coeff store {info.coeff_bytes / 1e9:.2f} GB, tiles {info.n_tiles_uniform}u/"
            f"{info.n_tiles_brick}b, setup {time.perf_counter() - t_setup:.1f} s")

    comm_mode = "none"
    if (world > 1 or force_comm) and os.environ.get("M3_BENCH_BACKEND", "nccl") == "nccl":
        # one in-library RCCL all-reduce per step; torch.distributed only carries the unique id
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
                eng.comm_init(uid[0], rank, world)
                ok = torch.ones(1, device="cuda")
            except Exception as e:  # keep the scaling run alive: exchange the bins through torch.distributed instead
                log(f"rank {rank}: in-library RCCL init failed ({e}); falling back to torch.distributed all_reduce")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            comm_mode = "rccl-allreduce(f64[n_bins]) in-library, one per step"
        else:
            comm_mode = "torch.distributed all_reduce of the bin vector (fallback)"
    elif world > 1:
        comm_mode = "none (rehearsal backend: ranks do not exchange bins)"

    fallback = comm_mode.startswith("torch.distributed")
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

    def step(i):
        par[:] = laps[i % len(laps)]
        if not fallback:
            return sh.Step()
        sh.Reweight()  # partial bins of this rank's shard
        reduce_bins()
        return sh.GetLikelihood()

    t_spin = time.perf_counter()
    n_spin = 0
    while time.perf_counter() - t_spin < args.spinup_seconds:
        step(n_spin)
        n_spin += 1
    for i in range(args.warmup):
        step(i)
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
    eng.timing_enable(False)

    if rank == 0:
        tr = pmc_traffic(args.config, per_gpu)
        cfg_index = {"c1": 0, "c2": 1, "c3": 2, "c4": 3, "c5": 4}.get(base.name, "?")
        stat_name = {0: "Poisson", 1: "Barlow-Beeston", 2: "IceCube", 3: "Pearson", 4: "Dembinski-Abdelmotteleb"}.get(int(spec.test_statistic), "?")
        if info.n_tiles_brick and info.n_tiles_uniform:
            kernel_name = "k_reweight_uniform<fill> + k_reweight_brick<fill> (spline eval + weight product + bin fill, fused; both launches timed together)"
        elif info.n_tiles_brick:
            kernel_name = "k_reweight_brick<fill> (ragged groups: spline eval + weight product + bin fill, fused)"
        else:
            kernel_name = "k_reweight_uniform<fill> (spline eval + weight product + bin fill, fused)"

        out = {
            "metric": "LLH evaluations/sec (MCMC steps/sec)",
            "value": n_gpus * args.steps / el,
            "unit": "steps/s" if n_gpus == 1 else f"steps/s x shards (one {per_gpu}-event shard per GPU)",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[{cfg_index}] ({base.name}): {per_gpu} events/GPU, {spec.n_params} "
                            f"{'TSpline3/TF1 (alternating)' if spec.mixed else 'TSpline3'} systematics x {spec.n_knots} knots "
                            f"(occupancy {spec.occupancy:g}), {spec.n_norm} norm systematics, pre-staged osc table, "
                            f"{spec.total_bins} bins, {stat_name} LLH",
                "events_per_gpu": per_gpu,
                "total_events": per_gpu * n_gpus,
                "spline_responses_per_gpu": int(info.n_splines),
                "comm": comm_mode,
                "steps_per_s": args.steps / el,
                "last_llh": llh,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": info.alg_bytes_fused / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (info.alg_bytes_fused / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else None,
                "traffic": tr["bytes"] if tr else None,
                "traffic_source": tr["source"] if tr else None,
                "alg_bytes_per_launch": int(info.alg_bytes_fused),
                "avg_kernel_ms": k_ms,
                "launches_timed": int(k_n),
                "event_sampling": f"every {EVENT_EVERY}th launch of the timed region carries a hipEvent pair stamped with the dispatch's own start/end (hipExtLaunchKernelGGL)",
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(base.scaled(per_gpu))
            except Exception as e:  # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
