"""One A/B driver for kernel and step timings (replaces round 1's kbench*.py, ab_step.py, debug_*.py).

    python scripts/kbench.py shard  [--config c3,c5] [--events 2000000]   fused reweight kernel on a shard (dispatch-stamped events)
    python scripts/kbench.py occ    [--events 1000000]                    brick kernel against responses per brick (config-3 shape at
                                                                          occupancy 1/32 .. 1/4): cost per brick vs cost per response
    python scripts/kbench.py step   [--config c2] [--events 1000000]      step latency, host phases and device span of ONE library build
    python scripts/kbench.py pad    [--events 1000000]                    identity-padded tiles vs bricks on nearly uniform lists
    python scripts/kbench.py modes  [--events 1000000]                    engine-side grouping of shuffled interaction modes
    python scripts/kbench.py mix    [--events 2000000]                    brick kernel: short / long runs x with / without TF1 responses
    python scripts/kbench.py det    [--config c2,c3]                      cost of the deterministic fill
    python scripts/kbench.py tilebrick [--events 1000000]                 a sample of uniform tiles AND bricks (mode-sorted events + a ragged
                                                                          remainder): the two launches of a step against each half alone

Two builds are compared by running the script twice in ONE gpurun call with M3HIP_LIB=<path to the other libm3hip.so>
(boxes of the pool differ by 2-3 %).  Debug knobs of the uniform kernel: --unroll 8|16, --no-nt, --lds-pad BYTES."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mach3_amd import capi, synth
from mach3_amd.engine import Engine, SMonolith, SampleHandlerFD, build_from_synth

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["shard", "occ", "step", "pad", "modes", "det", "mix", "tilebrick"])
ap.add_argument("--config", default=None)
ap.add_argument("--events", type=int, default=None)
ap.add_argument("--unroll", type=int, default=8)
ap.add_argument("--no-nt", action="store_true")
ap.add_argument("--lds-pad", type=int, default=0)
args = ap.parse_args()
LIB = os.path.basename(os.environ.get("M3HIP_LIB", "libm3hip.so"))


def ready(spec, **kw):
    eng, mono, sh, par = build_from_synth(spec, **kw)
    eng.lib.m3hip_debug_set_knobs(eng.h, args.unroll, 0 if args.no_nt else 1, args.lds_pad)
    sh.Reweight()
    sh.AddData(sh.GetMCArray())
    return eng, sh, par, [synth.params(spec, i + 1) for i in range(16)]


def kernel_us(eng, sh, par, laps, n=40):
    eng.timing_enable(True)
    for i in range(20):
        par[:] = laps[i % 16]; sh.Step()
    eng.timing_reset()
    t = time.perf_counter()
    for i in range(n):
        par[:] = laps[i % 16]; sh.Step()
    step = (time.perf_counter() - t) / n * 1e6
    f, _ = eng.timing_read(0)
    return f * 1e3, step


def line(tag, info, k_us, step_us):
    print(f"[{LIB}] {tag}: S={info.n_splines} T={info.n_tf1} tiles {info.n_tiles_uniform}u ({info.n_tiles_padded} padded) / {info.n_tiles_brick}b: "
          f"kernel {k_us:.1f} us = {info.alg_bytes_fused / k_us / 1e3:.0f} GB/s algorithmic ({info.alg_bytes_fused / k_us / 1e3 / 8000:.3f} of 8 TB/s), "
          f"step {step_us:.1f} us", flush=True)


if args.what == "shard":
    for cfg in (args.config or "c3,c5").split(","):
        spec = synth.CONFIGS[cfg].scaled(args.events or 2_000_000)
        eng, sh, par, laps = ready(spec)
        line(f"{cfg} E={spec.n_events}", eng.info(), *kernel_us(eng, sh, par, laps))
        eng.close()
elif args.what == "occ":
    for occ in (1 / 32, 1 / 16, 1 / 8, 1 / 4):
        spec = synth.CONFIGS["c3"].scaled(args.events or 1_000_000, occupancy=occ)
        eng, sh, par, laps = ready(spec)
        info = eng.info()
        k, st = kernel_us(eng, sh, par, laps)
        nb = max(int(info.n_tiles_brick), 1)
        line(f"occupancy 1/{round(1 / occ)}: {(info.n_splines + info.n_tf1) / nb:.0f} responses per brick, {k / nb * 512:.1f} us per brick slot", info, k, st)
        eng.close()
elif args.what == "step":
    spec = synth.CONFIGS[args.config or "c2"].scaled(args.events or 1_000_000)
    eng, sh, par, laps = ready(spec)
    t = time.perf_counter(); i = 0
    while time.perf_counter() - t < 1.5:
        par[:] = laps[i % 16]; sh.Step(); i += 1
    for rep in range(3):
        t = time.perf_counter()
        for i in range(500):
            par[:] = laps[i % 16]; sh.Step()
        print(f"[{LIB}] {spec.name} step {(time.perf_counter() - t) / 500 * 1e6:.1f} us", flush=True)
    eng.lib.m3hip_debug_device_span(eng.h, 1, None)
    dev, acc, out = [], np.zeros(6), np.zeros(6)
    for i in range(200):
        par[:] = laps[i % 16]; sh.Step()
        eng.lib.m3hip_debug_host_times(eng.h, out.ctypes.data); acc += out
        d = (C.c_double * 5)(); eng.lib.m3hip_debug_device_span(eng.h, 0, C.cast(d, C.POINTER(C.c_double))); dev.append(list(d))
    dv = np.array(dev)
    off = np.min(dv[:, 4] - dv[:, 2])  # host-device clock offset from the fastest completion
    print(f"   host phases {{stage params, launch prep, launch reweight, launch llh, wait, collect}} us: {(acc / 200 * 1e6).round(1)}")
    print(f"   device span k_prep start -> k_llh end: median {np.median(dv[:, 0]):.1f} us; launch -> k_prep start median "
          f"{np.median(dv[:, 1] + off - dv[:, 3]):.1f} us; k_llh end -> result seen median {np.median(dv[:, 4] - (dv[:, 2] + off)):.1f} us")
    eng.close()
elif args.what == "pad":  # the reference drops an event's flat splines: nearly uniform lists
    for occ in (0.9, 0.8, 0.7):
        for pad in ("0.01", "2"):  # always padded / never
            os.environ["M3HIP_PAD_MIN_FILL"] = pad
            spec = synth.CONFIGS["c2"].scaled(args.events or 1_000_000, occupancy=occ)
            eng, sh, par, laps = ready(spec)
            line(f"occupancy {occ} pad_min_fill {pad}", eng.info(), *kernel_us(eng, sh, par, laps))
            eng.close()
elif args.what == "modes":  # four interaction modes, 96 of 128 parameters each, events in random order
    E, P, K, L = args.events or 1_000_000, 128, 7, 96
    rng = np.random.default_rng(1)
    modes = [np.sort(rng.choice(P, L, replace=False)).astype(np.int16) for _ in range(4)]

    def chunk(which, drop):  # drop: fraction of an event's splines that are flat and left out (SplineMonolith.cpp:160-207)
        keep = [np.ones(L, bool) if drop == 0 else rng.random(L) >= drop for _ in which]
        cnt = np.array([int(k.sum()) for k in keep], np.uint32)
        pn = np.concatenate([modes[m][k] for m, k in zip(which, keep)]).astype(np.int16)
        S = len(pn)
        start = np.concatenate([[0], np.cumsum(cnt[:-1], dtype=np.uint64)]).astype(np.uint32)
        return dict(n_events=len(which), n_splines=S, n_tf1=0, total_nknots=S * K, coeff_many=(1 + 0.01 * rng.standard_normal(S * K * 4)).astype(np.float32),
                    nknots_arr=(np.arange(S, dtype=np.uint64) * K).astype(np.uint32), param_no=pn, coeff_tf1=np.zeros(0, np.float32),
                    param_no_tf1=np.zeros(0, np.int16), npe_tf1=np.zeros(2 * len(which), np.uint32),
                    npe=np.stack([cnt, start], axis=1).reshape(-1))
    cx, npts, which = np.tile(np.arange(-3.0, 4.0, dtype=np.float32), P), np.full(P, K, np.int16), rng.integers(0, 4, E)
    for drop in (0.0, 0.15):  # 0.15: too ragged for identity padding (fill 0.85 < 0.88): grouped bricks against mixed bricks
        for label, w, env in (("random order, engine grouping", which, None), ("random order, no grouping", which, "0"), ("sorted by mode", np.sort(which), None)):
            os.environ.pop("M3HIP_GROUP_EVENTS", None) if env is None else os.environ.__setitem__("M3HIP_GROUP_EVENTS", env)
            eng = Engine(0)
            mono = SMonolith(eng, P, K, cx, npts, [chunk(w[i:i + 131072], drop) for i in range(0, E, 131072)])
            par = np.zeros(P)
            mono.setSplinePointers(par)
            sh = SampleHandlerFD(eng, 1, np.array([0, 200], np.int32), par, capi.kPoisson)
            sh.SetupEvents(0, rng.integers(0, 200, E).astype(np.int32), None, None, None)
            sh.Reweight(); sh.AddData(sh.GetMCArray())
            laps = [rng.normal(0, 0.3, P) for _ in range(16)]
            line(f"{label}, {drop:.0%} of the splines flat", eng.info(), *kernel_us(eng, sh, par, laps, n=100))
            eng.close()
elif args.what == "mix":  # what limits the brick kernel: run length or the TF1 side?  2 x 2: short / long runs, with / without TF1
    for P, occ in ((512, 1 / 16), (128, 0.5)):
        for mixed in (False, True):
            spec = synth.CONFIGS["c3"].scaled(args.events or 2_000_000, n_params=P, occupancy=occ, mixed=mixed)
            os.environ["M3HIP_PAD_MIN_FILL"] = "2"
            eng, sh, par, laps = ready(spec)
            info = eng.info()
            nb = max(int(info.n_tiles_brick), 1)
            line(f"P={P} occupancy {occ:g} {'TSpline3 + TF1' if mixed else 'TSpline3 only'}: {1024 * occ:.0f} responses per run", info, *kernel_us(eng, sh, par, laps))
            eng.close()
elif args.what == "det":
    for cfg in (args.config or "c2,c3").split(","):
        spec = synth.CONFIGS[cfg].scaled(args.events or (1_000_000 if cfg == "c2" else 2_000_000))
        eng, sh, par, laps = ready(spec)
        for det in (0, 1, 0, 1):
            eng.set_deterministic(bool(det))
            k, st = kernel_us(eng, sh, par, laps, n=200)
            print(f"[{LIB}] {cfg} E={spec.n_events} deterministic={det}: step {st:.1f} us (kernel {k:.1f} us)", flush=True)
        eng.close()
elif args.what == "tilebrick":
    # What real MC looks like once it is mode-sorted: most events in uniform tiles (four interaction modes, 96 of 128 parameters
    # each), a ragged remainder (every parameter with probability 0.3) in bricks.  A step launches k_reweight_uniform and then
    # k_reweight_brick on one stream: is the sum of the two launches the sum of the halves alone?
    E, P, K, L = args.events or 1_000_000, 128, 7, 96
    rng = np.random.default_rng(7)
    modes = [np.sort(rng.choice(P, L, replace=False)).astype(np.int16) for _ in range(4)]

    def chunk(lists):
        cnt = np.array([len(l) for l in lists], np.uint32)
        pn = np.concatenate(lists).astype(np.int16) if len(lists) else np.zeros(0, np.int16)
        S = len(pn)
        start = np.concatenate([[0], np.cumsum(cnt[:-1], dtype=np.uint64)]).astype(np.uint32)
        return dict(n_events=len(lists), n_splines=S, n_tf1=0, total_nknots=S * K, coeff_many=(1 + 0.01 * rng.standard_normal(S * K * 4)).astype(np.float32),
                    nknots_arr=(np.arange(S, dtype=np.uint64) * K).astype(np.uint32), param_no=pn, coeff_tf1=np.zeros(0, np.float32),
                    param_no_tf1=np.zeros(0, np.int16), npe_tf1=np.zeros(2 * len(lists), np.uint32), npe=np.stack([cnt, start], axis=1).reshape(-1))

    def lists_of(n_tile, n_ragged):
        which = np.sort(rng.integers(0, 4, n_tile))
        out = [modes[m] for m in which]
        for _ in range(n_ragged):
            out.append(np.flatnonzero(rng.random(P) < 0.3).astype(np.int16))
        return out
    cx, npts = np.tile(np.arange(-3.0, 4.0, dtype=np.float32), P), np.full(P, K, np.int16)
    n_t = int(0.8 * E) // 1024 * 1024
    res = {}
    for label, nt_, nr_ in (("tiles + bricks", n_t, E - n_t), ("the tiles alone", n_t, 0), ("the bricks alone", 0, E - n_t)):
        ev = lists_of(nt_, nr_)
        eng = Engine(0)
        mono = SMonolith(eng, P, K, cx, npts, [chunk(ev[i:i + 131072]) for i in range(0, len(ev), 131072)])
        par = np.zeros(P)
        mono.setSplinePointers(par)
        sh = SampleHandlerFD(eng, 1, np.array([0, 200], np.int32), par, capi.kPoisson)
        sh.SetupEvents(0, rng.integers(0, 200, len(ev)).astype(np.int32), None, None, None)
        sh.Reweight(); sh.AddData(sh.GetMCArray())
        laps = [rng.normal(0, 0.3, P) for _ in range(16)]
        k, st = kernel_us(eng, sh, par, laps, n=200)
        res[label] = (k, st)
        line(label, eng.info(), k, st)
        eng.close()
    both, a, b = res["tiles + bricks"], res["the tiles alone"], res["the bricks alone"]
    print(f"[{LIB}] the two launches together {both[0]:.1f} us (first dispatch's start -> second's end) against {a[0]:.1f} + {b[0]:.1f} = {a[0] + b[0]:.1f} us alone "
          f"({100 * (both[0] / (a[0] + b[0]) - 1):+.1f} %); step {both[1]:.1f} us against {a[1]:.1f} / {b[1]:.1f} us", flush=True)
