"""One FULL-SIZE lap of the timed CPU baseline (oracle/libm3oracle_omp.so) on record, next to the bounded sample bench.py
extrapolates from: all events of a BASELINE configuration, as several monoliths of 2.5e6 events (the reference's `unsigned
int` offsets cannot hold configs[2] in one: 4.5e9 coefficient floats), arrays first-touched inside OpenMP loops, 16 threads
(the reference's own cap, Manager/Monitor.cpp:122-138).  Prints the measured full lap, the 3e5-event sample scaled by
events as bench.py does it, and their ratio.  TEST INFRASTRUCTURE (it runs the oracle); not part of the product.
usage: OMP_NUM_THREADS=16 OMP_PROC_BIND=close OMP_PLACES=cores python scripts/cpu_full_lap.py [config] [threads]"""
import gc
import os
import sys
import time

try:  # a parent whose main thread is pinned (its own OpenMP runtime under OMP_PROC_BIND) hands its one-core mask down: undo it
    os.sched_setaffinity(0, range(os.cpu_count() or 1))
except OSError:
    pass
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_binding as ob
from mach3_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
spec = synth.CONFIGS[cfg]
lib = ob.load("omp")
lib.m3o_set_num_threads(threads)
E, step = spec.n_events, 2_500_000
print(f"{cfg}: {E} events, {threads} threads (host has {os.cpu_count()} hardware threads)", flush=True)


def build(a, b):
    mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, a, b, variant="omp", localise=True)
    for k in ("coeff_many", "nknots_arr", "param_no", "coeff_tf1", "param_no_tf1", "npe", "npe_tf1"):
        ch[k] = None   # the oracle owns first-touched copies now
    gc.collect()
    return mono, samp, par


def laps(parts, n, first_lap):
    t_laps = []
    for i in range(n):
        p = synth.params(spec, first_lap + i)
        t0 = time.perf_counter()
        llh = 0.0
        for mono, samp, par in parts:
            par[:] = p
            samp.reweight()
            llh += samp.get_likelihood()
        t_laps.append(time.perf_counter() - t0)
    return t_laps, llh


# the bounded sample bench.py times (first 3e5 events), scaled by events
t0 = time.perf_counter()
small = [build(0, 300_000)]
small[0][1].reweight(); small[0][1].add_data(small[0][1].mc)
laps(small, 3, 1)
ts, _ = laps(small, 30, 100)
ms_small = 1e3 * float(np.mean(ts))
print(f"sample: first 300000 events, 30 laps: {ms_small:.2f} ms/lap -> scaled by events: {ms_small * E / 300_000:.1f} ms/lap = "
      f"{1e3 / (ms_small * E / 300_000):.3f} steps/s", flush=True)
for mono, samp, par in small:
    samp.close(); mono.close()
small = None
gc.collect()

parts = []
for a in range(0, E, step):
    parts.append(build(a, min(E, a + step)))
    print(f"  built events [{a}, {min(E, a + step)}) after {time.perf_counter() - t0:.0f} s", flush=True)
for mono, samp, par in parts:
    samp.reweight(); samp.add_data(samp.mc)
laps(parts, 1, 1)
tf, llh = laps(parts, 5, 100)
ms_full = 1e3 * float(np.mean(tf))
print(f"full size: {len(parts)} monoliths of <= {step} events, 5 laps: {' '.join(f'{1e3 * t:.1f}' for t in tf)} ms -> {ms_full:.1f} ms/lap = "
      f"{1e3 / ms_full:.3f} steps/s (LLH of the last lap {llh:.6f})")
print(f"ratio full lap / event-scaled sample: {ms_full / (ms_small * E / 300_000):.3f}")
