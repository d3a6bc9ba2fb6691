"""Step times of shapes off the beaten track, to catch costs that grow with something other than events x responses
(the way the bin count and the sample count did): many parameters per event, many functional shifts, 3-D boxes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd import synth, capi
from mach3_amd.engine import SampleHandlerFD, build_from_synth

rng = np.random.default_rng(9)


def timeit(fn, n=200, spin=0.7):
    t = time.perf_counter()
    while time.perf_counter() - t < spin:
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e6


def dense(E, P):
    spec = synth.CONFIGS["c2"].scaled(E, n_params=P)
    eng, mono, sh, par = build_from_synth(spec)
    info = eng.info()
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    eng.timing_enable(True)
    def step():
        par[:] = synth.params(spec, int(rng.integers(1, 60))); return sh.Step()
    t = timeit(step)
    k, _ = eng.timing_read(0)
    print(f"dense: E={E} x P={P} TSpline3: step {t:.1f} us, reweight kernel {k*1e3:.1f} us = {info.alg_bytes_fused/k/1e6:.0f} GB/s algorithmic", flush=True)
    eng.close()


def shifts(E, n_shifts):
    spec = synth.CONFIGS["c1"].scaled(E)
    eng, mono, sh0, par0 = build_from_synth(spec)
    ev = synth.events_chunk(spec, 0, E)
    par = np.concatenate([par0, np.zeros(n_shifts)])
    mono.setSplinePointers(par)
    ex, ey = np.linspace(0.0, 5.0, 41), np.linspace(-1.0, 1.0, 21)
    kin = np.stack([ev["erec"], rng.uniform(-1.0, 1.0, E)], axis=1)
    sid = rng.integers(0, 2, E).astype(np.int32)
    sh = SampleHandlerFD(eng, 2, None, par, capi.kBarlowBeeston, bin_edges=[[ex, ey]] * 2)
    sh.SetupKinematics(0, sid, kin); sh.ComputeBins()
    sh.SetupEvents(0, sh.GetEventBins(0, E), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    def step():
        par[:len(par0)] = synth.params(spec, int(rng.integers(1, 60))); par[len(par0):] = 0.01 * rng.standard_normal(n_shifts); return sh.Step()
    a = timeit(step)
    for i in range(n_shifts):
        sh.AddKinShift(i % 2 - 1 if i % 3 == 0 else i % 2, i % 2, 1 + i % 2, len(par0) + i)
    b = timeit(step)
    print(f"shifts: E={E}, 2 x 40 x 20 bins, {n_shifts} functional parameters: step {a:.1f} us without, {b:.1f} us with", flush=True)
    eng.close()


dense(100000, 1024)
dense(20000, 4096)
shifts(1000000, 1)
shifts(1000000, 8)
