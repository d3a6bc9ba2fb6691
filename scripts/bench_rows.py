"""Timings of the widened rows (binned splines, device binning with shifts, seam-2 eval path).
Numbers go into DESIGN.md section 8; not part of bench.py's contract."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
from mach3_amd import synth, capi
from mach3_amd.engine import Engine, SMonolith, SampleHandlerFD, BinnedSplineHandler, build_from_synth
from helpers import make_binned

def timeit(fn, n=200, spin=0.7):
    t=time.perf_counter()
    while time.perf_counter()-t<spin: fn()
    t=time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter()-t)/n*1e6

rng=np.random.default_rng(5)
only = sys.argv[1] if len(sys.argv) > 1 else "all"
# ---- many bins: 2-D samples whose bin vector is too large for LDS (direct global atomics, k_llh over thousands of bins)
def bins_row(E, nx, ny, n_samples):
    spec = synth.CONFIGS["c1"].scaled(E)
    eng, mono, sh0, par = build_from_synth(spec)
    ev = synth.events_chunk(spec, 0, E)
    ex, ey = np.linspace(0.0, 5.0, nx + 1), np.linspace(-1.0, 1.0, ny + 1)
    sid = rng.integers(0, n_samples, E).astype(np.int32)
    kin = np.stack([ev["erec"], rng.uniform(-1.0, 1.0, E)], axis=1)
    sh = SampleHandlerFD(eng, n_samples, None, par, capi.kBarlowBeeston, bin_edges=[[ex, ey]] * n_samples)
    sh.SetupKinematics(0, sid, kin); sh.ComputeBins()
    sh.SetupEvents(0, sh.GetEventBins(0, E), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    def step():
        par[:] = synth.params(spec, int(rng.integers(1, 60))); return sh.Step()
    t = timeit(step)
    eng.lib.m3hip_debug_device_span(eng.h, 1, None)
    import ctypes as C
    dev = []
    for _ in range(100):
        step(); d = (C.c_double * 5)(); eng.lib.m3hip_debug_device_span(eng.h, 0, C.cast(d, C.POINTER(C.c_double))); dev.append(d[0])
    print(f"many bins: E={E}, {n_samples} x {nx} x {ny} = {n_samples*nx*ny} bins: step {t:.1f} us (device span k_prep start -> k_llh end {np.median(dev):.1f} us)", flush=True)
    eng.close()
# ---- many samples / many norm parameters per event: the loops over samples in k_llh and over norm slots in the epilogue
def samples_row(E, n_samples, n_norm):
    spec = synth.CONFIGS["c1"].scaled(E)
    eng, mono, sh0, par0 = build_from_synth(spec)
    ev = synth.events_chunk(spec, 0, E)
    par = np.concatenate([par0, np.ones(n_norm)])
    mono.setSplinePointers(par)
    ex = np.linspace(0.0, 5.0, 101)
    sid = rng.integers(0, n_samples, E).astype(np.int32)
    sh = SampleHandlerFD(eng, n_samples, None, par, capi.kBarlowBeeston, bin_edges=[[ex]] * n_samples)
    sh.SetupKinematics(0, sid, ev["erec"]); sh.ComputeBins()
    cnt = rng.integers(max(0, n_norm - 3), n_norm + 1, E) if n_norm else np.zeros(E, np.int64)
    norm_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    norm_idx = np.concatenate([len(par0) + np.sort(rng.choice(n_norm, c, replace=False)) for c in cnt]).astype(np.int16) if n_norm else np.zeros(0, np.int16)
    sh.SetupEvents(0, sh.GetEventBins(0, E), ev["osc_idx"], norm_off, norm_idx)
    sh.SetOscillationWeights(synth.osc_table(spec))
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    def step():
        par[:len(par0)] = synth.params(spec, int(rng.integers(1, 60))); par[len(par0):] = 1 + 0.05 * rng.standard_normal(n_norm); return sh.Step()
    t = timeit(step)
    print(f"samples/norms: E={E}, {n_samples} samples x 100 bins, up to {n_norm} norm parameters per event: step {t:.1f} us", flush=True)
    eng.close()
if only == "samples":
    samples_row(100000, 1, 3)
    samples_row(100000, 30, 3)
    samples_row(100000, 1, 24)
    samples_row(1000000, 1, 3)
    samples_row(1000000, 1, 24)
    sys.exit(0)
if only == "bins":
    bins_row(100000, 20, 10, 1)
    bins_row(100000, 40, 20, 1)
    bins_row(100000, 50, 40, 1)
    bins_row(1000000, 20, 10, 1)
    bins_row(1000000, 50, 40, 1)
    bins_row(100000, 60, 40, 3)
    bins_row(100000, 100, 80, 4)
    sys.exit(0)
# ---- binned splines: 1e6 events, 50 unique parameters, 1e6 bin-splines, ~30 weights per event; both M3::float_t builds
def binned_row(float_t, sort_by_cell=False):
    P, cells, E = 50, 20000, 1_000_000
    bs = make_binned(P, cells, rng, float_t=float_t)
    cell = rng.integers(0, cells, E)
    if sort_by_cell:
        cell = np.sort(cell)  # events that share a spline bin sit next to each other (what ordering the MC by kinematic bin gives)
    mask = (rng.random((cells, P)) < 0.6)[cell]  # the pointer list belongs to the spline bin (sample / channel / mode / kinematic bin)
    cnt = mask.sum(1)
    w_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    w_idx = (cell[:, None] * P + np.arange(P)[None, :])[mask].astype(np.int32)
    par = np.zeros(P + 4); par[P:] = 1
    eng = Engine(0)
    h = BinnedSplineHandler(eng, E, bs); h.setSplinePointers(par)
    edges = np.linspace(0, 5, 201); kin = rng.gamma(3.0, 0.35, E); sid = np.zeros(E, np.int32)
    sh = SampleHandlerFD(eng, 1, np.array([0, 200], np.int32), par, capi.kBarlowBeeston)
    n_osc = int(os.environ.get("BINNED_OSC_N", "4000"))  # (204000: an atmospheric table; the events' look-ups are then a line each unless the engine orders them)
    sh.SetupEvents(0, synth.find_global_bin_uniform_1d(edges, kin, sid, 200), rng.integers(-1, n_osc, E).astype(np.int32), None, None)
    h.SetEventSplines(0, w_off, w_idx); sh.SetOscillationWeights(rng.random(n_osc).astype(float_t))
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    def step():
        par[:P] = rng.uniform(-2, 2, P); return sh.Step()
    eng.timing_enable(True); us = timeit(step); k_ms,_ = eng.timing_read(0)
    print(f"binned float_t={np.dtype(float_t).name} sorted_by_cell={sort_by_cell} osc_entries={n_osc}: E={E} M={P*cells} unique={len(bs['uniquecoeffindices'])} weights/event={cnt.mean():.1f}: "
          f"step {us:.1f} us, event kernel {k_ms*1e3:.1f} us ({(w_idx.nbytes + 8*E)/k_ms/1e6:.0f} GB/s of index+bin/osc bytes)", flush=True)
    eng.close()

if only in ("all", "binned"):
    binned_row(np.float32)
    binned_row(np.float64)
    binned_row(np.float32, True)
    binned_row(np.float64, True)
if only == "binned":
    sys.exit(0)

# ---- per-step oscillation-table upload (OscillationHandler::Evaluate output), config 2 at 1e6 events
if only in ("all", "osc"):
    spec = synth.CONFIGS["c2"].scaled(1_000_000, n_params=16)
    eng, mono, sh, par = build_from_synth(spec)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    for n_tbl in (4000, 200000):
        tbl = rng.random(n_tbl).astype(np.float32)
        sh.SetOscillationWeights(tbl)
        def step_a():
            par[:spec.n_par] = synth.params(spec, int(rng.integers(1, 60))); return sh.Step()
        def step_b():
            sh.SetOscillationWeights(tbl); return step_a()
        a = timeit(step_a); b = timeit(step_b)
        print(f"osc table of {n_tbl} floats re-uploaded every step: {b:.1f} us/step vs {a:.1f} us/step with a static table (+{b-a:.1f} us)", flush=True)
    eng.close()
if only == "osc":
    sys.exit(0)
n_ev_shift = int(os.environ.get("SHIFT_EVENTS", "1000000"))


# ---- device binning + one energy-scale shift on config 2 at 1e6 events
spec = synth.CONFIGS["c2"].scaled(n_ev_shift, n_params=16)
eng, mono, sh0, par = build_from_synth(spec)
ev = synth.events_chunk(spec, 0, spec.n_events)
par2 = np.concatenate([par, [0.0]])
sh = SampleHandlerFD(eng, 1, None, par2, bin_edges=[[synth.bin_edges(spec)]])
sh.SetupKinematics(0, ev["sample_id"], ev["erec"]); sh.ComputeBins()
sh.SetupEvents(0, sh.GetEventBins(0, spec.n_events), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
sh.SetOscillationWeights(synth.osc_table(spec))
sh.Reweight(); sh.AddData(sh.GetMCArray())
def step2():
    par2[:spec.n_par] = synth.params(spec, int(rng.integers(1, 60))); par2[-1] = 0.02*rng.standard_normal(); return sh.Step()
a = timeit(step2)
sh.AddKinShift(-1, 0, 1, spec.n_par)
b = timeit(step2)
print(f"device re-binning: E={n_ev_shift}, 1-D 200 bins, one scale shift: step {a:.1f} us without, {b:.1f} us with shifts (+{b-a:.1f} us)", flush=True)
if only == "shift":
    sys.exit(0)
# ---- seam 2: SMonolith::Evaluate + SynchroniseMemTransfer incl. the 4 MB device-to-host copy
mono.setSplinePointers(par)
def step3():
    par[:] = synth.params(spec, int(rng.integers(1, 60))); mono.Evaluate(); mono.SynchroniseMemTransfer()
c = timeit(step3, n=100)
print(f"seam 2 (Evaluate + sync, E=1e6 x 16 splines, 4 MB of weights D2H per step): {c:.1f} us/step")
eng.close()
