"""Soak: many fused steps in one process; reports latency percentiles per block of steps and the process RSS, to catch
leaks (host command lists, events) and drifts.  usage: soak.py [events] [steps] [config] [samples nx ny]
(with the last three: 2-D samples, device-side binning, Barlow-Beeston -- the many-bins form of the fill and of k_llh;
SOAK_SHIFT=1 additionally registers an energy-scale functional parameter: the per-step re-binning pass)"""
import sys, os, time, resource
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
cfg = sys.argv[3] if len(sys.argv) > 3 else "c2"
spec = synth.CONFIGS[cfg].scaled(ev, n_params=32 if cfg == "c2" else 96)
# SOAK_KEY=1: the engine stores the events in oscillation-bin order (m3hip_monolith_order_key); SOAK_MOVE=1: the oscillation
# table (SOAK_OSC_N entries) is replaced before every step through the host seam, two tables alternating
osc_n = int(os.environ.get("SOAK_OSC_N", "0"))
if osc_n:
    spec = spec.scaled(ev, osc_n=osc_n)
eng, mono, sh, par = build_from_synth(spec, order_by_osc=os.environ.get("SOAK_KEY") == "1")
if len(sys.argv) > 6:
    from mach3_amd import capi
    from mach3_amd.engine import SampleHandlerFD
    n_samp, nx, ny = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rng = np.random.default_rng(3)
    evs = synth.events_chunk(spec, 0, ev)
    ex, ey = np.linspace(0.0, 5.0, nx + 1), np.linspace(-1.0, 1.0, ny + 1)
    sh = SampleHandlerFD(eng, n_samp, None, par, capi.kBarlowBeeston, bin_edges=[[ex, ey]] * n_samp)
    sh.SetupKinematics(0, rng.integers(0, n_samp, ev).astype(np.int32), np.stack([evs["erec"], rng.uniform(-1.0, 1.0, ev)], axis=1))
    sh.ComputeBins()
    sh.SetupEvents(0, sh.GetEventBins(0, ev), evs["osc_idx"], evs["norm_off"], evs["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
shift = os.environ.get("SOAK_SHIFT") == "1" and len(sys.argv) > 6   # a functional parameter: every event re-binned every step
if shift:
    par = np.concatenate([par, [0.0]])
    sh.par = par
    sh.AddKinShift(-1, 0, 1, len(par) - 1)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps=[synth.params(spec,i+1) for i in range(64)]
if shift:
    laps = [np.concatenate([l, [0.01 * ((i % 7) - 3)]]) for i, l in enumerate(laps)]
def rss_mb():
    with open("/proc/self/statm") as f: return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
blk = steps // 10
llh0 = None
move = os.environ.get("SOAK_MOVE") == "1"
tables = [synth.osc_table(spec), (0.5 * synth.osc_table(spec) + 0.25).astype(np.float32)]
for b in range(10):
    ts = np.empty(blk)
    for i in range(blk):
        t0 = time.perf_counter(); par[:] = laps[i % 64]
        if move: sh.SetOscillationWeights(tables[(i // 64) & 1])   # (64 steps per table: step i % 64 == 0 always sees table 0 or 1 in turn ... compared per table below)
        l = sh.Step(); ts[i] = time.perf_counter() - t0
        if i % 128 == 0:   # same parameters, same table -> bit-identical likelihood, every time
            if llh0 is None: llh0 = l
            assert l == llh0, (l, llh0)
    ts *= 1e6
    free, total = eng.mem_info() if hasattr(eng, "mem_info") else (0, 0)
    print(f"block {b}: {blk} steps, median {np.median(ts):.1f} us, p99 {np.percentile(ts,99):.1f}, p99.9 {np.percentile(ts,99.9):.1f}, max {ts.max():.0f} us, RSS {rss_mb():.0f} MB", flush=True)
print("OK")
