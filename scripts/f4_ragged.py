"""SURVEY 8f row 4 on a RAGGED sample: BASELINE configs[2] (bricks) with one energy-scale functional parameter registered --
step time with and without the shift (every event re-binned per step: Samples/SampleHandlerFD.cpp:545-564,
Samples/SampleStructs.h:577-613).  usage: python scripts/f4_ragged.py [events=10000000] [config=c3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import SampleHandlerFD, build_from_synth

n_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
spec = synth.CONFIGS[cfg].scaled(n_ev)
rng = np.random.default_rng(5)
eng, mono, sh0, par = build_from_synth(spec)
info = eng.info()
par2 = np.concatenate([par, [0.0]])
sh = SampleHandlerFD(eng, 1, None, par2, bin_edges=[[synth.bin_edges(spec)]])
blk = 1 << 20
for e0 in range(0, spec.n_events, blk):
    ev = synth.events_chunk(spec, e0, min(spec.n_events, e0 + blk))
    sh.SetupKinematics(e0, ev["sample_id"], ev["erec"])
sh.ComputeBins()
for e0 in range(0, spec.n_events, blk):
    e1 = min(spec.n_events, e0 + blk)
    ev = synth.events_chunk(spec, e0, e1)
    sh.SetupEvents(e0, sh.GetEventBins(e0, e1 - e0), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
sh.SetOscillationWeights(synth.osc_table(spec))
sh.SetTestStatistic(int(spec.test_statistic))
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps = [synth.params(spec, i + 1) for i in range(16)]


def timeit(n=60, spin=20, walk=0.0):
    """walk = 0: the scale parameter is drawn afresh every step, N(0, 0.02) -- about half of the events change bin, the pass's worst
    case; walk > 0: a chain, theta += walk * N(0, 1) (a Metropolis step at ~520 parameters moves each by about a tenth of its width)"""
    th = 0.0

    def nxt():
        nonlocal th
        th = th + walk * rng.standard_normal() if walk else 0.02 * rng.standard_normal()
        return th
    for i in range(spin):
        par2[:spec.n_par] = laps[i % 16]; par2[-1] = nxt(); sh.Step()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for i in range(n):
            par2[:spec.n_par] = laps[i % 16]; par2[-1] = nxt(); sh.Step()
        best = min(best, (time.perf_counter() - t) / n * 1e6)
    return best


a = timeit()
sh.AddKinShift(-1, 0, 1, spec.n_par)
b = timeit()
c = timeit(walk=0.002)
print(f"[{os.path.basename(os.environ.get('M3HIP_LIB', 'libm3hip.so'))}] f4 on a ragged sample: {cfg} E={n_ev}, bricks {info.n_tiles_brick}, 1-D {spec.total_bins} bins, one scale shift: "
      f"step {a:.1f} us without, {b:.1f} us with the shift drawn afresh every step (+{b - a:.1f} us = +{100 * (b - a) / a:.1f} %), "
      f"{c:.1f} us with the shift walking like a chain's (+{c - a:.1f} us = +{100 * (c - a) / a:.1f} %)", flush=True)
eng.close()
