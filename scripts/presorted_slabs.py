"""What would physically sorting a brick's events by TF1 code count (inside every 1024-event slab, at upload) be worth?  The planner
hands the TF1 lane-slots of a slab its events in that order (sigma) and the kernel moves the spline products and sample words to
the slots' lanes through LDS; MC that ARRIVES in that order needs neither (identity sigma: no exchange).  This script uploads a shard
of a configuration twice -- as generated, and with the {count, start} pairs of every slab pre-sorted by TF1 count (only the small
arrays move, as in the engine's own grouping) -- and times the fused kernel.  usage: python scripts/presorted_slabs.py <config> <events>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd import capi, synth
from mach3_amd.engine import Engine, SMonolith, SampleHandlerFD

cfg, ev = sys.argv[1], int(sys.argv[2])
spec = synth.CONFIGS[cfg].scaled(ev)
cx, npts = synth.knots(spec)
CH = 131072


def chunks(presort):
    for e0 in range(0, ev, CH):
        ch = synth.monolith_chunk(spec, e0, min(ev, e0 + CH))
        if presort:
            n = ch["n_events"]
            a, t = ch["npe"].reshape(n, 2).copy(), ch["npe_tf1"].reshape(n, 2).copy()
            for s0 in range(0, n, 1024):
                o = np.argsort(-t[s0:s0 + 1024, 0].astype(np.int64), kind="stable")
                a[s0:s0 + 1024] = a[s0:s0 + 1024][o]
                t[s0:s0 + 1024] = t[s0:s0 + 1024][o]
            ch["npe"], ch["npe_tf1"] = a.reshape(-1), t.reshape(-1)
        yield ch


for presort in (False, True, False, True):
    eng = Engine(0)
    par = synth.params(spec, 0).copy()
    mono = SMonolith(eng, spec.n_params, spec.n_knots, cx, npts, chunks(presort), expect_events=ev)
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(eng, spec.n_samples, synth.sample_bin_offsets(spec), par, spec.test_statistic)
    for e0 in range(0, ev, 1 << 20):
        e = synth.events_chunk(spec, e0, min(ev, e0 + (1 << 20)))
        sh.SetupEvents(e0, e["bin_idx"], e["osc_idx"], e["norm_off"], e["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    laps = [synth.params(spec, i + 1) for i in range(16)]
    eng.timing_enable(True)
    for i in range(30):
        par[:] = laps[i % 16]; sh.Step()
    res = []
    for rep in range(3):
        eng.timing_reset()
        for i in range(40):
            par[:] = laps[i % 16]; sh.Step()
        res.append(eng.timing_read(0)[0] * 1e3)
    info = eng.info()
    print(f"{cfg} E={ev} G={info.brick_groups} slabs {'pre-sorted by TF1 count' if presort else 'as generated'}: kernel {' / '.join(f'{r:.1f}' for r in res)} us "
          f"-> {info.alg_bytes_fused / min(res) / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    eng.close()
