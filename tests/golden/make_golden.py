"""Writes tests/golden/*.npz: small self-contained fixtures (reference-layout inputs + expected
outputs).  The reference cannot run in this environment (ROOT etc. absent) and ships no vectors of
its own, so the expected outputs are those of the SERIAL CPU oracle (oracle/m3_oracle.cpp) -- i.e.
these files pin the oracle and the HIP path against regressions; they are not reference outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob  # noqa: E402
from mach3_amd import capi, synth  # noqa: E402

CASES = {
    "dense_c2_300x8": synth.CONFIGS["c2"].scaled(300, n_params=8, n_bins=20),
    "sparse_c3_1500x24": synth.CONFIGS["c3"].scaled(1500, n_params=24, n_bins=25, occupancy=0.25, n_samples=2),
}

for name, spec in CASES.items():
    mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
    cx, npts = synth.knots(spec)
    out = dict(n_params=spec.n_params, n_knots=spec.n_knots, n_samples=spec.n_samples, n_bins=spec.n_bins,
               test_statistic=spec.test_statistic, coeff_x=cx, npts=npts, osc_table=synth.osc_table(spec),
               bin_edges=synth.bin_edges(spec))
    for k in ("coeff_many", "nknots_arr", "param_no", "coeff_tf1", "param_no_tf1", "npe", "npe_tf1"):
        out["mono_" + k] = ch[k]
    for k in ("sample_id", "erec", "osc_idx", "norm_off", "norm_idx", "bin_idx"):
        out["ev_" + k] = ev[k]
    laps = [synth.params(spec, i) for i in range(4)]
    out["params"] = np.stack(laps)
    tw, ew, mc, w2, llh, seg = [], [], [], [], [], []
    for i, p in enumerate(laps):
        par[:] = p
        samp.reweight()
        if i == 0:
            samp.add_data(samp.mc)
            out["data"] = samp.mc
        tw.append(mono.total_weights.copy())
        ew.append(samp.event_weights())
        mc.append(samp.mc)
        w2.append(samp.w2)
        seg.append(mono.segments.copy())
        llh.append([samp.get_likelihood()] + [samp.get_sample_likelihood(s) for s in range(spec.n_samples)])
    out.update(exp_spline_w=np.stack(tw), exp_event_w=np.stack(ew), exp_mc=np.stack(mc), exp_w2=np.stack(w2),
               exp_llh=np.array(llh), exp_segments=np.stack(seg))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("exp_")}, "llh", llh[-1])
