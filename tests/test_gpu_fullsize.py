"""BASELINE configs[1] at FULL size (1e6 events x 128 TSpline3 x 7 knots) through the C ABI, checked by
size-independent properties plus exact comparison on event windows the oracle can afford:
  - event windows [e0, e1) re-generated on the CPU (the generator is chunk-independent): spline weights
    and full event weights bit-exact against the serial oracle,
  - Asimov: LLH == 0 at the data point,
  - conservation: sum of bins == sum of the positive, in-range event weights (fp64),
  - W2 frozen after the first Reweight,
  - sharding: bins of two half-sample engines add up to the bins of the whole (what the all-reduce sums).
"""
import numpy as np
import pytest

import oracle_binding as ob
from mach3_amd import synth
from mach3_amd.engine import Engine, build_from_synth

pytestmark = pytest.mark.gpu


def test_config2_full_size_properties():
    spec = synth.CONFIGS["c2"]
    E = spec.n_events
    eng, mono, sh, par = build_from_synth(spec, chunk_events=131072)
    info = eng.info()
    assert info.n_events == E and info.n_splines == E * spec.n_params and info.n_tiles_brick == 0
    assert info.alg_bytes_fused == 16 * info.n_splines + 8 * E
    sh.RecordEventWeights(True)
    sh.Reweight()
    data = sh.GetMCArray()
    w2_first = sh.GetW2Array()
    sh.AddData(data)
    assert sh.GetLikelihood() == 0.0                      # Asimov (every bin: data == mc)
    windows = [(0, 3000), (500_000, 502_000), (E - 1500, E)]
    oracles = [ob.build_oracle_from_synth(spec, a, b) for a, b in windows]
    bins_all = np.concatenate([synth.events_chunk(spec, a, min(E, a + 250_000))["bin_idx"] for a in range(0, E, 250_000)])
    for lap in (1, 2):
        par[:] = synth.params(spec, lap)
        llh = sh.Step()
        assert llh > 0
        mc = sh.GetMCArray()
        ew = sh.GetEventWeights(0, E)
        # conservation, in fp64 (the kernel adds float weights into double bins)
        sel = (ew > 0) & (bins_all >= 0)
        ref = np.bincount(bins_all[sel], weights=ew[sel].astype(np.float64), minlength=spec.total_bins)
        assert np.allclose(mc, ref, rtol=1e-11, atol=0)
        assert np.array_equal(sh.GetW2Array(), w2_first)
        for (a, b), (omono, osamp, opar, ch, ev) in zip(windows, oracles):
            opar[:] = par
            osamp.reweight()
            assert np.array_equal(sh.GetSplineWeights(a, b - a).view(np.uint32), omono.total_weights.view(np.uint32))
            assert np.array_equal(ew[a:b].view(np.uint32), osamp.event_weights().view(np.uint32))
    for omono, osamp, *_ in oracles:
        osamp.close()
        omono.close()
    mc_whole = sh.GetMCArray()
    eng.close()

    # sharding: two ranks' worth of events on two handles (what m3hip_comm_init would all-reduce)
    half = E // 2
    parts = []
    for a, b in ((0, half), (half, E)):
        e2, m2, s2, p2 = build_from_synth(spec, chunk_events=131072, event_range=(a, b))
        p2[:] = synth.params(spec, 2)
        s2.Reweight()
        parts.append(s2.GetMCArray())
        e2.close()
    assert np.allclose(parts[0] + parts[1], mc_whole, rtol=1e-11, atol=0)
