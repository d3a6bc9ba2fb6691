"""Pins the TIMED CPU baseline: libm3oracle_omp.so (the reference's OpenMP build: `#pragma omp simd
reduction(*:totalWeight)` in CalcTotalEventWeight, Splines/SplineMonolith.cpp:792-830, and the array-section
reduction of FillArray_MP, Samples/SampleHandlerFD.cpp:390-448) against libm3oracle_serial.so (strict
left-to-right products, the bit-exact comparator of the GPU tests) on identical seeded inputs, at 1, 8 and 16
threads (the reference's recommended and maximum team, Manager/Monitor.cpp:122-138).

Bars.  The two evaluation passes (CalcSplineWeights) are elementwise: bit-identical.  The per-event product is
re-associated by the simd reduction (8 partial products at AVX2), so an event with n factors may differ by up
to ~n/2 ulp from the left-to-right product (each float multiply rounds by <= 1/2 ulp in either order): the
bar is n/2 + 1 ulp per event (measured: <= 12 ulp at 32 factors, <= 2 ulp only for events with <= 4 factors --
a flat 2-ulp bar is not attainable for a re-associated product of 32...128 floats).  Integer bins equal;
per-bin rates inherit the weights' ulp-level differences (a bin holding one event inherits that event's):
(n_max/2 + 5) float ulp relative (and 1e-12 whenever the event weights happen to be bit-equal); total LLH to the north-star bar, 1e-6 relative (measured <= 7e-8: float-ulp weight differences, not summation order)."""
import numpy as np
import pytest

import oracle_binding as ob
from mach3_amd import synth

SPECS = {
    "c1": synth.CONFIGS["c1"].scaled(6000),
    "c2": synth.CONFIGS["c2"].scaled(3000, n_params=40),
    "c3": synth.CONFIGS["c3"].scaled(20000),
    "c5": synth.CONFIGS["c5"].scaled(12000),
}


def _ulp_distance(a, b):
    """distance in representable floats between two float32 arrays of equal sign (weights here are positive)"""
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


@pytest.mark.parametrize("name", list(SPECS))
def test_openmp_build_agrees_with_serial_build(name):
    spec = SPECS[name]
    n = spec.n_events
    smono, ssamp, spar, ch, sev = ob.build_oracle_from_synth(spec, 0, n, variant="serial")
    omono, osamp, opar, _, oev = ob.build_oracle_from_synth(spec, 0, n, variant="omp", localise=True)
    n_fac = (ch["npe"][0::2].astype(np.int64) + ch["npe_tf1"][0::2])   # factors of the event's spline product
    bin_tol = float(n_fac.max() // 2 + 5) * 2.0 ** -23   # a bin with one event inherits that event's ulp distance
    olib = ob.load("omp")
    assert olib.m3o_is_multithread() == 1 and ob.load("serial").m3o_is_multithread() == 0
    try:
        ssamp.reweight()
        ssamp.add_data(ssamp.mc)
        osamp.reweight()
        osamp.add_data(ssamp.mc)          # the SAME data vector on both sides
        assert np.array_equal(ssamp.event_bins(), osamp.event_bins())
        for threads in (1, 8, 16):
            olib.m3o_set_num_threads(threads)
            assert olib.m3o_max_threads() == threads
            for lap in (1, 2, 3):
                spar[:] = synth.params(spec, lap)
                opar[:] = spar
                ssamp.reweight()
                osamp.reweight()
                assert np.array_equal(smono.segments, omono.segments)
                # the two evaluation passes are elementwise: identical in both builds
                assert np.array_equal(smono.spline_weights.view(np.uint32), omono.spline_weights.view(np.uint32))
                assert np.array_equal(smono.tf1_weights.view(np.uint32), omono.tf1_weights.view(np.uint32))
                # the per-event product is where the OpenMP build may re-associate
                assert np.all(_ulp_distance(smono.total_weights, omono.total_weights) <= n_fac // 2 + 1)
                sw, ow = ssamp.event_weights(), osamp.event_weights()
                ok = sw > 0   # an event weight of exactly 0 (NC-like osc entries) is 0 on both sides
                assert np.array_equal(sw[~ok], ow[~ok])
                assert np.all(_ulp_distance(sw[ok], ow[ok]) <= n_fac[ok] // 2 + 5)  # + norm, osc factors
                smc, omc = np.array(ssamp.mc), np.array(osamp.mc)
                if np.array_equal(sw.view(np.uint32), ow.view(np.uint32)):
                    assert np.allclose(omc, smc, rtol=1e-12, atol=0)   # only the fp64 summation order differs
                assert np.allclose(omc, smc, rtol=bin_tol, atol=0)
                assert np.array_equal(np.array(ssamp.w2), np.array(osamp.w2)) or np.allclose(osamp.w2, ssamp.w2, rtol=2 * bin_tol, atol=0)
                a, b = ssamp.get_likelihood(), osamp.get_likelihood()
                assert abs(a - b) <= 1e-6 * max(abs(a), 1.0), (threads, lap, a, b)   # measured: <= 7e-8
    finally:
        olib.m3o_set_num_threads(min(8, olib.m3o_max_threads()))
        ssamp.close(); smono.close(); osamp.close(); omono.close()
