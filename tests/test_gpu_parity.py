"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded
inputs.  Bars: per-event float weights and integer bins bit-exact; per-bin rates 1e-12
relative (fp64 atomics reorder the sum); total LLH 1e-9 relative (north_star asks 1e-6)."""
import numpy as np
import pytest

import oracle_binding as ob
from helpers import DeviceArray, make_monolith, natural_cubic
from mach3_amd import capi, synth
from mach3_amd.engine import SMonolith, SampleHandlerFD, build_from_synth

pytestmark = pytest.mark.gpu

RTOL_BIN = 1e-12
RTOL_LLH = 1e-9


def _llh_close(a, b, scale=1.0):
    assert abs(a - b) <= RTOL_LLH * max(abs(a), abs(b), scale), (a, b)


@pytest.fixture(scope="module")
def dense_pair(gpu_engine):
    """C2-shaped (dense, 7 knots, Poisson) at 3000 events, 40 parameters (tail loop: 40 % 8 == 0 is
    avoided on purpose with 37)."""
    spec = synth.CONFIGS["c2"].scaled(3000, n_params=37)
    eng, mono, sh, par = build_from_synth(spec, engine=gpu_engine, chunk_events=2048)
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
    yield spec, mono, sh, par, omono, osamp, opar, ev
    osamp.close()
    omono.close()


def test_spline_eval_seam2_bit_exact(dense_pair):
    """SMonolith::Evaluate + SynchroniseMemTransfer: cpu_total_weights bit-equal to the serial oracle."""
    spec, mono, sh, par, omono, osamp, opar, ev = dense_pair
    for lap in (0, 1, 2, 3):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        mono.Evaluate()
        mono.SynchroniseMemTransfer()
        omono.evaluate()
        assert np.array_equal(mono.SplineSegments, omono.segments)
        assert np.array_equal(mono.ParamValues.view(np.uint32), omono.param_values.view(np.uint32))
        assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))


def test_fused_step_matches_oracle(dense_pair):
    spec, mono, sh, par, omono, osamp, opar, ev = dense_pair
    info = sh.eng.info()
    assert info.n_events == spec.n_events and info.n_splines == spec.n_events * spec.n_params
    assert info.alg_bytes_fused == 16 * info.n_splines + 8 * info.n_events
    sh.RecordEventWeights(True)
    # Asimov data at nominal (ref: Fitters/MaCh3Factory.h:142-153); W2 frozen there
    par[:] = synth.params(spec, 0)
    opar[:] = par
    sh.Reweight()
    osamp.reweight()
    mc, w2 = sh.GetMCArray(), sh.GetW2Array()
    assert np.allclose(mc, osamp.mc, rtol=RTOL_BIN, atol=0)
    assert np.allclose(w2, osamp.w2, rtol=RTOL_BIN, atol=0)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2, 3, 4, 5):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        sh.Reweight()
        llh = sh.GetLikelihood()
        osamp.reweight()
        # device-side FindSplineSegment state == oracle state
        seg, val = sh.GetSegments(spec.n_params)
        assert np.array_equal(seg, omono.segments)
        assert np.array_equal(val.view(np.uint32), omono.param_values.view(np.uint32))
        # per-event weights bit-exact (spline total weight and full CalcWeightTotal)
        assert np.array_equal(sh.GetSplineWeights(0, spec.n_events).view(np.uint32), omono.total_weights.view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(0, spec.n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        assert np.array_equal(sh.GetW2Array(), w2)          # frozen (SampleHandlerFD.cpp:342)
        assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
        assert llh > 0
    sh.RecordEventWeights(False)
    # Step() == Reweight()+GetLikelihood()
    assert sh.Step() == pytest.approx(llh, rel=1e-12)


def test_bin_indices_bit_exact(dense_pair):
    spec, mono, sh, par, omono, osamp, opar, ev = dense_pair
    assert np.array_equal(ev["bin_idx"], osamp.event_bins())


@pytest.mark.parametrize("kind", [capi.kPoisson, capi.kBarlowBeeston, capi.kDembinskiAbdelmotteleb, capi.kPearson,
                                  capi.kIceCube])
def test_every_test_statistic(dense_pair, kind):
    spec, mono, sh, par, omono, osamp, opar, ev = dense_pair
    par[:] = synth.params(spec, 7)
    opar[:] = par
    sh.SetTestStatistic(kind)
    osamp.set_test_statistic(kind)
    sh.Reweight()
    osamp.reweight()
    got, exp = sh.GetLikelihood(), osamp.get_likelihood()
    # IceCube runs in x87 long double + lgammal in the reference
    tol = RTOL_LLH   # kIceCube included: the device evaluates a rearranged, well-conditioned form in fp64 (2e-11 per bin)
    assert abs(got - exp) <= tol * max(abs(exp), 1.0), (kind, got, exp)
    assert sh.GetSampleLikelihood(0) == pytest.approx(got, rel=1e-12)
    sh.SetTestStatistic(spec.test_statistic)
    osamp.set_test_statistic(spec.test_statistic)


def test_low_mc_and_empty_bins(dense_pair):
    """data in bins the MC never reaches, MC below _LOW_MC_BOUND_ (ref: SampleHandlerBase.cpp:17-87)"""
    spec, mono, sh, par, omono, osamp, opar, ev = dense_pair
    par[:] = synth.params(spec, 3)
    opar[:] = par
    sh.Reweight()
    osamp.reweight()
    data = osamp.mc.copy()
    data[-5:] = [0.0, 3.0, 1e-6, 5e-6, 2.0]   # the last bins hold (almost) no MC
    data[0] = 0.0
    for kind in (capi.kPoisson, capi.kBarlowBeeston, capi.kPearson, capi.kDembinskiAbdelmotteleb):
        sh.SetTestStatistic(kind)
        osamp.set_test_statistic(kind)
        sh.AddData(data)
        osamp.add_data(data)
        _llh_close(sh.GetLikelihood(), osamp.get_likelihood())
    sh.SetTestStatistic(spec.test_statistic)
    osamp.set_test_statistic(spec.test_statistic)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)


@pytest.mark.parametrize("kind", [capi.kPoisson, capi.kBarlowBeeston, capi.kDembinskiAbdelmotteleb, capi.kPearson,
                                  capi.kIceCube])
def test_llh_kernel_on_a_grid_of_bins(fresh_engine, kind):
    """Kernel (iii) bin by bin (one sample per bin) on a grid of (data, mc, w2) triples covering
    every branch of GetTestStatLLH (ref: Samples/SampleHandlerBase.cpp:17-192): data == 0,
    mc below / at / above _LOW_MC_BOUND_, data <=> mc, w2 == 0, tiny and large w2."""
    eng = fresh_engine
    lib = ob.load("serial")
    vals_d = [0.0, 1e-7, 5e-6, 1e-5, 2e-5, 0.3, 1.0, 7.0, 38.875538093969226, 1234.0]
    vals_m = [0.0, 1e-7, 5e-6, 1e-5, 2e-5, 0.31, 1.0, 6.5, 38.15916385687888, 1200.0]
    vals_w = [0.0, 1e-12, 1e-3, 0.7, 29.382163026679336, 5e3]
    grid = np.array([(d, m, w) for d in vals_d for m in vals_m for w in vals_w if not (m == 0.0 and w > 0)])
    if kind == capi.kIceCube:
        # fp64 on the device (a rearranged form that stays well-conditioned) vs x87 long double + lgammal in the
        # reference, whose lgammal(data+a) - lgammal(a) loses digits once a = mc^2/w2 + 1 is huge (3e-6 absolute at
        # a = 1e12): the grid is kept to a <= 1e7 (MC stat error >= 3e-4), where the reference is good to 1e-10.
        grid = grid[(grid[:, 2] == 0) | (grid[:, 1] ** 2 / np.maximum(grid[:, 2], 1e-300) <= 1e7)]
    nb = len(grid)
    spec = synth.CONFIGS["c2"].scaled(256, n_params=2)
    eng, mono, sh0, par = build_from_synth(spec, engine=eng)
    sh = SampleHandlerFD(eng, nb, np.arange(nb + 1, dtype=np.int32), par, kind)
    sh.AddData(np.ascontiguousarray(grid[:, 0]))
    mc, w2 = np.ascontiguousarray(grid[:, 1]), np.ascontiguousarray(grid[:, 2])
    eng._chk(eng.lib.m3hip_debug_set_bins(eng.h, capi._ptr(mc, np.float64), capi._ptr(w2, np.float64), nb))
    total = sh.GetLikelihood()
    per = sh._per_sample.copy()
    exp = np.array([lib.m3o_test_stat_llh(kind, *g) for g in grid])
    both_nan = np.isnan(per) & np.isnan(exp)
    tol = 1e-9 if kind == capi.kIceCube else 1e-12
    ok = both_nan | (np.abs(per - exp) <= tol * np.maximum(np.abs(exp), 1.0))
    bad = np.nonzero(~ok)[0]
    assert len(bad) == 0, [(grid[i], per[i], exp[i]) for i in bad[:5]]
    fin = ~both_nan
    if fin.all():
        assert total == pytest.approx(exp.sum(), rel=1e-9 if kind == capi.kIceCube else 1e-12)


def test_exact_knot_hits_and_out_of_range(fresh_engine):
    """Segment edge cases through the DEVICE-side FindSplineSegment: x <= x0, x >= x_{n-1} -> n-2,
    x == knot with and without the previous-segment shortcut (ref: Splines/SplineBase.cpp:44-109)."""
    gpu_engine = fresh_engine
    X7 = np.arange(-3.0, 4.0)
    rng = np.random.default_rng(11)
    P, E = 3, 300
    events = [[(p, "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))) for p in range(P)] for _ in range(E)]
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    par = np.zeros(P)
    opar = np.zeros(P)
    omono.set_spline_pointers(opar)
    mono = SMonolith(gpu_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(gpu_engine, 1, np.array([0, 4], np.int32), par)
    sh.SetupEvents(0, np.arange(E, dtype=np.int32) % 4, None, None, None)
    sh.RecordEventWeights(True)
    seq = [[-5, 9, 1.5], [-3, 3, 1.0], [0.5, -0.5, 1.0], [1.0, 2.0, -2.5], [1.0, 2.5, 1.0], [0, 0, 0],
           [2.9999, -2.9999, 1e-30], [3.0000002, -3.0000002, -1e-30]]
    for vals in seq:
        par[:] = vals
        opar[:] = vals
        sh.Reweight()
        sh.GetLikelihood()
        omono.evaluate()
        seg, val = sh.GetSegments(P)
        assert np.array_equal(seg, omono.segments), vals
        assert np.array_equal(sh.GetSplineWeights(0, E).view(np.uint32), omono.total_weights.view(np.uint32)), vals
    omono.close()


def test_fill_semantics_on_device(fresh_engine):
    """w <= 0 dropped, bin -1 dropped, norm -> osc -> spline order, W2 frozen / UpdateW2
    (ref: Samples/SampleHandlerFD.cpp:316-463, 568-594); multi-sample likelihoods."""
    gpu_engine = fresh_engine
    X7 = np.arange(-3.0, 4.0)
    c = natural_cubic(X7, 1 + 0.1 * X7)
    E = 600
    events = [[(0, "spl", c)] for _ in range(E)]
    cx, npts, ch = make_monolith(1, 7, {0: X7}, events)
    rng = np.random.default_rng(3)
    par = np.array([0.0, 2.0, -1.0, 0.5])
    kin = rng.uniform(-0.5, 4.5, E)
    sample_id = (rng.random(E) < 0.5).astype(np.int32)
    nn = rng.integers(0, 3, E)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = rng.integers(1, 4, int(norm_off[-1])).astype(np.int16)
    osc_idx = rng.integers(-1, 3, E).astype(np.int32)
    osc = np.array([0.5, 0.25, 0.0], np.float32)     # 0.0: NC flavour-change events (M3::Zero)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, 1, 7, cx, npts, ch)
    opar = par.copy()
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    edges = np.array([0.0, 1.0, 2.0, 4.0])
    b.add_uniform([edges])
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, omono)
    bins = synth.find_global_bin_uniform_1d(edges, kin, sample_id, 3)
    assert np.array_equal(bins, osamp.event_bins())
    assert (bins == -1).any()

    mono = SMonolith(gpu_engine, 1, 7, cx, npts, [ch])
    mono.setSplinePointers(par, [0])
    sh = SampleHandlerFD(gpu_engine, 2, np.array([0, 3, 6], np.int32), par, capi.kBarlowBeeston)
    osamp.set_test_statistic(capi.kBarlowBeeston)
    sh.SetupEvents(0, bins, osc_idx, norm_off, norm_idx)
    sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    ew = osamp.event_weights()
    assert (ew < 0).any() and (ew == 0).any() and (ew > 0).any()
    assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), ew.view(np.uint32))
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN)
    assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN)
    data = np.round(osamp.mc * 1.1)
    sh.AddData(data)
    osamp.add_data(data)
    w2_first = sh.GetW2Array()
    par[:] = [0.7, 1.5, -0.5, 0.9]
    opar[:] = par
    sh.Reweight()
    osamp.reweight()
    assert np.array_equal(sh.GetW2Array(), w2_first)
    _llh_close(sh.GetLikelihood(), osamp.get_likelihood())
    for s in (0, 1):
        _llh_close(sh.GetSampleLikelihood(s), osamp.get_sample_likelihood(s))
    osamp.close()
    omono.close()

    # UpdateW2 = true: W2 follows every reweight
    omono = ob.OracleMonolith(lib, 1, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, omono)
    osamp.set_test_statistic(capi.kBarlowBeeston, True)
    sh2 = SampleHandlerFD(gpu_engine, 2, np.array([0, 3, 6], np.int32), par, capi.kBarlowBeeston, update_w2=True)
    for vals in ([0.0, 2.0, -1.0, 0.5], [0.7, 1.5, 0.5, 0.9]):
        par[:] = vals
        opar[:] = vals
        sh2.Reweight()
        osamp.reweight()
        assert np.allclose(sh2.GetW2Array(), osamp.w2, rtol=RTOL_BIN)
    osamp.close()
    omono.close()


def test_non_finite_weights_follow_the_reference(fresh_engine):
    """`if (totalweight <= 0.) continue;` (SampleHandlerFD.cpp:432) lets a NaN weight through: its bin, and with it
    the likelihood, turn NaN -- which is how a broken response shows up in a chain.  -inf is dropped, +inf fills."""
    X7 = np.arange(-3.0, 4.0)
    good = natural_cubic(X7, 1 + 0.1 * X7)
    E = 700
    events = []
    special = {5: np.nan, 300: np.inf, 301: -np.inf, 650: np.nan}
    for e in range(E):
        c = good.copy()
        if e in special:
            c[:, 0] = special[e]
        events.append([(0, "spl", c), (1, "spl", good)])
    cx, npts, ch = make_monolith(2, 7, {0: X7, 1: X7}, events)
    par = np.array([0.3, -0.4])
    kin = np.linspace(0.01, 3.99, E)
    edges = np.array([0.0, 1.0, 2.0, 3.0, 4.0])
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, 2, 7, cx, npts, ch)
    opar = par.copy()
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    sid = np.zeros(E, np.int32)
    off0 = np.zeros(E + 1, np.uint64)
    osamp = ob.OracleSample(lib, b, sid, kin, off0, np.zeros(0, np.int16), opar, np.full(E, -1, np.int32), np.ones(1, np.float32), omono)
    bins = osamp.event_bins()
    mono = SMonolith(fresh_engine, 2, 7, cx, npts, [ch])
    mono.setSplinePointers(par, [0, 1])
    sh = SampleHandlerFD(fresh_engine, 1, np.array([0, 4], np.int32), par, capi.kPoisson)
    sh.SetupEvents(0, bins, None, None, None)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    ew, gw = osamp.event_weights(), sh.GetEventWeights(0, E)
    assert np.isnan(ew[[5, 650]]).all() and ew[300] == np.inf and ew[301] == -np.inf
    assert np.array_equal(np.isnan(gw), np.isnan(ew))
    fin = ~np.isnan(ew)
    assert np.array_equal(gw[fin].view(np.uint32), ew[fin].view(np.uint32))
    mc, omc = sh.GetMCArray(), np.asarray(osamp.mc)
    assert list(bins[[5, 300, 301, 650]]) == [0, 1, 1, 3]
    assert np.isnan(omc[0]) and np.isnan(omc[3]) and omc[1] == np.inf and np.isfinite(omc[2])   # +inf filled, -inf was dropped
    assert np.array_equal(np.isnan(mc), np.isnan(omc))
    assert np.allclose(mc[~np.isnan(omc)], omc[~np.isnan(omc)], rtol=RTOL_BIN)
    data = np.array([100.0, 150.0, 170.0, 120.0])
    sh.AddData(data)
    osamp.add_data(data)
    got, want = sh.GetLikelihood(), osamp.get_likelihood()
    assert (np.isnan(got) and np.isnan(want)) or got == want
    assert np.isnan(want)
    osamp.close()
    omono.close()


def test_mixed_uniform_tf1_and_short_splines(fresh_engine):
    """TSpline3 parameters with 3/7/5 knots and TF1 parameters in one uniform event list; the
    product order is splines (event order) then TF1s (ref: SplineMonolith.cpp:792-830)."""
    gpu_engine = fresh_engine
    xs = {0: np.array([-1.0, 0, 1]), 2: np.arange(-3.0, 4.0), 3: np.array([-2.0, -1, 0, 1, 2])}
    rng = np.random.default_rng(1)
    E, P = 777, 6
    events = []
    for _ in range(E):
        ev = []
        for p in range(P):
            if p in xs:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(P, 7, xs, events)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    par = np.zeros(P + 2)
    opar = par.copy()
    # spline parameter i is driven by global entry i+2 (setSplinePointers with a non-trivial map)
    idx = np.arange(P) + 2
    omono.set_spline_pointers(opar, idx)
    mono = SMonolith(gpu_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par, idx)
    for k in range(6):
        par[2:] = rng.uniform(-3.5, 3.5, P)
        opar[:] = par
        mono.Evaluate()
        mono.SynchroniseMemTransfer()
        omono.evaluate()
        assert np.array_equal(mono.SplineSegments, omono.segments)
        assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    omono.close()


def test_api_error_paths(fresh_engine):
    gpu_engine = fresh_engine
    eng = gpu_engine
    spec = synth.CONFIGS["c2"].scaled(300, n_params=4)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, 300)
    bad = dict(ch)
    bad["param_no"] = ch["param_no"].copy()
    bad["param_no"][3] = 77
    eng.monolith_begin(4, 7, cx, npts)
    with pytest.raises(capi.M3HipError) as ei:
        eng.monolith_append(bad)
    assert ei.value.code == -1 and "paramNo" in str(ei.value)
    # a good upload still works afterwards
    mono = SMonolith(eng, 4, 7, cx, npts, [ch])
    par = synth.params(spec, 1)
    mono.setSplinePointers(par)
    mono.SplineSegments[:] = 0
    mono.ParamValues[:] = 0
    with pytest.raises(capi.M3HipError):
        seg = np.full(4, 6, np.int16)     # segment beyond nPts-2
        eng._chk(eng.lib.m3hip_eval_splines(eng.h, capi._ptr(mono.ParamValues, np.float32), capi._ptr(seg, np.int16), None))
    sh = SampleHandlerFD(eng, 1, np.array([0, 10], np.int32), par)
    with pytest.raises(capi.M3HipError):
        sh.SetupEvents(0, np.full(300, 10, np.int32), None, None, None)   # bin index out of range
    with pytest.raises(capi.M3HipError):
        sh.AddData(np.zeros(9))
    # indices the kernels use unguarded are checked against the table / vector sizes before anything is launched
    bins = np.zeros(300, np.int32)
    sh.SetupEvents(0, bins, np.full(300, 5, np.int32), None, None)
    with pytest.raises(capi.M3HipError, match="oscillation"):
        sh.Reweight()                                   # oscillation indices, no table
    sh.SetOscillationWeights(np.ones(5, np.float32))
    with pytest.raises(capi.M3HipError, match="oscillation"):
        sh.Reweight()                                   # index 5 of a 5-entry table
    sh.SetOscillationWeights(np.ones(6, np.float32))
    sh.Reweight()
    assert sh.GetMCArray().sum() > 0
    with pytest.raises(capi.M3HipError, match="norm_idx"):
        sh.SetupEvents(0, bins, None, np.arange(301, dtype=np.uint64), None)
    sh.SetupEvents(0, bins, None, np.arange(301, dtype=np.uint64), np.full(300, len(par), np.int16))
    with pytest.raises(capi.M3HipError, match="norm parameter"):
        sh.Reweight()                                   # norm index == len(par)
    mono.setSplinePointers(par, np.arange(4) + len(par) - 3)
    with pytest.raises(capi.M3HipError, match="map"):
        sh.Reweight()


@pytest.mark.parametrize("groups", [1, 2, 4])
@pytest.mark.parametrize("cfg,n_events,n_params", [("c3", 5000, 96), ("c5", 3000, 512), ("c3", 1024 * 9 + 77, 512)])
def test_ragged_bricks_match_oracle(fresh_engine, monkeypatch, cfg, n_events, n_params, groups):
    """Configs 3-5 shape: random sparse occupancy, mixed TSpline3 / TF1, Barlow-Beeston.  Goes
    through the brick kernel, in bricks of 1024, 2048 and 4096 events; weights bit-exact, bins 1e-12, LLH 1e-9."""
    monkeypatch.setenv("M3HIP_BRICK_GROUPS", str(groups))
    spec = synth.CONFIGS[cfg].scaled(n_events, n_params=n_params)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine, chunk_events=4096)
    info = eng.info()
    assert info.brick_groups == groups
    assert info.n_tiles_brick == sum(-(-min(4096, n_events - a) // (1024 * groups)) for a in range(0, n_events, 4096)) and info.n_tiles_uniform == 0
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, n_events)
    assert info.n_splines == ch["n_splines"] and info.n_tf1 == ch["n_tf1"]
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2, 3):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetSplineWeights(0, n_events).view(np.uint32), omono.total_weights.view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(0, n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    # seam 2 on the same data
    mono.Evaluate()
    mono.SynchroniseMemTransfer()
    omono.evaluate()
    assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    osamp.close()
    omono.close()


@pytest.mark.parametrize("groups", [1, 4])
@pytest.mark.parametrize("mask", [0, 4])
def test_streams_around_or_through_the_infinity_cache_give_the_same_numbers(fresh_engine, monkeypatch, mask, groups):
    """Whether the brick kernel loads the TF1 codes non-temporally (ReweightArgs::nt_codes, the kernel's NTC instantiation; the
    engine chooses by the size of the streams a step re-reads, M3HIP_NT_MASK forces the choice) is a matter of speed only: event
    weights bit-exact against the serial oracle, bins 1e-12, LLH 1e-9 either way."""
    monkeypatch.setenv("M3HIP_BRICK_GROUPS", str(groups))
    monkeypatch.setenv("M3HIP_NT_MASK", str(mask))
    n_events = 4096 + 1500
    spec = synth.CONFIGS["c3"].scaled(n_events, n_params=128)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    assert eng.info().n_tiles_brick > 0 and eng.info().n_tiles_uniform == 0
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, n_events)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    info = eng.info()
    assert info.tf1_codes_around == (1 if mask == 4 else 0) and 0 < info.cached_stream_bytes < 10_000_000
    osamp.close()
    omono.close()


@pytest.mark.parametrize("groups", [1, 4])
@pytest.mark.parametrize("n_norm", [40, 300])
def test_bricks_with_more_norm_parameters_than_the_packed_word_holds(fresh_engine, monkeypatch, groups, n_norm):
    """Up to six norm parameters per event: out of 40, the brick kernel takes an event's first three from its packed sample
    word and the rest from the index planes, in slot order (SampleHandlerFD.cpp:568-594); out of 300 (indices spread beyond
    the word's 6-bit codes) the word carries bin and oscillation index only and every norm slot comes from the planes.
    Event weights bit-exact either way."""
    monkeypatch.setenv("M3HIP_BRICK_GROUPS", str(groups))
    n_events = 4096 + 900
    spec = synth.CONFIGS["c3"].scaled(n_events, n_params=96, n_norm=n_norm, max_norm_per_event=6)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    assert eng.info().n_tiles_brick > 0 and eng.info().n_tiles_uniform == 0
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, n_events)
    assert int(np.diff(ev["norm_off"]).max()) == 6
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    osamp.close()
    omono.close()


def test_oscillation_weights_change_every_step(fresh_engine):
    """OscillationHandler::Evaluate runs inside every Reweight (SampleHandlerFD.cpp:319-324): the weight table
    behind the events' oscillation pointers is re-uploaded per step (asynchronously, no sync in between) and
    the fused result follows it."""
    spec = synth.CONFIGS["c2"].scaled(6000, n_params=12)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
    otbl = osamp.keep[7]                       # the oracle's events point into this array
    assert otbl.dtype == np.float32 and (ev["osc_idx"] >= 0).sum() > 4000
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    rng = np.random.default_rng(8)
    sh.RecordEventWeights(True)
    for lap in range(1, 8):
        otbl[:] = rng.random(len(otbl)).astype(np.float32)
        if lap == 5:
            otbl[::3] = 0.0                    # zero weights: those events drop out of the fill (w <= 0)
        sh.SetOscillationWeights(otbl)
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, spec.n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    # back-to-back uploads without a step in between: the last one wins
    sh.SetOscillationWeights(rng.random(len(otbl)).astype(np.float32))
    otbl[:] = rng.random(len(otbl)).astype(np.float32)
    sh.SetOscillationWeights(otbl)
    sh.SetOscillationWeights(otbl)
    llh = sh.Step()
    osamp.reweight()
    _llh_close(llh, osamp.get_likelihood())
    osamp.close()
    omono.close()


@pytest.mark.parametrize("cfg,n_params,osc_n,column", [("c2", 12, 4000, None), ("c3", 96, 4000, None), ("c3", 96, 40000, None), ("c3", 96, 40000, "1"), ("c5", 128, 40000, None)])
def test_oscillation_table_in_device_memory_is_read_in_place(monkeypatch, cfg, n_params, osc_n, column):
    """A GPU oscillation calculator leaves its weights on the device, and the reference's events alias that array
    (SampleHandlerFD.cpp:1108-1133); m3hip_set_osc_table_device reads it where it lies.  Uniform tiles and bricks; a small table
    and one beyond 64 KB (the size from which a RESTING table is folded into the packed words -- a moving one never is; with
    M3HIP_OSC_COLUMN=1 every table is, at once, and unfolded again whenever it moves).  The caller refills ONE array in
    place and announces it, lap after lap; then the table rests (word form), moves again through the host entry point, and
    comes back to the device.  Every lap against the oracle: event weights bit for bit."""
    if column is not None:
        monkeypatch.setenv("M3HIP_OSC_COLUMN", column)
    spec = synth.CONFIGS[cfg].scaled(9000, n_params=n_params, osc_n=osc_n)
    eng, mono, sh, par = build_from_synth(spec)
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
    otbl = osamp.keep[7]
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    sh.RecordEventWeights(True)
    rng = np.random.default_rng(81)
    d_tbl = DeviceArray(len(otbl))

    def check(lap):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, spec.n_events).view(np.uint32), osamp.event_weights().view(np.uint32)), lap
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())

    for lap in range(1, 6):                      # the calculator refills its array in place; the engine is told
        otbl[:] = rng.random(len(otbl)).astype(np.float32)
        if lap == 3:
            otbl[::5] = 0.0
        d_tbl.set(otbl)                          # ordering: the refill has completed before the reweight is called
        sh.SetOscillationWeightsDevice(d_tbl.ptr, len(otbl))
        check(lap)
    for lap in range(6, 11):                     # left alone: after a few steps the weights ride in the packed words (bricks, large table)
        check(lap)
    otbl[:] = rng.random(len(otbl)).astype(np.float32)
    d_tbl.set(otbl)                              # refilled and announced again: the words go back to the index form
    sh.SetOscillationWeightsDevice(d_tbl.ptr, len(otbl))
    check(11)
    otbl[:] = rng.random(len(otbl)).astype(np.float32)
    sh.SetOscillationWeights(otbl)               # a host table takes over: the engine's own copy is current again
    d_tbl.set(np.zeros(len(otbl), np.float32))   # ... and the caller's array is no longer read
    check(12)
    check(13)
    # another array of the caller's
    otbl[:] = rng.random(len(otbl)).astype(np.float32)
    d2 = DeviceArray(len(otbl))
    d2.set(otbl)
    sh.SetOscillationWeightsDevice(d2.ptr, len(otbl))
    check(14)
    # a host pointer is refused, not dereferenced on the device
    with pytest.raises(capi.M3HipError, match="device"):
        sh.SetOscillationWeightsDevice(otbl.ctypes.data, len(otbl))
    with pytest.raises(capi.M3HipError, match="empty"):
        sh.SetOscillationWeightsDevice(0, len(otbl))
    check(15)                                    # a refused call leaves the table in force
    osamp.close()
    omono.close()
    eng.close()
    d_tbl.free()
    d2.free()


@pytest.mark.parametrize("cfg,n_params,shape", [("c2", 24, (3, 60, 40)), ("c3", 96, (3, 60, 40)), ("c2", 24, (4, 100, 80)), ("c2", 24, (2, 50, 30))])
def test_thousands_of_bins(fresh_engine, cfg, n_params, shape):
    """Bin vectors too large to privatise in LDS (2-D samples: 3 x 60 x 40 = 7 200 and 4 x 100 x 80 = 32 000 global
    bins; 2 x 50 x 30 = 3 000 for the sizes in between): the fill goes straight into the workgroup's global replica
    (few replicas: the adds spread over the bins by themselves), the bin likelihood runs as one workgroup per 1 024
    bins + a final sum beyond 4 096 bins (sample boundaries fall inside workgroups) and as one 1 024-lane workgroup
    below; uniform tiles (c2 shape) and bricks (c3 shape), device-side 2-D binning, Barlow-Beeston with W2
    accumulated on the first step only."""
    E = 9000
    n_samp, nx, ny = shape
    spec = synth.CONFIGS[cfg].scaled(E, n_params=n_params)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, E)
    ev = synth.events_chunk(spec, 0, E)
    rng = np.random.default_rng(12)
    ex, ey = np.linspace(0.0, 5.0, nx + 1), np.linspace(-1.0, 1.0, ny + 1)
    sample_id = rng.integers(0, n_samp, E).astype(np.int32)
    kin = np.stack([ev["erec"], rng.uniform(-1.05, 1.05, E)], axis=1)
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    for _ in range(n_samp):
        b.add_uniform([ex, ey])
    assert b.n_bins() == n_samp * nx * ny
    par = synth.params(spec, 0)
    opar = par.copy()
    omono = ob.OracleMonolith(lib, spec.n_params, spec.n_knots, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, ev["norm_off"], ev["norm_idx"], opar, ev["osc_idx"], synth.osc_table(spec), omono)
    osamp.set_test_statistic(capi.kBarlowBeeston)

    mono = SMonolith(fresh_engine, spec.n_params, spec.n_knots, cx, npts, [ch])
    mono.setSplinePointers(par)
    info = fresh_engine.info()
    assert (info.n_tiles_brick > 0) == (cfg == "c3")
    sh = SampleHandlerFD(fresh_engine, n_samp, None, par, capi.kBarlowBeeston, bin_edges=[[ex, ey]] * n_samp)
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    bins = sh.GetEventBins(0, E)
    assert np.array_equal(bins, osamp.event_bins())
    sh.SetupEvents(0, bins, ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
    sh.Reweight()
    osamp.reweight()
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2, 3):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
        per = [sh.GetSampleLikelihood(i) for i in range(n_samp)]
        assert np.isclose(sum(per), llh, rtol=1e-12)
        for i in range(n_samp):
            _llh_close(per[i], osamp.get_sample_likelihood(i), scale=max(1.0, abs(llh)))
    osamp.close()
    omono.close()


def _ragged_events(rng, E, P, occ, xs, empty_frac=0.0, tf1_only_frac=0.0):
    events = []
    for _ in range(E):
        u = rng.random()
        ev = []
        if u >= empty_frac:
            for p in range(P):
                if rng.random() < occ:
                    if p in xs and u >= empty_frac + tf1_only_frac:
                        ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
                    elif p not in xs:
                        ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    return events


def _eval_matches(engine, P, K, cx, npts, ch, rng, laps=3):
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, K, cx, npts, ch)
    par, opar = np.zeros(P), np.zeros(P)
    omono.set_spline_pointers(opar)
    mono = SMonolith(engine, P, K, cx, npts, [ch])
    mono.setSplinePointers(par)
    for _ in range(laps):
        par[:] = rng.uniform(-3.5, 3.5, P)
        opar[:] = par
        mono.Evaluate()
        mono.SynchroniseMemTransfer()
        omono.evaluate()
        assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    info = engine.info()
    omono.close()
    return info


@pytest.mark.parametrize("case", ["one_event", "holes", "deep_brick", "padded", "mode_blocks", "many_params"])
def test_extreme_shapes(fresh_engine, case, monkeypatch):
    """Edges of the layout: a single event; ragged groups in which a third of the events carry no response
    at all (weight stays 1, SplineMonolith.cpp:798) and some carry TF1s only; a brick with ~190k responses
    in its 1024 events (24+ stages, close to the 30-stage limit); the same 80 %-occupied group laid out as
    identity-padded uniform tiles (what groups that carry at least 88 % of their union list become by default); 2 600 parameters."""
    rng = np.random.default_rng({"one_event": 1, "holes": 2, "deep_brick": 3, "padded": 3, "mode_blocks": 4, "many_params": 5}[case])
    X7, X5, X3 = np.arange(-3.0, 4.0), np.array([-2.0, -1, 0, 1, 2]), np.array([-1.0, 0, 1])
    if case == "one_event":
        xs = {0: X7}
        cx, npts, ch = make_monolith(1, 7, xs, [[(0, "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7)))]])
        info = _eval_matches(fresh_engine, 1, 7, cx, npts, ch, rng, laps=5)
        assert info.n_tiles_uniform == 1 and info.n_tiles_brick == 0
    elif case == "mode_blocks":
        # blocks of events sharing a list, boundaries off the 1024 grid: uniform tiles with small bricks in between
        P = 10
        quad = lambda: natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))
        events = [[(p, "spl", quad()) for p in (0, 1, 2, 3, 4)] for _ in range(1000)]
        events += [[(p, "spl", quad()) for p in (3, 4, 5, 6, 7, 8)] for _ in range(1300)]
        events += [[(p, "spl", quad()) for p in range(P) if rng.random() < 0.3] for _ in range(1500)]
        cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
        info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng)
        assert info.n_tiles_uniform == 7 and info.n_tiles_brick == 3
    elif case == "many_params":
        # more parameters than the kernels keep in LDS (1 024) and more runs than one brick stage holds (1 024): the
        # per-step tables are then read from global memory and the stages are cut by run count
        P = 2600
        xs = {p: (X3, X5)[p % 2] for p in range(P) if p % 5 != 4}   # four fifths TSpline3, one fifth TF1
        events = _ragged_events(rng, 1400, P, 0.012, xs, empty_frac=0.02)
        cx, npts, ch = make_monolith(P, 7, xs, events)
        info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng)
        assert info.n_tiles_brick == 2 and info.n_tiles_uniform == 0
        assert len(np.unique(ch["param_no"][: int(ch["npe"].reshape(-1, 2)[:1024, 0].sum())])) > 1024   # > 1 024 runs in the first brick
    elif case == "holes":
        P = 40
        xs = {p: (X7, X5, X3)[p % 3] for p in range(0, P, 2)}   # even parameters are TSpline3, odd are TF1
        events = _ragged_events(rng, 2500, P, 0.3, xs, empty_frac=0.33, tf1_only_frac=0.1)
        assert sum(1 for e in events if not e) > 700
        cx, npts, ch = make_monolith(P, 7, xs, events)
        info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng)
        assert info.n_tiles_brick == 3
    else:
        P = 230
        xs = {p: (X7, X5)[p % 2] for p in range(P) if p % 3 != 2}   # two thirds TSpline3, one third TF1
        events = _ragged_events(rng, 1024, P, 0.8, xs)
        n_resp = sum(len(e) for e in events)
        assert 180_000 < n_resp < 200_000
        cx, npts, ch = make_monolith(P, 7, xs, events)
        # 80 % full: a brick by default (>= 88 % would be padded)
        monkeypatch.setenv("M3HIP_PAD_MIN_FILL", "2" if case == "deep_brick" else "0.7")
        info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng, laps=2)
        if case == "deep_brick":
            assert info.n_tiles_brick == 1 and info.n_tiles_padded == 0
        else:
            assert info.n_tiles_brick == 0 and info.n_tiles_uniform == 4 and info.n_tiles_padded == 4
            assert 0.15 < info.pad_responses / (info.pad_responses + n_resp) < 0.25


def test_many_distinct_uniform_lists(fresh_engine):
    """90 uniform tiles, each with its own list of 104 parameters: 9 360 distinct slots, more than k_prep resolves
    inline (8 192), so the per-step slot resolution runs as its own kernel (k_resolve)."""
    rng = np.random.default_rng(77)
    X7 = np.arange(-3.0, 4.0)
    P, T, L = 200, 90, 100
    xs = {p: X7 for p in range(0, P, 50)}             # parameters 0, 50, 100, 150 are TSpline3, the rest TF1
    quads = [natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7)) for _ in range(64)]
    events = []
    for t in range(T):
        plist = sorted(set(range(t, t + L)) | set(xs))
        for _ in range(256):
            events.append([(p, "spl", quads[int(rng.integers(0, 64))]) if p in xs else (p, "tf1", (0.02 * rng.standard_normal(), 1.0))
                           for p in plist])
    cx, npts, ch = make_monolith(P, 7, xs, events)
    info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng, laps=4)
    assert info.n_tiles_uniform == T and info.n_tiles_brick == 0 and info.n_tiles_padded == 0


@pytest.mark.parametrize("per_event", [0.6, 1.7, 2.8, 3.8, 4.8])
def test_one_stage_bricks_of_every_record_length(fresh_engine, per_event):
    """A lane's response words of a stage are a record of 1 .. 5 uint16 read with one 12-byte load (m3hip_layout.cpp: records of
    2-byte granularity, so with 3 or 5 words every other record starts on a 2-byte boundary): one 1024-event brick whose TSpline3
    responses fill ONE stage with 1, 2, 3, 4 and 5 words per lane, a second, partly filled brick behind it; weights bit for bit."""
    rng = np.random.default_rng(int(per_event * 10))
    X7, X5 = np.arange(-3.0, 4.0), np.array([-2.0, -1, 0, 1, 2])
    P = 48
    xs = {p: (X7 if p % 2 else X5) for p in range(P) if p % 6}       # five sixths TSpline3, one sixth TF1
    events = _ragged_events(rng, 1024 + 300, P, per_event / len(xs), xs)
    n_spl = sum(1 for e in events[:1024] for r in e if r[1] == "spl")
    words = -(-n_spl // 1024)
    assert words == int(per_event) + 1 and n_spl <= 5120              # (run padding may add a few list entries: stays in one stage)
    cx, npts, ch = make_monolith(P, 7, xs, events)
    info = _eval_matches(fresh_engine, P, 7, cx, npts, ch, rng, laps=3)
    assert info.n_tiles_brick == 2 and info.n_tiles_uniform == 0


@pytest.mark.parametrize("grouping", ["on", "off"])
def test_shuffled_modes_are_grouped_by_the_engine(fresh_engine, grouping, monkeypatch):
    """Events of three interaction modes (each with its own parameter list, a tenth of the splines flat and dropped)
    in RANDOM order: as they come, every 256-event tile mixes the lists and the chunk would be all bricks.  The
    engine groups the events by list for processing -- uniform / padded tiles -- while every per-event array and
    result keeps the caller's order: weights, bins and likelihood against the oracle in the original order."""
    rng = np.random.default_rng(55)
    X7, X5 = np.arange(-3.0, 4.0), np.array([-2.0, -1, 0, 1, 2])
    P, E = 24, 5000
    xs = {p: (X7 if p % 2 else X5) for p in range(P) if p % 4}
    modes = [list(range(0, 12)), list(range(6, 20)), list(range(10, 24))]
    which = rng.integers(0, 3, E)
    events = []
    for e in range(E):
        ev = []
        for p in modes[which[e]]:
            if p in xs:
                if rng.random() < 0.1:
                    continue                                   # flat spline: dropped
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(P, 7, xs, events)
    monkeypatch.setenv("M3HIP_PAD_MIN_FILL", "0.8")   # this sample's tiles sit around the default (0.88): the test is about grouping
    if grouping == "off":
        monkeypatch.setenv("M3HIP_GROUP_EVENTS", "0")
    n_norm = 3
    par = np.zeros(P + n_norm)
    par[P:] = 1.0
    opar = par.copy()
    edges = np.linspace(0.0, 5.0, 41)
    kin = rng.gamma(3.0, 0.45, E)
    sample_id = rng.integers(0, 2, E).astype(np.int32)
    nn = rng.integers(0, 3, E)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, n_norm, int(norm_off[-1]))).astype(np.int16)
    osc_idx = rng.integers(-1, 20, E).astype(np.int32)
    osc = rng.random(20).astype(np.float32)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, omono)
    osamp.set_test_statistic(capi.kBarlowBeeston)

    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    info = fresh_engine.info()
    if grouping == "on":
        assert info.n_grouped_chunks == 1 and info.n_tiles_uniform >= 14 and info.n_tiles_brick <= 3
    else:
        assert info.n_grouped_chunks == 0 and info.n_tiles_uniform == 0 and info.n_tiles_brick == 5
    sh = SampleHandlerFD(fresh_engine, 2, np.array([0, 40, 80], np.int32), par, capi.kBarlowBeeston)
    sh.SetupEvents(0, synth.find_global_bin_uniform_1d(edges, kin, sample_id, 40), osc_idx, norm_off, norm_idx)
    sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in range(3):
        par[:P] = rng.uniform(-3.3, 3.3, P)
        par[P:] = 1 + 0.05 * rng.standard_normal(n_norm)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetSplineWeights(0, E).view(np.uint32), omono.total_weights.view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    mono.Evaluate()
    mono.SynchroniseMemTransfer()
    omono.evaluate()
    assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    osamp.close()
    omono.close()


def test_brick_of_ninety_stages(fresh_engine):
    """~390 000 responses in a sparse 1024-event group: one brick of ~90 stages (round 2's header block refused it at
    62); every weight bit-exact.  The refusal beyond M3_BRICK_MAX_STAGES = 250 (1.3 M responses in one brick) is
    exercised on the CPU by tests/cpp/layout_sanitize.cpp."""
    rng = np.random.default_rng(9)
    P = 640
    X3 = np.array([-1.0, 0, 1])
    xs = {p: X3 for p in range(P)}
    events = _ragged_events(rng, 1024, P, 0.6, xs)
    assert sum(len(e) for e in events) > 62 * 5120
    cx, npts, ch = make_monolith(P, 3, xs, events)
    info = _eval_matches(fresh_engine, P, 3, cx, npts, ch, rng, laps=2)
    assert info.n_tiles_brick == 1 and info.n_tiles_uniform == 0


def test_uniform_and_ragged_groups_in_one_sample(fresh_engine):
    """1024 dense events followed by ragged ones, plus events without any response."""
    P = 12
    X7 = np.arange(-3.0, 4.0)
    rng = np.random.default_rng(7)
    quad = lambda: natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))
    resp = lambda p: (p, "spl", quad()) if p % 3 else (p, "tf1", (0.1 * rng.standard_normal(), 1.0))
    events = [[resp(p) for p in range(P)] for _ in range(1024)]
    events += [[resp(p) for p in range(P) if rng.random() < 0.6] for _ in range(700)]
    events[1100] = []
    events[1500] = []
    E = len(events)
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P) if p % 3}, events)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    par = np.zeros(P)
    opar = np.zeros(P)
    omono.set_spline_pointers(opar)
    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    info = fresh_engine.info()
    assert info.n_tiles_uniform == 4 and info.n_tiles_brick == 1
    mono.setSplinePointers(par)
    for k in range(4):
        par[:] = rng.uniform(-3.2, 3.2, P)
        opar[:] = par
        mono.Evaluate()
        mono.SynchroniseMemTransfer()
        omono.evaluate()
        assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    assert mono.cpu_total_weights[1100] == 1.0 and mono.cpu_total_weights[1500] == 1.0
    omono.close()


def test_in_library_rccl_allreduce_single_rank(fresh_engine):
    """m3hip_comm_init + the per-step ncclAllReduce on a one-rank communicator: the RCCL code path
    (dlopen, ncclCommInitRank, in-place f64 sum on the engine's stream) runs and changes nothing."""
    spec = synth.CONFIGS["c2"].scaled(4000, n_params=16)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    sh.Reweight()
    sh.AddData(sh.GetMCArray())
    par[:] = synth.params(spec, 2)
    ref = sh.Step()
    mc_ref = sh.GetMCArray()
    uid = eng.comm_unique_id()
    assert len(uid) == capi.UNIQUE_ID_BYTES
    eng.comm_init(uid, 0, 1)
    got = sh.Step()
    assert got == pytest.approx(ref, rel=1e-12)
    assert np.allclose(sh.GetMCArray(), mc_ref, rtol=1e-12)
    with pytest.raises(capi.M3HipError):
        eng.comm_init(uid, 3, 2)   # rank out of range


def _dense_monolith(E, P, rng):
    X7 = np.arange(-3.0, 4.0)
    events = [[(p, "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))) for p in range(P)] for _ in range(E)]
    return make_monolith(P, 7, {p: X7 for p in range(P)}, events)


def test_device_side_binning_bit_exact(fresh_engine):
    """BinningHandler::FindGlobalBin / FindNominalBin on the device, integer-exact against the oracle:
    1-D and 2-D samples, values on edges, below the first and on/above the last edge."""
    rng = np.random.default_rng(21)
    E, P = 3000, 2
    cx, npts, ch = _dense_monolith(E, P, rng)
    e1 = np.array([0.0, 0.5, 1.0, 1.25, 2.0, 3.5, 5.0])
    ex, ey = np.linspace(0.0, 4.0, 11), np.array([-1.0, -0.2, 0.0, 0.3, 1.0, 2.5])
    sample_id = rng.integers(0, 2, E).astype(np.int32)
    kin = np.stack([rng.uniform(-0.3, 5.3, E), rng.uniform(-1.2, 2.7, E)], axis=1)
    kin[:40, 0] = rng.choice(np.concatenate([e1, ex]), 40)        # exactly on edges
    kin[40:60, 1] = rng.choice(ey, 20)
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    b.add_uniform([e1])
    b.add_uniform([ex, ey])
    par = np.zeros(P + 2)
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(par)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, par, None, None, omono)
    exp = osamp.event_bins()
    assert (exp == -1).any() and (exp >= 6).any()

    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 2, None, par, bin_edges=[[e1], [ex, ey]])
    assert sh.nBins == 6 + 50 and b.n_bins() == 56
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    assert np.array_equal(sh.GetEventBins(0, E), exp)
    # the fused step on device-computed bins
    sh.Reweight()
    osamp.reweight()
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    osamp.close()
    omono.close()


def _random_boxes(rng, nx, ny, x0, x1, y0, y1, drop=0):
    """irregular rectilinear tiling of [x0,x1] x [y0,y1] with some cells merged away (gaps)"""
    xs = np.sort(np.concatenate([[x0, x1], rng.uniform(x0, x1, nx - 1)]))
    boxes = []
    for i in range(nx):
        ys = np.sort(np.concatenate([[y0, y1], rng.uniform(y0, y1, ny - 1)]))   # different y cuts per column
        for j in range(ny):
            boxes.append([[xs[i], xs[i + 1]], [ys[j], ys[j + 1]]])
    boxes = np.array(boxes, np.float64)
    keep = np.ones(len(boxes), bool)
    keep[rng.choice(len(boxes), drop, replace=False)] = False
    return boxes[keep], xs


def test_device_side_nonuniform_binning_bit_exact(fresh_engine):
    """Non-uniform (box) samples on the device: mega-bin grid -> candidate scan with the (lo, hi] test
    (BinningHandler.cpp:277-290), mixed with a uniform sample, integer-exact against the oracle; with
    a shift registered every step re-bins through the cached nominal mega-bins."""
    rng = np.random.default_rng(31)
    E, P = 6000, 2
    cx, npts, ch = _dense_monolith(E, P, rng)
    boxes, xs = _random_boxes(rng, 9, 7, 0.0, 5.0, -1.0, 1.0, drop=5)
    boxes3 = np.array([[[0, 1], [0, 1], [0, 2]], [[1, 3], [0, 1], [0, 2]], [[0, 3], [1, 2], [0, 1]], [[0, 3], [1, 2], [1, 2]]], np.float64)
    e1 = np.linspace(0.0, 5.0, 26)
    sample_id = rng.integers(0, 3, E).astype(np.int32)
    kin = np.stack([rng.uniform(-0.2, 5.2, E), rng.uniform(-1.1, 1.1, E), rng.uniform(-0.1, 2.1, E)], axis=1)
    s2 = np.flatnonzero(sample_id == 2)
    kin[s2, 0] = rng.uniform(-0.1, 3.1, len(s2))
    kin[s2, 1] = rng.uniform(-0.1, 2.1, len(s2))
    kin[:200, 0] = rng.choice(np.concatenate([xs, np.linspace(0, 5, 11)]), 200)      # box edges and mega-bin edges
    kin[200:400, 1] = rng.choice(np.concatenate([boxes[:, 1, :].ravel(), np.linspace(-1, 1, 11)]), 200)
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    assert b.add_uniform([e1]) == 0
    assert b.add_nonuniform(boxes) == 1
    assert b.add_nonuniform(boxes3) == 2
    par = np.zeros(P + 1)
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, opar, None, None, omono)
    exp = osamp.event_bins()
    nb = 25 + len(boxes) + 4
    assert b.n_bins() == nb
    for s, lo, hi in ((0, 0, 25), (1, 25, 25 + len(boxes)), (2, 25 + len(boxes), nb)):
        got_s = exp[sample_id == s]
        assert (got_s == -1).any() and ((got_s >= lo) & (got_s < hi)).sum() > 0.5 * len(got_s)
        assert (((got_s >= lo) & (got_s < hi)) | (got_s == -1)).all()

    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 3, None, par, bin_edges=[[e1], boxes, boxes3])
    assert sh.nBins == nb
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    assert np.array_equal(sh.GetEventBins(0, E), exp)
    sh.Reweight()
    osamp.reweight()
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)

    # x *= 1 + theta on every sample: events migrate between boxes, mega-bins and out of range
    osamp.add_shift(-1, 0, 1, opar, P)
    sh.AddKinShift(-1, 0, 1, P)
    moved = 0
    for theta in (0.0, 0.013, -0.04, 0.11, -0.3, 0.0):
        par[:P] = opar[:P] = rng.normal(0, 0.4, P)
        par[P] = opar[P] = theta
        sh.Reweight()
        osamp.reweight()
        got = sh.GetEventBins(0, E)
        want = osamp.event_bins()
        assert np.array_equal(got, want), theta
        moved += int((want != exp).sum())
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        assert np.isclose(sh.GetLikelihood(), osamp.get_likelihood(), rtol=RTOL_LLH, atol=1e-9)
    assert moved > 1000
    osamp.close()
    omono.close()


def test_functional_shifts_rebin_every_step(fresh_engine):
    """Kinematic shifts driven by parameters (the reference's ApplyShifts + FindBin with the cached
    nominal bin: nominal hit, +-1 neighbour, binary search, migration out of range)."""
    rng = np.random.default_rng(22)
    E, P = 4000, 3
    cx, npts, ch = _dense_monolith(E, P, rng)
    ex, ey = np.linspace(0.0, 5.0, 41), np.linspace(-1.0, 1.0, 9)
    sample_id = rng.integers(0, 2, E).astype(np.int32)
    kin = np.stack([rng.gamma(3.0, 0.45, E), rng.uniform(-1.05, 1.05, E)], axis=1)
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    b.add_uniform([ex, ey])
    b.add_uniform([ex, ey])
    par = np.zeros(P + 2)     # [splines..., energy scale, y offset]
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, opar, None, None, omono)
    osamp.add_shift(-1, 0, 1, opar, P)        # x *= 1 + theta_P, all samples
    osamp.add_shift(1, 1, 2, opar, P + 1)     # y += theta_{P+1}, sample 1 only

    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 2, None, par, bin_edges=[[ex, ey], [ex, ey]])
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    nominal_bins = sh.GetEventBins(0, E)
    sh.AddKinShift(-1, 0, 1, P)
    sh.AddKinShift(1, 1, 2, P + 1)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    moved = 0
    for scale, off in [(0.0, 0.0), (0.01, 0.0), (-0.02, 0.05), (0.3, -0.4), (-0.5, 0.9), (1.5, 3.0), (0.0, 0.0)]:
        par[:P] = rng.uniform(-1, 1, P)
        par[P], par[P + 1] = scale, off
        opar[:] = par
        sh.Reweight()
        llh = sh.GetLikelihood()
        osamp.reweight()
        got, exp = sh.GetEventBins(0, E), osamp.event_bins()
        assert np.array_equal(got, exp), (scale, off)
        moved += int((got != nominal_bins).sum())
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    assert moved > E  # the shifts really migrated events
    assert np.array_equal(sh.GetEventBins(0, E), nominal_bins)
    sh.ClearKinShifts()
    assert np.array_equal(sh.GetEventBins(0, E), nominal_bins)
    osamp.close()
    omono.close()


@pytest.mark.parametrize("lists", ["ell", "shared_sorted", "shared_unsorted", "shared_forced"])
def test_binned_splines_match_oracle(fresh_engine, lists, monkeypatch):
    """BinnedSplineHandler::Evaluate (segment search + clamped cubic per bin-spline)
    and the per-event gather of those weights into CalcWeightTotal, against the oracle: bin-spline
    weights and event weights bit-exact, bins 1e-12, LLH 1e-9.  The events' pointer lists reach the device
    either as ELL planes (about one distinct list per event) or as list ids into the distinct lists (few distinct
    lists, MC in spline-bin order or in any order -- the engine then processes the events grouped by list; also
    forced onto events that each have a list of their own)."""
    from helpers import make_binned, make_binned_events
    from mach3_amd.engine import BinnedSplineHandler

    rng = np.random.default_rng(31)
    P, cells, E, n_norm = 9, 400, 5000, 4
    bs = make_binned(P, cells, rng)
    if lists == "shared_sorted":
        w_off, w_idx = make_binned_events(E, P, cells, rng, per_cell_lists=True, sort_by_cell=True)
    elif lists == "shared_unsorted":   # few distinct lists, events in any order: the engine processes them grouped by list
        w_off, w_idx = make_binned_events(E, P, cells, rng, per_cell_lists=True, sort_by_cell=False)
    else:
        w_off, w_idx = make_binned_events(E, P, cells, rng)
    if lists == "shared_forced":
        monkeypatch.setenv("M3HIP_BINNED_LISTS", "shared")
    par = np.zeros(P + n_norm)
    par[P:] = 1.0
    opar = par.copy()
    edges = np.linspace(0, 5, 41)
    kin = rng.gamma(3.0, 0.4, E)
    sample_id = np.zeros(E, np.int32)
    nn = rng.integers(0, 3, E)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, n_norm, int(norm_off[-1]))).astype(np.int16)
    osc_idx = rng.integers(-1, 50, E).astype(np.int32)
    osc = rng.random(50).astype(np.float32)

    lib = ob.load("serial")
    ob_binned = ob.OracleBinned(lib, bs)
    ob_binned.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, None)
    osamp.set_binned(ob_binned, w_off, w_idx)
    osamp.set_test_statistic(capi.kBarlowBeeston)

    handler = BinnedSplineHandler(fresh_engine, E, bs)
    handler.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 1, np.array([0, 40], np.int32), par, capi.kBarlowBeeston)
    sh.SetupEvents(0, synth.find_global_bin_uniform_1d(edges, kin, sample_id, 40), osc_idx, norm_off, norm_idx)
    handler.SetEventSplines(0, w_off, w_idx)
    sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    assert (ob_binned.weights == 1.0).sum() >= int(0.15 * P * cells)   # the flat ones
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    clamped = 0
    for lap in range(6):
        par[:P] = rng.uniform(-3.5, 3.5, P)
        par[P:] = 1 + 0.05 * rng.standard_normal(n_norm)
        opar[:] = par
        sh.Reweight()
        llh = sh.GetLikelihood()
        osamp.reweight()
        w_gpu, w_cpu = handler.GetWeights(), ob_binned.weights
        assert np.array_equal(w_gpu.view(np.uint32), w_cpu.view(np.uint32))
        clamped += int((w_cpu == 0).sum())
        seg, _ = sh.GetSegments(P)
        assert np.array_equal(seg, ob_binned.segments)
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    assert clamped > 0   # the weight < 0 -> 0 clamp was exercised
    osamp.close()
    ob_binned.close()


@pytest.mark.parametrize("float_t,P,cells,sort,kernel", [(np.float32, 130, 40, True, "lists"), (np.float32, 130, 40, False, "lists"), (np.float64, 130, 40, True, "lists"),
                                                        (np.float32, 12, 1500, False, "lists"), (np.float64, 12, 1500, True, "lists"), (np.float32, 130, 40, False, "tiles")])
def test_lists_kernel_long_lists_many_lists_and_reuploads(monkeypatch, float_t, P, cells, sort, kernel):
    """k_reweight_lists, the event kernel of a binned-only sample with shared pointer lists (what MaCh3Tutorial's far-detector sample
    runs, SampleHandlerFD.cpp:1196-1243): lists of more than 64 entries (78 on average of 130 parameters: the wave fetches
    them 64 at a time), waves with more distinct lists than the kernel fetches ahead (1 500 lists for 9 000 events: six events per
    list -- most lanes walk alone), both builds of M3::float_t, events in list order and in any order, a slice of bins and of
    oscillation indices re-uploaded after the first reweight (the processing-order copy is packed again), a slice of pointer lists
    replaced; `tiles`: the same sample kept on the uniform-tile kernel (M3HIP_LISTS_KERNEL=0).  Event weights bit for bit."""
    from helpers import make_binned, make_binned_events
    from mach3_amd.engine import BinnedSplineHandler, Engine
    if kernel == "tiles":
        monkeypatch.setenv("M3HIP_LISTS_KERNEL", "0")
    monkeypatch.setenv("M3HIP_BINNED_LISTS", "shared")
    rng = np.random.default_rng(57)
    E, n_norm = 9000, 4
    bs = make_binned(P, cells, rng, float_t=float_t)
    w_off, w_idx = make_binned_events(E, P, cells, rng, per_cell_lists=True, sort_by_cell=sort)
    assert P < 64 or int(np.diff(w_off).max()) > 64
    par = np.zeros(P + n_norm)
    par[P:] = 1.0
    opar = par.copy()
    edges = np.linspace(0, 5, 41)
    kin = rng.gamma(3.0, 0.4, E)
    sample_id = np.zeros(E, np.int32)
    nn = rng.integers(0, 3, E)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, n_norm, int(norm_off[-1]))).astype(np.int16)
    osc_idx = rng.integers(-1, 50, E).astype(np.int32)
    osc = rng.random(50).astype(float_t)
    bins = synth.find_global_bin_uniform_1d(edges, kin, sample_id, 40)

    lib = ob.load("serial_f64" if float_t == np.float64 else "serial")
    ob_binned = ob.OracleBinned(lib, bs)
    ob_binned.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    def oracle_sample():
        o = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx.copy(), osc, None)
        o.set_binned(ob_binned, w_off, w_idx)
        o.set_test_statistic(capi.kPoisson)
        return o
    osamp = oracle_sample()

    eng = Engine(0)
    handler = BinnedSplineHandler(eng, E, bs)
    handler.setSplinePointers(par)
    sh = SampleHandlerFD(eng, 1, np.array([0, 40], np.int32), par, capi.kPoisson)
    sh.SetupEvents(0, bins, osc_idx, norm_off, norm_idx)
    handler.SetEventSplines(0, w_off, w_idx)
    sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    data = osamp.mc
    sh.AddData(data)
    osamp.add_data(data)
    view = np.uint64 if float_t == np.float64 else np.uint32

    def laps(n, osamp):
        for _ in range(n):
            par[:P] = rng.uniform(-3.5, 3.5, P)
            par[P:] = 1 + 0.05 * rng.standard_normal(n_norm)
            opar[:] = par
            llh = sh.Step()
            osamp.reweight()
            assert np.array_equal(sh.GetEventWeights(0, E, float_t).view(view), osamp.event_weights().view(view))
            assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
            _llh_close(llh, osamp.get_likelihood())
    laps(3, osamp)
    # a slice of the events gets other oscillation indices (the bins stay what the kinematics say): the packed copy follows
    a0, a1 = 2048, 5000
    osc_idx[a0:a1] = rng.integers(-1, 50, a1 - a0)
    sh.SetupEvents(a0, bins[a0:a1], osc_idx[a0:a1], None, None)
    osamp.close()
    osamp = oracle_sample()                      # (the oracle takes its pointers at construction)
    osamp.reweight()
    osamp.add_data(data)
    laps(2, osamp)
    osamp.close()
    eng.close()


def test_binned_splines_double_build_match_oracle(fresh_engine):
    """The reference's default build, M3::float_t = double (Manager/Core.h:27-51): bin-spline tables,
    std::fma cubic on the parameter's own double value, float-cast value against double knots in the
    segment search, and the sample's weight product / w*w in double.  Bin-spline weights and event
    weights bit-exact in fp64, bins 1e-12, LLH 1e-9."""
    from helpers import make_binned, make_binned_events
    from mach3_amd.engine import BinnedSplineHandler

    rng = np.random.default_rng(41)
    P, cells, E, n_norm = 8, 300, 6000, 5
    bs = make_binned(P, cells, rng, float_t=np.float64)
    w_off, w_idx = make_binned_events(E, P, cells, rng)
    par = np.zeros(P + n_norm)
    par[P:] = 1.0
    opar = par.copy()
    edges = np.linspace(0, 5, 51)
    kin = rng.gamma(3.0, 0.4, E)
    sample_id = np.zeros(E, np.int32)
    nn = rng.integers(0, 4, E)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, n_norm, int(norm_off[-1]))).astype(np.int16)
    osc_idx = rng.integers(-1, 64, E).astype(np.int32)
    osc = rng.random(64)                       # double oscillation weights

    lib = ob.load("serial_f64")
    assert lib.float_t == np.float64
    ob_binned = ob.OracleBinned(lib, bs)
    ob_binned.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, None)
    osamp.set_binned(ob_binned, w_off, w_idx)
    osamp.set_test_statistic(capi.kBarlowBeeston)

    handler = BinnedSplineHandler(fresh_engine, E, bs)
    assert handler.float_t == np.float64
    handler.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 1, np.array([0, 50], np.int32), par, capi.kBarlowBeeston)
    sh.SetupEvents(0, synth.find_global_bin_uniform_1d(edges, kin, sample_id, 50), osc_idx, norm_off, norm_idx)
    handler.SetEventSplines(0, w_off, w_idx)
    sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    clamped = differs_from_float = 0
    knots = bs["xpts"].reshape(P, -1)
    for lap in range(8):
        par[:P] = rng.uniform(-3.5, 3.5, P)
        if lap == 3:
            par[0] = knots[0, 2]                     # a double knot: the float-cast value decides the segment
            par[1] = float(np.float32(knots[1, 1]))  # the float image of a knot
        par[P:] = 1 + 0.05 * rng.standard_normal(n_norm)
        opar[:] = par
        sh.Reweight()
        llh = sh.GetLikelihood()
        osamp.reweight()
        w_gpu, w_cpu = handler.GetWeights(), ob_binned.weights
        assert w_gpu.dtype == np.float64 and np.array_equal(w_gpu.view(np.uint64), w_cpu.view(np.uint64))
        clamped += int((w_cpu == 0).sum())
        differs_from_float += int((w_cpu != w_cpu.astype(np.float32)).sum())
        seg, _ = sh.GetSegments(P)
        assert np.array_equal(seg, ob_binned.segments)
        ew = sh.GetEventWeights(0, E, np.float64)
        assert np.array_equal(ew.view(np.uint64), osamp.event_weights().view(np.uint64))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    assert clamped > 0 and differs_from_float > 1000
    osamp.close()
    ob_binned.close()


def test_kinematic_cuts_on_shifted_variables(fresh_engine):
    """SampleHandlerFD::IsEventSelected (ref: Samples/SampleHandlerFD.cpp:281-294): cuts lo <= x < hi on
    a binning axis and on a cut-only column, evaluated after the functional shifts."""
    rng = np.random.default_rng(23)
    E, P = 3000, 2
    cx, npts, ch = _dense_monolith(E, P, rng)
    ex = np.linspace(0.0, 5.0, 26)
    sample_id = rng.integers(0, 2, E).astype(np.int32)
    kin = np.stack([rng.gamma(3.0, 0.45, E), rng.uniform(0.0, 10.0, E)], axis=1)   # column 1: cut-only variable
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    b.add_uniform([ex])
    b.add_uniform([ex])
    par = np.zeros(P + 1)
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, opar, None, None, omono)
    osamp.add_shift(-1, 1, 2, opar, P)      # the cut variable moves with a parameter
    osamp.add_cut(-1, 0, 0.5, 3.0)          # axis 0: [0.5, 3.0)
    osamp.add_cut(1, 1, 2.0, 7.5)           # sample 1 only, on the cut-only column

    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 2, None, par, bin_edges=[[ex], [ex]])
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    sh.AddKinShift(-1, 1, 2, P)
    sh.AddKinCut(-1, 0, 0.5, 3.0)
    sh.AddKinCut(1, 1, 2.0, 7.5)
    n_sel = []
    for off in (0.0, 1.5, -2.5, 0.0):
        par[:P] = rng.uniform(-1, 1, P)
        par[P] = off
        opar[:] = par
        sh.Reweight()
        osamp.reweight()
        got, exp = sh.GetEventBins(0, E), osamp.event_bins()
        assert np.array_equal(got, exp), off
        n_sel.append(int((got >= 0).sum()))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    assert 0 < min(n_sel) < max(n_sel) < E and n_sel[0] == n_sel[-1]
    osamp.close()
    omono.close()


def test_pymach3_style_names(fresh_engine):
    """the reference's python binding drives the monolith as EventSplineMonolith.evaluate / sync_mem_transfer /
    get_event_weight / set_param_value_array (python/splines.cpp:142-236); the harness answers to the same names"""
    rng = np.random.default_rng(77)
    E, P = 500, 5
    cx, npts, ch = _dense_monolith(E, P, rng)
    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    opar = np.zeros(P)
    omono.set_spline_pointers(opar)
    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    arr = np.zeros(P)
    mono.set_param_value_array(arr)
    for _ in range(3):
        arr[:] = rng.uniform(-3, 3, P)
        opar[:] = arr
        mono.evaluate()
        mono.sync_mem_transfer()
        omono.evaluate()
        assert all(np.float32(mono.get_event_weight(e)) == omono.total_weights[e] for e in (0, 1, E // 2, E - 1))
    omono.close()


def test_two_sample_handlers_interleaved():
    """A fitter with two sample handlers calls Reweight() on both and then GetLikelihood() on both
    (Fitters/MR2T2.cpp:66-76).  Each handler owns an engine handle (its own stream), so the two reweights overlap on
    the device; results must equal what each handler gives on its own."""
    spec_a = synth.CONFIGS["c2"].scaled(6000, n_params=20)
    spec_b = synth.CONFIGS["c3"].scaled(5000, n_params=64)
    eng_a, mono_a, sh_a, par_a = build_from_synth(spec_a)
    eng_b, mono_b, sh_b, par_b = build_from_synth(spec_b)
    try:
        for sh in (sh_a, sh_b):
            sh.Reweight()
            sh.AddData(sh.GetMCArray())
        alone = []
        for lap in (1, 2, 3):
            par_a[:] = synth.params(spec_a, lap)
            par_b[:] = synth.params(spec_b, lap)
            alone.append((sh_a.Step(), sh_b.Step(), sh_a.GetMCArray(), sh_b.GetMCArray()))
        for rep in range(3):
            for lap in (1, 2, 3):
                par_a[:] = synth.params(spec_a, lap)
                par_b[:] = synth.params(spec_b, lap)
                sh_a.Reweight()
                sh_b.Reweight()
                la, lb = sh_a.GetLikelihood(), sh_b.GetLikelihood()
                ea, eb, mca, mcb = alone[lap - 1]
                assert la == pytest.approx(ea, rel=1e-12) and lb == pytest.approx(eb, rel=1e-12)
                assert np.allclose(sh_a.GetMCArray(), mca, rtol=1e-12) and np.allclose(sh_b.GetMCArray(), mcb, rtol=1e-12)
    finally:
        eng_a.close()
        eng_b.close()


def test_nan_kinematics_are_out_of_range(fresh_engine):
    """A NaN kinematic value (bad input, or a shift by a non-finite parameter) is UNDER/OVERFLOW on the device, not the
    bin one past the last edge: in the reference's form of the range test a NaN passes, upper_bound returns the last
    edge, and the event is filled into the NEXT sample's first bin (or past the bin vector).  Its neighbours are untouched."""
    rng = np.random.default_rng(77)
    E, P = 2000, 2
    cx, npts, ch = _dense_monolith(E, P, rng)
    e1 = np.linspace(0.0, 5.0, 11)
    sample_id = (np.arange(E) % 2).astype(np.int32)
    kin = rng.uniform(0.1, 4.9, (E, 1))
    par = np.zeros(P + 1)
    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 2, None, par, bin_edges=[[e1], [e1]])
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    ref_bins = sh.GetEventBins(0, E).copy()
    sh.Reweight()
    mc_ref = sh.GetMCArray().copy()
    bad = np.array([10, 11, 500, 1999])
    kin2 = kin.copy()
    kin2[bad, 0] = np.nan
    sh.SetupKinematics(0, sample_id, kin2)
    sh.ComputeBins()
    got = sh.GetEventBins(0, E)
    assert (got[bad] == -1).all()
    keep = np.ones(E, bool)
    keep[bad] = False
    assert np.array_equal(got[keep], ref_bins[keep])
    sh.Reweight()
    mc = sh.GetMCArray()
    assert np.isfinite(mc).all() and mc.sum() < mc_ref.sum() and (mc <= mc_ref + 1e-9).all()


def test_failed_reweight_leaves_the_last_result_in_place(fresh_engine):
    """Preconditions are checked before the step takes a ring slot and k_prep empties the bin vectors: after an error
    return the bins and the likelihood are still those of the last good step, not those of an empty MC."""
    spec = synth.CONFIGS["c2"].scaled(3000, n_params=8)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    sh.Reweight()
    sh.AddData(sh.GetMCArray())
    par[:] = synth.params(spec, 1)
    llh = sh.Step()
    mc = sh.GetMCArray().copy()
    assert llh > 0 and mc.sum() > 0
    short = np.ascontiguousarray(par[: spec.n_params - 1])           # parameter vector shorter than the spline parameters
    rc = eng.lib.m3hip_reweight(eng.h, short.ctypes.data, len(short))
    assert rc != 0
    assert np.array_equal(sh.GetMCArray(), mc) and sh.GetLikelihood() == llh
    # the bin space redefined with fewer bins than the uploaded indices reach: refused by the next reweight, not read out of bounds
    off = np.array([0, 50], np.int32)
    assert eng.lib.m3hip_set_binning(eng.h, 1, off.ctypes.data) == 0
    rc = eng.lib.m3hip_reweight(eng.h, par.ctypes.data, len(par))
    assert rc != 0 and "bin index" in eng.lib.m3hip_last_error(eng.h).decode()


@pytest.mark.parametrize("cfg,n_params", [("c2", 12), ("c3", 64)])
def test_experiment_weight_slots_match_oracle(fresh_engine, cfg, n_params):
    """AddAdditionalWeightPointers / CalcWeightFunc (SampleHandlerFD.cpp:196, :428; SampleHandlerFD.h:289) on the fused path:
    a per-event weight (slot 0; some events at 0 and below: dropped by `w <= 0`), a table of 37 values that the "experiment"
    recomputes every step with an index per event, -1 for events without the pointer (slot 2; slot 1 stays unused), and a
    second per-event weight (slot 3).  Multiplied behind the spline weight in slot order, like the reference's pointer
    vector: event weights bit-exact against the serial oracle, on uniform tiles (c2) and on bricks (c3)."""
    rng = np.random.default_rng(99)
    spec = synth.CONFIGS[cfg].scaled(5000, n_params=n_params)
    E = spec.n_events
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine, chunk_events=2048)
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, E)
    col0 = rng.uniform(0.5, 1.5, E).astype(np.float32)
    col0[rng.choice(E, 40, replace=False)] = 0.0
    col0[rng.choice(E, 10, replace=False)] = -0.3
    col3 = rng.uniform(0.9, 1.1, E).astype(np.float32)
    idx = rng.integers(-1, 37, E).astype(np.int32)
    table = np.ones(37, np.float32)                      # the oracle's pointers alias THIS array: rewritten in place per step
    osamp.add_weight_pointers(col0)
    osamp.add_weight_pointers(table, idx)
    osamp.add_weight_pointers(col3)
    sh.SetEventWeights(0, 0, col0[:3000])
    sh.SetEventWeights(0, 3000, col0[3000:])             # in two pieces
    sh.SetWeightIndex(2, 0, idx)
    sh.SetEventWeights(3, 0, col3)
    with pytest.raises(capi.M3HipError, match="past its table"):
        sh.Reweight()                                    # indices uploaded, table not yet
    sh.SetWeightTable(2, table)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in (1, 2, 3):
        par[:] = synth.params(spec, lap)
        opar[:] = par
        table[:] = (1.0 + 0.2 * np.sin(np.arange(37) + lap)).astype(np.float32)   # CalcWeightFunc
        sh.SetWeightTable(2, table)
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        assert llh == pytest.approx(osamp.get_likelihood(), rel=1e-9)
    assert (osamp.event_weights() <= 0).sum() >= 50
    with pytest.raises(capi.M3HipError):
        sh.SetEventWeights(2, 0, col3)                   # slot 2 holds indices
    with pytest.raises(capi.M3HipError):
        sh.SetEventWeights(8, 0, col3)                   # no such slot
    osamp.close()
    omono.close()


@pytest.mark.parametrize("cfg,n_params,n_events", [("c2", 16, 40_000), ("c3", 64, 60_000)])
def test_deterministic_fill_is_bit_reproducible(cfg, n_params, n_events):
    """m3hip_set_deterministic: no floating-point atomics between workgroups, waves add in turn inside one.  Event weights are
    spread over twelve decades (an experiment weight slot), so that the fp64 bin sums really do round and the order of the
    adds shows: ten evaluations of the same parameter vector, interleaved with others, give bit-identical bins and
    likelihood; the values agree with the default (atomic) fill to rounding; and on the plain configuration (adds exact at
    these magnitudes) the bins of two half-sample engines add up to the whole sample's bit for bit."""
    rng = np.random.default_rng(5)
    spec = synth.CONFIGS[cfg].scaled(n_events, n_params=n_params)
    wide = (10.0 ** rng.uniform(-6, 6, n_events)).astype(np.float32)
    eng, mono, sh, par = build_from_synth(spec, chunk_events=16384)
    sh.SetEventWeights(0, 0, wide)
    sh.Reweight()
    sh.AddData(sh.GetMCArray())
    par[:] = synth.params(spec, 4)
    llh_atomic = sh.Step()
    mc_atomic = sh.GetMCArray().copy()
    eng.set_deterministic(True)
    seen = set()
    for rep in range(10):
        par[:] = synth.params(spec, 4)
        llh = sh.Step()
        seen.add((llh, sh.GetMCArray().tobytes(), sh.GetW2Array().tobytes()))
        par[:] = synth.params(spec, 5 + rep)   # something else in between
        sh.Step()
    assert len(seen) == 1
    llh_det, mc_det = next(iter(seen))[0], np.frombuffer(next(iter(seen))[1])
    # (the likelihood of such a sample cancels eight digits between its terms: 1e-6, the north-star bar, not 1e-9)
    assert np.allclose(mc_det, mc_atomic, rtol=1e-12, atol=0) and llh_det == pytest.approx(llh_atomic, rel=1e-6)
    eng.close()

    # shard additivity, bit for bit, on the configuration as it is (no twelve-decade weights)
    def bins(e0, e1):
        e, m, s, p = build_from_synth(spec, chunk_events=16384, event_range=(e0, e1))
        e.set_deterministic(True)
        p[:] = synth.params(spec, 4)
        s.Reweight()
        out = s.GetMCArray().copy(), s.GetW2Array().copy()
        e.close()
        return out
    cut = (n_events // 2 // 1024) * 1024
    (mc_a, w2_a), (mc_b, w2_b), (mc_w, w2_w) = bins(0, cut), bins(cut, n_events), bins(0, n_events)
    assert np.array_equal(mc_a + mc_b, mc_w) and np.array_equal(w2_a + w2_b, w2_w)


def test_deterministic_fill_needs_the_private_bin_vector(fresh_engine):
    spec = synth.CONFIGS["c2"].scaled(3000, n_params=4, n_samples=1, n_bins=3000)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    eng.set_deterministic(True)
    with pytest.raises(capi.M3HipError, match="deterministic"):
        sh.Reweight()
    eng.set_deterministic(False)
    sh.Reweight()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["lds_pass_into_word", "lds_pass_into_word_g2", "lds_pass_into_word_g4", "global_pass_into_word", "global_pass_into_word_g4",
                                  "lds_pass_no_word", "global_pass_no_word", "lds_pass_weight_words_rule", "lds_pass_weight_words_rule_g4",
                                  "global_pass_weight_words_forced_g4", "lds_pass_weight_words_forced", "general_lds_pass_into_word",
                                  "general_lds_pass_weight_words_forced"])
def test_functional_shift_on_a_ragged_sample_of_1e5_events(path, monkeypatch):
    """SURVEY 8f row 4 on a brick sample of a size where the launch has rounds of workgroups (120 000 events of configs[2]'s
    shape): an energy-scale functional parameter re-bins every event every step (ApplyShifts, Samples/SampleHandlerFD.cpp:545-564;
    FindBin with the cached nominal bin, Samples/SampleStructs.h:577-613).  Every device form against the oracle: the pass with
    the binning tables in LDS that rewrites the packed words' bin field (the default; bricks of 1024, 2048 and 4096 events), the
    passes over global tables (k_rebin_pack, k_find_bins), samples without the packed word.  Bins integer-exact, event weights bit-exact, bin contents 1e-12, LLH 1e-9."""
    from mach3_amd.engine import Engine
    if path.endswith("_g2"):
        monkeypatch.setenv("M3HIP_BRICK_GROUPS", "2")
    if path.endswith("_g4"):
        monkeypatch.setenv("M3HIP_BRICK_GROUPS", "4")
    if "no_word" in path:
        monkeypatch.setenv("M3HIP_PACKED_EVENTS", "0")
    if path.startswith("global"):
        monkeypatch.setenv("M3HIP_REBIN", "global")
    if path.startswith("general"):  # round 6: one sample in one variable runs k_rebin_1d; this keeps the pass for any shape on the same sample
        monkeypatch.setenv("M3HIP_REBIN", "lds")
    # round 5: a pre-staged oscillation table that rests is folded into the packed words (the weight's float bits instead of the
    # index, a 14-bit bin field for the passes to write into): by the engine's rule for a LARGE table (200 KB here: index words
    # for the first laps, weight words from the third reweight on, index words again when the table moves, ...), or forced
    if "forced" in path:
        monkeypatch.setenv("M3HIP_OSC_COLUMN", "1")
    spec = synth.CONFIGS["c3"].scaled(120_000, **({"osc_n": 50_000} if "weight_words_rule" in path else {}))
    E, NP = spec.n_events, spec.n_par
    lib = ob.load("serial")
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, E)
    ev = synth.events_chunk(spec, 0, E)
    opar = np.concatenate([synth.params(spec, 0), [0.0]])
    omono = ob.OracleMonolith(lib, spec.n_params, spec.n_knots, cx, npts, ch)
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    edges = synth.bin_edges(spec)
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, ev["sample_id"], ev["erec"], ev["norm_off"], ev["norm_idx"], opar, ev["osc_idx"], synth.osc_table(spec), omono)
    osamp.set_test_statistic(spec.test_statistic, False)
    osamp.add_shift(-1, 0, 1, opar, NP)

    eng = Engine(0)
    try:
        par = opar.copy()
        mono = SMonolith(eng, spec.n_params, spec.n_knots, cx, npts, [ch], expect_events=E)
        mono.setSplinePointers(par)
        sh = SampleHandlerFD(eng, 1, None, par, spec.test_statistic, bin_edges=[[edges]])
        sh.SetupKinematics(0, ev["sample_id"], ev["erec"])
        sh.ComputeBins()
        sh.SetupEvents(0, sh.GetEventBins(0, E), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
        sh.SetOscillationWeights(synth.osc_table(spec))
        sh.AddKinShift(-1, 0, 1, NP)
        info = eng.info()
        assert info.n_tiles_brick > 0 and info.n_tiles_uniform == 0
        assert int(info.brick_groups) == (2 if path.endswith("_g2") else 4 if path.endswith("_g4") else 1)
        sh.RecordEventWeights(True)
        sh.Reweight()
        osamp.reweight()
        sh.AddData(osamp.mc)
        osamp.add_data(osamp.mc)
        nominal = osamp.event_bins().copy()
        moved = 0
        for lap, scale in enumerate([0.0, 0.013, -0.02, 0.2, -0.35, 0.0, 0.05, -0.01, 0.02]):
            # the oscillation table is left alone for the first laps: from the third reweight on the packed words carry the events'
            # oscillation WEIGHTS (a 14-bit bin field the pass writes into); at lap 5 the table moves -- back to index words, rebuilt
            # from the bins the pass had left in the words only -- and then rests again
            if lap == 5:
                otbl = osamp.keep[7]
                otbl[:] = np.random.default_rng(9).random(len(otbl)).astype(np.float32)
                sh.SetOscillationWeights(otbl)
            par[:NP] = synth.params(spec, 1 + lap)
            par[NP] = scale
            opar[:] = par
            sh.Reweight()
            llh = sh.GetLikelihood()
            osamp.reweight()
            got, exp = sh.GetEventBins(0, E), osamp.event_bins()
            assert np.array_equal(got, exp), (path, scale)
            moved += int((exp != nominal).sum())
            assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32)), (path, scale)
            assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), (path, scale)
            _llh_close(llh, osamp.get_likelihood())
        assert moved > E // 4
        sh.ClearKinShifts()           # back to the nominal bins; the packed words follow
        par[NP] = opar[NP] = 0.3       # ... and the parameter no longer moves anything
        sh.Reweight()
        assert np.array_equal(sh.GetEventBins(0, E), nominal)
    finally:
        eng.close()
        osamp.close()
        omono.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_shifts,edges_kind", [(1, "equal"), (3, "equal"), (4, "variable"), (2, "variable")])
def test_rebin_1d_shift_kinds_cuts_and_variable_width_bins(n_shifts, edges_kind):
    """k_rebin_1d, the re-binning pass of ONE sample binned in ONE variable (round 6): up to four functional parameters on the axis in
    registration order -- a scale `x *= 1 + a`, an offset `x += b`, `x += c * x_nominal`, a second scale --, two selection cuts on the
    shifted value, bins of equal width and of variable width (where the pass's one guess from the mean bin width misses and the
    binary search answers), values that leave the axis on both sides and come back.  Bins integer-exact against the oracle every lap
    (FindBin, Samples/SampleStructs.h:577-613; ApplyShifts and IsEventSelected, Samples/SampleHandlerFD.cpp:545-564, 281-294),
    event weights bit for bit."""
    from mach3_amd.engine import Engine
    spec = synth.CONFIGS["c3"].scaled(70_000)
    E, NP = spec.n_events, spec.n_par
    rng = np.random.default_rng(91)
    lib = ob.load("serial")
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, E)
    ev = synth.events_chunk(spec, 0, E)
    if edges_kind == "equal":
        edges = synth.bin_edges(spec)
    else:
        w = rng.uniform(0.2, 3.0, spec.n_bins) ** 2
        edges = np.concatenate([[0.0], np.cumsum(w)]) * (5.0 / w.sum())
    kinds = [1, 2, 3, 1][:n_shifts]
    opar = np.concatenate([synth.params(spec, 0), np.zeros(n_shifts)])
    omono = ob.OracleMonolith(lib, spec.n_params, spec.n_knots, cx, npts, ch)
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, ev["sample_id"], ev["erec"], ev["norm_off"], ev["norm_idx"], opar, ev["osc_idx"], synth.osc_table(spec), omono)
    osamp.set_test_statistic(spec.test_statistic, False)
    for q, k in enumerate(kinds):
        osamp.add_shift(-1, 0, k, opar, NP + q)
    osamp.add_cut(-1, 0, 0.15, 4.6)
    osamp.add_cut(0, 0, 0.0, 4.2)
    eng = Engine(0)
    try:
        par = opar.copy()
        mono = SMonolith(eng, spec.n_params, spec.n_knots, cx, npts, [ch], expect_events=E)
        mono.setSplinePointers(par)
        sh = SampleHandlerFD(eng, 1, None, par, spec.test_statistic, bin_edges=[[edges]])
        sh.SetupKinematics(0, ev["sample_id"], ev["erec"])
        sh.ComputeBins()
        sh.SetupEvents(0, sh.GetEventBins(0, E), ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
        sh.SetOscillationWeights(synth.osc_table(spec))
        for q, k in enumerate(kinds):
            if k == 3:
                sh.AddKinShiftAux(-1, 0, 0, NP + q)
            else:
                sh.AddKinShift(-1, 0, k, NP + q)
        sh.AddKinCut(-1, 0, 0.15, 4.6)
        sh.AddKinCut(0, 0, 0.0, 4.2)
        assert eng.info().n_tiles_brick > 0 and eng.info().n_tiles_uniform == 0
        sh.RecordEventWeights(True)
        sh.Reweight()
        osamp.reweight()
        sh.AddData(osamp.mc)
        osamp.add_data(osamp.mc)
        nominal = osamp.event_bins().copy()
        moved = 0
        for lap in range(9):
            par[:NP] = synth.params(spec, 1 + lap)
            amp = [0.0, 0.01, 0.3, 0.02, 3.0, 0.002, 0.1, 0.0, 0.05][lap]  # small walks, a jump, everything off the axis, and back
            par[NP:] = amp * rng.standard_normal(n_shifts)
            opar[:] = par
            llh = sh.Step()
            osamp.reweight()
            got, exp = sh.GetEventBins(0, E), osamp.event_bins()
            assert np.array_equal(got, exp), (lap, par[NP:])
            moved += int((exp != nominal).sum())
            assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32))
            assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
            _llh_close(llh, osamp.get_likelihood())
        assert moved > E // 2
    finally:
        eng.close()
        osamp.close()
        omono.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["lds_pass_into_word", "lds_pass_no_word", "brick_tail", "global_pass_into_word"])
def test_functional_shifts_on_a_ragged_two_sample_2d_space(path, monkeypatch):
    """The re-binning forms on bricks with TWO kinematic columns and two samples (60 000 ragged events): a 2-D sample moved by an
    energy scale and by `y += theta * E` (the aux form), with a selection cut on the shifted y, beside a 1-D sample that no
    shift touches (its events keep their nominal bins and are neither read nor written by the LDS pass).  Bins integer-exact
    against the oracle, event weights bit-exact, bin contents 1e-12, LLH 1e-9."""
    from mach3_amd.engine import Engine
    if "no_word" in path:
        monkeypatch.setenv("M3HIP_PACKED_EVENTS", "0")
    if path.startswith("global"):
        monkeypatch.setenv("M3HIP_REBIN", "global")
    spec = synth.CONFIGS["c3"].scaled(60_000)
    E, NP = spec.n_events, spec.n_par
    rng = np.random.default_rng(77)
    lib = ob.load("serial")
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, E)
    ev = synth.events_chunk(spec, 0, E)
    sample_id = (rng.random(E) < 0.4).astype(np.int32)          # sample 0: 2-D, moved; sample 1: 1-D, untouched
    kin = np.stack([ev["erec"], rng.uniform(-1.05, 1.05, E)], axis=1)
    ex, ey = synth.bin_edges(spec), np.linspace(-1.0, 1.0, 7)
    opar = np.concatenate([synth.params(spec, 0), [0.0, 0.0]])
    omono = ob.OracleMonolith(lib, spec.n_params, spec.n_knots, cx, npts, ch)
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([ex, ey])
    b.add_uniform([ex])
    osamp = ob.OracleSample(lib, b, sample_id, kin, ev["norm_off"], ev["norm_idx"], opar, ev["osc_idx"], synth.osc_table(spec), omono)
    osamp.set_test_statistic(spec.test_statistic, False)
    osamp.add_shift(0, 0, 1, opar, NP)                 # E *= 1 + theta_0, sample 0
    osamp.add_shift(0, 1, 3 | (0 << 8), opar, NP + 1)  # y += theta_1 * E (nominal), sample 0
    osamp.add_cut(0, 1, -0.9, 0.95)

    eng = Engine(0)
    try:
        par = opar.copy()
        mono = SMonolith(eng, spec.n_params, spec.n_knots, cx, npts, [ch], expect_events=E)
        mono.setSplinePointers(par)
        sh = SampleHandlerFD(eng, 2, None, par, spec.test_statistic, bin_edges=[[ex, ey], [ex]])
        sh.SetupKinematics(0, sample_id, kin)
        sh.AddKinCut(0, 1, -0.9, 0.95)
        sh.ComputeBins()
        nominal = sh.GetEventBins(0, E)
        osamp.reweight()  # (the oracle bins at reweight; at theta = 0 that is the nominal binning)
        assert np.array_equal(nominal, osamp.event_bins())
        sh.SetupEvents(0, nominal, ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
        sh.SetOscillationWeights(synth.osc_table(spec))
        sh.AddKinShift(0, 0, 1, NP)
        sh.AddKinShiftAux(0, 1, 0, NP + 1)
        assert eng.info().n_tiles_brick > 0 and eng.info().n_tiles_uniform == 0
        sh.RecordEventWeights(True)
        sh.Reweight()
        sh.AddData(osamp.mc)
        osamp.add_data(osamp.mc)
        moved = 0
        for lap, (s0, s1) in enumerate([(0.0, 0.0), (0.02, 0.0), (-0.03, 0.05), (0.25, -0.1), (0.0, 0.3), (0.0, 0.0)]):
            par[:NP] = synth.params(spec, 1 + lap)
            par[NP], par[NP + 1] = s0, s1
            opar[:] = par
            sh.Reweight()
            llh = sh.GetLikelihood()
            osamp.reweight()
            got, exp = sh.GetEventBins(0, E), osamp.event_bins()
            assert np.array_equal(got, exp), (path, s0, s1)
            assert np.array_equal(got[sample_id == 1], nominal[sample_id == 1])   # the untouched sample stays put
            moved += int((exp != nominal).sum())
            assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32)), (path, s0, s1)
            assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), (path, s0, s1)
            _llh_close(llh, osamp.get_likelihood())
        assert moved > E // 10
    finally:
        eng.close()
        osamp.close()
        omono.close()


@pytest.mark.gpu
def test_parameter_handler_kernels_through_the_c_abi(fresh_engine):
    """m3hip_par_*: the prior term 0.5 d^T V^-1 d (ParameterHandlerBase::CalcLikelihood, Parameters/ParameterHandlerBase.cpp:816-841)
    and the correlated throw L z (CorrelateSteps, :726-745; M3::MatrixVectorMulti, Parameters/ParameterHandlerUtils.h:118-135)
    against numpy in extended precision, several sizes incl. odd ones, repeated launches (ring of two input slots), a matrix
    replaced between launches, and the error paths."""
    eng = fresh_engine
    rng = np.random.default_rng(11)
    with pytest.raises(capi.M3HipError, match="no such parameter-handler slot"):
        eng.par_launch(0, np.zeros(4), np.zeros(4))
    sl = eng.par_create()
    with pytest.raises(capi.M3HipError, match="before both matrices"):
        eng.par_launch(sl, np.zeros(4), np.zeros(4))
    with pytest.raises(capi.M3HipError, match="before par_launch"):
        eng.par_result(sl, 4)
    for n in (1, 5, 64, 65, 513, 1040):
        a = rng.standard_normal((n, n))
        cov = a @ a.T / n + 0.3 * np.eye(n)
        inv = np.linalg.inv(cov)
        inv = 0.5 * (inv + inv.T)
        L = np.linalg.cholesky(cov)
        eng.par_upload(sl, 0, inv)
        eng.par_upload(sl, 1, L)
        for rep in range(3):
            d, z = rng.standard_normal(n), rng.standard_normal(n)
            if rep == 1:
                d[::3] = 0.0      # flat priors
                z[1::4] = 0.0     # fixed parameters
            if rep == 2:          # a new throw matrix between two launches
                L = np.linalg.cholesky(cov + 0.1 * np.eye(n))
                eng.par_upload(sl, 1, L)
            eng.par_launch(sl, d, z)
            prior, thr = eng.par_result(sl, n)
            dl, il, Ll, zl = d.astype(np.longdouble), inv.astype(np.longdouble), L.astype(np.longdouble), z.astype(np.longdouble)
            ref_prior = float(sum(dl[i] * (il[i, :i] @ dl[:i] + 0.5 * il[i, i] * dl[i]) for i in range(n)))
            ref_thr = np.array([float(Ll[i, :i + 1] @ zl[:i + 1]) for i in range(n)])
            assert abs(prior - ref_prior) <= 1e-12 * max(abs(ref_prior), 1.0), (n, rep)
            assert np.allclose(thr, ref_thr, rtol=1e-12, atol=1e-13 * np.abs(ref_thr).max()), (n, rep)
            # bit-reproducible: the same launch again gives the same bits
            eng.par_launch(sl, d, z)
            prior2, thr2 = eng.par_result(sl, n)
            assert prior2 == prior and np.array_equal(thr2, thr)
    with pytest.raises(capi.M3HipError, match="n differs"):
        eng.par_launch(sl, np.zeros(3), np.zeros(3))
    with pytest.raises(capi.M3HipError, match="which is 0"):
        eng.par_upload(sl, 2, np.eye(3))
    eng.par_destroy(sl)
    with pytest.raises(capi.M3HipError, match="no such parameter-handler slot"):
        eng.par_result(sl, 3)


@pytest.mark.gpu
def test_two_parameter_handlers_share_one_engine(fresh_engine):
    """MCMCBase loops over ALL systematics objects (Fitters/MCMCBase.cpp:32-77): two handlers attached to one engine each get a
    slot of their own -- matrices, results and tickets do not mix, whether the two have the same number of parameters or not,
    and both launches may be in flight before either result is read (the order MCMCBase::EvaluateProposal produces)."""
    eng = fresh_engine
    rng = np.random.default_rng(5)

    def handler(n, shift):
        a = rng.standard_normal((n, n))
        cov = a @ a.T / n + shift * np.eye(n)
        inv = np.linalg.inv(cov)
        return 0.5 * (inv + inv.T), np.linalg.cholesky(cov)

    def ref(inv, L, d, z):
        n = len(d)
        return sum(d[i] * (inv[i, :i] @ d[:i] + 0.5 * inv[i, i] * d[i]) for i in range(n)), np.array([L[i, :i + 1] @ z[:i + 1] for i in range(n)])

    for na, nb in ((40, 40), (40, 73)):
        sa, sb = eng.par_create(), eng.par_create()
        assert sa != sb
        (inv_a, L_a), (inv_b, L_b) = handler(na, 0.3), handler(nb, 0.7)
        eng.par_upload(sa, 0, inv_a); eng.par_upload(sa, 1, L_a)
        eng.par_upload(sb, 0, inv_b); eng.par_upload(sb, 1, L_b)       # equal n: must not overwrite slot a's matrices
        for step in range(4):
            da, za, db, zb = rng.standard_normal(na), rng.standard_normal(na), rng.standard_normal(nb), rng.standard_normal(nb)
            eng.par_launch(sa, da, za)
            eng.par_launch(sb, db, zb)                                   # before slot a's result was read
            pa, ta = eng.par_result(sa, na)
            pb, tb = eng.par_result(sb, nb)
            ra, rta = ref(inv_a, L_a, da, za)
            rb, rtb = ref(inv_b, L_b, db, zb)
            assert abs(pa - ra) <= 1e-12 * max(abs(ra), 1) and abs(pb - rb) <= 1e-12 * max(abs(rb), 1), (na, nb, step)
            assert np.allclose(ta, rta, rtol=1e-12, atol=1e-13) and np.allclose(tb, rtb, rtol=1e-12, atol=1e-13)
        eng.par_destroy(sa)
        eng.par_destroy(sb)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rule", "always", "never", "rule_unpacked", "always_unpacked"])
def test_large_static_oscillation_table_becomes_a_per_event_column(mode, monkeypatch):
    """A pre-staged oscillation table of hundreds of KB (BASELINE configs[4]: 204 000 entries) does not stay in L2 beside the
    brick kernel's streams; once table and indices have been left alone for two steps the engine folds the weights INTO the events'
    packed sample words (round 5: the word then carries the weight's float bits instead of the index), or -- samples that do not
    use the packed word -- gathers them once into a per-event column the kernel streams.  Same numbers either way: event weights
    bit-exact against the oracle through the transitions gather -> weight words / column -> (table re-uploaded) gather -> ...,
    with events that have no oscillation pointer, zero weights, and a slice of indices re-uploaded.  `M3HIP_OSC_COLUMN` = 1 / 0
    forces / forbids it."""
    from mach3_amd.engine import Engine
    if mode.endswith("_unpacked"):
        monkeypatch.setenv("M3HIP_PACKED_EVENTS", "0")
        mode = mode[:-len("_unpacked")]
    if mode != "rule":
        monkeypatch.setenv("M3HIP_OSC_COLUMN", "1" if mode == "always" else "0")
    # 200 KB of weights: "large" by the engine's rule.  Poisson: the oracle sample is re-created half way (new indices), and a
    # Barlow-Beeston sample would freeze another W2 there than the engine's
    spec = synth.CONFIGS["c3"].scaled(40_000, n_params=64, osc_n=50_000, test_statistic=capi.kPoisson)
    eng = Engine(0)
    try:
        eng2, mono, sh, par = build_from_synth(spec, engine=eng)
        omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
        otbl = osamp.keep[7]
        osc_idx = osamp.keep[6]
        assert (osc_idx >= 0).sum() > 20_000 and (osc_idx < 0).sum() > 0 and eng.info().n_tiles_brick > 0
        sh.RecordEventWeights(True)
        sh.Reweight()
        osamp.reweight()
        data = osamp.mc.copy()
        sh.AddData(data)
        osamp.add_data(data)
        rng = np.random.default_rng(4)
        for lap in range(1, 16):
            if lap in (6, 7):       # the table moves for two steps: back to the gather
                otbl[:] = rng.random(len(otbl)).astype(np.float32)
                if lap == 7:
                    otbl[::5] = 0.0
                sh.SetOscillationWeights(otbl)
            if lap == 12:           # a slice of the events' oscillation indices changes (and with it their bins' contents)
                sl = slice(1024, 9000)
                osc_idx = osc_idx.copy()
                osc_idx[sl] = rng.integers(-1, spec.osc_n, sl.stop - sl.start).astype(np.int32)
                sh.SetupEvents(0, ev["bin_idx"], osc_idx, ev["norm_off"], ev["norm_idx"])
                # (the oracle resolves its events' pointers when the sample is created: a new sample over the new indices)
                osamp.close()
                b2 = ob.OracleBinning(ob.load("serial"))
                b2.add_uniform([synth.bin_edges(spec)])
                osamp = ob.OracleSample(ob.load("serial"), b2, ev["sample_id"], ev["erec"], ev["norm_off"], ev["norm_idx"], opar, osc_idx, otbl, omono)
                osamp.set_test_statistic(spec.test_statistic, False)
                osamp.reweight()
                osamp.add_data(data)
            par[:] = synth.params(spec, lap)
            opar[:] = par
            llh = sh.Step()
            osamp.reweight()
            assert np.array_equal(sh.GetEventWeights(0, spec.n_events).view(np.uint32), osamp.event_weights().view(np.uint32)), (mode, lap)
            assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), (mode, lap)
            _llh_close(llh, osamp.get_likelihood())
        osamp.close()
        omono.close()
    finally:
        eng.close()
