"""m3hip_monolith_order_key: the engine stores a chunk's events in the order of a caller-supplied key (their oscillation
index, so that a wavefront's look-ups into a large table that MOVES every step share lines) -- and nothing of it shows through
the interface: every per-event array goes in, and every per-event result comes out, in the CALLER's event numbering.  The
reference's events alias the calculator's array in MC order (Samples/SampleHandlerFD.cpp:1108-1133) and re-read it in every
Reweight (SampleHandlerFD.cpp:323, Samples/OscillationHandler.cpp:81-100).  Checked against the serial oracle, which knows
nothing of any order: event weights bit for bit, bins 1e-12, likelihood 1e-9."""
import numpy as np
import pytest

import oracle_binding as ob
from helpers import make_monolith, natural_cubic
from mach3_amd import capi, synth
from mach3_amd.engine import SMonolith, SampleHandlerFD, build_from_synth

pytestmark = pytest.mark.gpu

RTOL_BIN = 1e-12


def _llh_close(a, b):
    assert a == pytest.approx(b, rel=1e-9, abs=1e-9)


@pytest.mark.parametrize("cfg,n_params,n_events,chunk,groups", [
    ("c2", 12, 9000, 4096, None),          # uniform tiles
    ("c3", 96, 9000 + 77, 4096, "1"),      # bricks of 1024 events
    ("c5", 128, 3 * 8192 + 500, 8192, "4"),  # bricks of 4096 events, several samples
    ("c3", 64, 20000, 20000, None),        # one chunk that ends in a partial brick
])
def test_events_stored_in_oscillation_order_results_in_the_callers(fresh_engine, monkeypatch, cfg, n_params, n_events, chunk, groups):
    if groups:
        monkeypatch.setenv("M3HIP_BRICK_GROUPS", groups)
    spec = synth.CONFIGS[cfg].scaled(n_events, n_params=n_params, osc_n=40000)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine, chunk_events=chunk, order_by_osc=True)
    info = eng.info()
    assert info.n_keyed_chunks == 1 and info.n_key_refused == 0   # (keyed chunks are held back until 2^20 events are together: ONE chunk here)
    omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, n_events)
    otbl = osamp.keep[7]
    assert info.n_splines == ch["n_splines"] and info.n_tf1 == ch["n_tf1"]
    # the look-ups of 1024 consecutive events AS STORED: in the key's order (stable, -1 last)
    def lines(osc):
        return np.mean([len(np.unique(osc[a:a + 1024][osc[a:a + 1024] >= 0] >> 5)) for a in range(0, n_events - 1023, 1024)])
    stored = ev["osc_idx"][np.argsort(ev["osc_idx"].view(np.uint32), kind="stable")]
    assert info.osc_lines_per_kevent == pytest.approx(lines(stored), rel=1e-12) and lines(stored) < 0.25 * lines(ev["osc_idx"])
    assert np.array_equal(sh.GetEventBins(0, n_events), ev["bin_idx"])          # uploaded in the caller's order, read back in it
    assert np.array_equal(sh.GetEventBins(1000, 777), ev["bin_idx"][1000:1777])  # ... any range
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0)
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    rng = np.random.default_rng(5)
    for lap in range(1, 7):
        if lap != 4:                              # the table moves (lap 4: it rests for one step)
            otbl[:] = rng.random(len(otbl)).astype(np.float32)
            sh.SetOscillationWeights(otbl)
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetSplineWeights(0, n_events).view(np.uint32), omono.total_weights.view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(0, n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(513, 2000).view(np.uint32), osamp.event_weights()[513:2513].view(np.uint32))
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
        _llh_close(llh, osamp.get_likelihood())
    for lap in range(7, 12):                      # left alone: a large table's weights end up in the events' packed words
        par[:] = synth.params(spec, lap)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventWeights(0, n_events).view(np.uint32), osamp.event_weights().view(np.uint32))
        _llh_close(llh, osamp.get_likelihood())
    # seam 2 on the same engine: cpu_total_weights in the caller's order
    mono.Evaluate()
    mono.SynchroniseMemTransfer()
    omono.evaluate()
    assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32))
    # the sample side re-uploaded in slices that ignore the chunk borders
    ev2 = {k: v.copy() for k, v in ev.items()}
    ev2["bin_idx"][::7] = -1
    for a in range(0, n_events, 3000):
        b = min(n_events, a + 3000)
        no = ev2["norm_off"][a:b + 1] - ev2["norm_off"][a]
        ni = ev2["norm_idx"][ev2["norm_off"][a]:ev2["norm_off"][b]]
        sh.SetupEvents(a, ev2["bin_idx"][a:b], ev2["osc_idx"][a:b], no.astype(np.uint64), ni)
    assert np.array_equal(sh.GetEventBins(0, n_events), ev2["bin_idx"])
    osamp.close()
    omono.close()


def test_keyed_chunks_are_held_back_until_2e20_events_are_together(fresh_engine):
    """2.5e6 light events (configs[4]'s shape) uploaded in chunks of 131 072 with their oscillation indices as the key: the engine
    lays them out as chunks of 2^20, 2^20 and the rest -- three sorts, not twenty -- and a chunk WITHOUT a key in between flushes
    what is held back (the events keep their places).  Windows across the engine's own chunk borders bit for bit."""
    n_events, chunk = 2_500_000 + 333, 131072
    spec = synth.CONFIGS["c5"].scaled(n_events, n_params=128)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine, chunk_events=chunk, order_by_osc=True)
    info = eng.info()
    assert info.n_keyed_chunks == 3 and info.n_key_refused == 0 and info.n_events == n_events
    osc_all = np.concatenate([synth.events_chunk(spec, a, min(n_events, a + 500_000))["osc_idx"] for a in range(0, n_events, 500_000)])
    stored = np.concatenate([osc_all[a:a + (1 << 20)][np.argsort(osc_all[a:a + (1 << 20)].view(np.uint32), kind="stable")] for a in range(0, n_events, 1 << 20)])
    lines = np.mean([len(np.unique(stored[a:a + 1024][stored[a:a + 1024] >= 0] >> 5)) for a in range(0, n_events - 1023, 1024)])
    assert info.osc_lines_per_kevent == pytest.approx(lines, rel=1e-12) and lines < 12
    windows = [(0, 2000), ((1 << 20) - 1000, (1 << 20) + 1000), ((1 << 21) - 1000, (1 << 21) + 1000), (n_events - 1500, n_events)]
    oracles = [ob.build_oracle_from_synth(spec, a, b) for a, b in windows]
    sh.RecordEventWeights(True)
    rng = np.random.default_rng(3)
    for lap in (1, 2):
        tbl = rng.random(spec.osc_n).astype(np.float32)
        sh.SetOscillationWeights(tbl)
        par[:] = synth.params(spec, lap)
        sh.Step()
        for (a, b), (omono, osamp, opar, ch, ev) in zip(windows, oracles):
            osamp.keep[7][:] = tbl
            opar[:] = par
            osamp.reweight()
            assert np.array_equal(sh.GetSplineWeights(a, b - a).view(np.uint32), omono.total_weights.view(np.uint32)), (lap, a)
            assert np.array_equal(sh.GetEventWeights(a, b - a).view(np.uint32), osamp.event_weights().view(np.uint32)), (lap, a)
            assert np.array_equal(sh.GetEventBins(a, b - a), ev["bin_idx"])
    for omono, osamp, *_ in oracles:
        osamp.close()
        omono.close()
    # keyed, keyed, plain, keyed: the plain chunk flushes the two held back; the last one is flushed by monolith_end
    cx, npts = synth.knots(spec)
    eng.monolith_begin(spec.n_params, spec.n_knots, cx, npts)
    for i, a in enumerate(range(0, 4 * 4096, 4096)):
        ch = synth.monolith_chunk(spec, a, a + 4096)
        if i != 2:
            ch["order_key"] = synth.events_chunk(spec, a, a + 4096)["osc_idx"]
        eng.monolith_append(ch)
    eng.monolith_end()
    info = eng.info()
    assert info.n_keyed_chunks == 2 and info.n_events == 4 * 4096


def test_a_key_in_order_changes_nothing_and_a_key_for_the_wrong_chunk_is_refused(fresh_engine):
    spec = synth.CONFIGS["c3"].scaled(6000, n_params=64)
    cx, npts = synth.knots(spec)
    eng = fresh_engine
    ch = synth.monolith_chunk(spec, 0, 6000)
    eng.monolith_begin(spec.n_params, spec.n_knots, cx, npts)
    ch["order_key"] = np.arange(6000, dtype=np.int32) // 10          # already ascending: the stable sort moves nothing
    eng.monolith_append(ch)
    eng.monolith_end()
    info = eng.info()
    assert info.n_keyed_chunks == 0 and info.n_key_refused == 0 and info.n_events == 6000
    eng.monolith_begin(spec.n_params, spec.n_knots, cx, npts)
    ch["order_key"] = np.zeros(100, np.int32)
    with pytest.raises(capi.M3HipError, match="another number of events"):
        eng.monolith_append(ch)
    del ch["order_key"]
    eng.monolith_append(ch)                                           # the refused key is gone: the chunk goes up in its own order
    eng.monolith_end()
    assert eng.info().n_keyed_chunks == 0
    with pytest.raises(capi.M3HipError, match="outside"):
        eng._chk(eng.lib.m3hip_monolith_order_key(eng.h, None, 0))


def test_a_key_that_tears_the_modes_apart(fresh_engine):
    """Mode-structured MC (blocks of events with one parameter list each) and a key that is random against the modes: stored in
    the key's order the chunk is re-grouped by list for processing (the engine's own grouping, stable: inside a mode the key's
    order stays), or -- with the grouping switched off -- the key is refused because its order inflates the layout.  Either
    way the numbers are the oracle's."""
    rng = np.random.default_rng(17)
    E, P = 6 * 1024, 24
    X7 = np.arange(-3.0, 4.0)
    modes = [sorted(rng.choice(P, 10, replace=False)) for _ in range(6)]
    events = []
    for e in range(E):
        events.append([(int(p), "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))) for p in modes[e // 1024]])
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
    key = rng.integers(0, 5000, E).astype(np.int32)
    bins = rng.integers(-1, 40, E).astype(np.int32)
    osc = key.copy()
    tbl = rng.random(5000).astype(np.float32)
    lib = ob.load("serial")
    par = 0.3 * rng.standard_normal(P)
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    omono.evaluate()
    for grouping in ("1", "0"):
        import os
        os.environ["M3HIP_GROUP_EVENTS"] = grouping
        try:
            chk = dict(ch)
            chk["order_key"] = key
            mono = SMonolith(fresh_engine, P, 7, cx, npts, [chk])
        finally:
            del os.environ["M3HIP_GROUP_EVENTS"]
        info = fresh_engine.info()
        if grouping == "1":
            assert info.n_keyed_chunks == 1 and info.n_key_refused == 0 and info.n_grouped_chunks == 1 and info.n_tiles_brick == 0
        else:
            assert info.n_keyed_chunks == 0 and info.n_key_refused == 1 and info.n_tiles_brick == 0
        mono.setSplinePointers(par)
        sh = SampleHandlerFD(fresh_engine, 1, np.array([0, 40], np.int32), par)
        sh.SetupEvents(0, bins, osc, None, None)
        sh.SetOscillationWeights(tbl)
        sh.RecordEventWeights(True)
        sh.Reweight()
        w = sh.GetEventWeights(0, E)
        sw = sh.GetSplineWeights(0, E)
        assert np.array_equal(sw.view(np.uint32), omono.total_weights.view(np.uint32))
        assert np.array_equal(w.view(np.uint32), (tbl[osc] * omono.total_weights).view(np.uint32))   # CalcWeightTotal: osc first, then the spline weight
        exp = np.zeros(40)
        keep = (bins >= 0) & (w > 0)
        np.add.at(exp, bins[keep], w[keep].astype(np.float64))
        assert np.allclose(sh.GetMCArray(), exp, rtol=1e-12, atol=0)
    omono.close()


def test_kinematics_experiment_weights_and_shifts_under_a_key(fresh_engine):
    """The other per-event uploads -- nominal kinematics (rows of doubles) with device-side binning and a functional shift that
    re-bins every step, a per-event experiment weight, an index into a per-step weight table -- through an engine that stores its
    events in another order: bins integer-exact, event weights bit for bit, in the caller's numbering."""
    rng = np.random.default_rng(23)
    E, P = 9000, 40
    X7 = np.arange(-3.0, 4.0)
    events = []
    for e in range(E):
        ps = sorted(rng.choice(P, rng.integers(1, 9), replace=False))
        events.append([(int(p), "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))) for p in ps])
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
    key = rng.integers(-1, 3000, E).astype(np.int32)            # (-1: no oscillation pointer -- stored last)
    edges = np.linspace(0.0, 5.0, 41)
    sample_id = np.zeros(E, np.int32)
    kin = np.stack([rng.uniform(-0.2, 5.2, E), rng.uniform(0.0, 1.0, E)], axis=1)
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    par = np.concatenate([0.3 * rng.standard_normal(P), [0.0]])
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    tbl = rng.random(3000).astype(np.float32)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, opar, key, tbl, omono)
    osamp.add_shift(0, 0, 1, opar, P)                            # x *= (1 + theta_P)
    col = rng.uniform(0.5, 1.5, E).astype(np.float32)
    idx = rng.integers(-1, 11, E).astype(np.int32)
    wt = np.ones(11, np.float32)
    osamp.add_weight_pointers(col)
    osamp.add_weight_pointers(wt, idx)

    chk = dict(ch)
    chk["order_key"] = key
    mono = SMonolith(fresh_engine, P, 7, cx, npts, [chk])
    assert fresh_engine.info().n_keyed_chunks == 1 and fresh_engine.info().n_tiles_brick > 0
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 1, None, par, bin_edges=[[edges]])
    sh.SetupKinematics(0, sample_id[:5000], kin[:5000])
    sh.SetupKinematics(5000, sample_id[5000:], kin[5000:])
    sh.ComputeBins()
    assert np.array_equal(sh.GetEventBins(0, E), osamp.event_bins())
    sh.SetupEvents(0, sh.GetEventBins(0, E), key, None, None)
    sh.SetOscillationWeights(tbl)
    sh.AddKinShift(0, 0, 1, P)
    sh.SetEventWeights(0, 0, col)
    sh.SetWeightIndex(1, 0, idx)
    sh.SetWeightTable(1, wt)
    sh.RecordEventWeights(True)
    for lap in range(4):
        par[:P] = 0.3 * rng.standard_normal(P)
        par[P] = 0.05 * rng.standard_normal()
        opar[:] = par
        wt[:] = (1.0 + 0.2 * np.sin(np.arange(11) + lap)).astype(np.float32)
        sh.SetWeightTable(1, wt)
        sh.Reweight()
        osamp.reweight()
        assert np.array_equal(sh.GetEventBins(0, E), osamp.event_bins()), lap
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32)), lap
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0)
    osamp.close()
    omono.close()
