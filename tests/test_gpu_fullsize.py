"""BASELINE configs at FULL size through the C ABI -- configs[1] and configs[2] whole, configs[3] and configs[4] at the
shard one of their eight GPUs holds -- checked by size-independent properties plus exact comparison on event windows
the oracle can afford:
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


def _check_full_size(spec, e0, e1, windows, chunk_events=131072, laps=(1, 2), shard_check=True, order_key=False, move_table=False):
    """properties of events [e0, e1) of `spec` on one GPU (see the module docstring); windows are relative to e0.
    order_key: every chunk goes up with its events' oscillation indices as the order key (m3hip_monolith_order_key);
    move_table: a new oscillation table before every lap (the step of a fit that moves oscillation parameters)"""
    E = e1 - e0
    eng, mono, sh, par = build_from_synth(spec, chunk_events=chunk_events, event_range=(e0, e1), order_by_osc=order_key)
    info = eng.info()
    assert info.n_events == E
    ns, nt = 0, 0
    for a in range(e0, e1, 1 << 20):
        s_, t_ = synth.count(spec, a, min(e1, a + (1 << 20)))
        ns, nt = ns + s_, nt + t_
    assert (info.n_splines, info.n_tf1) == (ns, nt)
    assert info.alg_bytes_fused == 16 * ns + 8 * nt + 8 * E
    sh.RecordEventWeights(True)
    sh.Reweight()
    data = sh.GetMCArray()
    w2_first = sh.GetW2Array()
    sh.AddData(data)
    assert sh.GetLikelihood() == 0.0                      # Asimov (every bin: data == mc)
    oracles = [ob.build_oracle_from_synth(spec, e0 + a, e0 + b) for a, b in windows]
    bins_all = np.concatenate([synth.events_chunk(spec, a, min(e1, a + 500_000))["bin_idx"] for a in range(e0, e1, 500_000)])
    assert np.array_equal(sh.GetEventBins(0, 4096) if hasattr(sh, "GetEventBins") else bins_all[:4096], bins_all[:4096])
    mc = None
    for lap in laps:
        par[:] = synth.params(spec, lap)
        if move_table:
            tbl = np.random.default_rng(100 + lap).random(spec.osc_n).astype(np.float32)
            sh.SetOscillationWeights(tbl)
            for _, osamp, *_rest in oracles:
                osamp.keep[7][:] = tbl                    # the oracle's events alias this array
        llh = sh.Step()
        assert llh > 0
        mc = sh.GetMCArray()
        ew = sh.GetEventWeights(0, E)
        sel = (ew > 0) & (bins_all >= 0)                  # conservation, in fp64 (float weights into double bins)
        ref = np.bincount(bins_all[sel], weights=ew[sel].astype(np.float64), minlength=spec.total_bins)
        assert np.allclose(mc, ref, rtol=1e-10, atol=0)
        assert np.array_equal(sh.GetW2Array(), w2_first)  # W2 frozen after the first Reweight
        for (a, b), (omono, osamp, opar, ch, ev) in zip(windows, oracles):
            opar[:] = par
            osamp.reweight()
            assert np.array_equal(sh.GetSplineWeights(a, b - a).view(np.uint32), omono.total_weights.view(np.uint32))
            assert np.array_equal(ew[a:b].view(np.uint32), osamp.event_weights().view(np.uint32))
    for omono, osamp, *_ in oracles:
        osamp.close()
        omono.close()
    info = eng.info()  # (again: with what the reweights planned)
    eng.close()
    if shard_check:  # two ranks' worth of events on two handles: what the all-reduce would sum
        cut = e0 + ((E // 2 + 1023) // 1024) * 1024
        parts = []
        for a, b in ((e0, cut), (cut, e1)):
            e2, m2, s2, p2 = build_from_synth(spec, chunk_events=chunk_events, event_range=(a, b))
            p2[:] = synth.params(spec, laps[-1])
            s2.Reweight()
            parts.append(s2.GetMCArray())
            e2.close()
        assert np.allclose(parts[0] + parts[1], mc, rtol=1e-10, atol=0)
    return info


def test_config3_full_size_properties():
    """configs[2]: 1e7 events, 512 mixed TSpline3 / TF1 parameters at occupancy 1/16, Barlow-Beeston: 9 766 bricks.  In the
    reference's layout this monolith holds more than 2^32 coefficient floats -- past its `unsigned int` offsets; here it
    is uploaded in 77 chunks with chunk-local offsets, and the LAST events are checked bit for bit like the first."""
    spec = synth.CONFIGS["c3"]
    E = spec.n_events
    info = _check_full_size(spec, 0, E, [(0, 2500), (5_000_000, 5_002_048), (E - 3000, E)])
    assert info.brick_groups == 1 and info.n_tiles_brick == (E + 1023) // 1024 and info.n_tiles_uniform == 0
    assert info.n_splines * spec.n_knots * 4 > 2 ** 32
    # the small streams a step re-reads: stage tables + sample words + TF1 bookkeeping stay in the Infinity Cache (about 165 MB),
    # the TF1 codes (176 MB more) do not fit beside them and go around it (m3hip_step.hip choose_nt_codes)
    assert info.tf1_codes_around == 1 and 150e6 < info.cached_stream_bytes < 200e6


def test_config4_shard_properties():
    """configs[3]: four samples x 1e7 events (800 bins); the shard one of eight GPUs holds: 5e6 events of all four samples"""
    spec = synth.CONFIGS["c4"]
    per = spec.n_events // 8
    info = _check_full_size(spec, 3 * per, 4 * per, [(0, 2500), (per - 2048, per)], laps=(1,))
    assert info.n_samples == 4 and info.n_bins == 800
    assert info.n_tiles_brick == 5120   # 4 883 bricks of 1024 events = 19.07 per compute unit: 948 near the end are cut at 768 (20 each)


def test_config5_shard_properties():
    """configs[4]: 1e8 events, 1 024 parameters at occupancy 1/64, beam + atmospheric oscillation tables (204 000 pre-staged
    weights); the shard one of eight GPUs holds: 1.25e7 events"""
    spec = synth.CONFIGS["c5"]
    per = spec.n_events // 8
    info = _check_full_size(spec, 5 * per, 6 * per, [(0, 3000), (per - 2048, per)], laps=(1,), shard_check=False)
    # 16 responses per event and 12 208 groups of 1024 events: the engine lays this shard out in bricks of 4096 events (chunk by chunk)
    assert info.n_params == 1024 and info.brick_groups == 4
    assert info.n_tiles_brick == sum(-(-min(131072, per - a) // 4096) for a in range(0, per, 131072))


def test_config5_shard_stored_in_oscillation_order_table_replaced_every_step():
    """the same shape (another GPU's share) with the engine storing every 2^20-event chunk in oscillation-bin order
    (m3hip_monolith_order_key) and the 204 000-entry table replaced before every step: what a fit that moves oscillation
    parameters runs (Samples/SampleHandlerFD.cpp:323, Samples/OscillationHandler.cpp:81-100).  Windows at the start, across a
    chunk border and at the end bit for bit in the CALLER's event numbering; conservation over all 1.25e7 events."""
    spec = synth.CONFIGS["c5"]
    per = spec.n_events // 8
    info = _check_full_size(spec, 2 * per, 3 * per, [(0, 3000), ((1 << 20) - 1024, (1 << 20) + 1024), (per - 2048, per)], chunk_events=1 << 20,
                            laps=(1, 2), shard_check=False, order_key=True, move_table=True)
    assert info.n_keyed_chunks == -(-per // (1 << 20)) and info.n_key_refused == 0 and info.brick_groups == 4
    assert info.osc_lines_per_kevent < 10            # (858 in the MC's own order)


def test_more_than_1024_parameters_at_mid_size():
    """the BIGP instantiation of the brick kernel (more spline parameters than its LDS tables hold: {segment, dx} and the TF1
    parameter values come from global memory) beyond toy size: 150 000 events x 1 536 mixed parameters at occupancy 1/64,
    two windows bit for bit against the serial oracle, conservation, W2 freeze, shard additivity"""
    spec = synth.CONFIGS["c5"].scaled(150_000, n_params=1536)
    info = _check_full_size(spec, 0, spec.n_events, [(0, 3000), (spec.n_events - 2500, spec.n_events)], chunk_events=65536)
    assert info.n_params == 1536 and info.n_tiles_brick > 0 and info.n_tiles_uniform == 0


def test_short_bricks_even_out_a_launch_at_mid_size(monkeypatch):
    """250 bricks' worth of config-3 events on 256 compute units, the engine's rule for an even launch forced on (it asks for eight
    rounds otherwise): told the size, the engine converts 18 full bricks near the end into 24 of 768 events (m3hip_upload.hip
    ShortBrickPlan) -- one brick per unit; the short bricks' events (the last windows) bit for bit against the serial oracle,
    conservation, W2 freeze as for every other sample.  config 4's share of one GPU in eight (5e6 events) gets the rule by itself:
    test_config4_shard_properties."""
    monkeypatch.setenv("M3HIP_SHORT_BRICKS", "2")
    spec = synth.CONFIGS["c3"].scaled(250 * 1024)
    E = spec.n_events
    info = _check_full_size(spec, 0, E, [(0, 2500), (E - 21_000, E - 18_000), (E - 2500, E)], chunk_events=65536, shard_check=False)
    assert info.brick_groups == 1 and info.n_tiles_brick == 256 and info.n_tiles_uniform == 0
