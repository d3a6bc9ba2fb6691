"""Seeded random shapes through the whole fused step, against the oracle: event counts that are not multiples of
anything, parameters with 3 / 5 / 7 knots or TF1 or nothing, uniform and ragged groups in one sample, events with
no response, several samples, few and many bins, every test statistic, W2 frozen or updated.  The cases are
deterministic (the seed is the test id); they are meant to find corner cases the hand-written tests miss."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from helpers import make_monolith, natural_cubic
from mach3_amd import capi, synth
from mach3_amd.engine import SMonolith, SampleHandlerFD

pytestmark = pytest.mark.gpu

RTOL_BIN = 1e-12
RTOL_LLH = 1e-9
# M3_FUZZ_SCALE=N runs N times as many seeds (occasional deep sweeps; the default is what the suite runs)
SCALE = int(os.environ.get("M3_FUZZ_SCALE", "1"))
KNOTS = {3: np.array([-1.0, 0.0, 1.0]), 5: np.array([-2.0, -1.0, 0.0, 1.0, 2.0]), 7: np.arange(-3.0, 4.0)}


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    P = int(rng.integers(1, 60))
    E = int(rng.integers(1, max(2, min(6000, 90_000 // P))))
    kinds = rng.choice(["s3", "s5", "s7", "tf1", "none"], P, p=[0.15, 0.15, 0.3, 0.3, 0.1])
    xs = {p: KNOTS[int(k[1])] for p, k in enumerate(kinds) if k[0] == "s"}
    shape = rng.choice(["uniform", "ragged", "blocks"])
    occ = float(rng.uniform(0.05, 0.95))
    block_lists = [rng.random(P) < occ for _ in range(4)]
    events = []
    for e in range(E):
        if shape == "uniform":
            mask = np.ones(P, bool)
        elif shape == "ragged":
            mask = rng.random(P) < occ
            if rng.random() < 0.1:
                mask[:] = False
        else:  # runs of events that share a list (uniform tiles) next to ragged stretches
            blk = (e // 1500) % 5
            mask = block_lists[blk] if blk < 4 else rng.random(P) < occ
        ev = []
        for p in range(P):
            if not mask[p] or kinds[p] == "none":
                continue
            if p in xs:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0 + 0.01 * rng.standard_normal())))
        events.append(ev)
    n_samples = int(rng.integers(1, 4))
    nb = int(rng.choice([3, 17, 200, 1500]))
    n_norm = int(rng.integers(0, 5))
    stat = int(rng.integers(0, 5))
    return dict(rng=rng, P=P, E=E, xs=xs, events=events, n_samples=n_samples, nb=nb, n_norm=n_norm, stat=stat,
                update_w2=bool(rng.random() < 0.3), use_osc=bool(rng.random() < 0.7), shape=str(shape))


@pytest.mark.parametrize("seed", range(64 * SCALE))
def test_random_shape(fresh_engine, seed):
    _run_case(fresh_engine, _case(seed), seed)


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_random_shape_many_bins_and_samples(fresh_engine, seed):
    """the same sweep with the bin space of 2-D / atmospheric analyses: up to 24 samples, up to 9 600 bins (the
    replica count, the LDS / global form of the fill and the one- / two-kernel form of the bin likelihood all
    depend on these two numbers)"""
    c = _case(1000 + seed)
    c["n_samples"] = int([1, 2, 5, 24][seed % 4])
    c["nb"] = int([40, 400, 1100, 2600, 4800][(seed // 2) % 5] if c["n_samples"] < 5 else [40, 200, 400][seed % 3])
    _run_case(fresh_engine, c, 1000 + seed)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_random_shape_stored_in_the_order_of_a_key(fresh_engine, seed):
    """the same sweep with every upload chunk handed an order key (m3hip_monolith_order_key): the events' oscillation indices, a
    random key, or a coarse descending one -- the engine stores the events in another order (or refuses the key where it would
    inflate the layout), every per-event array and result keeps the oracle's numbering"""
    _run_case(fresh_engine, _case(2000 + seed), 2000 + seed, key_mode=("osc", "random", "coarse")[seed % 3])


def _run_case(fresh_engine, c, seed, key_mode=None):
    rng, P, E = c["rng"], c["P"], c["E"]
    cx, npts, ch = make_monolith(P, 7, c["xs"], c["events"])
    nPar = P + c["n_norm"]
    par = np.zeros(nPar)
    par[P:] = 1.0
    opar = par.copy()
    edges = np.linspace(0.0, 5.0, c["nb"] + 1)
    kin = rng.gamma(3.0, 0.45, E)
    kin[: min(E, 5)] = rng.choice(edges, min(E, 5))          # exactly on edges
    sample_id = rng.integers(0, c["n_samples"], E).astype(np.int32)
    nn = rng.integers(0, 3, E) if c["n_norm"] else np.zeros(E, np.int64)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, max(c["n_norm"], 1), int(norm_off[-1]))).astype(np.int16)
    osc = rng.random(40).astype(np.float32)
    osc[3] = 0.0
    osc_idx = rng.integers(-1, 40, E).astype(np.int32) if c["use_osc"] else None

    lib = ob.load("serial")
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    for _ in range(c["n_samples"]):
        b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc if c["use_osc"] else None, omono)
    osamp.set_test_statistic(c["stat"], c["update_w2"])

    # the engine takes the monolith in event chunks (every chunk but the last a multiple of 1024 events)
    chunks = [ch]
    if E > 2048 and seed % 2:
        cut = 1024 * int(rng.integers(1, E // 1024))
        chunks = [make_monolith(P, 7, c["xs"], c["events"][:cut])[2], make_monolith(P, 7, c["xs"], c["events"][cut:])[2]]
    if key_mode is not None:
        key = (osc_idx if key_mode == "osc" and osc_idx is not None else
               rng.integers(-1, 50, E).astype(np.int32) if key_mode != "coarse" else ((E - np.arange(E)) // 7).astype(np.int32))
        a0 = 0
        for chk in chunks:
            chk["order_key"] = key[a0:a0 + chk["n_events"]]
            a0 += chk["n_events"]
    # bricks of 1024, 2048 or 4096 events, by the seed (the engine reads the switch when the first chunk arrives)
    forced = os.environ.get("M3HIP_BRICK_GROUPS")
    os.environ["M3HIP_BRICK_GROUPS"] = forced or str((1, 2, 4)[seed % 3])
    # the events' {bin, osc, norm} indices as one packed word (the default when they fit) or as three arrays
    os.environ["M3HIP_PACKED_EVENTS"] = "0" if seed % 4 == 3 else "b" if seed % 4 == 1 else "1"  # arrays / word without norm codes / full word
    try:
        mono = SMonolith(fresh_engine, P, 7, cx, npts, chunks)
    finally:
        if forced is None:
            del os.environ["M3HIP_BRICK_GROUPS"]
    if key_mode is not None and E > 1:
        info = fresh_engine.info()
        assert info.n_keyed_chunks + info.n_key_refused >= 1 or np.all(np.diff(key.view(np.uint32).astype(np.int64)) >= 0), (seed, key_mode)
    mono.setSplinePointers(par)
    off = (np.arange(c["n_samples"] + 1) * c["nb"]).astype(np.int32)
    sh = SampleHandlerFD(fresh_engine, c["n_samples"], off, par, c["stat"], update_w2=c["update_w2"])
    bins = synth.find_global_bin_uniform_1d(edges, kin, sample_id, c["nb"])
    assert np.array_equal(bins, osamp.event_bins())
    sh.SetupEvents(0, bins, osc_idx, norm_off, norm_idx)
    if c["use_osc"]:
        sh.SetOscillationWeights(osc)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    data = np.round(osamp.mc * rng.uniform(0.8, 1.2, len(osamp.mc)))
    sh.AddData(data)
    osamp.add_data(data)
    tag = (seed, c["shape"], P, E, c["nb"], c["stat"])
    for lap in range(3):
        par[:P] = rng.uniform(-3.6, 3.6, P)
        if lap == 1 and P > 1:
            par[0], par[1] = -1.0, 2.0                      # exact knots
        par[P:] = 1 + 0.1 * rng.standard_normal(c["n_norm"])
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetSplineWeights(0, E).view(np.uint32), omono.total_weights.view(np.uint32)), tag
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), osamp.event_weights().view(np.uint32)), tag
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), tag
        assert np.allclose(sh.GetW2Array(), osamp.w2, rtol=RTOL_BIN, atol=0), tag
        want = osamp.get_likelihood()
        if np.isfinite(want):
            assert abs(llh - want) <= RTOL_LLH * max(abs(want), 1.0) or (c["stat"] == capi.kIceCube and abs(llh - want) <= 1e-6 * max(abs(want), 1.0)), (tag, llh, want)
        else:
            assert not np.isfinite(llh) or llh > 1e200, (tag, llh, want)
    # seam 2 on the same monolith
    mono.Evaluate()
    mono.SynchroniseMemTransfer()
    omono.evaluate()
    assert np.array_equal(mono.cpu_total_weights.view(np.uint32), omono.total_weights.view(np.uint32)), tag
    osamp.close()
    omono.close()
    os.environ.pop("M3HIP_PACKED_EVENTS", None)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_random_dynamic_binning(fresh_engine, seed):
    """device-side FindGlobalBin with random mixes of uniform 1-D / 2-D and non-uniform (box) samples, scale and
    offset shifts, cuts on binning and cut-only columns: bins integer-exact, fused LLH 1e-9, over several steps"""
    rng = np.random.default_rng(5000 + seed)
    E = int(rng.integers(300, 5000))
    P = int(rng.integers(1, 6))
    X7 = KNOTS[7]
    ragged = bool(rng.random() < 0.5)
    events = [[(p, "spl", natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))) for p in range(P) if not ragged or rng.random() < 0.6]
              for _ in range(E)]
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
    n_samples = int(rng.integers(1, 4))
    lib = ob.load("serial")
    b = ob.OracleBinning(lib)
    spec_bins, nb_tot = [], 0
    for s_ in range(n_samples):
        kind = rng.choice(["u1", "u2", "box"])
        if kind == "u1":
            e = np.sort(rng.uniform(0.0, 5.0, int(rng.integers(3, 40))))
            e[0], e[-1] = 0.0, 5.0
            b.add_uniform([e]); spec_bins.append([e]); nb_tot += len(e) - 1
        elif kind == "u2":
            ex, ey = np.linspace(0.0, 5.0, int(rng.integers(3, 30))), np.linspace(-1.0, 1.0, int(rng.integers(2, 12)))
            b.add_uniform([ex, ey]); spec_bins.append([ex, ey]); nb_tot += (len(ex) - 1) * (len(ey) - 1)
        else:
            nx, ny = int(rng.integers(2, 8)), int(rng.integers(2, 6))
            xs_ = np.sort(np.concatenate([[0.0, 5.0], rng.uniform(0.0, 5.0, nx - 1)]))
            boxes = []
            for i in range(nx):
                ys_ = np.sort(np.concatenate([[-1.0, 1.0], rng.uniform(-1.0, 1.0, ny - 1)]))
                boxes += [[[xs_[i], xs_[i + 1]], [ys_[j], ys_[j + 1]]] for j in range(ny) if rng.random() < 0.9]
            boxes = np.array(boxes, np.float64)
            assert b.add_nonuniform(boxes) == s_
            spec_bins.append(boxes); nb_tot += len(boxes)
    assert b.n_bins() == nb_tot
    sample_id = rng.integers(0, n_samples, E).astype(np.int32)
    kin = np.stack([rng.uniform(-0.3, 5.3, E), rng.uniform(-1.1, 1.1, E), rng.uniform(0.0, 10.0, E)], axis=1)
    n_sh = int(rng.integers(0, 4))
    par = np.zeros(P + 3)
    opar = par.copy()
    omono = ob.OracleMonolith(lib, P, 7, cx, npts, ch)
    omono.set_spline_pointers(opar)
    osamp = ob.OracleSample(lib, b, sample_id, kin, None, None, opar, None, None, omono)
    mono = SMonolith(fresh_engine, P, 7, cx, npts, [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, n_samples, None, par, capi.kPoisson, bin_edges=spec_bins)
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    assert np.array_equal(sh.GetEventBins(0, E), osamp.event_bins()), seed
    def add_shifts():
        for k in range(n_sh):
            smp, dim, kd, pi = int(rng.integers(-1, n_samples)), int(rng.integers(0, 2)), int(rng.integers(1, 4)), P + int(rng.integers(0, 3))
            if kd == 3:                      # x += theta * (another nominal column), e.g. a scale on part of the energy
                aux = int(rng.integers(0, 3))
                osamp.add_shift(smp, dim, 3 | (aux << 8), opar, pi)
                sh.AddKinShiftAux(smp, dim, aux, pi)
            else:
                osamp.add_shift(smp, dim, kd, opar, pi)
                sh.AddKinShift(smp, dim, kd, pi)

    def add_cuts():
        for k in range(int(rng.integers(0, 3))):
            smp, col = int(rng.integers(-1, n_samples)), int(rng.integers(0, 3))
            lo = float(rng.uniform(0.0, 2.0)) if col != 1 else float(rng.uniform(-1.0, 0.0))
            hi = lo + (float(rng.uniform(1.0, 8.0)) if col != 1 else float(rng.uniform(0.3, 1.5)))
            osamp.add_cut(smp, col, lo, hi)
            sh.AddKinCut(smp, col, lo, hi)

    for f in ((add_shifts, add_cuts) if seed % 2 else (add_cuts, add_shifts)):   # either registration order
        f()
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    for lap in range(4):
        par[:P] = rng.uniform(-3.2, 3.2, P)
        par[P:] = rng.normal(0.0, 0.08, 3)
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(sh.GetEventBins(0, E), osamp.event_bins()), (seed, lap)
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), (seed, lap)
        want = osamp.get_likelihood()
        assert abs(llh - want) <= RTOL_LLH * max(abs(want), 1.0), (seed, lap, llh, want)
    osamp.close()
    omono.close()


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_random_binned_spline_samples(fresh_engine, seed, monkeypatch):
    """BinnedSplineHandler samples: both M3::float_t builds, events unordered / ordered by spline bin (ELL or shared
    pointer lists), with and without functional-parameter re-binning (the double build runs it as a separate pass)."""
    from helpers import make_binned, make_binned_events
    from mach3_amd.engine import BinnedSplineHandler

    rng = np.random.default_rng(9000 + seed)
    float_t = np.float64 if seed % 2 else np.float32
    variant = "serial_f64" if seed % 2 else "serial"
    P, cells, E, n_norm = int(rng.integers(1, 12)), int(rng.integers(5, 300)), int(rng.integers(200, 6000)), int(rng.integers(0, 4))
    bs = make_binned(P, cells, rng, float_t=float_t)
    per_cell = bool(rng.random() < 0.6)
    ordered = bool(rng.random() < 0.5)
    w_off, w_idx = make_binned_events(E, P, cells, rng, per_cell_lists=per_cell, sort_by_cell=ordered)
    if rng.random() < 0.25:
        monkeypatch.setenv("M3HIP_BINNED_LISTS", "shared" if rng.random() < 0.5 else "ell")
    nPar = P + n_norm + 1
    par = np.zeros(nPar)
    par[P:P + n_norm] = 1.0
    opar = par.copy()
    edges = np.linspace(0.0, 5.0, int(rng.integers(4, 120)))
    nb = len(edges) - 1
    kin = rng.gamma(3.0, 0.4, E)
    sample_id = np.zeros(E, np.int32)
    nn = rng.integers(0, 3, E) if n_norm else np.zeros(E, np.int64)
    norm_off = np.concatenate([[0], np.cumsum(nn)]).astype(np.uint64)
    norm_idx = (P + rng.integers(0, max(n_norm, 1), int(norm_off[-1]))).astype(np.int16)
    osc_idx = rng.integers(-1, 30, E).astype(np.int32)
    osc = rng.random(30).astype(float_t)
    with_shift = bool(rng.random() < 0.5)

    lib = ob.load(variant)
    ob_binned = ob.OracleBinned(lib, bs)
    ob_binned.set_spline_pointers(opar)
    b = ob.OracleBinning(lib)
    b.add_uniform([edges])
    osamp = ob.OracleSample(lib, b, sample_id, kin, norm_off, norm_idx, opar, osc_idx, osc, None)
    osamp.set_binned(ob_binned, w_off, w_idx)
    osamp.set_test_statistic(capi.kBarlowBeeston)

    handler = BinnedSplineHandler(fresh_engine, E, bs)
    handler.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, 1, None, par, capi.kBarlowBeeston, bin_edges=[[edges]])
    sh.SetupKinematics(0, sample_id, kin)
    sh.ComputeBins()
    sh.SetupEvents(0, sh.GetEventBins(0, E), osc_idx, norm_off, norm_idx)
    handler.SetEventSplines(0, w_off, w_idx)
    sh.SetOscillationWeights(osc)
    if with_shift:
        osamp.add_shift(-1, 0, 1, opar, nPar - 1)
        sh.AddKinShift(-1, 0, 1, nPar - 1)
    sh.RecordEventWeights(True)
    sh.Reweight()
    osamp.reweight()
    sh.AddData(osamp.mc)
    osamp.add_data(osamp.mc)
    bits = np.uint64 if float_t == np.float64 else np.uint32
    for lap in range(4):
        par[:P] = rng.uniform(-3.4, 3.4, P)
        par[P:P + n_norm] = 1 + 0.05 * rng.standard_normal(n_norm)
        par[-1] = 0.05 * rng.standard_normal() if with_shift else 0.0
        opar[:] = par
        llh = sh.Step()
        osamp.reweight()
        assert np.array_equal(handler.GetWeights().view(bits), ob_binned.weights.view(bits)), (seed, lap)
        assert np.array_equal(sh.GetEventBins(0, E), osamp.event_bins()), (seed, lap)
        assert np.array_equal(sh.GetEventWeights(0, E, float_t).view(bits), osamp.event_weights().view(bits)), (seed, lap)
        assert np.allclose(sh.GetMCArray(), osamp.mc, rtol=RTOL_BIN, atol=0), (seed, lap)
        want = osamp.get_likelihood()
        assert abs(llh - want) <= RTOL_LLH * max(abs(want), 1.0), (seed, lap, llh, want)
    osamp.close()
    ob_binned.close()
