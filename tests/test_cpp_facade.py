"""The C++ facade (include/mach3_amd/: SMonolith, SMonolithGPU, SampleHandlerFD, BinningHandler)
driven by examples/mr2t2_dragrace.cpp: MasterSpline -> PrepareForGPU flattening -> Evaluate vs
TSpline3_red::Eval in double, DragRace, and a Metropolis chain in MR2T2::ProposeStep order."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "mr2t2_dragrace")


def test_facade_headers_compile():
    """Syntax-check the facade headers on their own (no GPU, no link)."""
    for h in ("Core.h", "SplineStructs.h", "SplineMonolith.h", "SampleHandlerFD.h", "BinnedSplineHandler.h", "SplineBuilders.h", "Fitters.h", "Sharding.h"):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                            "-x", "c++", os.path.join(ROOT, "include", "mach3_amd", h)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cpp_driver_runs_a_chain():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    r = subprocess.run([EXE, "20000", "16", "30", "400"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "agree with TSpline3_red::Eval" in out and "DragRace" in out and out.strip().endswith("OK")
    acc = float(out.split("acceptance ")[1].split(",")[0])
    assert 0.02 < acc < 0.98
    assert "MCMC + adaptive proposal" in out and "DelayedMR2T2" in out and "functional parameter" in out
    assert "reloaded monolith are bit-identical" in out
    assert "experiment weight slots" in out
    assert "LLH scan: 3 parameters x 9 points equal steps made by hand" in out


@pytest.mark.gpu
def test_likelihood_fit_calc_chi2_under_nelder_mead_returns_the_asimov_point():
    """examples/likelihood_fit_nm.cpp: LikelihoodFit::CalcChi2 (Fitters/LikelihoodFit.cpp:39-123: the minimisers' Reweight +
    GetLikelihood pair) over the GPU sample, minimised by a Nelder-Mead simplex from half a sigma away: chi2 = 0 at the Asimov
    point, mirroring at a bound, and the fit returns the point the data were made at to 1e-6 in every parameter."""
    exe = os.path.join(ROOT, "build", "likelihood_fit_nm")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "largest |fitted - true|" in r.stdout and r.stdout.strip().endswith("OK")


@pytest.mark.gpu
def test_cpp_facade_binned_double_build_and_box_bins():
    """examples/facade_check.cpp: BinnedSplineHandlerT<double>, a uniform + a non-uniform sample binned on the
    device, a selection cut, double oscillation weights -- all through the C++ facade, against a host
    evaluation in the same program."""
    exe = os.path.join(ROOT, "build", "facade_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-identical over 6 steps" in r.stdout and r.stdout.strip().endswith("OK")


def test_event_order_by_parameter_list():
    """SMonolith::EventOrderByParameterList groups events by the parameters they carry a response object for
    (flat one-knot splines included: they mark the event's mode even though PrepareForGPU drops them); stable
    within a group; events without responses last"""
    import ctypes as C
    import numpy as np
    so = os.path.join(ROOT, "build", "libm3facade_test.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    shim = C.CDLL(so)
    shim.facade_event_order.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    E, P = 500, 6
    is_tf1 = np.array([0, 0, 1, 0, 1, 0], np.int32)
    modes = [np.array([1, 1, 3, 0, 0, 0]), np.array([0, 1, 3, 1, 3, 1]), np.array([0, 0, 0, 0, 0, 0]), np.array([1, 0, 0, 1, 3, 0])]
    which = rng.integers(0, 4, E)
    kind = np.stack([modes[m] for m in which]).astype(np.int32)
    flat = (kind == 1) & (rng.random((E, P)) < 0.1)
    kind[flat] = 2                                              # a tenth of the splines are flat
    order = np.zeros(E, np.uint32)
    n = shim.facade_event_order(E, P, kind.ctypes.data, is_tf1.ctypes.data, order.ctypes.data)
    assert n == E and sorted(order.tolist()) == list(range(E))
    mode_of = which[order]
    # each mode is one contiguous block, in original order inside it; the empty mode comes last
    blocks = [m for i, m in enumerate(mode_of) if i == 0 or mode_of[i - 1] != m]
    assert len(blocks) == 4 and blocks[-1] == 2
    for m in range(4):
        ev = [int(e) for e in order if which[e] == m]
        assert ev == sorted(ev)


@pytest.mark.gpu
def test_reference_call_sites_of_SMonolithGPU_run_unchanged():
    """tests/cpp/seam2_callsites.cpp: the reference's own calls into class SMonolithGPU (Splines/SplineMonolith.cpp:82-83,
    271-303, 702-707, 854, 621-625) -- InitGPU_Segments / InitGPU_Vals through a null handler, `new SMonolithGPU()`,
    CopyToGPU_SplineMonolith with the reference's eleven arguments (vectors by value, no knot counts), RunGPU, the free
    SynchroniseSplines(), the clean-up calls -- compiled against the facade and run on a ragged mixed sample, uploaded in
    three slices; the weights that come back in the pinned cpu_total_weights are bit-identical to the serial oracle's."""
    import ctypes as C
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from mach3_amd import synth

    so = os.path.join(ROOT, "build", "libm3seam2_test.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    lib = C.CDLL(so)
    spec = synth.CONFIGS["c3"].scaled(2048 + 700, n_params=48)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, spec.n_events)
    omono, osamp, opar, _, _ = ob.build_oracle_from_synth(spec, 0, spec.n_events)
    laps = 3
    vals = np.zeros((laps, spec.n_params), np.float32)
    segs = np.zeros((laps, spec.n_params), np.int16)
    ref = np.zeros((laps, spec.n_events), np.float32)
    for lap in range(laps):
        opar[:] = synth.params(spec, lap + 1)
        omono.evaluate()
        vals[lap], segs[lap], ref[lap] = omono.param_values, omono.segments, omono.total_weights
    out = np.zeros((laps, spec.n_events), np.float32)
    err = C.create_string_buffer(512)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.seam2_callsites.restype = C.c_int
    rc = lib.seam2_callsites(C.c_uint(spec.n_events), C.c_short(spec.n_params), C.c_short(spec.n_knots), p(cx), p(ch["coeff_many"]),
                             C.c_uint(ch["total_nknots"]), p(ch["nknots_arr"]), p(ch["param_no"]), C.c_uint(ch["n_splines"]),
                             p(ch["coeff_tf1"]), p(ch["param_no_tf1"]), C.c_uint(ch["n_tf1"]), p(ch["npe"]), p(ch["npe_tf1"]),
                             C.c_int(laps), p(vals), p(segs), p(out), err, C.c_int(512))
    assert rc == 0, err.value.decode()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    osamp.close()
    omono.close()


@pytest.mark.gpu
def test_configs4_step_loop_example_runs():
    """examples/c5_step_loop.cpp at a small size: 1 040 parameters, adaptive proposal taking over part way, the host's
    O(P^2) work overlapped with the device step; the run reports its phases and ends with OK"""
    exe = os.path.join(ROOT, "build", "c5_step_loop")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    r = subprocess.run([exe, "200000", "1500", "100", "1200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert out.strip().endswith("OK") and "1024 spline parameters + 16 norm" in out
    acc = float(out.split("acceptance ")[1].split(",")[0])
    assert 0.05 < acc < 0.99
    assert "out of bounds 0" in out and "host work on the step's critical path" in out


@pytest.mark.gpu
def test_device_prior_and_throw_equal_the_host_loops():
    """SURVEY 8f row 2 at BASELINE configs[4]'s size: the prior term (ParameterHandlerBase::CalcLikelihood, ref
    Parameters/ParameterHandlerBase.cpp:816-841) and the correlated throw (CorrelateSteps, :726-745) computed on the device
    (AttachDevice -> m3hip_par_*) against the host loops of the same class, 1 040 parameters with a random SPD covariance, some
    flat priors and a fixed parameter, 40 steps of one chain walked twice; a throw matrix replaced while a device throw is in
    flight must be dropped.  Bar: 1e-12 relative (fp64 sums in another order)."""
    import ctypes as C
    import numpy as np
    subprocess.run(["make", "-C", ROOT, "build/libm3facade_test.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "build", "libm3facade_test.so"))
    lib.facade_par_device_check.argtypes = [C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
    for n, steps in ((1040, 40), (7, 12), (257, 20)):
        out = np.zeros(3)
        rc = lib.facade_par_device_check(n, 20260318 + n, steps, out.ctypes.data)
        assert rc == 0, (n, rc)
        assert out[2] == steps and out[0] <= 1e-12 and out[1] <= 1e-12, (n, out)


def _facade_lib():
    import ctypes as C
    subprocess.run(["make", "-C", ROOT, "build/libm3facade_test.so"], check=True, stdout=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "build", "libm3facade_test.so"))


def test_cpp_shard_ranges_equal_the_python_harness():
    """M3::ShardRanges (include/mach3_amd/Sharding.h) -- the rule a C++ host uses to give every rank its block of events --
    makes the cuts mach3_amd/dist.py:shard_ranges makes (the harness of bench.py and the tests): 20 random weight vectors
    (responses per event of sparse samples, incl. long stretches of empty events and ties on a rounding boundary), even splits,
    more ranks than 1024-event groups."""
    import ctypes as C
    import numpy as np
    from mach3_amd import dist
    lib = _facade_lib()
    lib.facade_shard_ranges.argtypes = [C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20260405)

    def cpp(n, ranks, w):
        cuts = np.zeros(2 * ranks, np.uint64)
        wp = None if w is None else np.ascontiguousarray(w, np.float64).ctypes.data
        assert lib.facade_shard_ranges(n, ranks, wp, cuts.ctypes.data) == 0
        return [(int(cuts[2 * r]), int(cuts[2 * r + 1])) for r in range(ranks)]

    for trial in range(20):
        n = int(rng.integers(1, 60)) * 1024 + int(rng.integers(0, 1024))
        ranks = int(rng.choice([1, 2, 3, 4, 8]))
        kind = trial % 4
        if kind == 0:
            w = rng.poisson(32, n).astype(np.float32)                       # config 3's shape
        elif kind == 1:
            w = (rng.random(n) < 0.3) * rng.integers(1, 40, n).astype(np.float32)   # most events empty
        elif kind == 2:
            w = np.repeat(rng.integers(0, 50, -(-n // 1024)), 1024)[:n].astype(np.float32)  # block averages, as bench.py counts them
        else:
            w = np.ones(n, np.float32)                                        # ties: targets fall exactly on events
        assert cpp(n, ranks, w) == dist.shard_ranges(n, ranks, weights=w), (trial, n, ranks)
        assert cpp(n, ranks, None) == dist.shard_ranges(n, ranks), (trial, n, ranks)
    for n, ranks in ((1, 4), (1024, 8), (5000, 8), (10_000_000, 8)):
        assert cpp(n, ranks, None) == dist.shard_ranges(n, ranks)
    assert lib.facade_shard_ranges(10, 0, None, np.zeros(2, np.uint64).ctypes.data) == -1


@pytest.mark.gpu
def test_two_parameter_handlers_attached_to_one_engine():
    """AttachDevice kept its state in the engine's single par block, so a second handler overwrote the first.
    Every attached handler now owns a slot (m3hip_par_create): two handlers with equal and with different numbers of
    parameters, both launches in flight together, priors equal to each handler's own host loop (1e-12) and the proposals
    (device throw) equal to the host-only twins' over 12 steps."""
    import ctypes as C
    import numpy as np
    lib = _facade_lib()
    lib.facade_two_handlers_one_engine.argtypes = [C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
    for na, nb in ((40, 40), (40, 131), (300, 17)):
        out = np.zeros(2)
        rc = lib.facade_two_handlers_one_engine(na, nb, 77 + na, 12, out.ctypes.data)
        assert rc == 0, (na, nb, rc)
        assert out[1] == 12 and out[0] <= 1e-12, (na, nb, out)


def _run_sharded(tmp, n_ranks, exchange, events=60000, params=48, steps=150, timeout=300, order_key=0):
    exe = os.path.join(ROOT, "build", "mr2t2_sharded")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(r), str(n_ranks), tmp, exchange, str(events), str(params), str(steps), "0", str(order_key)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n_ranks)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{n_ranks}-rank sharded chain timed out")
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = []
    for r in range(n_ranks):
        line = open(os.path.join(tmp, f"result_{r}.txt")).read().strip()
        kv = dict(tok.split("=", 1) for tok in line.split()[1:])
        res.append(kv)
    return res


@pytest.mark.gpu
def test_sharded_chain_through_the_facade_takes_the_same_decisions_on_every_rank():
    """examples/mr2t2_sharded.cpp: the multi-GPU seam from the C++ side -- M3::ShardRanges, SMonolith + SampleHandlerFD over a
    rank's own events, SampleHandlerFD::CreateExchange / OpenExchange (handles through files), the same seeded MR2T2 on every
    rank.  Two processes on the one card of the test box (the one-shot exchange; RCCL refuses two ranks on a device): the
    accept / reject sequence, the final likelihood and the final bins are IDENTICAL on both ranks (bit for bit: the exchange
    sums in rank order), and equal to the one-process chain over the whole sample -- same decisions, likelihood to 1e-9, bins to
    1e-12.  Reference: none (Manager/gpuUtils.cu:71, one device); caller loop Fitters/MR2T2.cpp:25-115."""
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
        one = _run_sharded(t1, 1, "none")[0]
        two = _run_sharded(t2, 2, "oneshot")
        b1 = np.fromfile(os.path.join(t1, "bins_0.bin"))
        b20, b21 = np.fromfile(os.path.join(t2, "bins_0.bin")), np.fromfile(os.path.join(t2, "bins_1.bin"))
    # ... and SampleHandlerFD::InitComm (RCCL, with the validator on) with the one rank a single card allows: the same chain again
    with tempfile.TemporaryDirectory() as t3:
        rccl = _run_sharded(t3, 1, "rccl")[0]
        b3 = np.fromfile(os.path.join(t3, "bins_0.bin"))
    assert rccl["acc"] == one["acc"] and float(rccl["logl"]) == pytest.approx(float(one["logl"]), rel=1e-12) and np.allclose(b3, b1, rtol=1e-12, atol=0)
    a, b = two
    assert a["events"] != b["events"] and a["events"].startswith("[0,") and b["events"].endswith(",60000)")
    assert a["acc"] == b["acc"] and a["logl"] == b["logl"] and a["accepted"] == b["accepted"]       # identical on both ranks
    assert np.array_equal(b20, b21)
    assert a["acc"] == one["acc"] and len(one["acc"]) == 150                                            # ... and the one-process chain's
    assert 5 < int(one["accepted"]) < 145
    assert float(a["logl"]) == pytest.approx(float(one["logl"]), rel=1e-9)
    assert np.allclose(b20, b1, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_chain_over_events_the_engine_stored_in_oscillation_order():
    """SMonolithGPU::next_event_order_key -> m3hip_monolith_order_key from the C++ side: the same seeded MR2T2 over the same
    sample, once as the MC is numbered and once with the engine storing its events in oscillation-bin order.  Nothing else
    changes in the caller (AddEvent in MC order): the same accept / reject sequence, likelihood to 1e-9, bins to 1e-12 (the same
    sums in another order).  Reference: the events alias the calculator's array in MC order, Samples/SampleHandlerFD.cpp:1108-1133."""
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as t1, tempfile.TemporaryDirectory() as t2:
        plain = _run_sharded(t1, 1, "none")[0]
        keyed = _run_sharded(t2, 1, "none", order_key=1)[0]
        b1, b2 = np.fromfile(os.path.join(t1, "bins_0.bin")), np.fromfile(os.path.join(t2, "bins_0.bin"))
    assert plain["keyed_chunks"] == "0" and keyed["keyed_chunks"] == "1" and float(keyed["osc_lines"]) < 0.5 * float(plain["osc_lines"])
    assert keyed["acc"] == plain["acc"] and len(plain["acc"]) == 150 and 5 < int(plain["accepted"]) < 145
    assert float(keyed["logl"]) == pytest.approx(float(plain["logl"]), rel=1e-9)
    assert np.allclose(b2, b1, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_two_live_sample_handlers_in_one_fitter():
    """examples/two_samples_one_fitter.cpp: the reference's step is `for samples: Reweight(); for samples: GetLikelihood()`
    (Fitters/MR2T2.cpp:66-76).  Two live handlers -- an SMonolith brick sample (float) and a BinnedSplineHandlerT<double> sample,
    each on its own engine handle and stream -- in ONE MR2T2 over one ParameterHandlerBase: every step's per-handler likelihood
    equals the handler stepped alone at the same parameters (1e-9), the chain accepts and rejects, and the interleaved pair
    takes less wall time per step than the two handlers one after the other (the launches and waits overlap)."""
    exe = os.path.join(ROOT, "build", "two_samples_one_fitter")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    r = subprocess.run([exe, "150000", "200000", "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PAIR")][0]
    kv = dict(tok.split("=") for tok in line.split()[1:])
    assert int(kv["in_bounds"]) >= 150 and 0 < int(kv["accepted"]) < 200
    assert float(kv["worst_rel"]) <= 1e-9
    assert float(kv["us_pair"]) < float(kv["us_A_alone"]) + float(kv["us_B_alone"])
    assert r.stdout.strip().endswith("OK")


def test_calc_norms_bins_matches_a_plain_restatement():
    """SampleHandlerFD::CalcNormsBins of the facade (which normalisation parameters apply to an event: ref
    Samples/SampleHandlerFD.cpp:667-731 with MatchCondition, Samples/SampleHandlerBase.h:196-202, and PassesSelection,
    Samples/SampleHandlerFD.cpp:2101-2130) against the rule written out in python, 400 random events x 12 random parameters:
    empty lists admit everything, oscillated NC events get no parameter, kinematic ranges are (lo, hi] and any one of them passes,
    the indices come out in the order of the parameter list."""
    import ctypes as C
    import numpy as np
    lib = _facade_lib()
    lib.facade_calc_norms_bins.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(42)
    n_par = 12
    modes_all, pdgs_all, targets_all = list(range(6)), [12, -12, 14, -14, 16], [12, 16, 1]
    pars = []
    for p in range(n_par):
        def pick(pool):
            return [] if rng.random() < 0.4 else sorted(rng.choice(pool, int(rng.integers(1, len(pool))), replace=False).tolist())
        kin = []
        if rng.random() < 0.5:
            col = int(rng.integers(0, 2))
            rs = []
            for _ in range(int(rng.integers(1, 3))):
                lo = float(rng.uniform(0, 4)); rs += [lo, lo + float(rng.uniform(0.2, 2.0))]
            kin = [float(col)] + rs
        pars.append(dict(index=40 + p, lists=[pick(modes_all), pick(pdgs_all), pick(pdgs_all), pick(targets_all)], kin=kin))
    index = np.array([p["index"] for p in pars], np.int32)
    off, values, kin_off, kinv = [0], [], [0], []
    for p in pars:
        for l in p["lists"]:
            values += l; off.append(len(values))
        kinv += p["kin"]; kin_off.append(len(kinv))
    off, values, kin_off, kinv = np.array(off, np.int32), np.array(values + [0], np.int32), np.array(kin_off, np.int32), np.array(kinv + [0.0])

    def rule(ev, kv):
        if ev["nc"] and ev["pdg"] != ev["pdg0"]:
            return []
        out = []
        for p in pars:
            m, g, g0, t = p["lists"]
            if (t and ev["target"] not in t) or (g and ev["pdg"] not in g) or (g0 and ev["pdg0"] not in g0) or (m and ev["mode"] not in m):
                continue
            if p["kin"]:
                x, r = kv[int(p["kin"][0])], p["kin"][1:]
                if not any(r[i] < x <= r[i + 1] for i in range(0, len(r), 2)):
                    continue
            out.append(p["index"])
        return out

    hits = 0
    for e in range(400):
        ev = dict(mode=int(rng.choice(modes_all)), target=int(rng.choice(targets_all)), pdg=int(rng.choice(pdgs_all)), pdg0=int(rng.choice(pdgs_all)), nc=bool(rng.random() < 0.2))
        if rng.random() < 0.6:
            ev["pdg0"] = ev["pdg"]
        kv = np.array([rng.uniform(0, 5), rng.uniform(0, 5)])
        if e % 50 == 0 and any(p["kin"] for p in pars):          # a value exactly on a range's upper edge: inside; on its lower edge: outside
            p = next(p for p in pars if p["kin"])
            kv[int(p["kin"][0])] = p["kin"][2] if e % 100 == 0 else p["kin"][1]
        out = np.zeros(n_par, np.int16)
        n = lib.facade_calc_norms_bins(n_par, index.ctypes.data, off.ctypes.data, values.ctypes.data, kin_off.ctypes.data, kinv.ctypes.data, ev["mode"], ev["target"],
                                       ev["pdg"], ev["pdg0"], int(ev["nc"]), 2, kv.ctypes.data, out.ctypes.data)
        exp = rule(ev, kv)
        assert n == len(exp) and out[:n].tolist() == exp, (e, ev, kv, out[:n].tolist(), exp)
        hits += n
    assert hits > 100          # (the rule admitted parameters in a fair share of the cases)


def test_binned_spline_index_event_fan_out():
    """BinnedSplineIndex::GetEventSplines (the facade's twin of BinnedSplineHandler::GetEventSplines, ref
    Splines/BinnedSplineHandler.cpp:832-884, and of the fan-out in SampleHandlerFD::SetSplinePointers, Samples/SampleHandlerFD.cpp:1196-1243)
    against the look-up written out in python: three spline axes with TAxis::FindBin's [lo, hi) bins, events outside any axis get
    nothing, a systematic contributes the spline of the event's bins when its mode list holds the event's spline mode and the spline is
    not flat, systematics in order; values on edges included."""
    import ctypes as C
    import numpy as np
    lib = _facade_lib()
    lib.facade_event_splines.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(7)
    axes = [np.array([0.0, 0.5, 1.0, 2.0, 5.0]), np.array([-1.0, 0.0, 0.3, 1.0]), np.array([-1e9, 1e9])]   # a "1-D" sample: one bin on the third axis
    nb = [len(a) - 1 for a in axes]
    cells = nb[0] * nb[1] * nb[2]
    systs, next_id = [], 0
    for s in range(7):
        modes = sorted(rng.choice(5, int(rng.integers(1, 4)), replace=False).tolist())
        ids = np.arange(next_id, next_id + len(modes) * cells, dtype=np.int32)
        next_id += len(ids)
        systs.append((modes, ids))
    isflat = (rng.random(next_id) < 0.2).astype(np.int8)
    n_ev = 600
    ev_mode = rng.integers(-1, 6, n_ev).astype(np.int32)
    ev_vars = np.stack([rng.uniform(-0.5, 5.5, n_ev), rng.uniform(-1.3, 1.3, n_ev), np.zeros(n_ev)], axis=1)
    ev_vars[::37, 0] = rng.choice(axes[0], len(ev_vars[::37]))       # exactly on edges
    ev_vars[5::41, 1] = rng.choice(axes[1], len(ev_vars[5::41]))

    def find(a, v):
        return int(np.searchsorted(a, v, side="right")) - 1

    exp_off, exp_idx = [0], []
    for i in range(n_ev):
        b = [find(axes[k], ev_vars[i, k]) for k in range(3)]
        if ev_mode[i] >= 0 and all(0 <= b[k] < nb[k] for k in range(3)):
            for modes, ids in systs:
                for im, m in enumerate(modes):
                    if m == ev_mode[i]:
                        sid = int(ids[((im * nb[0] + b[0]) * nb[1] + b[1]) * nb[2] + b[2]])
                        if not isflat[sid]:
                            exp_idx.append(sid)
        exp_off.append(len(exp_idx))
    n_edges = np.array([len(a) for a in axes], np.int32)
    edges = np.concatenate(axes)
    mode_off = np.cumsum([0] + [len(m) for m, _ in systs]).astype(np.int32)
    modes = np.concatenate([m for m, _ in systs]).astype(np.int32)
    id_off = np.cumsum([0] + [len(i) for _, i in systs]).astype(np.int32)
    ids = np.concatenate([i for _, i in systs]).astype(np.int32)
    out_off, out_idx = np.zeros(n_ev + 1, np.int32), np.zeros(len(exp_idx) + 8, np.int32)
    n = lib.facade_event_splines(n_edges.ctypes.data, edges.ctypes.data, len(systs), mode_off.ctypes.data, modes.ctypes.data, id_off.ctypes.data, ids.ctypes.data,
                                 len(isflat), isflat.ctypes.data, n_ev, ev_mode.ctypes.data, np.ascontiguousarray(ev_vars).ctypes.data, out_off.ctypes.data,
                                 out_idx.ctypes.data, len(out_idx))
    assert n == len(exp_idx) and out_off.tolist() == exp_off and out_idx[:n].tolist() == exp_idx
    assert len(exp_idx) > 300 and exp_off.count(0) < n_ev


def test_event_order_by_oscillation_bin():
    """SampleHandlerFD::EventOrderByOscillationBin: ascending oscillation index, stable inside a bin, events without an
    oscillation pointer (index -1) last"""
    import ctypes as C
    import numpy as np
    so = os.path.join(ROOT, "build", "libm3facade_test.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True)
    shim = C.CDLL(so)
    shim.facade_osc_order.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    E = 4000
    osc = rng.integers(-1, 300, E).astype(np.int32)
    order = np.zeros(E, np.uint32)
    assert shim.facade_osc_order(E, osc.ctypes.data, order.ctypes.data) == E
    assert sorted(order.tolist()) == list(range(E))
    key = osc[order].astype(np.int64)
    key[key < 0] = 1 << 40
    assert np.all(np.diff(key) >= 0)
    for b in (-1, 0, 17, 299):
        ev = order[osc[order] == b]
        assert np.all(np.diff(ev.astype(np.int64)) > 0)
