"""Host-side re-layout (reference flat arrays -> segment-major tiles): pure data movement,
checked on the CPU against the source arrays through the m3hip_debug_layout probe."""
import numpy as np
import pytest

from helpers import debug_layout, make_monolith, natural_cubic
from mach3_amd import synth

TE = 256


def check_uniform_tiles(n_params, K, npts, ch, lay):
    spl, tf1, meta, desc = lay["spl"], lay["tf1"], lay["meta"], lay["desc"]
    cm = ch["coeff_many"].reshape(-1, 4)
    ct = ch["coeff_tf1"].reshape(-1, 2)
    npe = ch["npe"].reshape(-1, 2)
    npt = ch["npe_tf1"].reshape(-1, 2)
    for t, d in enumerate(desc):
        spl_off, tf1_off, meta_off, js, jt, nev, kind, nplanes = [int(v) for v in d]
        assert kind == 0
        e0 = t * TE
        assert nev == min(TE, ch["n_events"] - e0)
        m = meta[meta_off: meta_off + 2 * js + jt]
        plane = 0
        for j in range(js):
            p = int(m[2 * j])
            assert int(m[2 * j + 1]) == plane
            nseg = int(npts[p]) - 1
            lanes = np.arange(nev)
            s = npe[e0 + lanes, 1] + j
            assert (ch["param_no"][s] == p).all()
            for k in range(nseg):
                got = spl[spl_off + (plane + k) * TE + lanes]
                exp = cm[ch["nknots_arr"][s].astype(np.int64) + k]
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
                pad = spl[spl_off + (plane + k) * TE + np.arange(nev, TE)]
                assert (pad == np.array([1, 0, 0, 0], np.float32)).all()
            plane += nseg
        assert plane == nplanes
        for j in range(jt):
            p = int(m[2 * js + j])
            lanes = np.arange(nev)
            s = npt[e0 + lanes, 1] + j
            assert (ch["param_no_tf1"][s] == p).all()
            got = tf1[tf1_off + j * TE + lanes]
            assert np.array_equal(got.view(np.uint32), ct[s].view(np.uint32))


@pytest.mark.parametrize("n_events,n_params", [(256, 8), (700, 5), (1, 3), (1024 + 17, 16)])
def test_dense_layout_roundtrip(n_events, n_params):
    spec = synth.CONFIGS["c2"].scaled(n_events, n_params=n_params)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, n_events)
    rc, err, lay = debug_layout(n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    assert lay["n_tiles"] == (n_events + TE - 1) // TE
    assert lay["spl"].shape[0] == lay["n_tiles"] * n_params * (spec.n_knots - 1) * TE
    check_uniform_tiles(n_params, spec.n_knots, npts, ch, lay)


def test_mixed_uniform_layout_with_different_knot_counts():
    # parameters with 3, 7 and 5 knots + two TF1 parameters; every event carries all of them
    xs = {0: np.array([-1.0, 0, 1]), 2: np.arange(-3.0, 4.0), 3: np.array([-2.0, -1, 0, 1, 2])}
    rng = np.random.default_rng(1)
    events = []
    for _ in range(300):
        ev = []
        for p in range(5):
            if p in xs:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(5, 7, xs, events)
    rc, err, lay = debug_layout(5, 7, npts, ch)
    assert rc == 0, err
    assert int(lay["desc"][0][7]) == 2 + 6 + 4
    check_uniform_tiles(5, 7, npts, ch, lay)


def test_layout_rejects_inconsistent_input():
    spec = synth.CONFIGS["c2"].scaled(300, n_params=4)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, 300)
    bad = dict(ch)
    bad["param_no"] = ch["param_no"].copy()
    bad["param_no"][5] = 9                      # parameter out of range
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "paramNo" in err
    bad = dict(ch)
    bad["npe"] = ch["npe"].copy()
    bad["npe"][-2] += 1                         # last event's run exceeds n_splines
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "exceeds" in err
    bad = dict(ch)
    bad["total_nknots"] = ch["total_nknots"] - 3  # knots exceed the coefficient array
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "total_nknots" in err
    npts2 = npts.copy()
    npts2[1] = -999                             # a spline on a parameter that has no knots
    rc, err, _ = debug_layout(4, 7, npts2, ch)
    assert rc != 0 and "knot count" in err
