"""Synthetic generator: seeded, chunk-independent, reference-layout invariants."""
import numpy as np

from mach3_amd import synth


def test_chunking_independent():
    spec = synth.CONFIGS["c3"].scaled(3000)
    whole = synth.monolith_chunk(spec, 0, 3000)
    a = synth.monolith_chunk(spec, 0, 1024)
    b = synth.monolith_chunk(spec, 1024, 3000)
    assert a["n_splines"] + b["n_splines"] == whole["n_splines"]
    assert np.array_equal(np.concatenate([a["coeff_many"], b["coeff_many"]]), whole["coeff_many"])
    assert np.array_equal(np.concatenate([a["param_no_tf1"], b["param_no_tf1"]]), whole["param_no_tf1"])
    assert np.array_equal(np.concatenate([a["npe"][0::2], b["npe"][0::2]]), whole["npe"][0::2])
    assert b["npe"][1] == 0 and b["nknots_arr"][0] == 0  # chunk-local offsets
    ev_w = synth.events_chunk(spec, 0, 3000)
    ev_b = synth.events_chunk(spec, 1024, 3000)
    assert np.array_equal(ev_w["erec"][1024:], ev_b["erec"])
    assert np.array_equal(ev_w["bin_idx"][1024:], ev_b["bin_idx"])
    off = int(ev_w["norm_off"][1024])
    assert np.array_equal(ev_w["norm_idx"][off:], ev_b["norm_idx"])


def test_reference_layout_invariants():
    spec = synth.CONFIGS["c3"].scaled(2000)
    ch = synth.monolith_chunk(spec, 0, 2000)
    npe = ch["npe"].reshape(-1, 2)
    assert (np.cumsum(npe[:, 0]) - npe[:, 0] == npe[:, 1]).all()     # SplineMonolith.cpp:209-215
    assert (ch["nknots_arr"] == np.arange(ch["n_splines"]) * spec.n_knots).all()
    assert (ch["param_no"] % 2 == 0).all() and (ch["param_no_tf1"] % 2 == 1).all()  # mixed: odd = TF1
    for e in (0, 7, 1999):                                           # ascending parameters inside an event
        s, c = int(npe[e, 1]), int(npe[e, 0])
        assert (np.diff(ch["param_no"][s:s + c]) > 0).all()
    occ = (ch["n_splines"] + ch["n_tf1"]) / (2000 * spec.n_params)
    assert abs(occ - spec.occupancy) < 0.01
    cm = ch["coeff_many"].reshape(-1, spec.n_knots, 4)
    assert (cm[:, -1, 1:] == 0).all()                                # last knot carries {y,0,0,0}
    assert (cm[:, :, 0] >= 0.05).all()
    # natural cubic: evaluating segment k at dx = 1 reproduces y[k+1]
    y1 = cm[:, :-1, 0] + cm[:, :-1, 1] + cm[:, :-1, 2] + cm[:, :-1, 3]
    assert np.allclose(y1, cm[:, 1:, 0], atol=2e-6)


def test_params_and_tables():
    spec = synth.CONFIGS["c2"].scaled(10)
    p0 = synth.params(spec, 0)
    assert (p0[:spec.n_params] == 0).all() and (p0[spec.n_params:] == 1).all()
    p1, p1b, p2 = synth.params(spec, 1), synth.params(spec, 1), synth.params(spec, 2)
    assert np.array_equal(p1, p1b) and not np.array_equal(p1, p2)
    assert abs(p1[:spec.n_params].std() - 0.3) < 0.08
    t = synth.osc_table(spec)
    assert t.min() >= 0 and t.max() < 1 and len(t) == spec.osc_n
    cx, npts = synth.knots(spec)
    assert (cx.reshape(spec.n_params, -1)[0] == np.arange(-3, 4)).all() and (npts == 7).all()
