"""SURVEY 8f row 3 (set-up ingestion): TSpline3_red coefficient builders.  The facade
(include/mach3_amd/SplineBuilders.h, product code) against the oracle's restatement of
TSpline3_red::SetFunc (Splines/SplineStructs.h:311-652), bit for bit, plus known answers; and the
facade's BinningHandler against the oracle's FindGlobalBin (integer-exact).  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "build", "libm3facade_test.so")
kTSpline3, kLinear, kMonotonic, kAkima, kKochanekBartels, kLinearFunc = range(6)


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", ROOT, "build/libm3facade_test.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(SHIM)
    lib.facade_build_spline.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.facade_root_coeffs.argtypes = [C.c_int] + [C.c_void_p] * 5
    lib.facade_spline_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    lib.facade_spline_eval.restype = C.c_double
    lib.facade_find_global_bin.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]
    return lib


def root_coeffs(olib, x, y):
    n = len(x)
    b, c, d = np.zeros(n), np.zeros(n), np.zeros(n)
    olib.m3o_root_tspline3_coeffs(n, x.ctypes.data, y.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data)
    return b, c, d


def oracle_build(olib, interp, x, y):
    b, c, d = root_coeffs(olib, x, y)
    bcd = np.ascontiguousarray(np.stack([b, c, d], axis=1))
    out = np.zeros((len(x), 5), np.float32)
    assert olib.m3o_build_tspline3_red(interp, len(x), x.ctypes.data, y.ctypes.data, bcd.ctypes.data, out.ctypes.data) == 0
    return out


def facade_build(shim, interp, x, y):
    out = np.zeros((len(x), 5), np.float32)
    assert shim.facade_build_spline(interp, len(x), x.ctypes.data, y.ctypes.data, out.ctypes.data) == 0
    return out


def test_root_tspline3_restatements_agree_and_interpolate(shim):
    olib = ob.load("serial")
    rng = np.random.default_rng(1)
    for n in (2, 3, 4, 7, 12):
        x = np.sort(rng.uniform(-3, 3, n)) if n > 2 else np.array([-1.0, 2.0])
        y = 1 + 0.2 * rng.standard_normal(n)
        b, c, d = root_coeffs(olib, x, y)
        fb, fc, fd = np.zeros(n), np.zeros(n), np.zeros(n)
        shim.facade_root_coeffs(n, x.ctypes.data, y.ctypes.data, fb.ctypes.data, fc.ctypes.data, fd.ctypes.data)
        assert np.array_equal(b, fb) and np.array_equal(c, fc) and np.array_equal(d, fd)
        # a cubic spline: each segment reaches the next knot, with continuous first derivative
        h = np.diff(x)
        assert np.allclose(y[:-1] + h * (b[:-1] + h * (c[:-1] + h * d[:-1])), y[1:], atol=1e-12)
        if n > 2:
            assert np.allclose(b[:-1] + h * (2 * c[:-1] + 3 * d[:-1] * h), b[1:], atol=1e-10)
    # four points on a cubic are reproduced exactly (the end conditions are exact for cubics)
    x = np.array([-2.0, -0.5, 1.0, 2.5, 3.0])
    f = lambda t: 1 + 0.3 * t - 0.2 * t ** 2 + 0.05 * t ** 3
    b, c, d = root_coeffs(olib, x, f(x))
    assert np.allclose(b, 0.3 - 0.4 * x + 0.15 * x ** 2, atol=1e-12) and np.allclose(d[:-1], 0.05, atol=1e-12)


def test_root_tspline3_restatement_satisfies_the_fmm_equations():
    """ROOT's TSpline3 without end-point options is Forsythe-Malcolm-Moler's `spline`: the unique C2 cubic
    interpolant whose third derivative in the first (last) segment equals the third divided difference of the
    first (last) four points.  The restatement is checked against these defining equations rather than against
    another implementation (it is NOT scipy's not-a-knot spline for n > 4; for n = 4 both reduce to the cubic
    through the four points)."""
    from scipy.interpolate import CubicSpline
    olib = ob.load("serial")
    rng = np.random.default_rng(3)

    def third_divided_difference(x, y):
        t = [float(v) for v in y]
        for k in range(1, 4):
            t = [(t[i + 1] - t[i]) / (x[i + k] - x[i]) for i in range(len(t) - 1)]
        return t[0]

    for n in (4, 5, 7, 12, 30):
        for _ in range(20):
            x = np.cumsum(rng.uniform(0.5, 1.5, n))
            y = 1 + 0.2 * rng.standard_normal(n)
            b, c, d = root_coeffs(olib, x, y)
            h = np.diff(x)
            assert np.allclose(y[:-1] + h * (b[:-1] + h * (c[:-1] + h * d[:-1])), y[1:], atol=1e-12)        # interpolates
            assert np.allclose(b[:-1] + h * (2 * c[:-1] + 3 * d[:-1] * h), b[1:], atol=1e-11)                 # C1
            assert np.allclose(2 * c[:-2] + 6 * d[:-2] * h[:-1], 2 * c[1:-1], atol=1e-11)                     # C2
            assert abs(d[0] - third_divided_difference(x[:4], y[:4])) < 1e-12                                 # FMM end conditions
            assert abs(d[n - 2] - third_divided_difference(x[-4:], y[-4:])) < 1e-12
            if n == 4:
                cs = CubicSpline(x, y, bc_type="not-a-knot")
                assert np.allclose(b[:-1], cs.c[2], atol=1e-12) and np.allclose(d[:-1], cs.c[0], atol=1e-12)


@pytest.mark.parametrize("interp", [kTSpline3, kLinear, kMonotonic, kAkima, kKochanekBartels, kLinearFunc])
def test_facade_builders_match_oracle_bitwise(shim, interp):
    olib = ob.load("serial")
    rng = np.random.default_rng(100 + interp)
    for trial in range(200):
        n = int(rng.integers(3, 10))
        x = np.sort(rng.choice(np.arange(-40, 41), n, replace=False)) / 8.0 if trial % 2 else np.sort(rng.uniform(-3, 3, n))
        y = 1 + 0.3 * rng.standard_normal(n)
        if trial % 5 == 0:
            y[rng.integers(0, n)] = y[rng.integers(0, n)]          # flat secants / repeated values
        if trial % 7 == 0:
            y = 1 + 0.1 * x                                         # collinear: the "keep linear segments" branch
        o = oracle_build(olib, interp, x, y)
        f = facade_build(shim, interp, x, y)
        assert np.array_equal(o.view(np.uint32), f.view(np.uint32)), (interp, trial, o, f)
    if interp == kMonotonic:  # two knots: plain linear interpolation (SplineStructs.h:472-481)
        x, y = np.array([-1.0, 1.0]), np.array([0.8, 1.3])
        assert np.array_equal(oracle_build(olib, interp, x, y), facade_build(shim, interp, x, y))


def test_builder_known_answers(shim):
    x = np.array([-3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0])
    y = np.array([0.7, 0.8, 0.95, 1.0, 1.2, 1.1, 1.4])
    lin = facade_build(shim, kLinear, x, y)
    assert np.allclose(lin[:-1, 2], np.diff(y), rtol=1e-6) and (lin[:, 3:] == 0).all() and lin[-1, 2] == lin[-2, 2]
    for interp in (kTSpline3, kMonotonic, kAkima, kKochanekBartels):
        s = facade_build(shim, interp, x, y)
        assert np.array_equal(s[:, 0], x.astype(np.float32)) and np.array_equal(s[:, 1], y.astype(np.float32))
        # every segment interpolates its right knot
        end = s[:-1, 1] + s[:-1, 2] + s[:-1, 3] + s[:-1, 4]
        assert np.allclose(end, y[1:], atol=3e-6), interp
        xs, ys, bcd = s[:, 0].copy(), s[:, 1].copy(), np.ascontiguousarray(s[:, 2:])
        assert shim.facade_spline_eval(7, xs.ctypes.data, ys.ctypes.data, bcd.ctypes.data, -0.5) == pytest.approx(
            ys[2] + 0.5 * (bcd[2, 0] + 0.5 * (bcd[2, 1] + 0.5 * bcd[2, 2])), rel=1e-12)
    mono = facade_build(shim, kMonotonic, x, y)
    # local maximum at x=1 and minimum at x=2: zero tangents there, and no overshoot inside the segments
    assert mono[4, 2] == 0 and mono[5, 2] == 0
    t = np.linspace(0, 1, 50)
    for i in range(6):
        v = mono[i, 1] + t * (mono[i, 2] + t * (mono[i, 3] + t * mono[i, 4]))
        assert v.min() >= min(y[i], y[i + 1]) - 1e-6 and v.max() <= max(y[i], y[i + 1]) + 1e-6
    kb = facade_build(shim, kKochanekBartels, x, y)   # Catmull-Rom: interior tangent = mean of the secants
    assert kb[3, 2] == pytest.approx((0.05 + 0.2) / 2, rel=1e-6)
    assert shim.facade_build_spline(kAkima, 2, x.ctypes.data, y.ctypes.data, np.zeros(10, np.float32).ctypes.data) == -1


def test_facade_binning_handler_matches_oracle(shim):
    """the facade's host FindGlobalBin (product code) is integer-exact against the oracle, with the
    nominal-bin cache set to the true nominal bin, a neighbour, a far bin and -1"""
    olib = ob.load("serial")
    rng = np.random.default_rng(9)
    e1 = np.array([0.0, 0.5, 1.0, 1.25, 2.0, 3.5, 5.0])
    ex, ey = np.linspace(0.0, 4.0, 11), np.array([-1.0, -0.2, 0.0, 0.3, 1.0, 2.5])
    b = ob.OracleBinning(olib)
    b.add_uniform([e1])
    b.add_uniform([ex, ey])
    n_dim = np.array([1, 2], np.int32)
    n_edges = np.array([len(e1), len(ex), len(ey)], np.int32)
    edges = np.concatenate([e1, ex, ey])
    for _ in range(3000):
        s = int(rng.integers(0, 2))
        kin = np.array([rng.uniform(-0.5, 5.5), rng.uniform(-1.3, 2.8)])
        if rng.random() < 0.1:
            kin[0] = rng.choice(e1 if s == 0 else ex)
        ndim = 1 + s
        true_nom = [b.find_nominal_bin(s, d, kin[d]) for d in range(ndim)]
        mode = rng.integers(0, 4)
        nmax = [len(e1) - 2] if s == 0 else [len(ex) - 2, len(ey) - 2]
        if mode == 0:
            nom = true_nom
        elif mode == 1:
            nom = [int(np.clip(t + rng.choice([-1, 1]), 0, m)) if t >= 0 else -1 for t, m in zip(true_nom, nmax)]
        elif mode == 2:
            nom = [int(rng.integers(0, m + 1)) for m in nmax]
        else:
            nom = [-1] * ndim
        nom_a = np.array(nom + [0] * (2 - ndim), np.int32)
        exp = b.find_global_bin(s, kin[:ndim], nom_a[:ndim])
        got = shim.facade_find_global_bin(2, n_dim.ctypes.data, n_edges.ctypes.data, edges.ctypes.data, s, kin.ctypes.data, nom_a.ctypes.data)
        assert got == exp, (s, kin, nom, got, exp)


def test_facade_nonuniform_binning_matches_oracle(shim):
    """non-uniform (box) bins through the 10x10 mega-bin grid: facade == oracle, including points on box
    edges (the reference's (lo,hi] boxes looked up through a [lo,hi) grid)"""
    shim.facade_find_nonuniform_bin.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    olib = ob.load("serial")
    # an irregular tiling of [0,4] x [0,3]
    boxes = np.array([[[0, 1], [0, 1]], [[1, 2.5], [0, 1]], [[2.5, 4], [0, 1.5]], [[0, 2.5], [1, 3]], [[2.5, 4], [1.5, 3]]], np.float64)
    b = ob.OracleBinning(olib)
    assert b.add_nonuniform(boxes) == 0
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(-0.2, 4.2, 4000), rng.uniform(-0.2, 3.2, 4000)], axis=1)
    pts[:300, 0] = rng.choice([0, 1, 2.5, 4, 0.4, 0.8, 1.2, 1.6, 2.0, 2.4, 2.8], 300)   # box and grid edges
    pts[300:600, 1] = rng.choice([0, 1, 1.5, 3, 0.3, 0.6, 0.9, 1.2, 2.1], 300)
    hits = set()
    for k in pts:
        nom = np.array([b.find_nominal_bin(0, 0, k[0]), b.find_nominal_bin(0, 1, k[1])], np.int32)
        exp = b.find_global_bin(0, k, nom)
        got = shim.facade_find_nonuniform_bin(len(boxes), 2, boxes.ctypes.data, np.ascontiguousarray(k).ctypes.data, nom.ctypes.data)
        assert got == exp, (k, got, exp)
        hits.add(got)
    assert hits == {-1, 0, 1, 2, 3, 4}
