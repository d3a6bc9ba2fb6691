"""Test helpers: hand-built monoliths in the reference's flat layout, natural-cubic
coefficients, and the debug-layout probe."""
from __future__ import annotations

import ctypes as C

import numpy as np

from mach3_amd import capi


def natural_cubic(x, y):
    """{b,c,d} per knot of the natural cubic spline through (x,y); last knot gets zeros.
    float64 in, float32 {y,b,c,d}[n] out (cast as SplineMonolith.cpp:673-677 does)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    n = len(x)
    h = np.diff(x)
    A = np.zeros((n, n))
    r = np.zeros(n)
    A[0, 0] = A[-1, -1] = 1.0
    for i in range(1, n - 1):
        A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2 * (h[i - 1] + h[i]), h[i]
        r[i] = 6 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
    M = np.linalg.solve(A, r)
    out = np.zeros((n, 4), np.float32)
    out[:, 0] = y
    for i in range(n - 1):
        out[i, 1] = (y[i + 1] - y[i]) / h[i] - h[i] * (2 * M[i] + M[i + 1]) / 6
        out[i, 2] = M[i] / 2
        out[i, 3] = (M[i + 1] - M[i]) / (6 * h[i])
    return out


def make_monolith(n_params, max_knots, knots_per_param, events):
    """events: list (one per event) of lists of (param, 'spl', ybcd[n][4]) or (param, 'tf1', (a, b)),
    in the order the reference would visit them (ascending parameter number).
    knots_per_param: dict param -> x array (TSpline3 parameters only).
    Returns (coeff_x, npts, chunk dict) in the reference layout (SplineMonolith.cpp:53-250)."""
    coeff_x = np.full(n_params * max_knots, -999.0, np.float32)
    npts = np.full(n_params, -999, np.int16)
    for p, xs in knots_per_param.items():
        coeff_x[p * max_knots: p * max_knots + len(xs)] = np.asarray(xs, np.float32)
        # GetSplineCoeff_SepMany initialises the tail of x_tmp to 1.0 (SplineMonolith.cpp:645-651)
        coeff_x[p * max_knots + len(xs): (p + 1) * max_knots] = 1.0
        npts[p] = len(xs)
    cm, nk, pn, ct, pt = [], [], [], [], []
    npe, npe_t = [], []
    knot = 0
    for ev in events:
        s0, t0 = len(pn), len(pt)
        for p, kind, c in ev:
            if kind == "spl":
                c = np.asarray(c, np.float32).reshape(-1, 4)
                assert len(c) == npts[p]
                cm.append(c.reshape(-1))
                nk.append(knot)
                knot += len(c)
                pn.append(p)
            else:
                ct.append(np.asarray(c, np.float32))
                pt.append(p)
        npe += [len(pn) - s0, s0]
        npe_t += [len(pt) - t0, t0]
    ch = dict(
        n_events=len(events), n_splines=len(pn), n_tf1=len(pt), total_nknots=knot,
        coeff_many=np.concatenate(cm).astype(np.float32) if cm else np.zeros(0, np.float32),
        nknots_arr=np.asarray(nk, np.uint32), param_no=np.asarray(pn, np.int16),
        coeff_tf1=np.concatenate(ct).astype(np.float32) if ct else np.zeros(0, np.float32),
        param_no_tf1=np.asarray(pt, np.int16),
        npe=np.asarray(npe, np.uint32), npe_tf1=np.asarray(npe_t, np.uint32),
    )
    return coeff_x, npts, ch


def debug_layout(n_params, max_knots, npts, ch):
    """Run the engine's host-side re-layout only.  Returns (rc, err, dict of stores)."""
    lib = capi.load_m3hip()
    sizes = np.zeros(8, np.uint64)
    args = [n_params, max_knots, capi._ptr(npts, np.int16), ch["n_events"], capi._ptr(ch["coeff_many"], np.float32),
            ch["total_nknots"], capi._ptr(ch["nknots_arr"], np.uint32), capi._ptr(ch["param_no"], np.int16),
            ch["n_splines"], capi._ptr(ch["coeff_tf1"], np.float32), capi._ptr(ch["param_no_tf1"], np.int16),
            ch["n_tf1"], capi._ptr(ch["npe"], np.uint32), capi._ptr(ch["npe_tf1"], np.uint32)]
    rc = lib.m3hip_debug_layout(*args, sizes.ctypes.data, None, None, None, None, None, None)
    if rc != 0:
        return rc, lib.m3hip_last_error(None).decode(), None
    nt, ns, ntf, nm, nbr, nw, dict_off, n_dict = [int(v) for v in sizes]
    spl = np.zeros(max(1, ns) * 4, np.float32)
    tf1 = np.zeros(max(1, ntf) * 2, np.float32)
    meta = np.zeros(max(1, nm), np.int32)
    desc = np.zeros((max(1, nt), 8), np.uint64)
    bdesc = np.zeros((max(1, nbr), 8), np.uint64)
    words = np.zeros(max(1, nw), np.uint32)
    rc = lib.m3hip_debug_layout(*args, sizes.ctypes.data, spl.ctypes.data, tf1.ctypes.data, meta.ctypes.data,
                                desc.ctypes.data, bdesc.ctypes.data, words.ctypes.data)
    return rc, "", dict(n_tiles=nt, spl=spl.reshape(-1, 4), tf1=tf1.reshape(-1, 2), meta=meta, desc=desc[:nt],
                        n_bricks=nbr, bdesc=bdesc[:nbr], words=words[:nw],
                        tf1_dict=words[dict_off:nw].view(np.uint16)[:n_dict])


def make_binned(n_params, n_cells, rng, max_knots=7, flat_frac=0.2, negative_frac=0.1, float_t=np.float32):
    """Synthetic BinnedSplineHandler members (ref: Splines/BinnedSplineHandler.h:116-140): one
    bin-spline per (cell, parameter); a fraction is flat (no coefficients, weight stays 1), some dip
    below zero (exercising the weight < 0 -> 0 clamp, BinnedSplineHandler.cpp:337).  Parameters have
    different knot counts.  float_t = float64 gives the default (non-low-memory) build's tables, with
    knots that are not representable in float."""
    knots = {p: np.linspace(-3, 3, [7, 5, 3][p % 3]) for p in range(n_params)}
    if float_t == np.float64:
        knots = {p: xs + 0.01 * np.sqrt(2.0) * (1 + np.arange(len(xs))) for p, xs in knots.items()}
    xpts = np.full((n_params, max_knots), -999.0, float_t)
    npts = np.zeros(n_params, np.int16)
    for p, xs in knots.items():
        xpts[p, :len(xs)] = xs
        npts[p] = len(xs)
    coeffindexvec, usv, uniq, xcoeff, many = [], [], [], [], []
    ci = 0
    for cell in range(n_cells):
        for p in range(n_params):
            i = cell * n_params + p
            coeffindexvec.append(ci)
            usv.append(p)
            if rng.random() < flat_frac:
                continue  # flat: CoeffIndex does not advance (BinnedSplineHandler.cpp:1023-1036)
            xs = knots[p]
            y = 1 + 0.15 * rng.standard_normal(len(xs))
            if rng.random() < negative_frac:
                y[rng.integers(0, len(xs))] = -0.4
            c = natural_cubic(xs, y)
            uniq.append(i)
            xcoeff.append(np.asarray(xs, float_t))
            many.append(c.reshape(-1))
            ci += len(xs)
    return dict(n_params=n_params, max_knots=max_knots, monolith_size=n_cells * n_params,
                xpts=np.ascontiguousarray(xpts.reshape(-1)), npts=npts,
                coeffindexvec=np.asarray(coeffindexvec, np.int32), uniquesplinevec=np.asarray(usv, np.int32),
                uniquecoeffindices=np.asarray(uniq, np.int32),
                xcoeff=np.concatenate(xcoeff).astype(float_t), manycoeff=np.concatenate(many).astype(float_t))


def make_binned_events(n_events, n_params, n_cells, rng, per_cell_lists=False, sort_by_cell=False):
    """per event: its cell's bin-splines for a subset of the parameters, ascending.  per_cell_lists: the subset
    belongs to the cell (events of one spline bin carry identical pointer lists, as in the reference where the
    list follows from sample / channel / mode / kinematic bin); sort_by_cell: events ordered by cell."""
    cells = rng.integers(0, n_cells, n_events)
    if sort_by_cell:
        cells = np.sort(cells)
    cell_mask = rng.random((n_cells, n_params)) < 0.6 if per_cell_lists else None
    w_off, w_idx = [0], []
    for e in range(n_events):
        cell = int(cells[e])
        mask = cell_mask[cell] if per_cell_lists else rng.random(n_params) < 0.6
        for p in range(n_params):
            if mask[p]:
                w_idx.append(cell * n_params + p)
        w_off.append(len(w_idx))
    return np.asarray(w_off, np.uint64), np.asarray(w_idx, np.int32)


class DeviceArray:
    """n elements of device memory that are NOT the engine's: what a GPU oscillation calculator owns.  Through the HIP runtime
    libm3hip.so already brought into the process (ctypes resolves the same soname), so that no second runtime is loaded."""
    _hip = None

    def __init__(self, n, dtype=np.float32):
        if DeviceArray._hip is None:
            capi.load_m3hip()
            hip = C.CDLL("libamdhip64.so.7")
            hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            hip.hipFree.argtypes = [C.c_void_p]
            DeviceArray._hip = hip
        self.n, self.dtype = int(n), np.dtype(dtype)
        p = C.c_void_p()
        rc = self._hip.hipMalloc(C.byref(p), self.n * self.dtype.itemsize)
        assert rc == 0 and p.value, f"hipMalloc: {rc}"
        self.ptr = p.value

    def set(self, host):
        """synchronous refill (hipMemcpy returns once the copy has completed: ordered before any later reweight)"""
        host = np.ascontiguousarray(host, self.dtype)
        assert len(host) == self.n
        rc = self._hip.hipMemcpy(self.ptr, host.ctypes.data, host.nbytes, 1)
        assert rc == 0, f"hipMemcpy: {rc}"

    def free(self):
        if self.ptr:
            self._hip.hipFree(self.ptr)
            self.ptr = None
