"""Committed fixtures (tests/golden/*.npz: reference-layout inputs + expected outputs written by
make_golden.py from the serial oracle).  CPU: the oracle reproduces them from the stored inputs
alone (no generator involved).  GPU: so does the HIP path through the C ABI."""
import glob
import os

import numpy as np
import pytest

import oracle_binding as ob
from mach3_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    g = dict(np.load(path))
    ch = {k[5:]: g[k] for k in g if k.startswith("mono_")}
    E = len(g["ev_bin_idx"])
    ch.update(n_events=E, n_splines=len(ch["param_no"]), n_tf1=len(ch["param_no_tf1"]), total_nknots=len(ch["coeff_many"]) // 4)
    return g, ch


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path):
    g, ch = load(path)
    lib = ob.load("serial")
    P, K = int(g["n_params"]), int(g["n_knots"])
    par = g["params"][0].copy()
    mono = ob.OracleMonolith(lib, P, K, g["coeff_x"], g["npts"], ch)
    mono.set_spline_pointers(par)
    b = ob.OracleBinning(lib)
    for _ in range(int(g["n_samples"])):
        b.add_uniform([g["bin_edges"]])
    s = ob.OracleSample(lib, b, g["ev_sample_id"], g["ev_erec"], g["ev_norm_off"], g["ev_norm_idx"], par, g["ev_osc_idx"],
                        g["osc_table"], mono)
    s.set_test_statistic(int(g["test_statistic"]))
    assert np.array_equal(s.event_bins(), g["ev_bin_idx"])
    for i, p in enumerate(g["params"]):
        par[:] = p
        s.reweight()
        if i == 0:
            s.add_data(g["data"])
        assert np.array_equal(mono.segments, g["exp_segments"][i])
        assert np.array_equal(mono.total_weights.view(np.uint32), g["exp_spline_w"][i].view(np.uint32))
        assert np.array_equal(s.event_weights().view(np.uint32), g["exp_event_w"][i].view(np.uint32))
        assert np.array_equal(s.mc, g["exp_mc"][i]) and np.array_equal(s.w2, g["exp_w2"][i])
        assert s.get_likelihood() == g["exp_llh"][i][0]
    s.close()
    mono.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(fresh_engine, path):
    from mach3_amd.engine import SMonolith, SampleHandlerFD

    g, ch = load(path)
    P, K, nS, nB = int(g["n_params"]), int(g["n_knots"]), int(g["n_samples"]), int(g["n_bins"])
    par = g["params"][0].copy()
    mono = SMonolith(fresh_engine, P, K, g["coeff_x"], g["npts"], [ch])
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(fresh_engine, nS, (np.arange(nS + 1) * nB).astype(np.int32), par, int(g["test_statistic"]))
    sh.SetupEvents(0, g["ev_bin_idx"], g["ev_osc_idx"], g["ev_norm_off"], g["ev_norm_idx"])
    sh.SetOscillationWeights(g["osc_table"])
    sh.RecordEventWeights(True)
    E = len(g["ev_bin_idx"])
    for i, p in enumerate(g["params"]):
        par[:] = p
        sh.Reweight()
        if i == 0:
            sh.AddData(g["data"])
        llh = sh.GetLikelihood()
        seg, _ = sh.GetSegments(P)
        assert np.array_equal(seg, g["exp_segments"][i])
        assert np.array_equal(sh.GetSplineWeights(0, E).view(np.uint32), g["exp_spline_w"][i].view(np.uint32))
        assert np.array_equal(sh.GetEventWeights(0, E).view(np.uint32), g["exp_event_w"][i].view(np.uint32))
        assert np.allclose(sh.GetMCArray(), g["exp_mc"][i], rtol=1e-12, atol=0)
        assert np.allclose(sh.GetW2Array(), g["exp_w2"][i], rtol=1e-12, atol=0)
        assert abs(llh - g["exp_llh"][i][0]) <= 1e-9 * max(abs(g["exp_llh"][i][0]), 1.0)
        for s in range(nS):
            assert abs(sh.GetSampleLikelihood(s) - g["exp_llh"][i][1 + s]) <= 1e-9 * max(abs(g["exp_llh"][i][1 + s]), 1.0)
