"""the ROOT-free step driver (include/mach3_amd/Fitters.h: ParameterHandlerBase,
MR2T2, DelayedMR2T2, adaptive proposal) on a toy target with a known posterior.  Host-only C++;
no GPU.  The reference has no tests for these classes either (chain-level validation only)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "build", "libm3facade_test.so")

V = np.array([[1.0, 0.6, 0.0], [0.6, 1.0, 0.0], [0.0, 0.0, 0.25]])
MU, SIG = np.array([0.5, -0.2, 0.1]), np.array([0.3, 0.5, 0.2])
PREC = np.linalg.inv(V) + np.diag(1 / SIG ** 2)
POST_COV = np.linalg.inv(PREC)
POST_MEAN = POST_COV @ (MU / SIG ** 2)


@pytest.fixture(scope="module")
def shim():
    subprocess.run(["make", "-C", ROOT, "build/libm3facade_test.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(SHIM)
    lib.facade_toy_chain.argtypes = [C.c_int, C.c_int, C.c_ulonglong, C.c_void_p]
    return lib


def run(shim, kind, n, seed):
    out = np.zeros(11)
    assert shim.facade_toy_chain(kind, n, seed, out.ctypes.data) == 0
    return dict(acc=out[0], delayed=out[1], oob=out[2], mean=out[3:6], var=out[6:9], cov01=out[9], n_reweight=out[10])


@pytest.mark.parametrize("kind,name", [(0, "MR2T2"), (1, "DelayedMR2T2"), (2, "MR2T2+adaptive")])
def test_chain_reproduces_the_posterior(shim, kind, name):
    n = 60000
    r = run(shim, kind, n, 777 + kind)
    sd = np.sqrt(np.diag(POST_COV))
    # means within a few Monte-Carlo standard errors (effective sample size ~ n / 30)
    assert np.all(np.abs(r["mean"] - POST_MEAN) < 5 * sd / np.sqrt(n * 0.8 / 30)), (name, r["mean"], POST_MEAN)
    # the delayed scheme (second try centred on the rejected proposal, DRAM-style acceptance without the
    # proposal-density ratio) is followed as the reference implements it; it over-disperses mildly
    rtol = 0.30 if kind == 1 else 0.15
    assert np.allclose(r["var"], np.diag(POST_COV), rtol=rtol), (name, r["var"], np.diag(POST_COV))
    assert r["cov01"] == pytest.approx(POST_COV[0, 1], rel=0.35, abs=0.004)
    assert 0.02 < r["acc"] < 0.75, (name, r["acc"])   # the prior-shaped proposal is wide for this posterior
    if kind == 0:
        # out-of-bounds proposals are rejected WITHOUT a Reweight() (ref: Fitters/MR2T2.cpp:51-76)
        assert r["oob"] > 0 and r["n_reweight"] == n + 1 - r["oob"]
    if kind == 1:
        assert r["delayed"] > 100       # second-stage (shrunken) proposals do get accepted


def test_delayed_rejection_raises_acceptance(shim):
    a = run(shim, 0, 20000, 5)["acc"]
    b = run(shim, 1, 20000, 5)["acc"]
    assert b > 1.5 * a


def test_adaption_repairs_a_poor_proposal(shim):
    # kind 2 starts from a throw matrix of 0.01 * identity; without adaption that chain random-walks
    # with tiny steps.  After adaption the moments are right (previous test) and the acceptance settles
    # near the usual Metropolis range instead of ~1.
    r = run(shim, 2, 40000, 11)
    assert r["acc"] < 0.7


def test_duplicate_sample_titles_are_refused(shim):
    """FitterBase::AddSampleHandler throws when a sub-sample title is registered twice (FitterBase.cpp:262-279)"""
    assert shim.facade_duplicate_title_refused() == 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_parameter_handler_step_semantics(shim, seed):
    """fixed parameters, circular bounds, flips, flat priors, bounds, random start
    (ref: Parameters/ParameterHandlerBase.cpp:591-626, 652-780, 816-867, 960-1080)"""
    shim.facade_parameter_handler_checks.argtypes = [C.c_ulonglong]
    assert shim.facade_parameter_handler_checks(seed) == 0


def test_annealing_follows_the_reference_formula(shim):
    """MR2T2::AcceptanceProbability with AnnealTemp (Fitters/MR2T2.cpp:109-110): the likelihood difference is divided
    by exp(-step / T), which goes from 1 towards 0 -- uphill moves die out and the chain settles at the minimum, so its
    late spread is far below the posterior's and the acceptance rate falls below the plain chain's"""
    plain, ann = np.zeros(16), np.zeros(16)
    assert shim.facade_toy_chain(0, 40000, 11, plain.ctypes.data) == 0
    assert shim.facade_toy_chain(3, 40000, 11, ann.ctypes.data) == 0
    assert ann[0] < 0.6 * plain[0]                 # acceptance
    assert (ann[6:9] < 0.5 * plain[6:9]).all()     # variances of the kept part of the chain


@pytest.mark.parametrize("n", [5, 100, 300, 1040])
def test_throw_matrix_cholesky(shim, n):
    """SetThrowMatrix (ref: Parameters/ParameterHandlerBase.cpp:1140-1200): the panel factorisation reproduces the matrix, does
    not depend on the size of its thread team (bit for bit), leaves the upper triangle empty, and a singular matrix is made
    decomposable by a diagonal shift as M3::MakeMatrixPosDef does (Parameters/ParameterHandlerUtils.h:512-545)"""
    shim.facade_cholesky_check.argtypes = [C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
    out = np.zeros(3)
    assert shim.facade_cholesky_check(n, 99 + n, 4, out.ctypes.data) == 0
    assert out[0] < 1e-13 and out[1] == 1.0 and out[2] == 1.0, out


@pytest.mark.gpu
@pytest.mark.parametrize("kind,name", [(10, "MR2T2"), (12, "MR2T2+adaptive"), (11, "DelayedMR2T2")])
def test_chain_with_the_parameter_handler_on_the_device(shim, kind, name):
    """the same toy chains with `ParameterHandlerBase::AttachDevice`: every step's prior term and correlated throw come from the
    device (m3hip_par_*) through MCMCBase::EvaluateProposal -- launch beside the sample's Reweight, collect behind its
    GetLikelihood, host fallback for out-of-bounds proposals, throw matrix re-uploaded by the adaptive proposal"""
    n = 40000
    r = run(shim, kind, n, 777 + kind)
    sd = np.sqrt(np.diag(POST_COV))
    assert np.all(np.abs(r["mean"] - POST_MEAN) < 5 * sd / np.sqrt(n * 0.8 / 30)), (name, r["mean"], POST_MEAN)
    assert np.allclose(r["var"], np.diag(POST_COV), rtol=0.30 if kind == 11 else 0.15), (name, r["var"], np.diag(POST_COV))
    assert 0.02 < r["acc"] < 0.75, (name, r["acc"])
    if kind == 10:
        assert r["oob"] > 0 and r["n_reweight"] == n + 1 - r["oob"]
        # ... and the device chain walks the host chain's path: same seeds, same decisions (the sums differ by rounding only)
        h = run(shim, 0, n, 777 + kind)
        assert abs(r["acc"] - h["acc"]) < 0.02 and np.allclose(r["mean"], h["mean"], atol=0.02)
