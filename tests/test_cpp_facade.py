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
    for h in ("Core.h", "SplineStructs.h", "SplineMonolith.h", "SampleHandlerFD.h", "BinnedSplineHandler.h", "SplineBuilders.h", "Fitters.h"):
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
