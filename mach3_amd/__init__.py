"""mach3_amd -- MI355X (gfx950) per-step likelihood engine for MaCh3.

Scope: the one hot path SplineBase::Evaluate -> SampleHandlerFD::Reweight ->
GetLikelihood.  The product is libm3hip.so (hand-written HIP kernels
behind the C ABI of include/m3hip.h) plus the C++ facade in include/mach3_amd/; this
package is the ctypes harness used by tests/ and bench.py."""
from .capi import (M3HipError, kPoisson, kBarlowBeeston, kIceCube, kPearson, kDembinskiAbdelmotteleb,
                   LARGE_LOGL, LOW_MC_BOUND, UNDER_OVERFLOW_BIN)

__all__ = ["M3HipError", "kPoisson", "kBarlowBeeston", "kIceCube", "kPearson", "kDembinskiAbdelmotteleb",
           "LARGE_LOGL", "LOW_MC_BOUND", "UNDER_OVERFLOW_BIN"]
