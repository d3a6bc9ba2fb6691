"""Seeded synthetic far-detector MC in the reference's flat layout.

Thin numpy front end of libm3synth.so.  The named configurations are BASELINE.json's
`configs`; `scaled()` gives event-count-reduced versions of the same distributions for
parity tests.  Event data are a pure function of (seed, event id), so any event range can
be generated on any rank."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace

import numpy as np

from . import capi


@dataclass(frozen=True)
class SynthSpec:
    name: str
    n_events: int
    n_params: int
    n_knots: int = 7
    mixed: bool = False          # odd parameters are TF1
    occupancy: float = 1.0
    n_norm: int = 16
    max_norm_per_event: int = 3
    n_samples: int = 1
    n_bins: int = 200            # per sample
    bin_lo: float = 0.0
    bin_hi: float = 5.0
    osc_n: int = 4000            # 4 channels x 1000 energy bins
    nc_fraction: float = 0.1
    seed: int = 20260313
    test_statistic: int = capi.kPoisson
    osc_sorted: bool = False     # MC registered in oscillation-bin order (event e points at entry e * osc_n // n_events)

    @property
    def n_par(self) -> int:
        """length of the global parameter vector: spline parameters, then norm parameters"""
        return self.n_params + self.n_norm

    @property
    def total_bins(self) -> int:
        return self.n_bins * self.n_samples

    def scaled(self, n_events: int, **kw) -> "SynthSpec":
        return replace(self, n_events=int(n_events), name=f"{self.name}@{int(n_events)}", **kw)

    def cconfig(self) -> capi.SynthConfig:
        return capi.SynthConfig(self.seed, self.n_params, self.n_knots, int(self.mixed), self.n_norm,
                                self.max_norm_per_event, self.n_samples, self.n_bins, self.osc_n,
                                self.occupancy, self.bin_lo, self.bin_hi, self.nc_fraction,
                                self.n_events if self.osc_sorted else 0)


# BASELINE.json configs[0..4]
CONFIGS = {
    "c1": SynthSpec("c1", 100_000, 32, seed=20260314),
    "c2": SynthSpec("c2", 1_000_000, 128, seed=20260315),
    "c3": SynthSpec("c3", 10_000_000, 512, mixed=True, occupancy=1 / 16, seed=20260316,
                    test_statistic=capi.kBarlowBeeston),
    "c4": SynthSpec("c4", 40_000_000, 512, mixed=True, occupancy=1 / 16, n_samples=4, seed=20260317,
                    test_statistic=capi.kBarlowBeeston),
    "c5": SynthSpec("c5", 100_000_000, 1024, mixed=True, occupancy=1 / 64, osc_n=4000 + 200_000, seed=20260318,
                    test_statistic=capi.kBarlowBeeston),
}


def knots(spec: SynthSpec):
    """coeff_x[P*K] and FastSplineInfo.nPts[P]"""
    lib = capi.load_m3synth()
    cx = np.empty(spec.n_params * spec.n_knots, np.float32)
    npts = np.empty(spec.n_params, np.int16)
    cfg = spec.cconfig()
    lib.m3synth_knots(C.byref(cfg), cx.ctypes.data, npts.ctypes.data)
    return cx, npts


def count(spec: SynthSpec, e0: int, e1: int):
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    ns, nt = C.c_uint64(0), C.c_uint64(0)
    lib.m3synth_count(C.byref(cfg), e0, e1, C.byref(ns), C.byref(nt))
    return ns.value, nt.value


def monolith_chunk(spec: SynthSpec, e0: int, e1: int) -> dict:
    """Reference-layout monolith arrays of events [e0, e1) (chunk-local offsets)."""
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    n = e1 - e0
    ns, nt = count(spec, e0, e1)
    K = spec.n_knots
    out = dict(
        n_events=n, n_splines=ns, n_tf1=nt, total_nknots=ns * K,
        coeff_many=np.empty(ns * K * 4, np.float32),
        nknots_arr=np.empty(ns, np.uint32),
        param_no=np.empty(ns, np.int16),
        coeff_tf1=np.empty(nt * 2, np.float32),
        param_no_tf1=np.empty(nt, np.int16),
        npe=np.empty(2 * n, np.uint32),
        npe_tf1=np.empty(2 * n, np.uint32),
    )
    if ns * K * 4 >= 2 ** 32:
        raise ValueError("chunk too large for the reference's unsigned offsets; use smaller chunks")
    lib.m3synth_monolith(C.byref(cfg), e0, e1, out["coeff_many"].ctypes.data, out["nknots_arr"].ctypes.data,
                         out["param_no"].ctypes.data, out["coeff_tf1"].ctypes.data,
                         out["param_no_tf1"].ctypes.data, out["npe"].ctypes.data, out["npe_tf1"].ctypes.data)
    return out


def bin_edges(spec: SynthSpec) -> np.ndarray:
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    e = np.empty(spec.n_bins + 1, np.float64)
    lib.m3synth_bin_edges(C.byref(cfg), e.ctypes.data)
    return e


def find_global_bin_uniform_1d(edges: np.ndarray, x: np.ndarray, sample_id: np.ndarray, n_bins: int) -> np.ndarray:
    """Global bin index for static 1-D uniform binning.

    Same result as BinningHandler::FindGlobalBin with NomBin from FindNominalBin
    (ref: Samples/BinningHandler.cpp:257-307): bins are [lo, hi), values outside
    [edges[0], edges[-1]) give UnderOverFlowBin (-1); GlobalOffset = sample * n_bins."""
    b = np.searchsorted(edges, x, side="right").astype(np.int64) - 1  # std::upper_bound - 1
    out = (x < edges[0]) | (x >= edges[-1])
    g = b + sample_id.astype(np.int64) * n_bins
    g[out] = capi.UNDER_OVERFLOW_BIN
    return g.astype(np.int32)


def events_chunk(spec: SynthSpec, e0: int, e1: int) -> dict:
    """Sample-side per-event data of events [e0, e1) (+ the global bin index)."""
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    n = e1 - e0
    out = dict(
        sample_id=np.empty(n, np.int32), erec=np.empty(n, np.float64), osc_idx=np.empty(n, np.int32),
        norm_off=np.empty(n + 1, np.uint64), norm_idx=np.empty(max(1, n * spec.max_norm_per_event), np.int16),
    )
    lib.m3synth_events(C.byref(cfg), e0, e1, out["sample_id"].ctypes.data, out["erec"].ctypes.data,
                       out["osc_idx"].ctypes.data, out["norm_off"].ctypes.data, out["norm_idx"].ctypes.data)
    out["norm_idx"] = out["norm_idx"][: int(out["norm_off"][-1])].copy() if n else out["norm_idx"][:0]
    out["bin_idx"] = find_global_bin_uniform_1d(bin_edges(spec), out["erec"], out["sample_id"], spec.n_bins)
    return out


def osc_table(spec: SynthSpec) -> np.ndarray:
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    t = np.empty(spec.osc_n, np.float32)
    lib.m3synth_osc_table(C.byref(cfg), t.ctypes.data)
    return t


def params(spec: SynthSpec, lap: int) -> np.ndarray:
    """Global parameter vector of lap `lap` (0 = nominal)."""
    lib = capi.load_m3synth()
    cfg = spec.cconfig()
    p = np.empty(spec.n_par, np.float64)
    lib.m3synth_params(C.byref(cfg), lap, p.ctypes.data)
    return p


def sample_bin_offsets(spec: SynthSpec) -> np.ndarray:
    return (np.arange(spec.n_samples + 1, dtype=np.int64) * spec.n_bins).astype(np.int32)
