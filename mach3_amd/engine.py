"""Python mirror of the reference's operator interface for the per-step path, on top of the
C ABI (include/m3hip.h).  Same names and argument meaning as the reference:

  SMonolith          Splines/SplineMonolith.h:11-149   Evaluate / SynchroniseMemTransfer /
                                                       setSplinePointers / cpu_total_weights
  SampleHandlerFD    Samples/SampleHandlerFD.h:21-413  Reweight / GetLikelihood /
                                                       GetSampleLikelihood / GetMCArray / ...

This is the test and benchmark harness; the production host side is the C++ facade in
include/mach3_amd/.  Nothing here computes on the CPU: if libm3hip.so or a HIP device is
missing, construction raises M3HipError."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import M3HipError, _ptr


class Engine:
    """Owns one m3hip handle (one device, one stream)."""

    def __init__(self, device_id: int = 0):
        self.lib = capi.load_m3hip()
        h = C.c_void_p()
        rc = self.lib.m3hip_create(device_id, C.byref(h))
        if rc != 0:
            raise M3HipError(rc, self.lib.m3hip_last_error(None).decode())
        self.h = h

    def _chk(self, rc: int):
        if rc != 0:
            raise M3HipError(rc, self.lib.m3hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.m3hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- monolith upload (CopyToGPU_SplineMonolith) ----
    def monolith_begin(self, n_params, max_knots, coeff_x, npts):
        self._chk(self.lib.m3hip_monolith_begin(self.h, n_params, max_knots, _ptr(coeff_x, np.float32), _ptr(npts, np.int16)))

    def monolith_append(self, ch: dict):
        key = ch.get("order_key")  # optional: the engine may store this chunk's events in the key's order (m3hip_monolith_order_key)
        if key is not None:
            key = np.ascontiguousarray(key, np.int32)
            self._chk(self.lib.m3hip_monolith_order_key(self.h, _ptr(key, np.int32), int(key.size)))
        self._chk(self.lib.m3hip_monolith_append(
            self.h, ch["n_events"], _ptr(ch["coeff_many"], np.float32), ch["total_nknots"],
            _ptr(ch["nknots_arr"], np.uint32), _ptr(ch["param_no"], np.int16), ch["n_splines"],
            _ptr(ch["coeff_tf1"], np.float32), _ptr(ch["param_no_tf1"], np.int16), ch["n_tf1"],
            _ptr(ch["npe"], np.uint32), _ptr(ch["npe_tf1"], np.uint32)))

    def monolith_expect_events(self, n_events: int):
        """optional hint before the first append: the events this device will hold (picks the brick size)"""
        if hasattr(self.lib, "m3hip_monolith_expect_events"):  # absent only in older builds loaded through M3HIP_LIB
            self._chk(self.lib.m3hip_monolith_expect_events(self.h, int(n_events)))

    def monolith_end(self):
        self._chk(self.lib.m3hip_monolith_end(self.h))

    def info(self) -> capi.Info:
        out = capi.Info()
        self._chk(self.lib.m3hip_get_info(self.h, C.byref(out)))
        return out

    def mem_info(self):
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.lib.m3hip_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def sync(self):
        self._chk(self.lib.m3hip_sync(self.h))

    # ---- timing of the dominant kernel (hipEvents on the engine's stream) ----
    def timing_enable(self, on=True):
        self._chk(self.lib.m3hip_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._chk(self.lib.m3hip_timing_reset(self.h))

    def timing_read(self, which=0):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._chk(self.lib.m3hip_timing_read(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- multi-GPU ----
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
        rc = self.lib.m3hip_comm_get_unique_id(buf)
        if rc != 0:
            raise M3HipError(rc, self.lib.m3hip_last_error(None).decode())
        return buf.raw

    def set_deterministic(self, on=True):
        self._chk(self.lib.m3hip_set_deterministic(self.h, int(on)))

    def comm_available(self) -> bool:
        return bool(self.lib.m3hip_comm_available())

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        assert len(unique_id) == capi.UNIQUE_ID_BYTES
        self._chk(self.lib.m3hip_comm_init(self.h, C.c_char_p(unique_id), rank, n_ranks))

    def comm_destroy(self):
        self._chk(self.lib.m3hip_comm_destroy(self.h))

    def xch_create(self, rank: int, n_ranks: int) -> bytes:
        """one-shot exchange: allocate this rank's buffer, return its IPC handle (to be all-gathered by the caller)"""
        buf = C.create_string_buffer(capi.XCH_HANDLE_BYTES)
        self._chk(self.lib.m3hip_xch_create(self.h, rank, n_ranks, buf))
        return buf.raw

    def xch_open(self, handles):
        """handles of all ranks in rank order; switches the engine's exchange to the one-shot form"""
        blob = b"".join(handles)
        assert len(blob) % capi.XCH_HANDLE_BYTES == 0
        self._chk(self.lib.m3hip_xch_open(self.h, C.c_char_p(blob)))

    # ---- the parameter handler's prior and correlated throw on the device (m3hip_par_*); one slot per handler
    def par_create(self) -> int:
        slot = C.c_int32(-1)
        self._chk(self.lib.m3hip_par_create(self.h, C.byref(slot)))
        return slot.value

    def par_destroy(self, slot: int):
        self._chk(self.lib.m3hip_par_destroy(self.h, slot))

    def par_upload(self, slot: int, which: int, matrix):
        m = np.ascontiguousarray(matrix, np.float64)
        assert m.ndim == 2 and m.shape[0] == m.shape[1]
        self._chk(self.lib.m3hip_par_upload(self.h, slot, which, m.ctypes.data, m.shape[0]))

    def par_launch(self, slot: int, d, z):
        d = np.ascontiguousarray(d, np.float64)
        z = np.ascontiguousarray(z, np.float64)
        self._chk(self.lib.m3hip_par_launch(self.h, slot, d.ctypes.data, z.ctypes.data, len(d)))

    def par_result(self, slot: int, n: int):
        prior = C.c_double()
        throw = np.empty(n, np.float64)
        self._chk(self.lib.m3hip_par_result(self.h, slot, C.byref(prior), throw.ctypes.data, n))
        return prior.value, throw

    def xch_set_timeout(self, seconds: float):
        self._chk(self.lib.m3hip_xch_set_timeout(self.h, C.c_double(seconds)))

    def xch_destroy(self):
        self._chk(self.lib.m3hip_xch_destroy(self.h))

    def comm_set_validate(self, on=True):
        self._chk(self.lib.m3hip_comm_set_validate(self.h, int(on)))

    def comm_last_validation(self):
        rel, n = C.c_double(0), C.c_uint64(0)
        self._chk(self.lib.m3hip_comm_last_validation(self.h, C.byref(rel), C.byref(n)))
        return rel.value, n.value


class SMonolith:
    """Event-by-event spline monolith evaluated on the GPU.

    Mirrors SMonolith (ref: Splines/SplineMonolith.h:11-149): `setSplinePointers`,
    `Evaluate`, `SynchroniseMemTransfer`, `cpu_total_weights`, `GetNParams`.  The
    constructor takes the flattened arrays PrepareForGPU would have produced (one dict per
    event chunk, see synth.monolith_chunk) instead of a MasterSpline of ROOT objects."""

    def __init__(self, engine: Engine, n_params, max_knots, coeff_x, npts, chunks, expect_events=0):
        self.eng = engine
        self.nParams = int(n_params)
        self._max_knots = int(max_knots)
        self.coeff_x = np.ascontiguousarray(coeff_x, np.float32)
        self.npts = np.ascontiguousarray(npts, np.int16)
        engine.monolith_expect_events(expect_events)
        engine.monolith_begin(n_params, max_knots, self.coeff_x, self.npts)
        for ch in chunks:
            engine.monolith_append(ch)
        engine.monolith_end()
        self.NEvents = int(engine.info().n_events)
        # FastSplineInfo state (ref: Splines/SplineStructs.h:21-44)
        self.CurrSegment = np.zeros(n_params, np.int16)
        self.SplineSegments = np.zeros(n_params, np.int16)          # SplineMonolith.cpp:91
        self.ParamValues = np.full(n_params, -999.0, np.float32)    # SplineMonolith.cpp:92
        self.cpu_total_weights = np.zeros(self.NEvents, np.float32)
        self._par = None
        self._idx = np.arange(n_params)

    def GetName(self):
        return "SplineMonolith"

    def GetNParams(self):
        return self.nParams

    def setSplinePointers(self, par_vector: np.ndarray, index=None):
        """`par_vector` plays ParameterHandlerBase::_fPropVal; `index[i]` says which entry
        drives spline parameter i.  Values are read at Evaluate() time, as through the
        reference's stored `const double*`."""
        self._par = par_vector
        self._idx = np.arange(self.nParams) if index is None else np.asarray(index, np.int64)
        gi = np.ascontiguousarray(self._idx, np.int32)
        self.eng._chk(self.eng.lib.m3hip_set_spline_param_map(self.eng.h, _ptr(gi, np.int32), self.nParams))

    def FindSplineSegment(self):
        """Host half of Evaluate(): SplineBase::FindSplineSegment (ref: Splines/SplineBase.cpp:44-109).
        O(P) work; stays on the host exactly as in the reference's GPU build."""
        x = self.coeff_x.reshape(self.nParams, self._max_knots)
        for i in range(self.nParams):
            n = int(self.npts[i])
            xvar = np.float32(self._par[self._idx[i]])
            self.ParamValues[i] = xvar
            if n <= 0:
                continue
            xa = x[i]
            prev = int(self.CurrSegment[i])
            seg, khigh = 0, n - 1
            if xvar <= xa[0]:
                seg = 0
            elif xvar >= xa[n - 1]:
                seg = khigh
            elif xa[prev + 1] > xvar and xvar >= xa[prev]:
                seg = prev
            else:
                while khigh - seg > 1:
                    khalf = (seg + khigh) // 2
                    if xvar > xa[khalf]:
                        seg = khalf
                    else:
                        khigh = khalf
            if seg >= n - 1 and n > 1:
                seg = n - 2
            self.CurrSegment[i] = seg
            self.SplineSegments[i] = seg

    def Evaluate(self):
        """ref: Splines/SplineMonolith.cpp:695-708 (GPU branch): segments on the host, then
        RunGPU_SplineMonolith (asynchronous)."""
        self.FindSplineSegment()
        self.eng._chk(self.eng.lib.m3hip_eval_splines(self.eng.h, _ptr(self.ParamValues, np.float32),
                                                      _ptr(self.SplineSegments, np.int16),
                                                      _ptr(self.cpu_total_weights, np.float32)))

    def SynchroniseMemTransfer(self):
        self.eng.sync()

    def retPointer(self, event: int):
        return self.cpu_total_weights[event:event + 1]

    # ---- the names pyMaCh3 gives the same operations (ref: python/splines.cpp:142-236, class EventSplineMonolith)
    def evaluate(self):
        self.Evaluate()

    def sync_mem_transfer(self):
        self.SynchroniseMemTransfer()

    def get_event_weight(self, event: int) -> float:
        return float(self.cpu_total_weights[event])

    def set_param_value_array(self, array: np.ndarray):
        """the array's entries become the spline parameters, read at evaluate() time (the binding stores pointers
        into the numpy buffer, python/splines.cpp:97-131)"""
        assert array.dtype == np.float64 and array.ndim == 1 and len(array) == self.nParams
        self.setSplinePointers(array)


class BinnedSplineHandler:
    """Bin-by-bin splines on the GPU: the per-step part of BinnedSplineHandler
    (ref: Splines/BinnedSplineHandler.cpp:295-341).  `bs` holds the handler's flat members
    (coeffindexvec, uniquesplinevec, uniquecoeffindices, xcoeff, manycoeff, xpts, npts).  The dtype of
    xcoeff selects M3::float_t: float32 (low-memory build) or float64 (the reference's default build,
    Manager/Core.h:27-51), which also switches the sample's weight product to double."""

    def __init__(self, engine: Engine, n_events: int, bs: dict):
        self.eng = engine
        self.bs = bs
        self.nParams = int(bs["n_params"])
        self.MonolithSize = int(bs["monolith_size"])
        self.float_t = np.float64 if bs["xcoeff"].dtype == np.float64 else np.float32
        up = engine.lib.m3hip_binned_upload_f64 if self.float_t == np.float64 else engine.lib.m3hip_binned_upload
        engine._chk(up(
            engine.h, n_events, self.nParams, bs["max_knots"], _ptr(bs["xpts"], self.float_t), _ptr(bs["npts"], np.int16),
            self.MonolithSize, _ptr(bs["coeffindexvec"], np.int32), _ptr(bs["uniquesplinevec"], np.int32),
            len(bs["uniquecoeffindices"]), _ptr(bs["uniquecoeffindices"], np.int32), len(bs["xcoeff"]),
            _ptr(bs["xcoeff"], self.float_t), _ptr(bs["manycoeff"], self.float_t)))

    def GetName(self):
        return "BinnedSplineHandler"

    def GetNParams(self):
        return self.nParams

    def setSplinePointers(self, par_vector, index=None):
        idx = np.arange(self.nParams) if index is None else np.asarray(index)
        gi = np.ascontiguousarray(idx, np.int32)
        self.eng._chk(self.eng.lib.m3hip_set_spline_param_map(self.eng.h, _ptr(gi, np.int32), self.nParams))

    def SetEventSplines(self, event_offset, w_off, w_idx):
        """per event the indices into weightvec_Monolith, in total_weight_pointers order
        (ref: Samples/SampleHandlerFD.cpp:1196-1243)"""
        wo = np.ascontiguousarray(w_off, np.uint64)
        wi = np.ascontiguousarray(w_idx, np.int32)
        self.eng._chk(self.eng.lib.m3hip_binned_upload_events(self.eng.h, event_offset, len(wo) - 1, _ptr(wo, np.uint64), _ptr(wi, np.int32)))

    def GetWeights(self):
        out = np.empty(self.MonolithSize, self.float_t)
        get = self.eng.lib.m3hip_get_binned_weights_f64 if self.float_t == np.float64 else self.eng.lib.m3hip_get_binned_weights
        self.eng._chk(get(self.eng.h, _ptr(out, self.float_t), self.MonolithSize))
        return out


class SampleHandlerFD:
    """The fused sample operator: SampleHandlerBase::Reweight / GetLikelihood on the GPU
    (ref: Samples/SampleHandlerFD.cpp:316-463, 1262-1300).  Built from flat per-event
    arrays instead of YAML + ROOT files."""

    def __init__(self, engine: Engine, n_samples: int, sample_bin_offset, par_vector: np.ndarray,
                 test_statistic=capi.kPoisson, update_w2=False, bin_edges=None):
        """bin_edges = [[edges_dim0, edges_dim1, ...] per sample] switches to device-side binning
        (SetupKinematics / ComputeBins / AddKinShift); a sample given as an ndarray
        [n_boxes][n_dim][2] is NON-UNIFORM (one {lo, hi} extent per axis and bin).  Otherwise
        global bins come precomputed through SetupEvents and `sample_bin_offset` defines the
        global bin space."""
        self.eng = engine
        self.nSamples = int(n_samples)
        if bin_edges is not None:
            assert len(bin_edges) == n_samples
            engine._chk(engine.lib.m3hip_binning_begin(engine.h))
            nb = []
            for s in bin_edges:
                if isinstance(s, np.ndarray) and s.ndim == 3:
                    ext = np.ascontiguousarray(s, np.float64)
                    engine._chk(engine.lib.m3hip_binning_add_nonuniform(engine.h, ext.shape[0], ext.shape[1], _ptr(ext, np.float64)))
                    nb.append(ext.shape[0])
                else:
                    n_edges = np.array([len(e) for e in s], np.int32)
                    edges = np.ascontiguousarray(np.concatenate([np.asarray(e, np.float64) for e in s]))
                    engine._chk(engine.lib.m3hip_binning_add_uniform(engine.h, len(s), _ptr(n_edges, np.int32), _ptr(edges, np.float64)))
                    nb.append(int(np.prod([len(e) - 1 for e in s])))
            engine._chk(engine.lib.m3hip_binning_end(engine.h))
            off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
        else:
            off = np.ascontiguousarray(sample_bin_offset, np.int32)
            engine._chk(engine.lib.m3hip_set_binning(engine.h, n_samples, _ptr(off, np.int32)))
        self.sample_bin_offset = off
        self.nBins = int(off[-1])
        self.par = par_vector  # plays _fPropVal; read at Reweight() time
        self._step_cache = None
        self.SetTestStatistic(test_statistic, update_w2)

    def GetName(self):
        return "SampleHandlerFD"

    def GetNsamples(self):
        return self.nSamples

    def SetTestStatistic(self, kind, update_w2=False):
        self.eng._chk(self.eng.lib.m3hip_set_test_statistic(self.eng.h, int(kind), int(update_w2)))

    def SetupEvents(self, event_offset, bin_idx, osc_idx, norm_off, norm_idx):
        n = len(bin_idx)
        self.eng._chk(self.eng.lib.m3hip_upload_events(
            self.eng.h, event_offset, n, _ptr(np.ascontiguousarray(bin_idx, np.int32), np.int32),
            _ptr(np.ascontiguousarray(osc_idx, np.int32), np.int32) if osc_idx is not None else None,
            _ptr(np.ascontiguousarray(norm_off, np.uint64), np.uint64) if norm_off is not None else None,
            _ptr(np.ascontiguousarray(norm_idx, np.int16), np.int16) if norm_idx is not None else None))

    # ---- experiment weights behind the splines (AddAdditionalWeightPointers / CalcWeightFunc, SampleHandlerFD.cpp:196, :428)
    def SetEventWeights(self, slot, event_offset, w):
        w = np.ascontiguousarray(w, np.float32)
        self.eng._chk(self.eng.lib.m3hip_upload_event_weights(self.eng.h, int(slot), event_offset, len(w), _ptr(w, np.float32)))

    def SetWeightIndex(self, slot, event_offset, idx):
        idx = np.ascontiguousarray(idx, np.int32)
        self.eng._chk(self.eng.lib.m3hip_upload_weight_index(self.eng.h, int(slot), event_offset, len(idx), _ptr(idx, np.int32)))

    def SetWeightTable(self, slot, table):
        table = np.ascontiguousarray(table, np.float32)
        self.eng._chk(self.eng.lib.m3hip_set_weight_table(self.eng.h, int(slot), _ptr(table, np.float32), len(table)))

    # ---- device-side binning (ref: Samples/BinningHandler.cpp:257-307, SampleStructs.h:577-613)
    def SetupKinematics(self, event_offset, sample_id, kin):
        kin = np.ascontiguousarray(kin, np.float64)
        if kin.ndim == 1:
            kin = kin.reshape(-1, 1)
        sid = np.ascontiguousarray(sample_id, np.int32) if sample_id is not None else None
        self.eng._chk(self.eng.lib.m3hip_upload_kinematics(self.eng.h, event_offset, kin.shape[0],
                                                           _ptr(sid, np.int32) if sid is not None else None,
                                                           _ptr(kin, np.float64), kin.shape[1]))

    def ComputeBins(self):
        self.eng._chk(self.eng.lib.m3hip_compute_bins(self.eng.h))

    def AddKinShift(self, sample, dim, kind, param_index):
        self.eng._chk(self.eng.lib.m3hip_add_kin_shift(self.eng.h, int(sample), int(dim), int(kind), int(param_index)))

    def AddKinShiftAux(self, sample, dim, aux_column, param_index):
        """x[dim] += theta * (nominal value of column aux_column)"""
        self.eng._chk(self.eng.lib.m3hip_add_kin_shift_aux(self.eng.h, sample, dim, aux_column, param_index))

    def AddKinCut(self, sample, column, lower, upper):
        self.eng._chk(self.eng.lib.m3hip_add_kin_cut(self.eng.h, int(sample), int(column), float(lower), float(upper)))

    def ClearKinShifts(self):
        self.eng._chk(self.eng.lib.m3hip_clear_kin_shifts(self.eng.h))

    def GetEventBins(self, off, n):
        out = np.empty(n, np.int32)
        self.eng._chk(self.eng.lib.m3hip_get_event_bins(self.eng.h, _ptr(out, np.int32), off, n))
        return out

    def SetOscillationWeights(self, table):
        """float64 tables go through m3hip_set_osc_table_f64 (M3::float_t = double build)"""
        if np.asarray(table).dtype == np.float64:
            t = np.ascontiguousarray(table, np.float64)
            self.eng._chk(self.eng.lib.m3hip_set_osc_table_f64(self.eng.h, _ptr(t, np.float64), len(t)))
            return
        t = np.ascontiguousarray(table, np.float32)
        self.eng._chk(self.eng.lib.m3hip_set_osc_table(self.eng.h, _ptr(t, np.float32), len(t)))

    def SetOscillationWeightsDevice(self, device_ptr: int, n: int, float_t=np.float32):
        """the table where a GPU calculator left it (m3hip_set_osc_table_device): `device_ptr` is the address of n floats (or
        doubles) in this device's memory, read in place; call again after every refill"""
        fn = self.eng.lib.m3hip_set_osc_table_device_f64 if float_t == np.float64 else self.eng.lib.m3hip_set_osc_table_device
        self.eng._chk(fn(self.eng.h, C.c_void_p(int(device_ptr)), int(n)))

    def AddData(self, data):
        d = np.ascontiguousarray(data, np.float64)
        self.eng._chk(self.eng.lib.m3hip_set_data(self.eng.h, _ptr(d, np.float64), len(d)))

    def Reweight(self):
        p = np.ascontiguousarray(self.par, np.float64)
        self.eng._chk(self.eng.lib.m3hip_reweight(self.eng.h, _ptr(p, np.float64), len(p)))

    def GetLikelihood(self) -> float:
        tot = C.c_double(0)
        self._per_sample = np.zeros(self.nSamples, np.float64)
        self.eng._chk(self.eng.lib.m3hip_get_likelihood(self.eng.h, C.byref(tot), _ptr(self._per_sample, np.float64)))
        return tot.value

    def GetSampleLikelihood(self, isample: int) -> float:
        self.GetLikelihood()
        return float(self._per_sample[isample])

    def Step(self) -> float:
        """Reweight() + GetLikelihood(): one proposal's sample part (ref: Fitters/MR2T2.cpp:66-76)."""
        p = self.par
        c = self._step_cache
        if c is None or c[0] is not p:  # the pointer of the caller's parameter array is looked up once
            assert isinstance(p, np.ndarray) and p.dtype == np.float64 and p.flags["C_CONTIGUOUS"]
            tot = C.c_double(0)
            c = self._step_cache = (p, _ptr(p, np.float64), len(p), tot, C.byref(tot))
        rc = self.eng.lib.m3hip_step(self.eng.h, c[1], c[2], c[4], None)
        if rc:
            self.eng._chk(rc)
        return c[3].value

    def _bins(self):
        mc = np.empty(self.nBins, np.float64)
        w2 = np.empty(self.nBins, np.float64)
        self.eng._chk(self.eng.lib.m3hip_get_bins(self.eng.h, _ptr(mc, np.float64), _ptr(w2, np.float64), self.nBins))
        return mc, w2

    def GetMCArray(self):
        return self._bins()[0]

    def GetW2Array(self):
        return self._bins()[1]

    # ---- parity helpers ----
    def RecordEventWeights(self, on=True):
        self.eng._chk(self.eng.lib.m3hip_record_event_weights(self.eng.h, int(on)))

    def GetEventWeights(self, off, n, float_t=np.float32):
        out = np.empty(n, float_t)
        get = self.eng.lib.m3hip_get_event_weights_f64 if float_t == np.float64 else self.eng.lib.m3hip_get_event_weights
        self.eng._chk(get(self.eng.h, _ptr(out, float_t), off, n))
        return out

    def GetSplineWeights(self, off, n):
        out = np.empty(n, np.float32)
        self.eng._chk(self.eng.lib.m3hip_get_spline_weights(self.eng.h, _ptr(out, np.float32), off, n))
        return out

    def GetSegments(self, n_params):
        seg = np.empty(n_params, np.int16)
        val = np.empty(n_params, np.float32)
        self.eng._chk(self.eng.lib.m3hip_get_segments(self.eng.h, _ptr(seg, np.int16), _ptr(val, np.float32), n_params))
        return seg, val


def build_from_synth(spec, device_id=0, chunk_events=131072, event_range=None, engine=None, order_by_osc=False):
    """Generate `spec` (events [e0,e1) of it) and upload it through the C ABI.
    Returns (engine, SMonolith, SampleHandlerFD, par_vector).  order_by_osc: every chunk goes up with its events'
    oscillation indices as the order key (m3hip_monolith_order_key)."""
    from . import synth

    e0, e1 = (0, spec.n_events) if event_range is None else event_range
    eng = engine or Engine(device_id)
    cx, npts = synth.knots(spec)

    def chunks():
        for a in range(e0, e1, chunk_events):
            ch = synth.monolith_chunk(spec, a, min(e1, a + chunk_events))
            if order_by_osc:
                ch["order_key"] = synth.events_chunk(spec, a, min(e1, a + chunk_events))["osc_idx"]
            yield ch

    mono = SMonolith(eng, spec.n_params, spec.n_knots, cx, npts, chunks(), expect_events=e1 - e0)
    par = synth.params(spec, 0)
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(eng, spec.n_samples, synth.sample_bin_offsets(spec), par, spec.test_statistic)
    for a in range(e0, e1, chunk_events):
        ev = synth.events_chunk(spec, a, min(e1, a + chunk_events))
        sh.SetupEvents(a - e0, ev["bin_idx"], ev["osc_idx"], ev["norm_off"], ev["norm_idx"])
    sh.SetOscillationWeights(synth.osc_table(spec))
    return eng, mono, sh, par
