/*
 * m3hip.h -- C ABI of the MI355X (gfx950) per-step likelihood engine for MaCh3.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each
 * entry point names the reference interface it replaces (file:line under the MaCh3
 * v2.4.2 tree).  The C++ facade in include/mach3_amd/ re-exposes the reference's class
 * names (SMonolithGPU, SMonolith, SampleHandlerFD) on top of exactly these calls.
 *
 * Conventions
 *   - every function returns 0 on success, a negative M3HIP_E* code otherwise; no C++ exception leaves the
 *     library (host allocation failures come back as M3HIP_ENOMEM);
 *     m3hip_last_error() gives the message (the reference logs + throws
 *     MaCh3Exception, Manager/MaCh3Exception.h:29-38; the facade rethrows).
 *   - host arrays are copied during the call; the caller keeps ownership.
 *   - one handle <-> one device <-> one HIP stream; calls on a handle must come from
 *     one host thread at a time (the reference's step loop is single threaded).
 *   - there is NO CPU fallback: without a usable HIP device m3hip_create fails.
 */
#ifndef M3HIP_H
#define M3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3HIP_OK 0
#define M3HIP_EINVAL (-1)  /* bad argument / inconsistent input arrays */
#define M3HIP_EHIP (-2)    /* HIP runtime error */
#define M3HIP_ESTATE (-3)  /* call out of order */
#define M3HIP_ENOMEM (-4)
#define M3HIP_ECOMM (-5)   /* RCCL error */

/* Samples/SampleStructs.h:105-112 -- same numeric values as enum TestStatistic */
#define M3HIP_kPoisson 0
#define M3HIP_kBarlowBeeston 1
#define M3HIP_kIceCube 2
#define M3HIP_kPearson 3
#define M3HIP_kDembinskiAbdelmotteleb 4

/* Manager/Core.h:80-91 */
#define M3HIP_LARGE_LOGL 1234567890.0
#define M3HIP_LOW_MC_BOUND 0.00001
#define M3HIP_UNDER_OVERFLOW_BIN (-1)

typedef struct m3hip_ctx* m3hip_handle;

/* ------------------------------------------------------------------------------------
 * Lifetime.  Replaces: SMonolithGPU::SMonolithGPU / ~SMonolithGPU
 * (Splines/gpuSplineUtils.cu:77-110), SetDevice/ResetDevice (Manager/gpuUtils.cu:85-111).
 * ---------------------------------------------------------------------------------- */
int m3hip_create(int device_id, m3hip_handle* out);
int m3hip_destroy(m3hip_handle h);
/* message of the last failure on h (h == NULL: last failure of m3hip_create) */
const char* m3hip_last_error(m3hip_handle h);
/* run on a caller-owned hipStream_t instead of the handle's own stream (NULL restores it) */
int m3hip_set_stream(m3hip_handle h, void* hip_stream);
/* number of usable HIP devices (no handle needed).  Replaces the cudaGetDeviceCount of PrintNdevices / SetDevice
 * (Manager/gpuUtils.cu:60-72, 85-100); one process per GPU picks device = local rank % count. */
int m3hip_device_count(int* out);
/* device memory in use / total, bytes.  Replaces checkGpuMem (Manager/gpuUtils.cu:40-56) */
int m3hip_mem_info(m3hip_handle h, size_t* free_bytes, size_t* total_bytes);

/* ------------------------------------------------------------------------------------
 * Seam 2: the spline-monolith backend (class SMonolithGPU, Splines/gpuSplineUtils.cuh:63-218)
 * ---------------------------------------------------------------------------------- */

/* Pinned host buffers.  Replaces InitGPU_Segments / InitGPU_Vals and the pinned
 * cpu_total_weights of InitGPU_SplineMonolith (gpuSplineUtils.cu:116-171), and
 * CleanupGPU_Segments / the cudaFreeHost in CleanupGPU_SplineMonolith (:519-570). */
/* h may be NULL (current device): the reference allocates its pinned segment / value arrays through a still-null
 * gpu_spline_handler, before the GPU object exists (SplineMonolith.cpp:82-83, 271). */
int m3hip_alloc_pinned(m3hip_handle h, void** ptr, size_t bytes);
int m3hip_free_pinned(m3hip_handle h, void* ptr);
/* SynchroniseSplines() (gpuSplineUtils.cu:71-73: cudaDeviceSynchronize), a free function in the reference */
int m3hip_device_synchronize(void);

/* Optional, before the first m3hip_monolith_append: the number of events this device is going to hold (the reference
 * knows it when it sizes its buffers: InitGPU_SplineMonolith's n_events, Splines/gpuSplineUtils.cuh:76-82).  Uploads are
 * streamed, so the engine cannot see the total by itself; with it, samples of many light events are laid out in bricks of
 * 2048 or 4096 events instead of 1024.  0 = unknown. */
int m3hip_monolith_expect_events(m3hip_handle h, uint64_t n_events);
/*
 * One-time upload of the flattened monolith.  Replaces InitGPU_SplineMonolith +
 * CopyToGPU_SplineMonolith (gpuSplineUtils.cu:116-333).  Input is the reference's flat
 * layout as built by SMonolith::PrepareForGPU (Splines/SplineMonolith.cpp:53-250):
 *   coeff_x[n_params*max_knots]      knot x per parameter (SplineCommon.h:36)
 *   npts[n_params]                   FastSplineInfo.nPts per parameter (SplineStructs.h:37);
 *                                    <= 1 for parameters without a TSpline3 (TF1 / unused)
 *   coeff_many[4*total_nknots]       AoS {y,b,c,d} per knot (SplineCommon.h:39)
 *   nknots_arr[n_splines]            first knot of each spline (SplineCommon.h:42)
 *   param_no[n_splines]              parameter of each spline (SplineCommon.h:45)
 *   coeff_tf1[2*n_tf1], param_no_tf1[n_tf1]   (SplineMonolith.h:130-137, _nTF1Coeff_ = 2)
 *   n_param_per_event[2*n_events]    {count,start} (SplineMonolith.h:124)
 *   n_param_per_event_tf1[2*n_events]{count,start} (SplineMonolith.h:127)
 * Unlike the reference, n_params is a run-time value (the reference needs a recompile,
 * gpuSplineUtils.cu:211-216) and the data are re-laid-out on upload.
 *
 * Large monoliths (> 2^32 coefficients, which the reference's unsigned offsets cannot
 * address) are streamed in event-ordered chunks: begin / append / end.  Offsets inside a
 * chunk are chunk-local.  Every chunk but the last must hold a multiple of 1024 events.
 *
 * What the upload does with the events (none of it is visible through this interface: every per-event
 * array handed over later, and every per-event result, is in the event order of the upload):
 *   - 256 events that carry the same ordered parameter list form a "uniform tile" (the streaming fast path);
 *   - lists that differ only by a few missing entries (the reference drops an event's flat one-knot splines)
 *     are padded to their union with identity responses {1,0,0,0} / {0,1}, which leave the float product
 *     unchanged bit for bit (at least 88 % of the padded slots must be real; M3HIP_PAD_MIN_FILL);
 *   - a chunk whose events mix several lists (interaction modes in random order) is PROCESSED grouped by
 *     list -- a lane permutation inside the kernels -- when that avoids the slow path (M3HIP_GROUP_EVENTS=0
 *     switches it off);
 *   - what remains ragged is laid out as "bricks" (requires ascending parameter order inside an event).
 * m3hip_get_info reports the outcome (n_tiles_uniform, n_tiles_padded, pad_responses, n_tiles_brick,
 * n_grouped_chunks).
 */
int m3hip_monolith_begin(m3hip_handle h, int n_params, int max_knots, const float* coeff_x,
                         const int16_t* npts);
/* Optional, before an m3hip_monolith_append: a key per event of THAT chunk by which the engine may order the chunk's events in its
 * own memory (stable sort, ascending as unsigned: -1 sorts last).  Meant for the events' oscillation index: the reference's events
 * alias the calculator's weight array (Samples/SampleHandlerFD.cpp:1108-1133) and read it in MC order; with the events of a chunk
 * stored in table order, the 64 look-ups of a wavefront fall into a few 128-byte lines of a large table instead of 64, which is what a
 * table that MOVES every step costs (one GPU's share of BASELINE configs[4], 485 us in MC order, 437 in table order).
 * Invisible through this interface: every per-event array handed over later and every per-event result keeps the CALLER's event
 * numbering (the library scatters / gathers at the boundary; set-up and diagnostics only -- a step moves no per-event data).  The
 * bin contents are the same sums in another order.  A chunk whose layout the key's order would inflate by more than 2 % (it tears
 * runs of equal parameter lists apart) keeps its order: m3hip_info::n_key_refused.  The key is consumed by the next append.
 * How the caller slices its upload does not matter: keyed chunks are held back (copied into the engine's own host arrays) until
 * 2^20 events are together and laid out as one chunk -- m3hip_info::n_keyed_chunks counts those -- so an error in a keyed chunk's
 * arrays may be reported by a LATER append or by m3hip_monolith_end; a chunk of 2^20 events and more goes straight through. */
int m3hip_monolith_order_key(m3hip_handle h, const int32_t* key, uint64_t n_events);
int m3hip_monolith_append(m3hip_handle h, uint64_t n_events, const float* coeff_many,
                          uint64_t total_nknots, const unsigned int* nknots_arr,
                          const int16_t* param_no, uint64_t n_splines, const float* coeff_tf1,
                          const int16_t* param_no_tf1, uint64_t n_tf1,
                          const unsigned int* n_param_per_event,
                          const unsigned int* n_param_per_event_tf1);
int m3hip_monolith_end(m3hip_handle h);
/* begin + one append + end */
int m3hip_upload_monolith(m3hip_handle h, uint64_t n_events, int n_params, int max_knots,
                          const float* coeff_x, const int16_t* npts, const float* coeff_many,
                          uint64_t total_nknots, const unsigned int* nknots_arr,
                          const int16_t* param_no, uint64_t n_splines, const float* coeff_tf1,
                          const int16_t* param_no_tf1, uint64_t n_tf1,
                          const unsigned int* n_param_per_event,
                          const unsigned int* n_param_per_event_tf1);

/*
 * Per step.  Replaces RunGPU_SplineMonolith(cpu_total_weights, vals, segment, n_splines,
 * n_tf1) (gpuSplineUtils.cu:444-512): evaluates every TSpline3 / TF1 response and the
 * per-event product (EvalOnGPU_Splines + EvalOnGPU_TF1 + EvalOnGPU_TotWeight, :337-438) in
 * ONE kernel and, if cpu_total_weights != NULL, starts the async device-to-host copy of
 * the n_events total weights.  vals / segment are what SplineBase::FindSplineSegment
 * (Splines/SplineBase.cpp:44-109) produced on the host.  Asynchronous.
 */
int m3hip_eval_splines(m3hip_handle h, const float* vals, const int16_t* segment,
                       float* cpu_total_weights);
/* Replaces SynchroniseSplines (gpuSplineUtils.cu:71-73) */
int m3hip_sync(m3hip_handle h);

/* ------------------------------------------------------------------------------------
 * Seam 3: the sample operator (SampleHandlerBase::Reweight / GetLikelihood,
 * Samples/SampleHandlerBase.h:40-49, as implemented by SampleHandlerFD.cpp:316-463,
 * 568-594, 1262-1300).  Spline evaluation, weight product, bin fill and bin LLH all stay
 * on the device; per step only the parameter vector goes in and the LLH comes out.
 * ---------------------------------------------------------------------------------- */

/* Which entry of the global parameter vector drives spline parameter i.
 * Replaces SMonolith::setSplinePointers (Splines/SplineMonolith.h:44-46), whose
 * `const double*` are addresses inside ParameterHandlerBase::_fPropVal. */
int m3hip_set_spline_param_map(m3hip_handle h, const int32_t* global_index, int n_params);

/* Global binning: sample s owns global bins [sample_bin_offset[s], sample_bin_offset[s+1])
 * (BinningHandler::SetGlobalBinNumbers, Samples/BinningHandler.cpp:341-355). */
int m3hip_set_binning(m3hip_handle h, int n_samples, const int32_t* sample_bin_offset);

/*
 * Per-event sample-side data for events [event_offset, event_offset+n), any chunking:
 *   bin_idx   global bin of the event (FindGlobalBin result, BinningHandler.cpp:257-291);
 *             -1 = under/overflow.  Static kinematics (no functional shifts).
 *   osc_idx   index into the oscillation-weight table, -1 = no oscillation pointer
 *             (SampleHandlerFD.cpp:1108-1133: NC / unity weights are not pushed)
 *   norm_off[n+1], norm_idx[]   CSR of global parameter indices whose values are the
 *             event's norm_pointers (FarDetectorCoreInfoStruct.h:109), chunk-local offsets
 * Must be called after m3hip_monolith_end.
 */
int m3hip_upload_events(m3hip_handle h, uint64_t event_offset, uint64_t n,
                        const int32_t* bin_idx, const int32_t* osc_idx,
                        const uint64_t* norm_off, const int16_t* norm_idx);

/*
 * Binned splines: the per-step part of BinnedSplineHandler
 * (Splines/BinnedSplineHandler.cpp:295-341) -- FindSplineSegment over the handler's unique spline
 * parameters, then weight = max(0, cubic) per non-flat bin-spline -- plus the per-event gather of
 * those weights (SampleHandlerFD::SetSplinePointers, Samples/SampleHandlerFD.cpp:1196-1243, puts
 * pointers into weightvec_Monolith on every event's total_weight_pointers).  float_t = float build.
 * Arrays are the handler's own members: coeffindexvec[M], uniquesplinevec_Monolith[M],
 * uniquecoeffindices[U], xcoeff_arr[C], manycoeff_arr[4C] (BinnedSplineHandler.h:116-140); xpts/npts are
 * SplineInfoArray (knots of each unique parameter).  A sample uses EITHER this OR the event-by-event
 * monolith (m3hip_upload_monolith), as in the reference.  Then: m3hip_set_spline_param_map,
 * m3hip_set_binning..., m3hip_upload_events (bins, osc, norms) and m3hip_binned_upload_events
 * (per event the CSR of indices into weightvec_Monolith, in pointer order).
 * Device memory: the tables are kept segment-major with max_knots - 1 planes for EVERY non-flat bin-spline, i.e.
 * 5 x sizeof(float_t) x (max_knots - 1) bytes each whatever its own knot count; an upload that would not fit the device's free
 * memory fails with M3HIP_ENOMEM and a message that says so.
 * Events that share their pointer list (at most one distinct list per 8 events of the first m3hip_binned_upload_events call) are
 * run by a kernel of their own on a processing-order copy of the event side (k_reweight_lists): event weights bit-identical to
 * the reference's loop in any MC order.
 */
int m3hip_binned_upload(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const float* xpts,
                        const int16_t* npts, int monolith_size, const int32_t* coeffindexvec,
                        const int32_t* uniquesplinevec, int n_unique, const int32_t* uniquecoeffindices,
                        int n_coeff, const float* xcoeff, const float* manycoeff);
int m3hip_binned_upload_events(m3hip_handle h, uint64_t event_offset, uint64_t n, const uint64_t* w_off,
                               const int32_t* w_idx);
int m3hip_get_binned_weights(m3hip_handle h, float* out, int monolith_size);

/*
 * The same for the reference's DEFAULT build, M3::float_t = double (Manager/Core.h:27-51; float only with
 * MaCh3_LOW_MEMORY_STRUCTS): knots, xcoeff_arr, manycoeff_arr and weightvec_Monolith are double, the cubic
 * uses std::fma on the parameter's own double value (BinnedSplineHandler.cpp:320-337), FindSplineSegment
 * compares the float-cast value with the double knots (SplineBase.cpp:51-54), and the whole sample side --
 * norm casts, oscillation weights, the weight product, w*w -- runs in double
 * (SampleHandlerFD.cpp:568-594, 432-446).  Uploading bin-splines through this entry point switches the
 * handle's sample epilogue to that build; oscillation weights then come from m3hip_set_osc_table_f64 (a
 * float table is widened), per-event weights from m3hip_get_event_weights_f64.  That build has no
 * event-by-event SMonolith (its weights are float), so the event space carries bin-splines only.
 */
int m3hip_binned_upload_f64(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const double* xpts,
                            const int16_t* npts, int monolith_size, const int32_t* coeffindexvec,
                            const int32_t* uniquesplinevec, int n_unique, const int32_t* uniquecoeffindices,
                            int n_coeff, const double* xcoeff, const double* manycoeff);
int m3hip_get_binned_weights_f64(m3hip_handle h, double* out, int monolith_size);

/*
 * Device-side binning: instead of precomputed bin_idx, hand over the
 * binning and the nominal kinematic variables and let the device run
 * BinningHandler::FindGlobalBin / SampleBinningInfo::FindBin (Samples/BinningHandler.cpp:257-307,
 * Samples/SampleStructs.h:577-613).  Uniform N-D binning, N <= 4.
 *   m3hip_set_binning_edges  per sample: n_dim[s] axes with n_edges[..] increasing edges each,
 *                            edges concatenated in (sample, axis) order (SampleBinningInfo::InitUniform,
 *                            SampleStructs.h:250-287); also defines the global bin space, so it
 *                            replaces m3hip_set_binning
 *   m3hip_upload_kinematics  EventInfo.NominalSample and the KinVar values, kin[n][kin_stride]
 *   m3hip_compute_bins       nominal bins (FindNominalBin) and the global bin of every event
 *   m3hip_add_kin_shift      a functional-parameter shift of axis `dim` driven by params[param_index]:
 *                            kind 1: x *= (1 + theta), kind 2: x += theta.  The reference applies
 *                            arbitrary std::function callbacks per event (SampleHandlerFD.cpp:545-564);
 *                            the device supports this closed set.  With shifts registered, every
 *                            m3hip_reweight re-bins all events (nominal bin, +-1 neighbour, binary search).
 *   m3hip_binning_begin / _add_uniform / _add_nonuniform / _end
 *                            the same, sample by sample (BinningHandler::SetupSampleBinning,
 *                            Samples/BinningHandler.cpp:15-60), and the only way to declare a NON-UNIFORM
 *                            sample: n_boxes bins given as extents[n_boxes][n_dim][2] = {lo, hi} per axis
 *                            (SampleBinningInfo::InitNonUniform, Samples/SampleStructs.h:465-530; 2..4
 *                            dimensions as there).  The library builds the reference's 10-per-axis
 *                            mega-bin grid and its BinGridMapping (SampleStructs.h:397-462); an event lands
 *                            in the first candidate box with lo < x <= hi on every axis
 *                            (BinInfo::IsEventInside, SampleStructs.h:186-220), global bin = box number +
 *                            the sample's offset.  Samples get their global offsets in the order added.
 */
int m3hip_set_binning_edges(m3hip_handle h, int n_samples, const int32_t* n_dim, const int32_t* n_edges,
                            const double* edges);
int m3hip_binning_begin(m3hip_handle h);
int m3hip_binning_add_uniform(m3hip_handle h, int n_dim, const int32_t* n_edges, const double* edges);
int m3hip_binning_add_nonuniform(m3hip_handle h, int n_boxes, int n_dim, const double* extents);
int m3hip_binning_end(m3hip_handle h);
int m3hip_upload_kinematics(m3hip_handle h, uint64_t event_offset, uint64_t n, const int32_t* sample_id,
                            const double* kin, int kin_stride);
int m3hip_compute_bins(m3hip_handle h);
int m3hip_add_kin_shift(m3hip_handle h, int sample, int dim, int kind, int param_index);
/* kind 3: x[dim] += theta * (nominal value of kinematic column aux_column) -- e.g. E_rec = E_lep + E_had with a scale
 * on the hadronic part only: keep E_had in a cut-only column and shift E_rec by theta * E_had */
int m3hip_add_kin_shift_aux(m3hip_handle h, int sample, int dim, int aux_column, int param_index);
int m3hip_clear_kin_shifts(m3hip_handle h);
/* SampleHandlerFD::IsEventSelected (Samples/SampleHandlerFD.cpp:281-294): an event is kept while
 * lower <= x < upper for every cut of its sample; `column` addresses the kinematics array (binning
 * axes first, further columns may hold cut-only variables).  Evaluated on the shifted values. */
int m3hip_add_kin_cut(m3hip_handle h, int sample, int column, double lower, double upper);
int m3hip_get_event_bins(m3hip_handle h, int32_t* out, uint64_t event_offset, uint64_t n);

/* Pre-staged oscillation weights (what OscillationHandler::Evaluate leaves behind the
 * event pointers, Samples/OscillationHandler.cpp:81-119).  May be re-uploaded per step. */
int m3hip_set_osc_table(m3hip_handle h, const float* table, uint64_t n);
int m3hip_set_osc_table_f64(m3hip_handle h, const double* table, uint64_t n);
/* The same table where a GPU calculator LEFT it: `d_table` is device memory of the engine's device and is read in place by the
 * kernels of the following reweights -- no copy, no host round trip.  This is the reference's own arrangement: the events'
 * oscillation pointers alias the calculator's weight array (Samples/SampleHandlerFD.cpp:1108-1133, NuOscillator's
 * ReturnWeightPointer, Samples/OscillationHandler.cpp:103-119), which OscillationHandler::Evaluate refills inside every Reweight
 * (SampleHandlerFD.cpp:323, OscillationHandler.cpp:81-100).  Call it after every refill (cheap: no device work, the pointer is
 * checked the first time only); the call is what tells the engine that the weights changed.
 * ORDERING is by the handle's stream: the refill must have completed, or have been enqueued on the handle's stream
 * (m3hip_set_stream makes the calculator's stream the engine's), before m3hip_reweight / m3hip_step is called; and the array
 * must not be overwritten before that step has completed (m3hip_get_likelihood / m3hip_step returned, or m3hip_sync) --
 * unless the writer is enqueued on the same stream behind it.  The array stays the caller's; it must outlive its use here
 * (until the next m3hip_set_osc_table* call or m3hip_destroy).  Errors: M3HIP_EINVAL for a host pointer or another device's
 * memory. */
int m3hip_set_osc_table_device(m3hip_handle h, const float* d_table, uint64_t n);
int m3hip_set_osc_table_device_f64(m3hip_handle h, const double* d_table, uint64_t n);

/* Experiment weights behind oscillation and splines.  SampleHandlerFD::AddAdditionalWeightPointers
 * (Samples/SampleHandlerFD.cpp:196) lets an experiment append pointers to an event's total_weight_pointers;
 * CalcWeightTotal multiplies them in that order after the spline weights (:584-591), and the virtual
 * CalcWeightFunc(iEvent) (SampleHandlerFD.h:289, called at :428) refreshes what they point at every step.
 * Here such a pointer is a SLOT (0..7, multiplied in ascending order behind the spline weight):
 *   m3hip_upload_event_weights  the pointer targets a per-event variable that does not change between steps
 *                               (flux / POT / detector weights): one float per event;
 *   m3hip_upload_weight_index + m3hip_set_weight_table
 *                               the pointer targets a variable shared by a class of events that CalcWeightFunc
 *                               recomputes: an index per event (-1: the event has no such pointer) into a table the
 *                               host refreshes whenever it likes, also every step (asynchronous, like the oscillation
 *                               weights).
 * A weight that needs per-event arithmetic on the device beyond these two forms is not representable: such a sample
 * keeps seam 2 (m3hip_eval_splines) and fills its histograms on the host as the reference does. */
#define M3HIP_MAX_EXTRA_WEIGHTS 8
int m3hip_upload_event_weights(m3hip_handle h, int slot, uint64_t event_offset, uint64_t n, const float* w);
int m3hip_upload_weight_index(m3hip_handle h, int slot, uint64_t event_offset, uint64_t n, const int32_t* idx);
int m3hip_set_weight_table(m3hip_handle h, int slot, const float* table, uint64_t n);

/* SampleHandlerFD::AddData for all global bins (SampleHandlerFD.cpp:1027-1044) */
int m3hip_set_data(m3hip_handle h, const double* data, int n_bins);
/* Deterministic bin fill ("histogram determinism"; the reference's OpenMP array reduction has no fixed order
 * either, SampleHandlerFD.cpp:415).  on = 1: no floating-point atomics between workgroups.  Inside a workgroup the waves add
 * their events in turn; every workgroup stores its bin vector to a row of its own, and two small kernels sum the rows in
 * index order.  Bins and likelihood of a parameter vector are then bit-identical from call to call whatever the
 * hardware's arbitration does.  Needs the workgroup-private bin vector (at most 1280 bins); the float build of the
 * sample side.  Cost: two extra launches and one row of bins per workgroup written and read (+5 % per step on configs[1], +7 % on a
 * configs[2] shard). */
int m3hip_set_deterministic(m3hip_handle h, int on);
/* SampleHandlerBase::SetTestStatistic (SampleHandlerBase.h:185) + LikelihoodOptions:UpdateW2
 * (SampleHandlerFD.cpp:62-65) */
int m3hip_set_test_statistic(m3hip_handle h, int kind, int update_w2);

/* SampleHandlerFD::Reweight() (SampleHandlerFD.cpp:316-343) for parameter vector
 * params[n_par] (the proposed values, ParameterHandlerBase.h:205).  Asynchronous.
 * Before anything is launched the indices the kernels use unguarded are checked against what they point into:
 * the spline parameter map and the events' norm indices against n_par, the events' oscillation indices against
 * the uploaded table, kinematic shift / cut columns against kin_stride (M3HIP_EINVAL, nothing runs).
 * Non-finite weights behave as in the reference: `w <= 0` drops the event, so a NaN weight is filled and turns
 * its bin and the likelihood NaN. */
int m3hip_reweight(m3hip_handle h, const double* params, int n_par);
/* SampleHandlerFD::GetLikelihood() (+ GetSampleLikelihood(i) into per_sample[n_samples],
 * may be NULL).  Runs the bin-LLH kernel, copies the result back and synchronises. */
int m3hip_get_likelihood(m3hip_handle h, double* llh_total, double* per_sample);
/* reweight + get_likelihood: one MCMC step's `samples[i]->Reweight(); llh +=
 * samples[i]->GetLikelihood()` pair (Fitters/MR2T2.cpp:66-76) */
int m3hip_step(m3hip_handle h, const double* params, int n_par, double* llh_total,
               double* per_sample);

/* GetMCArray / GetW2Array (SampleHandlerFD.h:162-187).  Synchronises. */
int m3hip_get_bins(m3hip_handle h, double* mc, double* w2, int n_bins);
/* Device-side results for parity checks: the monolith's per-event total weight
 * (SMonolith::cpu_total_weights) and the full CalcWeightTotal weight.  Synchronises. */
int m3hip_get_spline_weights(m3hip_handle h, float* out, uint64_t event_offset, uint64_t n);
/* enable = 1: subsequent m3hip_reweight calls also store both per-event weights (off by
 * default: the fused step writes nothing per event) */
int m3hip_record_event_weights(m3hip_handle h, int enable);
int m3hip_get_event_weights(m3hip_handle h, float* out, uint64_t event_offset, uint64_t n);
int m3hip_get_event_weights_f64(m3hip_handle h, double* out, uint64_t event_offset, uint64_t n);
/* state written by the device-side FindSplineSegment of the last reweight */
int m3hip_get_segments(m3hip_handle h, int16_t* segments, float* vals, int n_params);

/* ------------------------------------------------------------------------------------
 * Multi-GPU (new; the reference is single device, Manager/gpuUtils.cu:71).  Events are
 * sharded over ranks; m3hip_reweight then ends with ONE in-place ncclAllReduce(sum) of the
 * per-bin rate vector (and of w2 while it is being updated) before the LLH kernel.
 * ---------------------------------------------------------------------------------- */
#define M3HIP_UNIQUE_ID_BYTES 128
/* 1 when librccl can be loaded in this process (no communicator is created): lets every rank agree on
 * "in-library exchange or not" BEFORE the collective m3hip_comm_init, which would otherwise hang on the
 * ranks that can while one cannot. */
int m3hip_comm_available(void);
int m3hip_comm_get_unique_id(void* out_id /*128 bytes*/);
int m3hip_comm_init(m3hip_handle h, const void* unique_id, int rank, int n_ranks);
int m3hip_comm_destroy(m3hip_handle h);
/* Validator of the exchange (also M3HIP_COMM_VALIDATE=1 in the environment of m3hip_comm_init):
 * every m3hip_reweight additionally all-gathers the ranks' partial bin vectors, copies them to the host,
 * sums them there in rank order and fails with M3HIP_ECOMM if the all-reduced vector differs by more than
 * a few ulps per bin.  Costs a device synchronisation per step: for tests and first runs on a new node. */
int m3hip_comm_set_validate(m3hip_handle h, int on);

/* One-shot exchange of the bin vector (the message is 1.6 - 6.4 KB, latency-bound; a ring all-reduce is
 * 2 (N-1) hops, this is one).  Every rank owns a buffer of 2 x N slots in fine-grained device memory; per step a rank
 * stores its folded [mc | w2] vector and a ticket into its slot of EVERY rank's buffer (peer stores over xGMI through
 * IPC mappings), and the bin likelihood starts once the N tickets of its own buffer carry the step number, from the
 * rank-ordered sum of the N slots: bit-identical bins on every rank.  Transport-free like m3hip_comm_init: the caller
 * carries the 64-byte handles between the ranks (MPI, torch.distributed, a file).
 *   m3hip_xch_create  after the bin space is defined; allocates this rank's buffer, returns its IPC handle
 *   m3hip_xch_open    handles of all ranks in rank order (this rank's own entry is ignored); from here on
 *                     m3hip_reweight / m3hip_step exchange through the buffers instead of RCCL
 *                     Ranks that live in the SAME process (one process driving several engines) are recognised by their
 *                     handles and wired up directly; such a process must issue m3hip_reweight for all of its ranks before it
 *                     waits for the likelihood of any (the step is a collective).
 *   m3hip_xch_set_timeout  how long a rank waits for its peers' vectors of a step (default 2 s, at most 50 s -- the spin runs on
 *                     the device, and the host-side waits around it give up after 60 s; also M3HIP_XCH_TIMEOUT_S
 *                     in the environment of m3hip_xch_create).  Choose it above the largest host-side skew between two
 *                     ranks of one step (chain output, checkpoints, a new throw matrix).
 * A peer that does not push in time (a dead rank) is reported as M3HIP_ECOMM by whichever call reads the step's bins next
 * (m3hip_get_likelihood / m3hip_step / m3hip_get_bins); nothing on the device waits for ever.  The rank then holds its own
 * PARTIAL bins while its peers may have completed the step: M3HIP_ECOMM from the exchange is fatal for the whole group
 * (stop every rank; there is no resynchronisation).  A slot that already carries a LATER step -- this rank fell a whole
 * step behind -- is the same error.  The reference has no counterpart (one device: Manager/gpuUtils.cu:71). */
#define M3HIP_XCH_HANDLE_BYTES 64
int m3hip_xch_create(m3hip_handle h, int rank, int n_ranks, void* handle_out /* M3HIP_XCH_HANDLE_BYTES */);
int m3hip_xch_open(m3hip_handle h, const void* handles /* n_ranks x M3HIP_XCH_HANDLE_BYTES */);
int m3hip_xch_destroy(m3hip_handle h);
int m3hip_xch_set_timeout(m3hip_handle h, double seconds);
int m3hip_comm_last_validation(m3hip_handle h, double* max_rel_diff, uint64_t* n_checked);

/* ------------------------------------------------------------------------------------
 * The parameter handler's O(nPar^2) work of a Metropolis step, on the device (the callers of the hot
 * path; at the 1 040 parameters of BASELINE configs[4] these sweeps take longer on one host core than the GPU's step).
 *   m3hip_par_create             a context for ONE parameter handler (its matrices, its results, its ticket); a fit with several
 *                                handlers on one engine (MCMCBase loops over all systematics objects, Fitters/MCMCBase.cpp:32-77)
 *                                creates one each.  Up to M3HIP_MAX_PAR_SLOTS; m3hip_par_destroy gives a slot back.
 *   m3hip_par_upload(which = 0)  InvCovMatrix, nPar x nPar row-major (lower triangle read)
 *   m3hip_par_upload(which = 1)  the lower Cholesky factor of the throw matrix (ParameterHandlerBase::SetThrowMatrix,
 *                                Parameters/ParameterHandlerBase.cpp:1140-1200); again whenever the adaptive proposal replaces it
 *   m3hip_par_launch             enqueue, without waiting: the prior term 0.5 d^T V^-1 d of ParameterHandlerBase::CalcLikelihood
 *                                (Parameters/ParameterHandlerBase.cpp:816-841; d = proposed - prior value, 0 where the prior is
 *                                flat) and the correlated throw L z of CorrelateSteps (:726-745, M3::MatrixVectorMulti,
 *                                Parameters/ParameterHandlerUtils.h:118-135; z = the random vector of Randomize, :690-707, drawn
 *                                on the host).  Runs on a stream of its own beside the step's reweight kernel.
 *   m3hip_par_result             waits for the slot's last launch; prior and corr_throw[nPar].  Fixed summation order:
 *                                reproducible, equal to the host loops up to fp64 rounding.
 * Slots are independent: launches of different slots may be in flight together, and a slot's results stay until ITS next launch.
 * include/mach3_amd/Fitters.h: ParameterHandlerBase::AttachDevice uses these (one slot per attached handler). */
#define M3HIP_MAX_PAR_SLOTS 16
int m3hip_par_create(m3hip_handle h, int* slot_out);
int m3hip_par_destroy(m3hip_handle h, int slot);
int m3hip_par_upload(m3hip_handle h, int slot, int which, const double* matrix, int n);
int m3hip_par_launch(m3hip_handle h, int slot, const double* d, const double* z, int n);
int m3hip_par_result(m3hip_handle h, int slot, double* prior, double* corr_throw, int n);

/* ------------------------------------------------------------------------------------
 * Introspection / measurement
 * ---------------------------------------------------------------------------------- */
typedef struct {
  uint64_t n_events, n_splines, n_tf1;
  uint64_t n_tiles_uniform, n_tiles_brick;
  uint64_t coeff_bytes;     /* device bytes of the re-laid-out coefficient store */
  uint64_t alg_bytes_eval;  /* 16*S + 8*T + 4*E : algorithmic bytes of one spline-eval launch */
  uint64_t alg_bytes_fused; /* 16*S + 8*T + 8*E : ... of one fused reweight launch */
  int n_params, max_knots, n_bins, n_samples;
  uint64_t n_tiles_padded;  /* uniform tiles that hold identity-padded ragged events (counted in n_tiles_uniform) */
  uint64_t pad_responses;   /* identity responses added by that padding (TSpline3 + TF1); not in n_splines / n_tf1 */
  uint64_t n_grouped_chunks; /* upload chunks whose events the engine grouped by parameter list (processing order only:
                              * every per-event array and result keeps the caller's event order) */
  uint64_t brick_groups;     /* G: a brick holds up to G x 1024 events (1, 2 or 4; 0 before the first chunk) */
  uint64_t cached_stream_bytes; /* brick kernel, as of the last fused reweight: bytes of the small streams a step re-reads that keep
                                 * cached loads, i.e. that the engine plans to find in the Infinity Cache again next step (stage
                                 * tables, sample words, TF1 bookkeeping, and the TF1 codes unless tf1_codes_around) */
  uint64_t tf1_codes_around;    /* 1: the TF1 codes are loaded non-temporally, around the Infinity Cache (they did not fit beside the others) */
  double osc_lines_per_kevent;  /* distinct 128-byte lines of the (float) oscillation table that 1024 consecutive events point at, averaged
                                 * over the uploaded events (0: none uploaded).  Near 1: the MC is registered in oscillation-bin order and a
                                 * wave's look-ups share a line or two; hundreds: every event reads a line of its own -- free while the table
                                 * rests (the weights are folded into the events' words), +11 % on BASELINE configs[4]'s kernel when a fit moves
                                 * oscillation parameters.  m3hip_monolith_order_key lets the engine store the events in table order itself
                                 * (the figure is then of ITS order); SampleHandlerFD::EventOrderByOscillationBin gives the order to register the MC in. */
  uint64_t n_keyed_chunks;      /* chunks (of >= 2^20 events, merged from the caller's keyed chunks) stored in the order of the key (m3hip_monolith_order_key) */
  uint64_t n_key_refused;       /* ... that kept their order because the key's would have cost more than 2 % of layout bytes */
} m3hip_info;
int m3hip_get_info(m3hip_handle h, m3hip_info* out);
/* Average duration (ms) of the dominant kernel over the launches since the last reset,
 * measured with hipEvents on the stream the kernel runs on.  which: 0 = fused reweight
 * kernel, 1 = spline-eval kernel, 2 = with a communicator: from the end of the fused reweight kernel to the point where the
 * bin likelihood may start (replica fold + all-reduce + un-park, taken on the same sampled steps from an event recorded in
 * front of the likelihood kernel).  enable = N > 0 times every N-th fused launch (1: all of them), 0 switches
 * the events off.  The fused launch is an extended launch whose two events receive the dispatch's own
 * start / end timestamps -- what a profiler reports -- at ~3 us per timed step; events recorded around the
 * launch instead cost ~8 us and read 2-3 % long. */
int m3hip_timing_enable(m3hip_handle h, int enable);
int m3hip_timing_read(m3hip_handle h, int which, double* avg_ms, uint64_t* n_launches);
int m3hip_timing_reset(m3hip_handle h);

/* Test hook: overwrite the device-side bin vectors (mc, w2) so that m3hip_get_likelihood can
 * be checked on arbitrary (data, mc, w2) triples. */
int m3hip_debug_set_bins(m3hip_handle h, const double* mc, const double* w2, int n_bins);

/* Tuning knobs of the reweight kernel: loads in flight per lane (8 / 16), non-temporal
 * coefficient loads (0 / 1), dynamic-LDS pad in bytes (caps workgroups per CU; 0 = none). */
int m3hip_debug_set_knobs(m3hip_handle h, int unroll, int nt, int lds_pad);

/* Number of replicas of the bin vector the reweight kernel spreads its flushes over
 * (default 32; 1 = every workgroup adds into the same addresses).  Before m3hip_set_binning. */
int m3hip_debug_set_hist_replicas(m3hip_handle h, int n);

/* Host-side seconds spent in the phases of the last m3hip_reweight / m3hip_get_likelihood:
 * {stage params, launch prep, launch reweight(+all-reduce), launch llh, wait, collect}. */
int m3hip_debug_host_times(m3hip_handle h, double* out6);

/* Diagnostics: device-side time (us, 100 MHz wall clock) from the start of k_prep to the end of
 * k_llh of the last step; enable = 1 allocates the two clock words.  last_us[5] = {span, device
 * clock at k_prep start, at k_llh end, host clock at the k_prep launch, at completion seen}. */
int m3hip_debug_device_span(m3hip_handle h, int enable, double* last_us);

/* Host-only probe of the rule that evens out a brick launch (m3hip_upload.hip, ShortBrickPlan): the bricks an all-ragged
 * sample of expect_events events ends up in when it arrives in chunks of chunk_events events on a device of n_cu compute
 * units, bricks of brick_groups x 1024 events; n_short: how many of them are three-quarter bricks.  Used by the CPU tests. */
int m3hip_debug_short_brick_plan(uint64_t expect_events, uint64_t chunk_events, int brick_groups, int n_cu,
                                 uint64_t* n_bricks, uint64_t* n_short);

/* Host-only probe of the upload re-layout (no device needed): sizes[8] = {n_tiles,
 * spline store in float4, TF1 store in float2, slot-meta in int32, n_bricks, brick words,
 * first word of the bricks' TF1 parameter dictionary inside the brick words, its entries};
 * with non-NULL output buffers of those sizes it also fills them,
 * tile_desc[n_tiles][8] = {spl_off, tf1_off, meta_off, n_spl_slots, n_tf1_slots, n_events,
 * first_event, n_planes} and brick_desc[n_bricks][8] = {spl_off, tf1_off, words_off,
 * n_spl_float4, n_tf1_float2, n_events, first_event, n_words}.  Used by the CPU tests. */
int m3hip_debug_layout(int n_params, int max_knots, const int16_t* npts, uint64_t n_events,
                       const float* coeff_many, uint64_t total_nknots,
                       const unsigned int* nknots_arr, const int16_t* param_no, uint64_t n_splines,
                       const float* coeff_tf1, const int16_t* param_no_tf1, uint64_t n_tf1,
                       const unsigned int* n_param_per_event,
                       const unsigned int* n_param_per_event_tf1, uint64_t* sizes, float* spl,
                       float* tf1, int32_t* meta, uint64_t* tile_desc, uint64_t* brick_desc,
                       uint32_t* brick_words);

const char* m3hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
