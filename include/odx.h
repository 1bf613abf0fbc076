/*
 * odx.h — C ABI of libodx.so: the MI355X (gfx950) hot path of hsp-iit/online-detection.
 *
 * The reference has no FFI of its own: it is pure Python and reaches this arithmetic
 * through third-party packages (falkon, torch/cuBLAS/cuSOLVER, maskrcnn_benchmark CUDA
 * ops).  Each entry point below names the reference call site it stands in for
 * (paths relative to the reference root).  Conventions, all functions:
 *   - extern "C", return int: 0 = ok, <0 = error (odx_last_error_string() has the text);
 *   - raw DEVICE pointers, element counts, leading dimensions in ELEMENTS, row-major;
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no allocation, no sync;
 *     workspace sizes come from the *_workspace_bytes twins;
 *   - one host thread per device.
 * Precision policy: the n x M Gaussian block K_nM is formed at f32 accuracy and
 * stored in f32 — by default on the f16 matrix cores through a two-term f16 split of every f32
 * operand value (odx_gauss_*_h2, v_mfma_f32_16x16x32_f16 with f32 accumulation), alternatively
 * on the f32-input MFMA (odx_gauss_*_f32, exact fmaf chain); everything M x M and every
 * M-vector, including the accumulations inside the K_nM passes, is f64.
 */
#ifndef ODX_H
#define ODX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ODX_OK 0
#define ODX_ERR_INVALID (-1)     /* bad argument (shape, alignment, null pointer) */
#define ODX_ERR_HIP (-2)         /* a HIP runtime call / launch failed */
#define ODX_ERR_UNSUPPORTED (-3) /* size outside what the kernels are built for */
#define ODX_ERR_WORKSPACE (-4)   /* workspace too small */

typedef void* odx_stream_t;

const char* odx_last_error_string(void);
int odx_version(void);
/* number of compute units of the current device (grid sizing for persistent kernels) */
int odx_device_cus(void);
/* Process-wide options of the library — the ONLY switches there are: no entry point reads the environment.  The host side sets
 * them once when it loads the library (odx/options.py; INTEGRATION.md has the table).  Defaults = the benched configuration.
 *
 *   name                      values                         default  meaning
 *   "h2_tile"                 0 | 128 | 256                  0        tile core of the split-f16 Gaussian kernels / row GEMMs:
 *                                                                     0 = 256 x 256 once a launch has >= 512 (GEMMs: 256) tiles
 *   "precond"                 0 | 1 | 2                      0        A factor of the preconditioner on the split-f16 core:
 *                                                                     0 = from 4096 centres on, 1 = never (all-f64 chain), 2 = always
 *   "chain_helpers"           -1 | 0 | 1                     -1       internal helper streams of the factorisation chains:
 *                                                                     -1 = from 4096 rows on, 0 = never, 1 = always
 *   "rls_force_nt_gram"       0 | 1                          0        TEST HOOK: every RLS Gram by the transposed-copy + NT-GEMM
 *                                                                     route (otherwise only D % 8 != 0 takes it)
 *   "rls_force_inverse_solve" 0 | 1                          0        TEST HOOK: every RLS solve by the explicit inverse
 *                                                                     (otherwise only D + 1 > 2048 takes it)
 *
 * odx_set_option: ODX_ERR_INVALID for an unknown name or a value outside the column above.  odx_option_default: the default. */
int odx_set_option(const char* name, int value);
int odx_get_option(const char* name, int* value);
int odx_option_default(const char* name, int* value);
/* Destroys the calling thread's internal helper streams of the factorisation chains (odx_falkon_precond_*, odx_potrf_f64 from 4096
 * rows on: look-ahead updates, the inverse beside T T'); they are made again on demand.  Does not wait: the runtime keeps a
 * destroyed stream until the work queued on it has completed.  Call it when a fit / a training step is queued: on this runtime
 * the helpers of a class-batched chain, idle but alive, slow every later small launch of the process (a one-image forward behind
 * the headline job: 4.7 -> 7.6 ms). */
int odx_release_helper_streams(void);
/* Two streams for the library to use as its helper streams (slot 0: the forked inverse, slot 1: the look-ahead updates) instead of
 * creating its own; (NULL, NULL): it creates them again.  Why: the hardware queue a stream created inside the library lands on
 * depends on how many streams the process has created before, and a helper that shares the main stream's queue serialises the
 * chain's GEMMs behind every K_nM build; the Python host side measures which streams sit on queues of their own
 * (odx/streams.py) and hands two of them in when it loads the library.  Handed-in streams are never destroyed by the library;
 * chains confined by a CU mask (odx_set_side_stream_cu_mask) keep making masked helpers of their own. */
int odx_set_helper_streams(odx_stream_t s0, odx_stream_t s1);
/* CU-partitioned execution (diagnostic: tools/cu_split_probe.py measured it and the job does NOT use it).
 * odx_stream_create_cu_mask: a HIP stream whose kernels run only on the compute units whose
 * bit is set in `mask` (`words` 32-bit words, bit i = logical CU i; hipExtStreamCreateWithCUMask) — the HBM-bound CG passes
 * of one class and the MFMA-bound K_nM build / scoring of its neighbours then run beside each other on disjoint parts of
 * the chip (odx/job.py; the reference has no counterpart: one class after the other,
 * OnlineRegionClassifier_incore.py:96-155).  odx_set_pass_cus: the CU count the persistent grids of the compact pass
 * kernels are sized for (0 = the device's; set it to the size of the partition their stream is confined to).
 * odx_debug_placement: where `blocks` one-wave workgroups launched on `stream` ran — out[3 b] = XCC id, out[3 b + 1] =
 * HW_ID register, out[3 b + 2] = arrival order (out: 3 * blocks + 1 int32 on the device; every workgroup holds its CU
 * for `spin` ticks of the 100 MHz clock).                                                                                */
int odx_stream_create_cu_mask(const uint32_t* mask, int words, odx_stream_t* stream);
int odx_stream_destroy(odx_stream_t stream);
int odx_set_pass_cus(int cus);
/* the CU mask the library's internal helper streams (look-ahead of the blocked Cholesky, the forked inverse) are created
 * with from now on; words = 0: the whole device.  Streams that exist keep theirs.                                       */
int odx_set_side_stream_cu_mask(const uint32_t* mask, int words);
int odx_debug_placement(int32_t* out, int blocks, int spin, odx_stream_t stream);

/* ---------------------------------------------------------------- A3: Gaussian kernel
 * falkon.kernels.GaussianKernel(sigma) as built at
 * src/modules/region-classifier/FALKONWrapper_with_centers_selection_incore.py:50 and
 * applied inside InCoreFalkon.fit (:68):  K_ij = exp(-||x_i - z_j||^2 / (2 sigma^2)).   */

/* out[i] = sum_d X[i,d]^2 (f32 fmaf chain).  ldx % 4 == 0. */
int odx_row_sqnorm_f32(const float* X, int64_t ldx, int64_t n, int D, float* out, odx_stream_t stream);
/* The same norms (bit for bit) and, from the same read, max |x| of the matrix into meta[1] (its IEEE bits; meta[1] must be 0
 * on entry) — the value odx_split_f16 scales by; odx_split_f16_premax then packs without a maximum pass of its own. */
int odx_row_sqnorm_absmax_f32(const float* X, int64_t ldx, int64_t n, int D, float* out, float* meta, odx_stream_t stream);

/* K (n x M, ldk % 4 == 0) = gauss(X (n x D), Z (M x D)); xsq/zsq from odx_row_sqnorm_f32.
 * Columns [M, ldk) of K are written as 0.  ldx % 4 == 0, ldz % 4 == 0. */
int odx_gauss_knm_f32(const float* X, int64_t ldx, const float* xsq, int64_t n,
                      const float* Z, int64_t ldz, const float* zsq, int64_t M, int D,
                      double sigma, float* K, int64_t ldk, odx_stream_t stream);

/* ---------------------------------------------------------------- A5 / A9: scoring
 * model.predict(X) = K(X, ny_points_) @ alpha_
 *   (FALKONWrapper_with_centers_selection_incore.py:75-82) and the batched multi-class
 * kernel.mmv(features, nystrom_parallel, alpha_parallel) of the test-time heads
 *   (mrcnn_modified/modeling/roi_heads/box_head/roi_box_predictors.py:140-160,
 *    mrcnn_modified/modeling/rpn/rpn.py:201-227,
 *    mrcnn_modified/modeling/roi_heads/mask_head/roi_mask_predictors.py:72-99).
 * V (Mtot x T, f64, ldv) plays alpha_ / alpha_parallel.  Its block structure is passed as
 * per-column row ranges: ranges[2c], ranges[2c+1] (DEVICE int32) bound the rows of V that
 * are non-zero in column c, and out[i, c] = sum_{j in range c} K_ij V[j, c]  (f64 accumulate,
 * rounded once to f32).  A dense V is the special case range = [0, Mtot) for every column.
 * K is never stored.                                                                      */
int odx_gauss_mmv_f32(const float* X, int64_t ldx, const float* xsq, int64_t n,
                      const float* Z, int64_t ldz, const float* zsq, int D, double sigma,
                      const double* V, int64_t ldv, const int32_t* ranges, int T,
                      float* out, int64_t ldo, odx_stream_t stream);

/* The same block for SMALL problems by direct differences: K_ij = exp(-sum_d (x_id - z_jd)^2 / (2 sigma^2)), the differences
 * formed directly and summed in f64 — the stored f32 entry is the exactly rounded one (no norm cancellation; one thread per
 * entry, O(n M D) on the vector ALU).  The Python host side takes it for blocks of at most 2^24 multiply-adds (toy problems
 * and fixtures, where tiny ill-conditioned fits feel the 1e-6 of the f32-accurate forms); same reference call site as
 * odx_gauss_knm_f32.  No alignment requirement on X / Z. */
int odx_gauss_knm_direct_f32(const float* X, int64_t ldx, int64_t n, const float* Z, int64_t ldz, int64_t M, int D,
                             double sigma, float* K, int64_t ldk, odx_stream_t stream);

/* ---------------------------------------------------------------- A3 / A5 on the f16 matrix cores
 * The same two operations (same reference call sites as odx_gauss_knm_f32 / odx_gauss_mmv_f32 above) with
 * the X Z' contraction on v_mfma_f32_16x16x32_f16 at f32 accuracy: every f32 value is split once into two
 * f16 terms (hi + lo, after a power-of-two scale) and  x.z = hi.hi + hi.lo + lo.hi  accumulates in f32.
 *
 * odx_split_f16 packs an f32 matrix for those kernels.  P: n rows of ldp 4-byte units, ldp % 4 == 0,
 * ldp >= roundup(D, 64); granule t (features 32 t .. 32 t + 31) of a row is 128 contiguous bytes, 32 f16 "hi"
 * then 32 f16 "lo"; features past D are zero.  meta: 2 DEVICE floats; on return meta[0] = the scale that was
 * applied (2^(13 - e) for max |x| = 1.m x 2^e), meta[1] = bits of max |x| (scratch).  Rows gathered from P
 * keep the same meta.                                                                              */
int odx_split_f16(const float* X, int64_t ldx, int64_t n, int D, void* P, int64_t ldp, float* meta,
                  odx_stream_t stream);
int odx_split_f16_premax(const float* X, int64_t ldx, int64_t n, int D, void* P, int64_t ldp, float* meta, odx_stream_t stream);
/* odx_set_h2_tile(t) = odx_set_option("h2_tile", t): pins the tile core (128 or 256; 0 = automatic, the default) for tests and
 * measurements.  odx_gauss_h2_tile: side of the square output tile odx_gauss_knm_h2 uses for an n x M block under the
 * current setting; 0 for an empty block. */
int odx_set_h2_tile(int tile);
int odx_gauss_h2_tile(int64_t n, int64_t M);
int odx_gauss_knm_h2(const void* PX, int64_t ldpx, const float* metax, const float* xsq, int64_t n,
                     const void* PZ, int64_t ldpz, const float* metaz, const float* zsq, int64_t M, int D,
                     double sigma, float* K, int64_t ldk, odx_stream_t stream);
/* Build with the right-hand side of the fit fused in (A4: b = K_nM' (y / n), the first of the passes
 * FALKONWrapper_with_centers_selection_incore.py:56-68 -> InCoreFalkon.fit makes over K_nM): writes K as
 * odx_gauss_knm_h2 does and ktw[j] = sum_i K_ij w_i (f64, fixed-order sums), so that pass over the stored K_nM is
 * never made.  Runs on the 256 x 256 tile core at every size.  w: n f64 weights; ktw: M f64. */
int64_t odx_gauss_knm_h2_rhs_workspace_bytes(int64_t n, int64_t M);
int odx_gauss_knm_h2_rhs(const void* PX, int64_t ldpx, const float* metax, const float* xsq, int64_t n,
                         const void* PZ, int64_t ldpz, const float* metaz, const float* zsq, int64_t M, int D,
                         double sigma, float* K, int64_t ldk, const double* w, double* ktw, void* workspace,
                         int64_t workspace_bytes, odx_stream_t stream);
/* Two tile cores serve both calls: 256 x 256 outputs per workgroup (half the L2 traffic per product) once a launch
 * has >= 512 such tiles, 128 x 128 below that; the option "h2_tile" (odx_set_option) pins one.
 * Scoring is tiled over (row block, group of 512 centre columns); the f64 partial sums of the groups pass through
 * `workspace` (odx_gauss_mmv_h2_workspace_bytes(n, max_range, T)) and are added in fixed order.  max_range must be
 * >= the longest per-column row range (ranges[2c+1] - ranges[2c]); pass the number of rows of V when unknown. */
int64_t odx_gauss_mmv_h2_workspace_bytes(int64_t n, int64_t max_range, int T);
int odx_gauss_mmv_h2(const void* PX, int64_t ldpx, const float* metax, const float* xsq, int64_t n,
                     const void* PZ, int64_t ldpz, const float* metaz, const float* zsq, int64_t max_range, int D,
                     double sigma, const double* V, int64_t ldv, const int32_t* ranges, int T,
                     float* out, int64_t ldo, void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* ---------------------------------------------------------------- A4: CG pass on stored K
 * falkon's incore_fdmmv on the stored K_nM (selected by store_kernel_d_threshold=250,
 * FALKONWrapper_with_centers_selection_incore.py:56):  out = K' (K v + w)  for one shard
 * of rows.  K (n x M) f32; v (M) f64 or NULL; w (n) f64 or NULL; out (M) f64.
 * One read of K per call.  Deterministic (fixed-order slab reduction).                  */
int64_t odx_knm_fwd_bwd_workspace_bytes(int64_t n, int64_t M);
/* The pass kernel is persistent (one workgroup per CU for M > 4096) and would otherwise hold every CU for its whole
 * run (6.5 ms at n = 1e6, M = 1e4).  Reserving `cus` CUs (0 = none, the default) lets small kernels of concurrent
 * streams — the preconditioner of the next class — make progress during the passes; a process-wide setting. */
int odx_set_pass_reserved_cus(int cus);
int odx_knm_fwd_bwd(const float* K, int64_t ldk, int64_t n, int64_t M,
                    const double* v, const double* w, double* out,
                    void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* Two products from ONE read of K:  out = K' (K v),  out2 = K' (K v2)  — the pass of a CG step whose periodic
 * full-residual recomputation (falkon: every 10th iteration, R = B - W x) rides along: W x_new = W x_old + a W p.
 * Available where both vectors fit in LDS (M <= 10 000); the workspace query returns ODX_ERR_UNSUPPORTED (< 0)
 * otherwise and callers issue two odx_knm_fwd_bwd passes. */
int64_t odx_knm_fwd_bwd2_workspace_bytes(int64_t n, int64_t M);
int odx_knm_fwd_bwd2(const float* K, int64_t ldk, int64_t n, int64_t M, const double* v, const double* v2,
                     double* out, double* out2, void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* Name of the kernel (as rocprofv3 lists it) a pass over a block of M columns stored as `fmt` launches, nv = 1 (one vector)
 * or 2 (two vectors from one read): the library's own dispatch rule, for tools that label measurements.  "" if no
 * configuration covers the shape.  The string lives in thread-local storage until the thread's next call.  */
const char* odx_knm_pass_kernel_name(int64_t M, int fmt, int nv);
/* ---------------------------------------------------------------- A4: compact storage of the stored K_nM
 * The CG passes above are HBM-bound at the chip's copy rate, so their time is the bytes per entry of the stored block
 * (falkon keeps it in the data's dtype: f32, FALKONWrapper_with_centers_selection_incore.py:56-68; config/defaults.py:466).
 * Formats (tools/precision_storage_study.py has the effect of each on the fitted alpha):
 *   ODX_KNM_F32   n x ld floats, ld = roundup(M, 4)                                          4 B / entry (parity format)
 *   ODX_KNM_U24   24-bit fixed point on [0, 1]: q = round(K 2^24) as a u16 plane (q >> 8, `K`) and a u8 plane (q & 255,
 *                 `Klo`), both n x ld, ld = roundup(M, 8).  Absolute step 2^-24 = f32's own on [0.5, 1)   3 B / entry
 *   ODX_KNM_BF16  K rounded to bf16, n x ld u16, ld = roundup(M, 8)  (BASELINE config 2's throughput storage)  2 B / entry
 * Pad columns [M, ld) are written as zero.  odx_gauss_knm_h2_store builds a block in any of the three on the 256 x 256
 * tile core (16-byte stores per lane) and, when w is given, leaves ktw = K' w over the values it STORED (the fused
 * right-hand side; workspace as odx_gauss_knm_h2_rhs_workspace_bytes).  odx_knm_fwd_bwd_q / odx_knm_fwd_bwd2_q are
 * odx_knm_fwd_bwd / odx_knm_fwd_bwd2 on the two compact formats (M <= 20440; two vectors: 4096 < M <= ~10 000). */
#define ODX_KNM_F32 0
#define ODX_KNM_U24 1
#define ODX_KNM_BF16 2
int64_t odx_knm_ld(int64_t M, int fmt);
int64_t odx_knm_bytes(int64_t n, int64_t M, int fmt);
int odx_gauss_knm_h2_store(const void* PX, int64_t ldpx, const float* metax, const float* xsq, int64_t n,
                           const void* PZ, int64_t ldpz, const float* metaz, const float* zsq, int64_t M, int D,
                           double sigma, int fmt, void* K, int64_t ldk, void* Klo, int64_t ldlo, const double* w,
                           double* ktw, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
int64_t odx_knm_fwd_bwd_q_workspace_bytes(int64_t n, int64_t M, int fmt);
int odx_knm_fwd_bwd_q(const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t n, int64_t M,
                      const double* v, const double* w, double* out, void* workspace, int64_t workspace_bytes,
                      odx_stream_t stream);
int64_t odx_knm_fwd_bwd2_q_workspace_bytes(int64_t n, int64_t M, int fmt);
int odx_knm_fwd_bwd2_q(const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t n, int64_t M,
                       const double* v, const double* v2, double* out, double* out2, void* workspace,
                       int64_t workspace_bytes, odx_stream_t stream);

/* ---------------------------------------------------------------- A3 / A5 with the fp8 contraction (BASELINE config 5)
 * "fp8 (OCP e4m3) inputs to the X Z' MFMA, f32 accumulate, stress / throughput only" (cfg 5; the reference's own
 * dtype is f32, config/defaults.py:466).  Each operand value is rounded ONCE to e4m3 (after a power-of-two scaling of the
 * whole matrix), the products run on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales — a third of the MFMA issue
 * slots of the f16-split kernels at a quarter of the operand bytes.  K entries come out ~1e-3 off, scores ~5e-3
 * (tools/precision_scoring_study.py): NOT a parity path, never the default; odx.options gauss="f8" selects it in the Python backend.
 * odx_split_f8: P8 row-major e4m3 bytes, row stride ldp8 bytes (ldp8 % 16 == 0, ldp8 >= roundup(D, 128)), features past D
 * zero; meta[0] = the scale, meta[1] = max |x| (as odx_split_f16); qsq (n floats, may be NULL) = the squared norms of the
 * ROUNDED rows, |e4m3(s x)|^2 / s^2 — what the f8 kernels take as xsq / zsq, so that d^2 is the squared distance of the
 * rounded points (0 for duplicates) instead of mixing exact norms with a rounded inner product.
 * odx_gauss_knm_f8_store = odx_gauss_knm_h2_store and odx_gauss_mmv_f8 = odx_gauss_mmv_h2 (same workspace queries) on such
 * operands, always on the 256 x 256 tile core. */
int odx_split_f8(const float* X, int64_t ldx, int64_t n, int D, void* P8, int64_t ldp8, float* meta, float* qsq,
                 odx_stream_t stream);
int odx_gauss_knm_f8_store(const void* PX, int64_t ldpx8, const float* metax, const float* xsq, int64_t n,
                           const void* PZ, int64_t ldpz8, const float* metaz, const float* zsq, int64_t M, int D,
                           double sigma, int fmt, void* K, int64_t ldk, void* Klo, int64_t ldlo, const double* w,
                           double* ktw, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
int odx_gauss_mmv_f8(const void* PX, int64_t ldpx8, const float* metax, const float* xsq, int64_t n,
                     const void* PZ, int64_t ldpz8, const float* metaz, const float* zsq, int64_t max_range, int D,
                     double sigma, const double* V, int64_t ldv, const int32_t* ranges, int T,
                     float* out, int64_t ldo, void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* ---------------------------------------------------------------- A4: preconditioner (f64)
 * FalkonPreconditioner.init as run by InCoreFalkon.fit with min_cuda_pc_size_*=0
 * (FALKONWrapper_with_centers_selection_incore.py:56):
 *   L_T = chol_lower(K_MM + eps*M*I) (T = L_T'),  L_A = chol_lower(T T'/M + lam*I) (A = L_A'),
 * and their explicit inverses so that the four triangular solves of a CG iteration are
 * row-dot matrix-vector products.
 * Outputs, all M x M f64 row-major with leading dimension ld (ld % 2 == 0, ld >= M):
 *   LTi  = L_T^-1 (lower)   => T^-T v = LTi v
 *   LTit = L_T^-T (upper)   => T^-1 v = LTit v
 *   LAi, LAit likewise for A.
 * Z (M x D) f32 centres.  info (device int32): 0 or index+1 of the first non-positive pivot. */
int64_t odx_falkon_precond_workspace_bytes(int64_t M, int D);
int odx_falkon_precond_f64(const float* Z, int64_t ldz, int64_t M, int D, double sigma,
                           double lam, double eps, double* LTi, double* LTit, double* LAi,
                           double* LAit, int64_t ld, int32_t* info,
                           void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* The same preconditioner for B <= 32 independent classes at once (the reference trains its classes one after the
 * other, OnlineRegionClassifier_incore.py:96-155 — each `classifier.train` call above builds one; the classes are
 * independent, so here every launch of the factorisation chain advances all B).  Z[b] / ldz[b] / M[b]: HOST arrays of
 * the classes' centre pointers (device), leading dimensions and centre counts, M[b] <= Mmax.  out: B blocks,
 * out_stride elements apart (>= 4 Mmax ld), each holding LTi | LTit | LAi | LAit as four Mmax x ld matrices whose
 * leading M[b] x M[b] blocks are class b's factors (the rest is the factor of an identity border).  info: B words. */
int64_t odx_falkon_precond_batched_workspace_bytes(int64_t Mmax, int D, int B);
int odx_falkon_precond_batched_f64(const float* const* Z, const int64_t* ldz, const int64_t* M, int B,
                                   int64_t Mmax, int D, double sigma, double lam, double eps,
                                   double* out, int64_t ld, int64_t out_stride, int32_t* info,
                                   void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* y = op(Tri) x for a triangular M x M f64 matrix, rows as dot products.
 * uplo: 0 = lower (uses columns j <= i), 1 = upper (j >= i).
 * y = alpha * Tri x + beta * z  (z may be NULL when beta == 0; y may alias z).          */
int odx_trmv_f64(const double* Tri, int64_t ld, int64_t M, int uplo, const double* x,
                 double alpha, double beta, const double* z, double* y, odx_stream_t stream);

/* ---------------------------------------------------------------- A4: CG vector updates
 * falkon ConjugateGradient.solve inner step for ONE right-hand side, all f64, length M,
 * scalars kept on device (state[0]=rs_old, state[1]=rs_new, state[2]=stop flag as double):
 *   odx_cg_init  : R = B; P = B; X = 0; rs_old = R.R
 *   odx_cg_step  : a = rs_old / (P.AP + cg_eps); X += a P;
 *                  if (!full_grad) R -= a AP;            (full_grad: caller recomputes R)
 *   odx_cg_finish: rs_new = R.R; stop |= sqrt(rs_new) < tol; P = R + (rs_new/(rs_old+cg_eps)) P;
 *                  rs_old = rs_new
 * Every kernel is a no-op once the stop flag is set (device-side early exit, no host sync). */
int odx_cg_init(const double* B, double* X, double* R, double* P, double* state, int64_t M,
                odx_stream_t stream);
int odx_cg_step(double* X, double* R, const double* P, const double* AP, double* state,
                double cg_eps, int full_grad, int64_t M, odx_stream_t stream);
int odx_cg_finish(const double* R, double* P, double* state, double cg_eps, double tol,
                  int64_t M, odx_stream_t stream);
/* R = B - (AX + a AP), a = state[3] (the step just taken): the periodic full residual B - W x_new from W x_old and W p
 * (x_new = x_old + a p), both of which one odx_knm_fwd_bwd2 pass delivers; no-op once the stop flag is up. */
int odx_cg_residual(const double* B, const double* AX, const double* AP, const double* state,
                    double* R, int64_t M, odx_stream_t stream);
/* y = a*x + b*y (f64), used for R = B - mmv(X) */
int odx_axpby_f64(double a, const double* x, double b, double* y, int64_t M, odx_stream_t stream);

/* The whole CG loop of one unsharded fit as a single call (the launches above in falkon's order; see
 * online-detection_amd/odx/solver.py for the same loop statement by statement, which is also the form used when rows are
 * sharded and a collective sits inside every iteration):
 *   B = A^-T T^-T b0;  maxiter x { AP = A^-T[T^-T K'K(T^-1 A^-1 P)/n_total + lam A^-1 P]; step; (every
 *   full_gradient_every-th: R = B - mmv(X)); finish };  alpha = T^-1 A^-1 X.
 * K: the stored n x M f32 block; LTi / LTit / LAi / LAit: the four inverse factors of odx_falkon_precond_f64 (ld = ldp);
 * b0 = K'(y / n_total) (M, f64: odx_gauss_knm_h2_rhs or odx_knm_fwd_bwd with w); alpha: M f64 out. */
int64_t odx_falkon_cg_workspace_bytes(int64_t n, int64_t M);
int odx_falkon_cg_f64(const float* K, int64_t ldk, int64_t n, int64_t M, const double* LTi, const double* LTit,
                      const double* LAi, const double* LAit, int64_t ldp, const double* b0, double n_total,
                      double lam, int maxiter, int full_gradient_every, double cg_epsilon, double cg_tolerance,
                      double* alpha, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* The same loop for B <= 32 independent fits in lock step — the classes of one Minibootstrap round, which the reference
 * trains one after the other (OnlineRegionClassifier_incore.py:96-155): every launch carries the class as a grid
 * dimension, so the round costs the launches of ONE fit; per class the arithmetic (and so alpha, bit for bit) is
 * odx_falkon_cg_f64's.  HOST arrays K / ldk / n / M / n_total describe the classes' stored blocks; P: the output block
 * of odx_falkon_precond_batched_f64 (four p_rows x ldp factors per class, p_stride apart); b0 and alpha: B vectors
 * vstride apart.  The classes must share one pass configuration (same bracket of M): the workspace query returns < 0
 * otherwise and callers fall back to one odx_falkon_cg_f64 per class. */
int64_t odx_falkon_cg_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M);
int odx_falkon_cg_batched_f64(int B, const float* const* K, const int64_t* ldk, const int64_t* n, const int64_t* M,
                              const double* P, int64_t ldp, int64_t p_rows, int64_t p_stride,
                              const double* b0, int64_t vstride, const double* n_total, double lam,
                              int maxiter, int full_gradient_every, double cg_epsilon, double cg_tolerance,
                              double* alpha, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* The same lock-step loops over blocks stored in a compact format (odx_gauss_knm_h2_store; fmt = ODX_KNM_U24: Khi / Klo are the
 * u16 / u8 planes, ODX_KNM_BF16: Khi the bf16 words, Klo / ldlo ignored): class b's passes are odx_knm_fwd_bwd_q's bit for bit.
 * Replaces, for a class batch, the per-class CG loops the reference's falkon runs one fit at a time
 * (FALKONWrapper_with_centers_selection_incore.py:56-68) when the stored blocks are not floats. */
int64_t odx_falkon_cg_batched_q_workspace_bytes(int B, const int64_t* n, const int64_t* M, int fmt);
int odx_falkon_cg_batched_q_f64(int B, const void* const* Khi, const int64_t* ldk, const void* const* Klo, const int64_t* ldlo,
                                int fmt, const int64_t* n, const int64_t* M, const double* P, int64_t ldp, int64_t p_rows,
                                int64_t p_stride, const double* b0, int64_t vstride, const double* n_total, double lam,
                                int maxiter, int full_gradient_every, double cg_epsilon, double cg_tolerance, double* alpha,
                                void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* ---------------------------------------------------------------- dense f64 building blocks
 * (exported for the parity tests and for the RLS path)                                   */
/* C (m x n) = alpha * A (m x k) B' (n x k) + beta * C ; flags below */
#define ODX_GEMM_LOWER_ONLY 1   /* skip output tiles strictly above the diagonal */
#define ODX_GEMM_A_UPPER 2      /* A[i,k] == 0 for k < i  */
#define ODX_GEMM_B_UPPER 4      /* B[j,k] == 0 for k < j  */
#define ODX_GEMM_A_LOWER 8      /* A[i,k] == 0 for k > i  */
#define ODX_GEMM_B_LOWER 16     /* B[j,k] == 0 for k > j  */
#define ODX_GEMM_STORE_T 32     /* store C transposed: Ct[j,i] (ldc applies to Ct) */
int odx_gemm_nt_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                    int64_t ldc, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                    int flags, odx_stream_t stream);
int odx_gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                    int64_t ldc, int64_t m, int64_t n, int64_t k, float alpha, float beta,
                    int flags, odx_stream_t stream);
/* in-place lower Cholesky of the lower triangle of A (M x M, f64); strict upper is zeroed.
 * Also returns the inverses of the nb x nb diagonal blocks if Dinv != NULL.             */
int64_t odx_potrf_workspace_bytes(int64_t M);
int odx_potrf_f64(double* A, int64_t lda, int64_t M, int32_t* info, void* workspace,
                  int64_t workspace_bytes, odx_stream_t stream);
/* Li = L^-1 (lower), Lit = L^-T (upper) from a lower-triangular L (strict upper == 0). */
int64_t odx_trtri_workspace_bytes(int64_t M);
int odx_trtri_f64(const double* L, int64_t ldl, int64_t M, double* Li, double* Lit, int64_t ld,
                  void* workspace, int64_t workspace_bytes, odx_stream_t stream);
int odx_convert_f32_f64(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows,
                        int64_t cols, odx_stream_t stream);
int odx_convert_f64_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows,
                        int64_t cols, odx_stream_t stream);

/* ---------------------------------------------------------------- A7: RLS box regressors
 * RegionRefinerTrainer.train / solve
 * (src/modules/region-refiner/region_refiner_trainer/train_region_refiner.py:25-119):
 * per class, f64:  G = [X 1]' [X 1] + lam I,  R = chol(G),  w_k = R^-T R^-1 [X 1]' y_k.
 * Step 1 (per row shard): gather rows idx[0..nc) of X (f32, n x D), append the bias column,
 *   cast to f64 and accumulate  G (D1 x D1, lower, D1 = D + 1)  and  XtY (4 x D1)  with
 *   Yt (4 x nc) f64 already whitened by the caller.  G and XtY are ACCUMULATED INTO
 *   (beta = 1) so shards / chunks can be summed (and all-reduced) before step 2.
 * Step 2: add lam to the diagonal, Cholesky, 4 right-hand sides -> W (4 x D1).           */
int64_t odx_rls_gram_workspace_bytes(int64_t nc, int D);
int odx_rls_gram_f64(const float* X, int64_t ldx, int D, const int64_t* idx, int64_t nc,
                     const double* Yt, int64_t ldy, double* G, int64_t ldg, double* XtY,
                     int64_t ldxy, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
int64_t odx_rls_solve_workspace_bytes(int D);
int odx_rls_solve_f64(double* G, int64_t ldg, int D, double lam, const double* XtY, int64_t ldxy,
                      double* W, int64_t ldw, int32_t* info, void* workspace,
                      int64_t workspace_bytes, odx_stream_t stream);

/* P (nc x 4, ldp) = [X[idx] 1] W'  (f64 accumulate; idx may be NULL = identity): the training
 * residuals behind the per-sample 'losses' of train_region_refiner.py:116, and the f64 form
 * of the apply  F W[:-1] + W[-1]  of predict_regions.py:45-46.                              */
int odx_rls_predict_rows_f64(const float* X, int64_t ldx, int D, const int64_t* idx, int64_t nc,
                             const double* W, int64_t ldw, double* P, int64_t ldp, odx_stream_t stream);
/* The same for the rows of up to 32 classes with one launch: idx holds the row ids class after class (class c's rows start at
 * seg_start[c], a HOST array of C ascending entries beginning with 0), W (C, 4, ldw) w_stride apart; P (total, 4). */
int odx_rls_predict_rows_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx, const int64_t* seg_start, int C,
                                     int64_t total, const double* W, int64_t ldw, int64_t w_stride, double* P, int64_t ldp,
                                     odx_stream_t stream);

/* The regressors of C <= 32 classes at once (the reference trains them one after the other,
 * train_region_refiner.py:27-98; they are independent): every kernel takes the class as a grid dimension.
 *   gram:  idx_pad (DEVICE, npad entries): the row ids class after class, each class's segment starting at a multiple of
 *          16 and padded with -1; seg_off / seg_len (HOST, C entries): start and true length of the segments; Yt (4 x ldy):
 *          the whitened targets in the same padded order.  G (C blocks g_stride apart, (D+1) x ldg lower) += Gram,
 *          XtY (C blocks xy_stride apart, 4 x ldxy) += Yt [X 1] — straight from the f32 rows when D % 8 == 0, ldx % 4 == 0
 *          and X is 16-byte aligned (one TN Gram launch on the f64 matrix cores + one sweep for X'Y); otherwise, or with
 *          the test hook "rls_force_nt_gram", through a transposed f64 copy of the rows and the NT GEMM.
 *   solve: per class the result of odx_rls_solve_f64 (to rounding: Cholesky, then block substitution with the factor — one
 *          workgroup per class — instead of the explicit inverse; the test hook "rls_force_inverse_solve" keeps that form); W: C blocks w_stride
 *          apart, 4 x ldw; info: C words.
 * Two calls so that a row-sharded caller can all-reduce G and XtY in between. */
int64_t odx_rls_gram_batched_workspace_bytes(int64_t npad, int D);
/* The gram step in two calls, so that the Grams (which need the rows only) can run while the caller is still deriving the
 * targets (train_region_refiner.py:58-68: mean, covariance, its eigen-decomposition, the whitening): odx_rls_gram_batched_f64
 * with Yt == NULL and XtY == NULL forms the Grams alone, odx_rls_xty_batched_f64 (same workspace size) then adds Yt [X 1] and
 * the bias row / column of G.  Both need the rows form (odx_rls_rows_form(X, ldx, D) != 0: D % 8 == 0, ldx % 4 == 0, X 16-byte
 * aligned, "rls_force_nt_gram" unset) and return ODX_ERR_UNSUPPORTED without it — the single call with Yt always works. */
int odx_rls_rows_form(const float* X, int64_t ldx, int D);
int odx_rls_xty_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                            const int64_t* seg_off, const int64_t* seg_len, int C, const double* Yt, int64_t ldy,
                            double* G, int64_t ldg, int64_t g_stride, double* XtY, int64_t ldxy, int64_t xy_stride,
                            void* workspace, int64_t workspace_bytes, odx_stream_t stream);
int odx_rls_gram_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                             const int64_t* seg_off, const int64_t* seg_len, int C,
                             const double* Yt, int64_t ldy, double* G, int64_t ldg, int64_t g_stride,
                             double* XtY, int64_t ldxy, int64_t xy_stride,
                             void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* The same without waiting for the statistics at all (train_region_refiner.py:58-68 derives them from the targets only): the
 * whitening is linear, Yw = (Y - 1 mu') T, so X' Yw = (X' Y - (X' 1) mu') T and 1' Yw = 0.  odx_rls_gram_raw_batched_f64 forms the
 * Grams AND the raw products O5 (C, 5, ldo) f64 = [Y 1]' X (rows 0 .. 3: the un-whitened targets Yraw (n, ldyr >= 4) f32 by row
 * id; row 4: the column sums; columns 0 .. D - 1) in ONE sweep over the rows — the targets' products ride on the vector ALU under
 * the Gram's matrix instructions; odx_rls_fold_whitened_f64 then writes XtY (4 x (D + 1) per class) and adds the Gram's bias row
 * from O5, stats (C, 9, 4) f64 = [mu; T; T_inv] per class and cnt (C) f64 = rows per class.  Rows form only. */
/* The padded row-id array the batched calls take, and its inverse maps, in one launch: run (total) = the row ids class after class
 * (seg_len[k] of class k); idx_pad (npad) gets class k's ids from seg_off[k] on and -1 elsewhere; for the i-th id of run gid[i] =
 * its class slot, pos[i] = its rank inside the class, dest[i] = its position in idx_pad; lens (C) = seg_len (all int64, device). */
int odx_rls_pad_index(const int64_t* run, int64_t total, const int64_t* seg_off, const int64_t* seg_len, int C, int64_t npad,
                      int64_t* idx_pad, int64_t* gid, int64_t* pos, int64_t* dest, int64_t* lens, odx_stream_t stream);
int odx_rls_gram_raw_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                                 const int64_t* seg_off, const int64_t* seg_len, int C, const float* Yraw, int64_t ldyr,
                                 double* G, int64_t ldg, int64_t g_stride, double* O5, int64_t ldo, odx_stream_t stream);
int odx_rls_fold_whitened_f64(const double* O5, int64_t ldo, int D, int C, const double* stats, const double* cnt,
                              double* G, int64_t ldg, int64_t g_stride, double* XtY, int64_t ldxy, int64_t xy_stride,
                              odx_stream_t stream);
int64_t odx_rls_solve_batched_workspace_bytes(int D, int C);
int odx_rls_solve_batched_f64(double* G, int64_t ldg, int64_t g_stride, int D, int C, double lam,
                              const double* XtY, int64_t ldxy, int64_t xy_stride,
                              double* W, int64_t ldw, int64_t w_stride, int32_t* info,
                              void* workspace, int64_t workspace_bytes, odx_stream_t stream);

/* ---------------------------------------------------------------- A11: RoIAlign forward, NMS
 * The two maskrcnn_benchmark CUDA ops on the on-line path:
 *   Pooler -> ROIAlign(14x14, spatial_scale 1/16, sampling_ratio 0)
 *     (mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:21-25,47)
 *   boxlist_nms (mrcnn_modified/modeling/rpn/inference.py:116-121;
 *     src/modules/accuracy-evaluator/OnlineDetectionPostProcessor.py:55-57)
 * feat (N, C, H, W) f32; rois (R, 5) = (batch index, x1, y1, x2, y2) in image pixels;
 * out (R, C, PH, PW), PH * PW <= 256; legacy (aligned = False) sampling, sampling_ratio 0 =
 * adaptive ceil(roi / bins).                                                               */
int odx_roi_align_fwd_f32(const float* feat, int N, int C, int H, int W, const float* rois, int R,
                          float spatial_scale, int PH, int PW, int sampling_ratio, float* out,
                          odx_stream_t stream);
/* odx_split_f16 of the 3 x 3 neighbourhood matrix of an NHWC row matrix Y (R * H * W rows of C channels, C % 8 == 0): row
 * (r, h, w) of P holds, tap (ky, kx) after tap, the channels of Y's row (r, h + ky - 1, w + kx - 1), zeros outside the map
 * — the operand of a 3 x 3 convolution with padding 1 run as a GEMM (K = 9 C), written in the packed form directly
 * (P: (R H W) x ldp 4-byte units, ldp >= roundup(9 C, 64); meta as in odx_split_f16).  */
int odx_split_f16_taps3x3(const float* Y, int64_t ldy, int64_t R, int H, int W, int C, void* P, int64_t ldp,
                          float* meta, odx_stream_t stream);
/* out (m x n) f32 = act(A B' + bias[col] + residual), A (m x K) and B (n x K) handed over in the packed two-term f16 form
 * of odx_split_f16 (with their meta words): the f32 product at f32 accuracy on the f16 matrix cores (3 MFMAs per product,
 * the Gaussian kernels' tile cores).  Replaces the convolutions of ResNet50Conv5ROIFeatureExtractor's head
 * (roi_box_feature_extractors.py:26-52) run as GEMMs over (RoIs x positions, channels) rows: bias = the folded frozen batch
 * norm, residual = the block's identity branch, relu = 1 for the ReLU behind it.  bias / residual may be NULL.  */
int odx_gemm_h2_f32(const void* PA, int64_t ldpa, const float* metaa, int64_t m, const void* PB, int64_t ldpb,
                    const float* metab, int64_t n, int K, const float* bias, const float* residual, int64_t ldr,
                    int relu, float* out, int64_t ldo, odx_stream_t stream);
/* The same product for a CHAIN of layers — the ResNet trunk (mrcnn_modified/modeling/detector/generalized_rcnn_getProposals.py:60
 * -> self.backbone; maskrcnn_benchmark's ResNet stages, STRIDE_IN_1X1) and the RPN head's 3 x 3 convolution
 * (mrcnn_modified/modeling/rpn/rpn.py:164-170) run as GEMMs over NHWC rows, as the conv5 head is: the launch also leaves
 * max |out| in out_meta[1] (IEEE bits of a non-negative float; must be 0 on entry, out_meta[0] is not touched), which is what
 * odx_split_f16_premax / odx_split_f16_taps3x3_premax scale the next layer's operand by, so no layer reads its input once
 * more for a maximum. */
int odx_gemm_h2_max_f32(const void* PA, int64_t ldpa, const float* metaa, int64_t m, const void* PB, int64_t ldpb,
                        const float* metab, int64_t n, int K, const float* bias, const float* residual, int64_t ldr,
                        int relu, float* out, int64_t ldo, float* out_meta, odx_stream_t stream);
int odx_split_f16_taps3x3_premax(const float* Y, int64_t ldy, int64_t R, int H, int W, int C, void* P, int64_t ldp,
                                 float* meta, odx_stream_t stream);
/* The 3 x 3 convolution of such a chain WITHOUT the 9 x neighbourhood matrix: out (R H W x n) = act(taps3x3(Y) B' + bias +
 * residual) with the gather done inside the product's operand loads (LDS-DMA from the neighbour's row, or from a zero row
 * outside the map).  PY: the packed rows of Y (R H W rows of C channels, odx_split_f16's form, row stride ldpy 4-byte units,
 * meta words metay) FOLLOWED BY ONE ALL-ZERO ROW; B (n x 9 C, K index (ky kx c)) packed as for odx_gemm_h2_f32; out_meta
 * as in odx_gemm_h2_max_f32, or NULL.  Served where odx_gemm_h2_taps_supported(m = R H W, n, C, ldpy) returns 1 (the
 * 256 x 256 tile core fills the chip, C % 32 == 0, the packed rows span < 2^31 bytes); elsewhere: odx_split_f16_taps3x3 +
 * odx_gemm_h2_f32.  Same sums in the same order as that pair (bit-identical results). */
int odx_gemm_h2_taps_supported(int64_t m, int64_t n, int C, int64_t ldpy);
int odx_gemm_h2_taps_f32(const void* PY, int64_t ldpy, const float* metay, int64_t R, int H, int W, int C,
                         const void* PB, int64_t ldpb, const float* metab, int64_t n, const float* bias,
                         const float* residual, int64_t ldr, int relu, float* out, int64_t ldo, float* out_meta,
                         void* out_packed, int64_t ldop, float bound_w, float bound_add, const float* residual_meta,
                         odx_stream_t stream);
/* The top-down step of a feature pyramid (maskrcnn_benchmark/modeling/backbone/fpn.py:57-63: last_inner = inner_lateral +
 * F.interpolate(last_inner, scale_factor=2, mode="nearest")) on NHWC rows, in place and in one pass: lat (B H W rows of C, row
 * stride ldl) += top (B Hp Wp rows) at row (h Hp / H, w Wp / W).  f32: meta != NULL receives max |sum| in meta[1] (bits of a
 * non-negative float; zero on entry), what odx_split_f16_premax needs to pack the 3 x 3 output convolution's operand without a
 * pass for the maximum.  16 bits (is_bf16 != 0: bfloat16, else IEEE half): added in f32, rounded once. */
int odx_upsample_add_rows_f32(float* lat, int64_t ldl, const float* top, int64_t ldt, int B, int H, int W, int Hp, int Wp, int C,
                              float* meta, odx_stream_t stream);
int odx_upsample_add_rows_16(void* lat, int64_t ldl, const void* top, int64_t ldt, int is_bf16, int B, int H, int W, int Hp, int Wp,
                             int C, odx_stream_t stream);
/* A layer of such a chain that writes its output AS the next layer's operand: odx_gemm_h2_max_f32 which also
 * (out_packed != NULL) stores the packed two-term split of its output straight from the accumulators — rows of ldop 4-byte
 * units (ldop % 4 == 0, ldop >= roundup(n, 64), columns n .. roundup(n, 64) zero), out_meta[0] = the scale, out_meta[1] =
 * max |out| — so no split pass is made over the activation; `out` may then be NULL (an activation only products read gets no
 * f32 copy).  The scale has to be fixed before the maximum is known; it comes from the caller's bound
 *     max |out| <= metaa[1] * bound_w + bound_add + residual_meta[1]
 * (metaa[1] = max |A|; bound_w >= sqrt(K) max_j |B_j|_2 by Cauchy-Schwarz, bound_add >= max |bias|; residual_meta: the meta
 * words of the residual's own packing / producer, NULL without a residual).  A bound 2^b above the true maximum costs b of
 * the ~17 binades below the maximum in which the split carries its full 22 bits; the absolute error of an entry stays below
 * 2^(b-39) of the maximum.  odx_gemm_h2_taps_f32 takes the same trailing arguments (out_packed NULL: as before).
 * odx_taps3x3_packed: odx_split_f16_taps3x3 for rows that are already packed (a gather of 16-byte pieces, same meta words)
 * — for the layers odx_gemm_h2_taps_supported does not serve. */
int odx_gemm_h2_chain_f32(const void* PA, int64_t ldpa, const float* metaa, int64_t m, const void* PB, int64_t ldpb,
                          const float* metab, int64_t n, int K, const float* bias, const float* residual, int64_t ldr,
                          int relu, float* out, int64_t ldo, float* out_meta, void* out_packed, int64_t ldop,
                          float bound_w, float bound_add, const float* residual_meta, odx_stream_t stream);
int odx_taps3x3_packed(const void* PY, int64_t ldpy, int64_t R, int H, int W, int C, void* P, int64_t ldp,
                       odx_stream_t stream);
/* The same layers for a forward run in a 16-bit type (BASELINE config 2's bf16; the reference's own dtype is f32,
 * config/defaults.py:466): out (m x n) = act(A B' + bias[col] + residual) for plain row-major bf16 (is_bf16 = 1) or f16
 * operands A (m x K), B (n x K) — lda / ldb in ELEMENTS, multiples of 8, >= roundup(K, 128), the elements beyond K zero;
 * 16-byte aligned.  One MFMA term per product (v_mfma_f32_16x16x32_bf16 / _f16), sums in f32; bias f32; out / residual are
 * f32 (out_16 = 0) or the operands' type (out_16 = 1: bias and residual are added in f32, ONE rounding).  */
int odx_gemm_b16(const void* A, int64_t lda, int64_t m, const void* B, int64_t ldb, int64_t n, int K, int is_bf16,
                 const float* bias, const void* residual, int64_t ldr, int relu, void* out, int64_t ldo, int out_16,
                 odx_stream_t stream);
/* odx_split_f16_taps3x3 for 16-bit rows: P ((R H W) x ldp elements, ldp % 8 == 0, ldp >= 9 C) = the 3 x 3 neighbourhood
 * matrix of Y ((R H W) x C, C % 8 == 0, ldy % 8 == 0), zeros outside the map and beyond 9 C: the operand of odx_gemm_b16.  */
int odx_taps3x3_16(const void* Y, int64_t ldy, int64_t R, int H, int W, int C, void* P, int64_t ldp, odx_stream_t stream);
/* The 3 x 3 layer of a 16-bit forward WITHOUT the neighbourhood matrix (as odx_gemm_h2_taps_f32 for f32): out = act(taps3x3(Y) B' +
 * bias + residual), the taps of the 16-bit NHWC rows Y (R H W rows of C channels, C % 64 == 0, row stride ldy elements, FOLLOWED
 * BY ONE ALL-ZERO ROW) gathered inside the product's operand loads; B (n x 9 C) and the rest as for odx_gemm_b16.  Served where
 * odx_gemm_b16_taps_supported(m = R H W, n, C, ldy) returns 1; elsewhere odx_taps3x3_16 + odx_gemm_b16. */
/* Measurement only (tools/ab_mfma_shape.py): odx_gemm_h2_f32 without bias / residual on the same 256 x 256 LDS-DMA loop built from
 * v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_16x16x32_f16 — the A/B of the MFMA shape the round-4 review asked for. */
int odx_debug_gemm_h2_mf32(const void* PA, int64_t ldpa, const float* metaa, int64_t m, const void* PB, int64_t ldpb,
                           const float* metab, int64_t n, int K, float* out, int64_t ldo, odx_stream_t stream);
int odx_gemm_b16_taps_supported(int64_t m, int64_t n, int C, int64_t ldy);
int odx_gemm_b16_taps(const void* Y, int64_t ldy, int64_t R, int H, int W, int C, const void* B, int64_t ldb, int64_t n,
                      int is_bf16, const float* bias, const void* residual, int64_t ldr, int relu, void* out, int64_t ldo,
                      int out_16, odx_stream_t stream);
/* The same bins for a consumer that starts with a stride-`step` 1 x 1 convolution (ResNet50Conv5ROIFeatureExtractor's
 * head, roi_box_feature_extractors.py:26-52 with STRIDE_IN_1X1): only the bins (ph, pw) with ph % step == pw % step == 0,
 * as rows of an (R * ceil(PH / step) * ceil(PW / step), C) matrix (NHWC) — a quarter of the grid at 14 x 14, step 2.  */
int odx_roi_align_rows_f32(const float* feat, int N, int C, int H, int W, const float* rois, int R,
                           float spatial_scale, int PH, int PW, int sampling_ratio, int step, float* out_rows,
                           odx_stream_t stream);
/* odx_roi_align_rows_f32 reading the map as the NHWC row matrix the trunk's GEMMs write (N * H * W rows of C channels, row
 * stride ldf floats; C % 4 == 0, ldf % 4 == 0, 16-byte aligned) — same bins, sample positions and sums per channel. */
int odx_roi_align_rows_nhwc_f32(const float* feat_rows, int64_t ldf, int N, int C, int H, int W, const float* rois, int R,
                                float spatial_scale, int PH, int PW, int sampling_ratio, int step, float* out_rows,
                                odx_stream_t stream);
/* Multi-level RoIAlign over an FPN pyramid: maskrcnn_benchmark's Pooler as FPN2MLPFeatureExtractor uses it
 * (mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:61-68,79; config/defaults.py:214-219 with the
 * R-50-FPN values POOLER_SCALES (1/4 .. 1/32), POOLER_RESOLUTION 7, POOLER_SAMPLING_RATIO 2).  feats[l] (N, C, H[l], W[l]),
 * scales[l] = 1 / stride of level l (halving from level to level; at most 4 levels); each RoI is pooled from the level
 *   clamp(floor(4 + log2(sqrt((x2 - x1 + 1)(y2 - y1 + 1)) / 224 + 1e-6)), -log2 scales[0], -log2 scales[levels-1]) + log2 scales[0]
 * with ONE launch for all RoIs (the reference: one RoIAlign per level over a boolean selection, then a scatter).
 * out (R, C, PH, PW); level_out (R ints, may be NULL) receives each RoI's level.  feats / H / W / scales are HOST arrays. */
int odx_roi_align_fpn_f32(const float* const* feats, const int* H, const int* W, const float* scales, int levels,
                          int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out,
                          int* level_out, odx_stream_t stream);
/* odx_roi_align_fpn_f32 reading the levels as the NHWC row matrices the pyramid's row GEMMs write (level l: N H_l W_l rows of C
 * channels, C % 4 == 0) and writing out (R, PH PW C): the rows fc6 multiplies when its weight's columns are ordered (ph, pw, c).
 * Same level rule, sample positions and sums per channel. */
int odx_roi_align_fpn_nhwc_f32(const float* const* feats, const int* H, const int* W, const float* scales, int levels,
                               int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out_rows,
                               int* level_out, odx_stream_t stream);
/* The same over a pyramid run natively in 16 bits (is_bf16 != 0: bfloat16, else IEEE half; levels 8-byte aligned, C % 4 == 0): the
 * samples are decoded to f32, everything after is the f32 entry's arithmetic, the crops leave as f32 rows. */
int odx_roi_align_fpn_nhwc_16(const void* const* feats, int is_bf16, const int* H, const int* W, const float* scales, int levels,
                              int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out_rows,
                              int* level_out, odx_stream_t stream);
/* Greedy NMS over boxes (R, 4) xyxy ALREADY SORTED by descending score, areas with the +1
 * pixel convention: keep[i] = 1 unless an earlier kept box overlaps i with IoU > threshold.  */
int64_t odx_nms_workspace_bytes(int R);
int odx_nms_f32(const float* boxes_sorted, int R, float iou_threshold, unsigned char* keep,
                void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* odx_nms_f32 with an upper bound on the survivors: keep[i] = 1 for the first max_keep of them only — what
 * `nms(...)[:post_nms_top_n]` selects (mrcnn_modified/modeling/rpn/inference.py:116-121) — without visiting the
 * candidates behind the last one.  */
int odx_nms_first_f32(const float* boxes_sorted, int R, float iou_threshold, int max_keep, unsigned char* keep,
                      void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* The per-class NMS loop of the detection post-processing (OnlineDetectionPostProcessor.py:51-66: for every
 * foreground class threshold, boxlist_nms, collect) as ONE launch pair: B independent box sets, set b = counts[b] <= Rmax
 * boxes sorted by descending score in slot b of boxes_sorted (B, Rmax, 4); keep (B, Rmax) u8.  counts is a DEVICE array:
 * no host read between the score threshold and the suppression.  */
int64_t odx_nms_batched_workspace_bytes(int Rmax, int B);
int odx_nms_batched_f32(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold,
                        unsigned char* keep, void* workspace, int64_t workspace_bytes, odx_stream_t stream);
/* odx_nms_batched_f32 with at most max_keep (> 0) survivors per set, the first ones in score order: the RPN proposals of a
 * batch of images (one set per image) with one launch pair (rpn/inference.py:116-121 per image).  */
int odx_nms_batched_first_f32(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold,
                              int max_keep, unsigned char* keep, void* workspace, int64_t workspace_bytes,
                              odx_stream_t stream);
/* RPNPostProcessor.forward_for_single_feature_map up to the suppression (rpn/inference.py:76-115) for B images of one size,
 * one launch: logits (B, A, H, W), deltas (B, 4 A, H, W), anchors (H W A, 4) in grid order (location-major, anchor type minor)
 * -> the k <= min(8192, A H W) best candidates of every image by objectness in descending order (ties: lower flat index
 * (h W + w) A + a first): boxes (B, k, 4) decoded (BoxCoder weights 1, +1 widths, dw / dh clamped at delta_clamp) and clipped to
 * [0, img_w - 1] x [0, img_h - 1], scores (B, k) = sigmoid(logit), index (B, k) flat indices (may be NULL).  */
int odx_rpn_topk_decode_f32(const float* logits, const float* deltas, const float* anchors, int B, int A, int H, int W, int k,
                            float img_w, float img_h, float delta_clamp, float* boxes, float* scores, int32_t* index,
                            odx_stream_t stream);
/* The first P kept boxes of each of B sorted sets (keep flags from odx_nms_*), in order, as a dense (B, P, 4) block; nkept[b]
 * = how many (<= P); slots behind them hold the box (0, 0, 15, 15).  counts (DEVICE, may be NULL = Rmax each).  */
int odx_nms_compact_f32(const float* boxes, const unsigned char* keep, const int32_t* counts, int Rmax, int B, int P, float* out,
                        int32_t* nkept, odx_stream_t stream);

/* Masker / paste_mask_in_image (mrcnn_modified/modeling/roi_heads/mask_head/inference.py:119-191), all detections
 * of one image at once: masks (R, S, S) f32 probabilities, boxes (R, 4) xyxy f32 -> out (R, im_h, im_w) u8 0/1:
 * zero-pad by `padding`, expand the box by (S + 2 padding) / S, truncate to integers, bilinear resize
 * (align_corners = False) to the box, `> thresh`, cropped to the image.                                        */
int odx_paste_masks_u8(const float* masks, const float* boxes, int R, int S, int im_h, int im_w, float thresh,
                       int padding, unsigned char* out, odx_stream_t stream);

/* What follows every convolution of the frozen-batch-norm ResNet trunk (mrcnn_modified/modeling/backbone/resnet.py
 * Bottleneck.forward: conv -> FrozenBatchNorm2d -> (+ identity) -> relu_; the norm folded into the weights and `bias`), in
 * place over a contiguous NCHW f32 map and in ONE pass:  y = act(y + bias[c] (+ residual)),  act = ReLU when relu != 0.
 * residual: same shape as y, or NULL.  The additions are made in the reference's order ((conv + bias) + identity).      */
int odx_bias_act_nchw_f32(float* y, const float* bias, const float* residual, int64_t N, int C, int64_t HW, int relu,
                          odx_stream_t stream);
/* The same over a 16-bit map (a trunk run natively in bf16 / f16): is_bf16 != 0: bfloat16, else IEEE half; every addition is
 * rounded to the map's type, as the separate operators round.                                                            */
int odx_bias_act_nchw_16(void* y, const void* bias, const void* residual, int is_bf16, int64_t N, int C, int64_t HW, int relu,
                         odx_stream_t stream);
/* The stem's tail in one pass (ResNet stem, maskrcnn_benchmark/modeling/backbone/resnet.py: relu(bn1(conv1(x))) -> max_pool2d(3, 2,
 * 1)): relu(x + bias) of the convolution's NCHW output x (B, C, H, W) with the frozen batch norm folded into weights and bias,
 * pooled 3 x 3 / stride 2 / padding 1, written as the NHWC rows (B Ho Wo, C) the stages' row GEMMs read (row stride ldr; Ho =
 * (H - 1) / 2 + 1).  f32: meta != NULL receives max |out| in meta[1] (zero on entry) for odx_split_f16_premax.  16 bits: bias in
 * the map's type, every sum rounded once (odx_bias_act_nchw_16's arithmetic). */
int odx_stem_pool_rows_f32(const float* x, const float* bias, int B, int C, int H, int W, float* rows, int64_t ldr, float* meta,
                           odx_stream_t stream);
int odx_stem_pool_rows_16(const void* x, const void* bias, int is_bf16, int B, int C, int H, int W, void* rows, int64_t ldr,
                          odx_stream_t stream);


/* ---------------------------------------------------------------- A11 / A12: harvest labelling
 * The device work in front of a harvester's host read (odx/harvest.py RPNHarvester.prepare / DetectorHarvester.prepare; the
 * reference's rpn_getProposals.py:265-449 and box_head_getProposals.py:117-226 label anchors / proposals against the ground-truth
 * boxes with ~45 tensor operations per image): one or two launches, the tensor form's arithmetic operation by operation in f32
 * (no fused multiply-add: the flags compare overlaps with thresholds), maxima keep the FIRST maximum.  G <= 64 boxes per image.
 *
 * odx_rpn_label_f32: n visible anchors (anchors (n, 4), cls (n) int64 anchor types 0..A-1) against gt (G, 4): ious (n) best overlap,
 * assoc (n, 4) the box of best overlap, neg_mask / over (n) bytes (overlap < neg_thr, > pos_thr), extra (G, n) bytes (anchor i is
 * associated with a box equal to box j and has the best overlap of all such anchors); counters (int32): [A candidates per type]
 * [G over-threshold anchors whose box shares a coordinate with box j][G box j has anchors][A over-threshold anchors per type]
 * [G x A extras of box j per type].  workspace: G uint32.
 * odx_det_label_f32: R proposals against gt (G, 4) with classes labels0 (G) int32 (0-based): both clamped to the image (prop (R, 4)
 * out), overlap (R, C) per-class maximum overlap, sel (G, R) bytes (proposal r regresses onto box j: its class's overlap > reg_min
 * and j the first box of largest, positive overlap), cmask (R, n_in) bytes (overlap of listed class in_image[k] < neg_thr);
 * counters (int32): [G pairs per box][n_in candidates per listed class].
 * odx_box_targets_f32: the regression targets of n (example, target) box pairs, ((gx - sx) / sw, (gy - sy) / sh, log(gw / sw),
 * log(gh / sh)) with widths x2 - x1 + 1. */
int odx_rpn_label_f32(const float* gt, int G, const float* anchors, const int64_t* cls, int n, int A, float neg_thr,
                      float pos_thr, float* ious, float* assoc, unsigned char* neg_mask, unsigned char* over,
                      unsigned char* extra, int32_t* counters, void* workspace, odx_stream_t stream);
int odx_det_label_f32(const float* gt, const int32_t* labels0, int G, const float* proposals, int R, int C, float img_w,
                      float img_h, float reg_min, float neg_thr, const int32_t* in_image, int n_in, float* prop, float* overlap,
                      unsigned char* sel, unsigned char* cmask, int32_t* counters, odx_stream_t stream);
int odx_box_targets_f32(const float* examples, const float* targets, int64_t n, float* out, odx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ODX_H */
