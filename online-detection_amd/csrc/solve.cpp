// The conjugate-gradient loop of one unsharded FALKON fit as ONE library call: the same sequence of libodx launches as
// online-detection_amd/odx/solver.py::falkon_fit issues one by one from Python (which stays the reference statement of
// the loop and the only form for row shards, where a collective sits inside every iteration).  A fit of the reference
// regime (M ~ 2000, n ~ 1e4) is ~ 250 launches of a few microseconds each; issued from here the host needs ~ 0.3 ms for
// them instead of ~ 3 ms, which is what bounds the class-stream minibootstrap.
//
// Reference: InCoreFalkon.fit -> ConjugateGradient.solve at FALKONWrapper_with_centers_selection_incore.py:56-68
// (falkon's defaults: 20 iterations, residual recomputed every 10).
#include "odx_internal.h"

using namespace odx;

extern "C" int64_t odx_falkon_cg_workspace_bytes(int64_t n, int64_t M) {
  if (M <= 0) return 0;
  const int64_t pass = odx_knm_fwd_bwd_workspace_bytes(n > 0 ? n : 1, M);
  if (pass < 0) return pass;
  const int64_t Mp = round_up(M, 2);
  const int64_t pass2 = odx_knm_fwd_bwd2_workspace_bytes(n > 0 ? n : 1, M);      // < 0: no two-vector pass at this M
  return (14 * Mp + 4) * (int64_t)sizeof(double) + round_up(pass2 > pass ? pass2 : pass, 16);
}

extern "C" int odx_falkon_cg_f64(const float* K, int64_t ldk, int64_t n, int64_t M, const double* LTi, const double* LTit,
                                 const double* LAi, const double* LAit, int64_t ldp, const double* b0, double n_total,
                                 double lam, int maxiter, int full_gradient_every, double cg_epsilon, double cg_tolerance,
                                 double* alpha, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && n >= 0 && K && LTi && LTit && LAi && LAit && b0 && alpha && n_total > 0 && maxiter >= 0 &&
                  full_gradient_every > 0,
              "odx_falkon_cg_f64: bad argument");
  const int64_t need = odx_falkon_cg_workspace_bytes(n, M);
  ODX_REQUIRE(need >= 0, "odx_falkon_cg_f64: M = %lld is outside the range of the pass kernels", (long long)M);
  if (workspace == nullptr || workspace_bytes < need || !aligned16(workspace)) {
    set_error("odx_falkon_cg_f64: workspace too small or not 16-byte aligned");
    return ODX_ERR_WORKSPACE;
  }
  const int64_t Mp = round_up(M, 2);
  double* w = static_cast<double*>(workspace);
  double *v = w, *t = w + Mp, *cc = w + 2 * Mp, *u = w + 3 * Mp, *B = w + 4 * Mp, *X = w + 5 * Mp, *R = w + 6 * Mp,
         *Pv = w + 7 * Mp, *AP = w + 8 * Mp, *v2 = w + 9 * Mp, *t2 = w + 10 * Mp, *cc2 = w + 11 * Mp, *u2 = w + 12 * Mp,
         *AX = w + 13 * Mp, *state = w + 14 * Mp;
  void* pass_ws = state + 4;
  const int64_t pass_bytes = workspace_bytes - (14 * Mp + 4) * (int64_t)sizeof(double);
  hipStream_t s = as_stream(stream);
  const bool two = odx_knm_fwd_bwd2_workspace_bytes(n > 0 ? n : 1, M) >= 0;

  // out = A^-T [ T^-T K'K (T^-1 A^-1 src) / n + lam A^-1 src ]
  auto mmv = [&](const double* src, double* out) -> int {
    ODX_PROPAGATE(odx_trmv_f64(LAit, ldp, M, 1, src, 1.0, 0.0, nullptr, v, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTit, ldp, M, 1, v, 1.0, 0.0, nullptr, t, stream));
    ODX_PROPAGATE(odx_knm_fwd_bwd(K, ldk, n, M, t, nullptr, cc, pass_ws, pass_bytes, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTi, ldp, M, 0, cc, 1.0 / n_total, lam, v, u, stream));
    return odx_trmv_f64(LAi, ldp, M, 0, u, 1.0, 0.0, nullptr, out, stream);
  };

  // out = W src and out2 = W src2 from ONE read of K_nM (the small algebra twice, the pass once)
  auto mmv2 = [&](const double* src, double* out, const double* src2, double* out2) -> int {
    ODX_PROPAGATE(odx_trmv_f64(LAit, ldp, M, 1, src, 1.0, 0.0, nullptr, v, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTit, ldp, M, 1, v, 1.0, 0.0, nullptr, t, stream));
    ODX_PROPAGATE(odx_trmv_f64(LAit, ldp, M, 1, src2, 1.0, 0.0, nullptr, v2, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTit, ldp, M, 1, v2, 1.0, 0.0, nullptr, t2, stream));
    ODX_PROPAGATE(odx_knm_fwd_bwd2(K, ldk, n, M, t, t2, cc, cc2, pass_ws, pass_bytes, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTi, ldp, M, 0, cc, 1.0 / n_total, lam, v, u, stream));
    ODX_PROPAGATE(odx_trmv_f64(LAi, ldp, M, 0, u, 1.0, 0.0, nullptr, out, stream));
    ODX_PROPAGATE(odx_trmv_f64(LTi, ldp, M, 0, cc2, 1.0 / n_total, lam, v2, u2, stream));
    return odx_trmv_f64(LAi, ldp, M, 0, u2, 1.0, 0.0, nullptr, out2, stream);
  };

  ODX_CHECK_HIP(hipMemsetAsync(state, 0, 4 * sizeof(double), s));
  ODX_PROPAGATE(odx_trmv_f64(LTi, ldp, M, 0, b0, 1.0, 0.0, nullptr, u, stream));
  ODX_PROPAGATE(odx_trmv_f64(LAi, ldp, M, 0, u, 1.0, 0.0, nullptr, B, stream));        // A^-T T^-T b0
  ODX_PROPAGATE(odx_cg_init(B, X, R, Pv, state, M, stream));
  const double tol = cg_tolerance * cg_tolerance;
  for (int it = 0; it < maxiter; ++it) {
    const int full = ((it + 1) % full_gradient_every) == 0;
    // falkon recomputes R = B - W x from scratch every full_gradient_every-th step.  W is linear and x_new = x_old + a p,
    // so W x_new = W x_old + a W p: W x_old is formed together with this step's W p by one two-vector pass over K_nM,
    // and the recomputation costs no pass of its own (same value up to f64 rounding, still free of recursive drift).
    const bool fold = two && full && it != maxiter - 1;
    if (fold) ODX_PROPAGATE(mmv2(Pv, AP, X, AX));
    else ODX_PROPAGATE(mmv(Pv, AP));
    ODX_PROPAGATE(odx_cg_step(X, R, Pv, AP, state, cg_epsilon, full, M, stream));
    if (it == maxiter - 1) break;      // the residual / direction update of the last step cannot change X
    if (full && fold) {
      ODX_PROPAGATE(odx_cg_residual(B, AX, AP, state, R, M, stream));                     // R = B - (W x_old + a W p)
    } else if (full) {
      ODX_PROPAGATE(mmv(X, AP));
      ODX_CHECK_HIP(hipMemcpyAsync(R, B, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, s));
      ODX_PROPAGATE(odx_axpby_f64(-1.0, AP, 1.0, R, M, stream));                        // R = B - mmv(X)
    }
    ODX_PROPAGATE(odx_cg_finish(R, Pv, state, cg_epsilon, tol, M, stream));
  }
  ODX_PROPAGATE(odx_trmv_f64(LAit, ldp, M, 1, X, 1.0, 0.0, nullptr, v, stream));
  return odx_trmv_f64(LTit, ldp, M, 1, v, 1.0, 0.0, nullptr, alpha, stream);            // alpha = T^-1 A^-1 beta
}

// ---------------------------------------------------------------- the same loop for a batch of classes in lock step
// B independent fits (the classes of one Minibootstrap round, OnlineRegionClassifier_incore.py:96-155) advance through
// the CG schedule together: every launch of the loop above is issued ONCE with the class as a grid dimension, so a round
// of 30 fits costs the ~170 launches of one.  Per class the arithmetic is exactly odx_falkon_cg_f64's (same kernels' bodies,
// same pass configuration and workgroup count, same slab order): alpha comes out bit-identical to the one-class call.
// The classes must share one pass configuration (odx_falkon_cg_batched_workspace_bytes < 0 otherwise).  A class whose
// residual meets the tolerance raises its own stop flag and coasts.
// The blocks a class-batched CG streams: f32 rows (K) or one of the compact formats (Khi / Klo planes, fmt).
struct BatchBlocks {
  int fmt = ODX_KNM_F32;
  const float* const* K = nullptr; const int64_t* ldk = nullptr;
  const void* const* Khi = nullptr; const void* const* Klo = nullptr; const int64_t* ldlo = nullptr;
};

static int64_t batched_pass_bytes(const BatchBlocks& kb, int B, const int64_t* n, const int64_t* M) {
  return kb.fmt == ODX_KNM_F32 ? knm_pass_batched_workspace_bytes(B, n, M) : knm_passq_batched_workspace_bytes(B, n, M, kb.fmt);
}

static int64_t cg_batched_bytes(const BatchBlocks& kb, int B, const int64_t* n, const int64_t* M) {
  if (B < 1 || B > ODX_MAX_ZBATCH || !n || !M) return ODX_ERR_INVALID;
  const int64_t pass = batched_pass_bytes(kb, B, n, M);
  if (pass < 0) return pass;
  int64_t mm = 1;
  for (int b = 0; b < B; ++b) mm = M[b] > mm ? M[b] : mm;
  const int64_t Mp = round_up(mm, 2);
  return ((int64_t)B * (9 * Mp + 4)) * (int64_t)sizeof(double) + round_up(pass, 16);
}

extern "C" int64_t odx_falkon_cg_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M) {
  return cg_batched_bytes(BatchBlocks(), B, n, M);
}

extern "C" int64_t odx_falkon_cg_batched_q_workspace_bytes(int B, const int64_t* n, const int64_t* M, int fmt) {
  BatchBlocks kb;
  kb.fmt = fmt;
  if (fmt != ODX_KNM_U24 && fmt != ODX_KNM_BF16) return ODX_ERR_UNSUPPORTED;
  return cg_batched_bytes(kb, B, n, M);
}

static int falkon_cg_batched(const char* who, const BatchBlocks& kb, int B, const int64_t* n, const int64_t* M, const double* P,
                             int64_t ldp, int64_t p_rows, int64_t p_stride, const double* b0, int64_t vstride, const double* n_total,
                             double lam, int maxiter, int full_gradient_every, double cg_epsilon, double cg_tolerance, double* alpha,
                             void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(B >= 1 && B <= ODX_MAX_ZBATCH && n && M && P && b0 && n_total && alpha && maxiter >= 0 && full_gradient_every > 0,
              "%s: bad argument", who);
  const int64_t need = cg_batched_bytes(kb, B, n, M);
  ODX_REQUIRE(need >= 0, "%s: the classes of a batch must share one pass configuration", who);
  if (workspace == nullptr || workspace_bytes < need || !aligned16(workspace)) {
    set_error("%s: workspace too small or not 16-byte aligned", who);
    return ODX_ERR_WORKSPACE;
  }
  VecBatch vb;
  vb.B = B;
  int64_t mm = 1;
  for (int b = 0; b < B; ++b) {
    ODX_REQUIRE(M[b] > 0 && M[b] <= p_rows && n[b] >= 0 && n_total[b] > 0, "%s: class %d: bad sizes", who, b);
    vb.M[b] = (int)M[b];
    vb.scale[b] = 1.0 / n_total[b];
    mm = M[b] > mm ? M[b] : mm;
  }
  ODX_REQUIRE(vstride >= mm && vstride % 2 == 0 && ldp % 2 == 0 && p_stride >= 4 * p_rows * ldp,
              "%s: vstride even >= max M, ldp even, p_stride >= 4 p_rows ldp", who);
  const int64_t Mp = round_up(mm, 2);
  double* w = static_cast<double*>(workspace);
  const int64_t V = (int64_t)B * Mp;            // one vector per class, Mp apart
  double *v = w, *t = w + V, *cc = w + 2 * V, *u = w + 3 * V, *Bv = w + 4 * V, *X = w + 5 * V, *R = w + 6 * V,
         *Pv = w + 7 * V, *AP = w + 8 * V, *state = w + 9 * V;
  void* pass_ws = state + 4 * B;
  const int64_t pass_bytes = workspace_bytes - ((int64_t)B * (9 * Mp + 4)) * (int64_t)sizeof(double);
  hipStream_t s = as_stream(stream);
  const double *LTi = P, *LTit = P + p_rows * ldp, *LAi = P + 2 * p_rows * ldp, *LAit = P + 3 * p_rows * ldp;

  auto pass = [&](const double* src, double* dst) -> int {
    if (kb.fmt == ODX_KNM_F32) return knm_pass_batched(B, kb.K, kb.ldk, n, M, src, Mp, dst, Mp, pass_ws, pass_bytes, s);
    return knm_passq_batched(B, kb.Khi, kb.ldk, kb.Klo, kb.ldlo, kb.fmt, n, M, src, Mp, dst, Mp, pass_ws, pass_bytes, s);
  };
  auto mmv = [&](const double* src, double* out) -> int {
    ODX_PROPAGATE(trmv_batched_f64(LAit, ldp, p_stride, 1, vb, src, Mp, false, 0.0, nullptr, 0, v, Mp, s));
    ODX_PROPAGATE(trmv_batched_f64(LTit, ldp, p_stride, 1, vb, v, Mp, false, 0.0, nullptr, 0, t, Mp, s));
    ODX_PROPAGATE(pass(t, cc));
    ODX_PROPAGATE(trmv_batched_f64(LTi, ldp, p_stride, 0, vb, cc, Mp, true, lam, v, Mp, u, Mp, s));
    return trmv_batched_f64(LAi, ldp, p_stride, 0, vb, u, Mp, false, 0.0, nullptr, 0, out, Mp, s);
  };

  ODX_CHECK_HIP(hipMemsetAsync(state, 0, (size_t)(4 * B) * sizeof(double), s));
  // b0 arrives vstride apart; the loop's vectors are Mp apart: u <- T^-T b0, Bv <- A^-T u
  ODX_PROPAGATE(trmv_batched_f64(LTi, ldp, p_stride, 0, vb, b0, vstride, false, 0.0, nullptr, 0, u, Mp, s));
  ODX_PROPAGATE(trmv_batched_f64(LAi, ldp, p_stride, 0, vb, u, Mp, false, 0.0, nullptr, 0, Bv, Mp, s));
  ODX_PROPAGATE(cg_init_batched(vb, Bv, X, R, Pv, state, Mp, s));
  const double tol = cg_tolerance * cg_tolerance;
  for (int it = 0; it < maxiter; ++it) {
    ODX_PROPAGATE(mmv(Pv, AP));
    const int full = ((it + 1) % full_gradient_every) == 0;
    ODX_PROPAGATE(cg_step_batched(vb, X, R, Pv, AP, state, cg_epsilon, full, Mp, s));
    if (it == maxiter - 1) break;
    if (full) {
      ODX_PROPAGATE(mmv(X, AP));
      ODX_PROPAGATE(cg_full_residual_batched(vb, Bv, AP, R, Mp, s));                  // R = B - mmv(X)
    }
    ODX_PROPAGATE(cg_finish_batched(vb, R, Pv, state, cg_epsilon, tol, Mp, s));
  }
  ODX_PROPAGATE(trmv_batched_f64(LAit, ldp, p_stride, 1, vb, X, Mp, false, 0.0, nullptr, 0, v, Mp, s));
  return trmv_batched_f64(LTit, ldp, p_stride, 1, vb, v, Mp, false, 0.0, nullptr, 0, alpha, vstride, s);
}

extern "C" int odx_falkon_cg_batched_f64(int B, const float* const* K, const int64_t* ldk, const int64_t* n, const int64_t* M,
                                         const double* P, int64_t ldp, int64_t p_rows, int64_t p_stride, const double* b0,
                                         int64_t vstride, const double* n_total, double lam, int maxiter,
                                         int full_gradient_every, double cg_epsilon, double cg_tolerance, double* alpha,
                                         void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(K && ldk, "odx_falkon_cg_batched_f64: bad argument");
  BatchBlocks kb;
  kb.K = K; kb.ldk = ldk;
  return falkon_cg_batched("odx_falkon_cg_batched_f64", kb, B, n, M, P, ldp, p_rows, p_stride, b0, vstride, n_total, lam, maxiter,
                           full_gradient_every, cg_epsilon, cg_tolerance, alpha, workspace, workspace_bytes, stream);
}

// The same lock-step loops over blocks stored in a compact format (ODX_KNM_U24: Khi = the u16 plane, Klo = the u8 plane;
// ODX_KNM_BF16: Khi = the bf16 words, Klo ignored): class b's passes are odx_knm_fwd_bwd_q's, bit for bit.
extern "C" int odx_falkon_cg_batched_q_f64(int B, const void* const* Khi, const int64_t* ldk, const void* const* Klo,
                                           const int64_t* ldlo, int fmt, const int64_t* n, const int64_t* M, const double* P,
                                           int64_t ldp, int64_t p_rows, int64_t p_stride, const double* b0, int64_t vstride,
                                           const double* n_total, double lam, int maxiter, int full_gradient_every,
                                           double cg_epsilon, double cg_tolerance, double* alpha, void* workspace,
                                           int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(Khi && ldk && (fmt == ODX_KNM_BF16 || (fmt == ODX_KNM_U24 && Klo && ldlo)), "odx_falkon_cg_batched_q_f64: bad argument");
  BatchBlocks kb;
  kb.fmt = fmt; kb.Khi = Khi; kb.ldk = ldk; kb.Klo = Klo; kb.ldlo = ldlo;
  return falkon_cg_batched("odx_falkon_cg_batched_q_f64", kb, B, n, M, P, ldp, p_rows, p_stride, b0, vstride, n_total, lam, maxiter,
                           full_gradient_every, cg_epsilon, cg_tolerance, alpha, workspace, workspace_bytes, stream);
}
