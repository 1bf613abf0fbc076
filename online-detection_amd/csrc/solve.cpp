// The conjugate-gradient loop of one unsharded FALKON fit as ONE library call: the same sequence of libodx launches as
// online-detection_amd/odx/solver.py::falkon_fit issues one by one from Python (which stays the reference statement of
// the loop and the only form for row shards, where a collective sits inside every iteration).  A fit of the reference
// regime (M ~ 2000, n ~ 1e4) is ~ 250 launches of a few microseconds each; issued from here the host needs ~ 0.3 ms for
// them instead of ~ 3 ms, which is what bounds the class-stream minibootstrap.
//
// Reference: InCoreFalkon.fit -> ConjugateGradient.solve at FALKONWrapper_with_centers_selection_incore.py:56-68
// (falkon's defaults: 20 iterations, residual recomputed every 10).
#include "odx_common.h"

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
