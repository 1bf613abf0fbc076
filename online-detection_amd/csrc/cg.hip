// Conjugate-gradient vector updates (f64, length M, one right-hand side), scalars on device.
// state[0] = rs_old, state[1] = rs_new, state[2] = stop flag (0 / 1), state[3] = last step size.
// One 1024-thread workgroup per call: M <= ~20k, so these are latency-sized, not bandwidth-sized.
#include "odx_internal.h"

namespace odx {

__device__ __forceinline__ double block_sum_1024(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // red may still be read from a previous reduction
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += red[q];
  return s;
}

__device__ __forceinline__ void cg_init_body(const double* __restrict__ B, double* __restrict__ X,
                                             double* __restrict__ R, double* __restrict__ P,
                                             double* __restrict__ state, int64_t M) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < M; i += 1024) {
    const double b = B[i];
    X[i] = 0.0;
    R[i] = b;
    P[i] = b;
    s = fma(b, b, s);
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) {
    state[0] = s;
    state[1] = s;
    state[2] = 0.0;
    state[3] = 0.0;
  }
}

__device__ __forceinline__ void cg_step_body(double* __restrict__ X, double* __restrict__ R,
                                             const double* __restrict__ P, const double* __restrict__ AP,
                                             double* __restrict__ state, double cg_eps, int full_grad, int64_t M) {
  __shared__ double red[16];
  if (state[2] != 0.0) return;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < M; i += 1024) s = fma(P[i], AP[i], s);
  s = block_sum_1024(s, red);
  const double a = state[0] / (s + cg_eps);
  for (int64_t i = threadIdx.x; i < M; i += 1024) {
    X[i] = fma(a, P[i], X[i]);
    if (!full_grad) R[i] = fma(-a, AP[i], R[i]);
  }
  if (threadIdx.x == 0) state[3] = a;
}

__device__ __forceinline__ void cg_finish_body(const double* __restrict__ R, double* __restrict__ P,
                                               double* __restrict__ state, double cg_eps, double tol, int64_t M) {
  __shared__ double red[16];
  if (state[2] != 0.0) return;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < M; i += 1024) s = fma(R[i], R[i], s);
  s = block_sum_1024(s, red);
  const double rs_old = state[0];
  if (sqrt(fabs(s)) < tol) {  // converged: leave P alone, raise the flag (falkon breaks out here)
    __syncthreads();
    if (threadIdx.x == 0) {
      state[1] = s;
      state[2] = 1.0;
    }
    return;
  }
  const double b = s / (rs_old + cg_eps);
  for (int64_t i = threadIdx.x; i < M; i += 1024) P[i] = fma(b, P[i], R[i]);
  __syncthreads();
  if (threadIdx.x == 0) {
    state[1] = s;
    state[0] = s;
  }
}

__global__ __launch_bounds__(1024) void cg_init_kernel(const double* __restrict__ B, double* __restrict__ X,
                                                       double* __restrict__ R, double* __restrict__ P,
                                                       double* __restrict__ state, int64_t M) {
  cg_init_body(B, X, R, P, state, M);
}
__global__ __launch_bounds__(1024) void cg_step_kernel(double* __restrict__ X, double* __restrict__ R,
                                                       const double* __restrict__ P, const double* __restrict__ AP,
                                                       double* __restrict__ state, double cg_eps, int full_grad,
                                                       int64_t M) {
  cg_step_body(X, R, P, AP, state, cg_eps, full_grad, M);
}
__global__ __launch_bounds__(1024) void cg_finish_kernel(const double* __restrict__ R, double* __restrict__ P,
                                                         double* __restrict__ state, double cg_eps, double tol,
                                                         int64_t M) {
  cg_finish_body(R, P, state, cg_eps, tol, M);
}

// ---- the same updates for the classes of a batch: one workgroup per class, vectors vstride apart, 4 state words each
__global__ __launch_bounds__(1024) void cg_init_batched_kernel(VecBatch vb, const double* __restrict__ B,
                                                               double* __restrict__ X, double* __restrict__ R,
                                                               double* __restrict__ P, double* __restrict__ state,
                                                               int64_t vs) {
  const int64_t b = blockIdx.x;
  cg_init_body(B + b * vs, X + b * vs, R + b * vs, P + b * vs, state + 4 * b, vb.M[b]);
}
__global__ __launch_bounds__(1024) void cg_step_batched_kernel(VecBatch vb, double* __restrict__ X, double* __restrict__ R,
                                                               const double* __restrict__ P, const double* __restrict__ AP,
                                                               double* __restrict__ state, double cg_eps, int full_grad,
                                                               int64_t vs) {
  const int64_t b = blockIdx.x;
  cg_step_body(X + b * vs, R + b * vs, P + b * vs, AP + b * vs, state + 4 * b, cg_eps, full_grad, vb.M[b]);
}
__global__ __launch_bounds__(1024) void cg_finish_batched_kernel(VecBatch vb, const double* __restrict__ R,
                                                                 double* __restrict__ P, double* __restrict__ state,
                                                                 double cg_eps, double tol, int64_t vs) {
  const int64_t b = blockIdx.x;
  cg_finish_body(R + b * vs, P + b * vs, state + 4 * b, cg_eps, tol, vb.M[b]);
}
// R = B - AX (the periodic full residual in its plain form: y = a x + b y of odx_axpby_f64 with a = -1, b = 1 on a copy of B)
__global__ __launch_bounds__(256) void cg_full_residual_batched_kernel(VecBatch vb, const double* __restrict__ B,
                                                                       const double* __restrict__ AX, double* __restrict__ R,
                                                                       int64_t vs) {
  const int64_t b = blockIdx.y, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < vb.M[b]) R[b * vs + i] = -1.0 * AX[b * vs + i] + 1.0 * B[b * vs + i];
}

// R = B - (AX + a AP) with a = state[3], the step cg_step_kernel has just taken: the full residual B - W x_new of the
// periodic recomputation, from W x_old and W p (W is linear, x_new = x_old + a p) — both products come out of one
// two-vector pass over K_nM (odx_knm_fwd_bwd2) made BEFORE the step.
__global__ __launch_bounds__(256) void cg_residual_kernel(const double* __restrict__ B, const double* __restrict__ AX,
                                                          const double* __restrict__ AP, const double* __restrict__ state,
                                                          double* __restrict__ R, int64_t M) {
  if (state[2] != 0.0) return;
  const double a = state[3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < M) R[i] = B[i] - fma(a, AP[i], AX[i]);
}

__global__ __launch_bounds__(256) void axpby_kernel(double a, const double* __restrict__ x, double b,
                                                    double* __restrict__ y, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < M) y[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}

static int vb_max(const VecBatch& vb) {
  int mm = 0;
  for (int b = 0; b < vb.B; ++b) mm = vb.M[b] > mm ? vb.M[b] : mm;
  return mm;
}

int cg_init_batched(const VecBatch& vb, const double* Bv, double* X, double* R, double* P, double* state, int64_t vstride,
                    hipStream_t stream) {
  hipLaunchKernelGGL(cg_init_batched_kernel, dim3((unsigned)vb.B), dim3(1024), 0, stream, vb, Bv, X, R, P, state, vstride);
  ODX_CHECK_LAUNCH("cg_init_batched");
  return ODX_OK;
}
int cg_step_batched(const VecBatch& vb, double* X, double* R, const double* P, const double* AP, double* state,
                    double cg_eps, int full_grad, int64_t vstride, hipStream_t stream) {
  hipLaunchKernelGGL(cg_step_batched_kernel, dim3((unsigned)vb.B), dim3(1024), 0, stream, vb, X, R, P, AP, state, cg_eps,
                     full_grad, vstride);
  ODX_CHECK_LAUNCH("cg_step_batched");
  return ODX_OK;
}
int cg_finish_batched(const VecBatch& vb, const double* R, double* P, double* state, double cg_eps, double tol,
                      int64_t vstride, hipStream_t stream) {
  hipLaunchKernelGGL(cg_finish_batched_kernel, dim3((unsigned)vb.B), dim3(1024), 0, stream, vb, R, P, state, cg_eps, tol,
                     vstride);
  ODX_CHECK_LAUNCH("cg_finish_batched");
  return ODX_OK;
}
int cg_full_residual_batched(const VecBatch& vb, const double* Bv, const double* AX, double* R, int64_t vstride,
                             hipStream_t stream) {
  const int mm = vb_max(vb);
  if (mm <= 0) return ODX_OK;
  hipLaunchKernelGGL(cg_full_residual_batched_kernel, dim3((unsigned)ceil_div(mm, 256), (unsigned)vb.B), dim3(256), 0,
                     stream, vb, Bv, AX, R, vstride);
  ODX_CHECK_LAUNCH("cg_full_residual_batched");
  return ODX_OK;
}

}  // namespace odx

using namespace odx;

extern "C" int odx_cg_init(const double* B, double* X, double* R, double* P, double* state, int64_t M,
                           odx_stream_t stream) {
  ODX_REQUIRE(B && X && R && P && state && M > 0, "odx_cg_init: bad argument");
  hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, as_stream(stream), B, X, R, P, state, M);
  ODX_CHECK_LAUNCH("odx_cg_init");
  return ODX_OK;
}

extern "C" int odx_cg_step(double* X, double* R, const double* P, const double* AP, double* state, double cg_eps,
                           int full_grad, int64_t M, odx_stream_t stream) {
  ODX_REQUIRE(X && R && P && AP && state && M > 0, "odx_cg_step: bad argument");
  hipLaunchKernelGGL(cg_step_kernel, dim3(1), dim3(1024), 0, as_stream(stream), X, R, P, AP, state, cg_eps,
                     full_grad, M);
  ODX_CHECK_LAUNCH("odx_cg_step");
  return ODX_OK;
}

extern "C" int odx_cg_finish(const double* R, double* P, double* state, double cg_eps, double tol, int64_t M,
                             odx_stream_t stream) {
  ODX_REQUIRE(R && P && state && M > 0, "odx_cg_finish: bad argument");
  hipLaunchKernelGGL(cg_finish_kernel, dim3(1), dim3(1024), 0, as_stream(stream), R, P, state, cg_eps, tol, M);
  ODX_CHECK_LAUNCH("odx_cg_finish");
  return ODX_OK;
}

extern "C" int odx_cg_residual(const double* B, const double* AX, const double* AP, const double* state, double* R,
                               int64_t M, odx_stream_t stream) {
  ODX_REQUIRE(B && AX && AP && state && R && M > 0, "odx_cg_residual: bad argument");
  hipLaunchKernelGGL(cg_residual_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, as_stream(stream), B, AX, AP,
                     state, R, M);
  ODX_CHECK_LAUNCH("odx_cg_residual");
  return ODX_OK;
}

extern "C" int odx_axpby_f64(double a, const double* x, double b, double* y, int64_t M, odx_stream_t stream) {
  if (M <= 0) return ODX_OK;
  ODX_REQUIRE(x && y, "odx_axpby_f64: null pointer");
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, as_stream(stream), a, x, b, y, M);
  ODX_CHECK_LAUNCH("odx_axpby_f64");
  return ODX_OK;
}
