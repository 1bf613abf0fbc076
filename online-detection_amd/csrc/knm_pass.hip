// The conjugate-gradient pass over a stored K_nM shard:  out = K' (K v + w).
// HBM-bound: every float of K is read exactly once per call.
//
// A persistent workgroup streams blocks of R rows.  Thread t owns the float4 column chunks
// t, t+NT, ... (CH of them) of every row: the running column sums of K' t live in registers
// as f64, the R x CH float4 of K of the current block too; v sits in LDS as f64.  Phase 1 forms the R row dots (f64) and reduces them over the workgroup (wave shuffles
// + one LDS exchange, ping-pong buffers => one barrier per block).  Phase 2 adds
// K[r, cols] * t_r into the column sums and, chunk by chunk, re-issues the loads of the NEXT
// block into the registers it has just finished with, so the block's worth of loads is in
// flight across the reduction.  Each workgroup writes its column sums as one slab; a second
// kernel adds the slabs in fixed order (bitwise reproducible, no float atomics).
//
// Requirements: ldk % 4 == 0, K 16-byte aligned, columns [M, roundup(M,4)) of K are zero
// (odx_gauss_knm_f32 writes them so).
#include "odx_internal.h"

namespace odx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// NV = 1: out = K' (K v + w).  NV = 2: two products from ONE read of K — out = K' (K v), out2 = K' (K v2) (no w) — which
// is how the CG's periodic residual recomputation rides along with the neighbouring iteration's pass instead of costing a
// pass of its own.  With NV = 2 both vectors sit in dynamic LDS (2 x roundup(M, 4) x 8 B, M <= 10 000 on the 512-thread
// configurations) and the slab of a workgroup holds its two column-sum vectors back to back.
template <int NT, int CH, int R, int NV>
__global__ __launch_bounds__(NT) void knm_pass_kernel(const float* __restrict__ K, int64_t ldk, int64_t n, int64_t M,
                                                      const double* __restrict__ v, const double* __restrict__ v2,
                                                      const double* __restrict__ w, double* __restrict__ slab,
                                                      int64_t slab_ld) {
  constexpr int NW = NT / 64;
  constexpr int VCAP = (NT * CH * 4 < 20000) ? NT * CH * 4 : 20000;  // 160,000 B of the 163,840 B LDS at most
  __shared__ __attribute__((aligned(16))) double vs_static[NV == 1 ? VCAP : 2];
  extern __shared__ __attribute__((aligned(16))) double vs_dyn[];       // NV == 2: [2][vcap2]
  __shared__ double red[2][NW][R * NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t nchunk = (M + 3) >> 2;
  const int64_t nblk = (n + R - 1) / R;
  const int vcap = NV == 1 ? VCAP : (int)(nchunk * 4);
  double* vs = NV == 1 ? vs_static : vs_dyn;

  // v lives in LDS (f64): its read traffic per block equals M * 8 B, a few % of LDS bandwidth.
  for (int i = tid; i < vcap; i += NT) {
    vs[i] = (v != nullptr && i < M) ? v[i] : 0.0;
    if (NV == 2) vs[vcap + i] = (i < M) ? v2[i] : 0.0;
  }

  double acc[NV][CH][4];
  bool cvalid[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    cvalid[c] = (tid + (int64_t)c * NT) < nchunk;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q][c][e] = 0.0;
  }

  f32x4 kr[R][CH];
  auto load_block = [&](int64_t blk, int c) {
    const int64_t ch = tid + (int64_t)c * NT;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = blk * R + r;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (cvalid[c] && row < n) x = *reinterpret_cast<const f32x4*>(K + row * ldk + ch * 4);
      kr[r][c] = x;
    }
  };

  int64_t blk = blockIdx.x;
  if (blk < nblk) {
#pragma unroll
    for (int c = 0; c < CH; ++c) load_block(blk, c);
  }
  __syncthreads();  // vs is complete
  int pp = 0;
  for (; blk < nblk; blk += gridDim.x) {
    double t[NV][R];
    if (v != nullptr) {
      // phase 1: row dots
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) t[q][r] = 0.0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          double vv[4] = {0.0, 0.0, 0.0, 0.0};
          if (cvalid[c]) {
            const f64x2 a = *reinterpret_cast<const f64x2*>(&vs[q * vcap + (tid + c * NT) * 4]);
            const f64x2 b = *reinterpret_cast<const f64x2*>(&vs[q * vcap + (tid + c * NT) * 4 + 2]);
            vv[0] = a[0]; vv[1] = a[1]; vv[2] = b[0]; vv[3] = b[1];
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[q][r] = fma((double)kr[r][c][e], vv[e], t[q][r]);
        }
      }
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double s = t[q][r];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
          if (lane == 0) red[pp][wave][q * R + r] = s;
        }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double s = 0.0;
#pragma unroll
          for (int u = 0; u < NW; ++u) s += red[pp][u][q * R + r];
          t[q][r] = s;
        }
      pp ^= 1;
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) t[q][r] = 0.0;
    }
    if (NV == 1 && w != nullptr) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t row = blk * R + r;
        if (row < n) t[0][r] += w[row];
      }
    }
    // phase 2: column sums, and the next block's loads re-issued chunk by chunk
    const int64_t nxt = blk + gridDim.x;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q][c][e] = fma((double)kr[r][c][e], t[q][r], acc[q][c][e]);
      if (nxt < nblk) load_block(nxt, c);
    }
  }
  double* my = slab + (int64_t)blockIdx.x * slab_ld * NV;
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int64_t ch = tid + (int64_t)c * NT;
      if (cvalid[c]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) my[q * slab_ld + ch * 4 + e] = acc[q][c][e];
      }
    }
}

// out[j] = sum_g slab[g][j] in a fixed order: a workgroup owns 64 columns; its 4 waves take the slabs g = w, w + 4, ...
// (coalesced 512-B row segments), then wave 0 adds the 4 partial sums in wave order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const double* __restrict__ slab, int64_t slab_ld, int nslab,
                                                          int64_t M, double* __restrict__ out) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (j < M)
    for (int g = wave; g < nslab; g += 4) s += slab[(int64_t)g * slab_ld + j];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < M) out[j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// the same fixed-order sum for the two vectors a two-product pass leaves per workgroup (slab g = [sums of v | sums of v2])
__global__ __launch_bounds__(256) void slab_reduce2_kernel(const double* __restrict__ slab, int64_t slab_ld, int nslab,
                                                           int64_t M, double* __restrict__ out, double* __restrict__ out2) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const double* base = slab + (int64_t)blockIdx.y * slab_ld;
  double s = 0.0;
  if (j < M)
    for (int g = wave; g < nslab; g += 4) s += base[(int64_t)g * 2 * slab_ld + j];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < M) (blockIdx.y ? out2 : out)[j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

static int g_reserved_cus = 0;

struct PassCfg {
  int nt, ch, r, wg_per_cu;
};

static bool pick_cfg(int64_t M, PassCfg* cfg) {
  const int64_t chunks = (M + 3) / 4;
  if (chunks <= 256) { *cfg = {256, 1, 16, 2}; return true; }
  if (chunks <= 512) { *cfg = {256, 2, 8, 2}; return true; }
  if (chunks <= 1024) { *cfg = {256, 4, 4, 2}; return true; }
  if (chunks <= 2048) { *cfg = {512, 4, 4, 1}; return true; }
  if (chunks <= 2560) { *cfg = {512, 5, 4, 1}; return true; }
  if (chunks <= 3072) { *cfg = {512, 6, 2, 1}; return true; }
  if (chunks <= 5000) { *cfg = {1024, 5, 1, 1}; return true; }
  return false;
}

static int grid_for(const PassCfg& cfg, int64_t n) {
  int cus = odx_device_cus();
  if (cus <= 0) cus = 256;
  const int64_t nblk = ceil_div(n, cfg.r);
  // One-workgroup-per-CU configurations can leave CUs to concurrent streams (odx_set_pass_reserved_cus).
  const int reserve = (cfg.wg_per_cu == 1 && g_reserved_cus < cus / 2) ? g_reserved_cus : 0;
  int64_t g = (int64_t)(cus - reserve) * cfg.wg_per_cu;
  if (g > nblk) g = nblk;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace odx

using namespace odx;

int odx::slab_reduce_f64(const double* slab, int64_t slab_ld, int nslab, int64_t M, double* out, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(M, 64)), dim3(256), 0, s, slab, slab_ld, nslab, M, out);
  ODX_CHECK_LAUNCH("slab_reduce_f64");
  return ODX_OK;
}

extern "C" int odx_set_pass_reserved_cus(int cus) {
  ODX_REQUIRE(cus >= 0, "odx_set_pass_reserved_cus: negative count");
  g_reserved_cus = cus;
  return ODX_OK;
}

extern "C" int64_t odx_knm_fwd_bwd_workspace_bytes(int64_t n, int64_t M) {
  PassCfg cfg;
  if (n <= 0 || M <= 0) return 0;
  if (!pick_cfg(M, &cfg)) return ODX_ERR_UNSUPPORTED;
  int cus = odx_device_cus();
  if (cus <= 0) cus = 256;
  // sized for the largest grid any n can get, so one workspace serves every shard size
  return (int64_t)cus * cfg.wg_per_cu * round_up(M, 4) * (int64_t)sizeof(double);
}

#define ODX_PASS_LAUNCH(NT_, CH_, R_)                                                                                \
  hipLaunchKernelGGL((knm_pass_kernel<NT_, CH_, R_, 1>), dim3(grid), dim3(NT_), 0, s, K, ldk, n, M, v, nullptr, w, \
                     slab, slab_ld)
#define ODX_PASS2_LAUNCH(NT_, CH_, R_)                                                                                  \
  do {                                                                                                                  \
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_pass_kernel<NT_, CH_, R_, 2>),                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                          \
    hipLaunchKernelGGL((knm_pass_kernel<NT_, CH_, R_, 2>), dim3(grid), dim3(NT_), lds2, s, K, ldk, n, M, v, v2,      \
                       nullptr, slab, slab_ld);                                                                         \
  } while (0)

extern "C" int odx_knm_fwd_bwd(const float* K, int64_t ldk, int64_t n, int64_t M, const double* v, const double* w,
                               double* out, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && out, "odx_knm_fwd_bwd: M <= 0 or null out");
  hipStream_t s = as_stream(stream);
  if (n <= 0) {
    ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)M * sizeof(double), s));
    return ODX_OK;
  }
  ODX_REQUIRE(K && (v || w), "odx_knm_fwd_bwd: null K, or both v and w null");
  ODX_REQUIRE(ldk % 4 == 0 && ldk >= round_up(M, 4) && aligned16(K),
              "odx_knm_fwd_bwd: K must be 16-byte aligned, ldk %% 4 == 0, ldk >= roundup(M, 4)");
  PassCfg cfg;
  if (!pick_cfg(M, &cfg)) {
    set_error("odx_knm_fwd_bwd: M = %lld exceeds the 20000 columns the pass kernels are built for", (long long)M);
    return ODX_ERR_UNSUPPORTED;
  }
  const int grid = grid_for(cfg, n);
  const int64_t slab_ld = round_up(M, 4);
  if (workspace == nullptr || workspace_bytes < (int64_t)grid * slab_ld * (int64_t)sizeof(double)) {
    set_error("odx_knm_fwd_bwd: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)((int64_t)grid * slab_ld * (int64_t)sizeof(double)));
    return ODX_ERR_WORKSPACE;
  }
  double* slab = static_cast<double*>(workspace);
  if (cfg.nt == 256 && cfg.ch == 1) ODX_PASS_LAUNCH(256, 1, 16);
  else if (cfg.nt == 256 && cfg.ch == 2) ODX_PASS_LAUNCH(256, 2, 8);
  else if (cfg.nt == 256 && cfg.ch == 4) ODX_PASS_LAUNCH(256, 4, 4);
  else if (cfg.nt == 512 && cfg.ch == 4) ODX_PASS_LAUNCH(512, 4, 4);
  else if (cfg.nt == 512 && cfg.ch == 5) ODX_PASS_LAUNCH(512, 5, 4);
  else if (cfg.nt == 512 && cfg.ch == 6) ODX_PASS_LAUNCH(512, 6, 2);
  else ODX_PASS_LAUNCH(1024, 5, 1);
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(M, 64)), dim3(256), 0, s, slab, slab_ld, grid, M,
                     out);
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd(reduce)");
  return ODX_OK;
}

// ---------------------------------------------------------------- two products from one read of K
// out = K' (K v), out2 = K' (K v2).  Supported where both vectors fit in LDS beside the reduction scratch and the
// configuration's register budget allows a second set of column sums: 4096 < M <= 10 000 (the headline's M = 1e4 included);
// wider blocks report ODX_ERR_UNSUPPORTED from the workspace query and callers issue two single passes instead.
// Only the one-workgroup-per-CU 512-thread configurations (4096 < M <= 10 000): below that a fit is launch-bound and the
// second vector's four extra triangular products cost more than the pass they save.  Two rows per block instead of four:
// the second set of f64 column sums takes the registers of two rows' worth of K (227 -> 256 VGPRs would spill otherwise).
static bool pick_cfg2(int64_t M, PassCfg* cfg) {
  if (!pick_cfg(M, cfg)) return false;
  if (cfg->nt != 512 || cfg->ch > 5) return false;
  cfg->r = 2;
  const int64_t lds = 2 * round_up(M, 4) * 8 + 2 * (cfg->nt / 64) * cfg->r * 2 * 8 + 64;
  return lds <= 163840;
}

extern "C" int64_t odx_knm_fwd_bwd2_workspace_bytes(int64_t n, int64_t M) {
  PassCfg cfg;
  if (n <= 0 || M <= 0) return 0;
  if (!pick_cfg2(M, &cfg)) return ODX_ERR_UNSUPPORTED;
  return 2 * odx_knm_fwd_bwd_workspace_bytes(n, M);
}

extern "C" int odx_knm_fwd_bwd2(const float* K, int64_t ldk, int64_t n, int64_t M, const double* v, const double* v2,
                                double* out, double* out2, void* workspace, int64_t workspace_bytes,
                                odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && out && out2, "odx_knm_fwd_bwd2: M <= 0 or null out");
  hipStream_t s = as_stream(stream);
  if (n <= 0) {
    ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)M * sizeof(double), s));
    ODX_CHECK_HIP(hipMemsetAsync(out2, 0, (size_t)M * sizeof(double), s));
    return ODX_OK;
  }
  ODX_REQUIRE(K && v && v2, "odx_knm_fwd_bwd2: null K, v or v2");
  ODX_REQUIRE(ldk % 4 == 0 && ldk >= round_up(M, 4) && aligned16(K),
              "odx_knm_fwd_bwd2: K must be 16-byte aligned, ldk %% 4 == 0, ldk >= roundup(M, 4)");
  PassCfg cfg;
  if (!pick_cfg2(M, &cfg)) {
    set_error("odx_knm_fwd_bwd2: M = %lld is outside the two-vector configurations (use two single passes)", (long long)M);
    return ODX_ERR_UNSUPPORTED;
  }
  const int grid = grid_for(cfg, n);
  const int64_t slab_ld = round_up(M, 4);
  if (workspace == nullptr || workspace_bytes < 2 * (int64_t)grid * slab_ld * (int64_t)sizeof(double)) {
    set_error("odx_knm_fwd_bwd2: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  double* slab = static_cast<double*>(workspace);
  const size_t lds2 = (size_t)(2 * slab_ld * sizeof(double));
  if (cfg.ch == 4) ODX_PASS2_LAUNCH(512, 4, 2);
  else ODX_PASS2_LAUNCH(512, 5, 2);
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd2");
  // the two column-sum vectors of a workgroup lie back to back in its slab: reduce them as one vector of 2 slab_ld
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3((unsigned)ceil_div(M, 64), 2), dim3(256), 0, s, slab, slab_ld, grid, M, out,
                     out2);
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd2(reduce)");
  return ODX_OK;
}
