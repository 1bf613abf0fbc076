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
#include <algorithm>

#include "odx_internal.h"

namespace odx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// NV = 1: out = K' (K v + w).  NV = 2: two products from ONE read of K — out = K' (K v), out2 = K' (K v2) (no w) — which
// is how the CG's periodic residual recomputation rides along with the neighbouring iteration's pass instead of costing a
// pass of its own.  With NV = 2 both vectors sit in dynamic LDS (2 x roundup(M, 4) x 8 B, M <= 10 000 on the 512-thread
// configurations) and the slab of a workgroup holds its two column-sum vectors back to back.
template <int NT, int CH, int R, int NV>
__device__ __forceinline__ void knm_pass_body(const float* __restrict__ K, int64_t ldk, int64_t n, int64_t M,
                                              const double* __restrict__ v, const double* __restrict__ v2,
                                              const double* __restrict__ w, double* __restrict__ slab, int64_t slab_ld,
                                              const int64_t wg, const int64_t nwg) {
  constexpr int NW = NT / 64;
  constexpr int VCAP = (NT * CH * 4 < 20000) ? NT * CH * 4 : 20000;  // 160,000 B of the 163,840 B LDS at most
  constexpr bool VFULL = NV == 1 && NT * CH * 4 <= VCAP;             // v zero-filled up to every chunk a thread walks
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

  // Loads are unconditional (a load under a condition makes its registers selects: zero-fills and copies around every
  // load of the streaming loop): a row past n is read as row n - 1 and its row dot is set to zero before phase 2; a chunk
  // past the row's end is read as the thread's chunk 0 — it meets zeros of v in phase 1 (VFULL: v is zero-filled up to the
  // chunks the threads walk; otherwise the guard on v stays) and its column sums are never stored.
  f32x4 kr[R][CH];
  uint32_t kcol[CH];       // element offset of the thread's chunk inside a row (32 bits; the row base is wave-uniform)
#pragma unroll
  for (int c = 0; c < CH; ++c) kcol[c] = cvalid[c] ? (uint32_t)(tid + c * NT) * 4u : 0u;
  auto load_block = [&](int64_t blk, int c) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = blk * R + r < n ? blk * R + r : n - 1;
      const float* rowp = K + row * ldk;
      kr[r][c] = *reinterpret_cast<const f32x4*>(rowp + kcol[c]);
    }
  };

  int64_t blk = wg;
  if (blk < nblk) {
#pragma unroll
    for (int c = 0; c < CH; ++c) load_block(blk, c);
  }
  __syncthreads();  // vs is complete
  int pp = 0;
  for (; blk < nblk; blk += nwg) {
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
          if (VFULL || cvalid[c]) {
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
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = blk * R + r;
      if (row >= n) {                        // (the last block only) a repeated row, not a zero one, was read for it
#pragma unroll
        for (int q = 0; q < NV; ++q) t[q][r] = 0.0;
      } else if (NV == 1 && w != nullptr) {
        t[0][r] += w[row];
      }
    }
    // the doubles phase 1 converted the block to must not stay live into phase 2 (R x CH x 4 of them: spills); an empty asm
    // makes the f32 registers opaque here, phase 2 converts again
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) asm volatile("" : "+v"(kr[r][c]));
    // phase 2: column sums, and the next block's loads re-issued chunk by chunk
    // (the loads are issued unconditionally — behind the last block they re-read it and nobody waits for them: a branch
    // around them makes every register of the block a loop-carried select, two register copies per register and trip)
    const int64_t nxt = blk + nwg < nblk ? blk + nwg : blk;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q][c][e] = fma((double)kr[r][c][e], t[q][r], acc[q][c][e]);
      load_block(nxt, c);
      __builtin_amdgcn_sched_barrier(0);      // the loads of chunk column c go into the registers just consumed, not ahead of them
    }
  }
  double* my = slab + wg * slab_ld * NV;
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


template <int NT, int CH, int R, int NV>
__global__ __launch_bounds__(NT) void knm_pass_kernel(const float* __restrict__ K, int64_t ldk, int64_t n, int64_t M,
                                                      const double* __restrict__ v, const double* __restrict__ v2,
                                                      const double* __restrict__ w, double* __restrict__ slab,
                                                      int64_t slab_ld) {
  knm_pass_body<NT, CH, R, NV>(K, ldk, n, M, v, v2, w, slab, slab_ld, blockIdx.x, gridDim.x);
}

// The same pass for the classes of a batch (blockIdx.y = class): class b's block is walked by grid[b] workgroups exactly as
// its own odx_knm_fwd_bwd launch would walk it (same workgroup -> rows assignment, same slab order), so the sums are the
// single-class call's bit for bit.  Vectors and slabs of the classes lie vstride / slab_stride elements apart.
struct PassBatch {
  const float* K[ODX_MAX_ZBATCH];
  int64_t ldk[ODX_MAX_ZBATCH];
  int64_t n[ODX_MAX_ZBATCH];
  int M[ODX_MAX_ZBATCH];
  int grid[ODX_MAX_ZBATCH];
};

template <int NT, int CH, int R>
__global__ __launch_bounds__(NT) void knm_pass_batched_kernel(PassBatch pb, const double* __restrict__ v, int64_t vstride,
                                                              double* __restrict__ slab, int64_t slab_ld,
                                                              int64_t slab_stride) {
  const int b = blockIdx.y;
  if ((int)blockIdx.x >= pb.grid[b]) return;
  knm_pass_body<NT, CH, R, 1>(pb.K[b], pb.ldk[b], pb.n[b], pb.M[b], v + (int64_t)b * vstride, nullptr, nullptr,
                              slab + (int64_t)b * slab_stride, slab_ld, blockIdx.x, pb.grid[b]);
}

// Sum over the slabs g = wave, wave + 4, ... of one column (`stride` doubles from slab to slab): four loads in flight per
// lane (a chain of one load + one add per slab was latency-bound: 77 us for 512 slabs of 1e4 columns).  ONE function for
// every reducer: the order of the additions depends on (nslab, wave) alone, so the class-batched reducers give, bit for
// bit, what the single-class ones give.
__device__ __forceinline__ double slab_column_sum(const double* __restrict__ col, int64_t stride, int nslab, int wave) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int g = wave;
  for (; g + 12 < nslab; g += 16) {
    const double a0 = col[(int64_t)g * stride], a1 = col[(int64_t)(g + 4) * stride];
    const double a2 = col[(int64_t)(g + 8) * stride], a3 = col[(int64_t)(g + 12) * stride];
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; g < nslab; g += 4) s0 += col[(int64_t)g * stride];
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(PassBatch pb, const double* __restrict__ slab,
                                                                  int64_t slab_ld, int64_t slab_stride,
                                                                  double* __restrict__ out, int64_t ostride) {
  __shared__ double part[4][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t M = pb.M[b];
  const int nslab = pb.grid[b];
  const double* base = slab + (int64_t)b * slab_stride;
  part[wave][lane] = j < M ? slab_column_sum(base + j, slab_ld, nslab, wave) : 0.0;
  __syncthreads();
  if (wave == 0 && j < M) out[(int64_t)b * ostride + j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// ---------------------------------------------------------------- two vectors, one read of K
// out = K' (K v1), out2 = K' (K v2): the same persistent row-block scheme with two sets of row dots and column sums.  The
// second set of f64 column sums needs the registers the addressing of the plain kernel takes (64-bit pointers per load,
// per-chunk validity predicates), so this kernel loads K through buffer descriptors instead: one descriptor per row,
// built from wave-uniform scalars (row base, row length in bytes — zero records for rows past n), the per-lane part is
// one 32-bit byte offset per chunk, and chunks past the row's end read as zero by the hardware range check (no branches
// in the streaming loop).  Both vectors sit in dynamic LDS as f64 (2 x roundup(M, 4) x 8 B: M <= ~10 200).
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));

template <int NT, int CH, int R>
__global__ __launch_bounds__(NT) void knm_pass2_kernel(const float* __restrict__ K, int64_t ldk, int64_t n, int64_t M,
                                                       const double* __restrict__ v1, const double* __restrict__ v2,
                                                       double* __restrict__ slab, int64_t slab_ld) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) double vs2[];       // [2][vcap]
  __shared__ double red[2][NW][2 * R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = (int)((M + 3) >> 2);
  const int vcap = nchunk * 4;
  const int64_t nblk = (n + R - 1) / R;
  for (int i = tid; i < vcap; i += NT) {
    vs2[i] = i < M ? v1[i] : 0.0;
    vs2[vcap + i] = i < M ? v2[i] : 0.0;
  }
  const int voff = tid * 16;        // the only per-lane part of a K address: byte offset inside a (row, chunk column) window
  double acc[2][CH][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[q][c][e] = 0.0;

  f32x4 kr[R][CH];
  const int row_bytes = nchunk * 16;
  auto load_block = [&](int64_t blk, int c) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // descriptor of the window [chunk column c of row `row`]: base and length are wave-uniform scalars; lanes whose
      // chunk lies past the row's end (or rows past n: zero records) get zeros from the range check
      const int64_t row = blk * R + r;
      const bool in = row < n;
      const float* base = K + (in ? row : 0) * ldk + c * NT * 4;
      const int left = row_bytes - c * NT * 16;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, (in && left > 0) ? left : 0, 0x00020000);
      const u32x4b raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
      kr[r][c] = __builtin_bit_cast(f32x4, raw);
    }
  };

  int64_t blk = blockIdx.x;
  if (blk < nblk) {
#pragma unroll
    for (int c = 0; c < CH; ++c) load_block(blk, c);
  }
  __syncthreads();  // vs2 is complete
  int pp = 0;
  for (; blk < nblk; blk += gridDim.x) {
    double t[2][R];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) t[q][r] = 0.0;
    // phase 1: row dots with both vectors.  The v entries a thread needs are the same for every block, so the compiler
    // would lift their LDS reads out of the loop and hold 2 x CH x 4 doubles (80 VGPRs) for good: an opaque zero added to
    // the index keeps the reads inside the loop (LDS traffic is a few % of its bandwidth here).
    int zofs;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zofs));
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ch = tid + c * NT + zofs;
        const int vi = (ch < nchunk ? ch : nchunk - 1) * 4;      // any valid entry where K reads as zero
        const f64x2 a = *reinterpret_cast<const f64x2*>(&vs2[q * vcap + vi]);
        const f64x2 b = *reinterpret_cast<const f64x2*>(&vs2[q * vcap + vi + 2]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          t[q][r] = fma((double)kr[r][c][0], a[0], t[q][r]);
          t[q][r] = fma((double)kr[r][c][1], a[1], t[q][r]);
          t[q][r] = fma((double)kr[r][c][2], b[0], t[q][r]);
          t[q][r] = fma((double)kr[r][c][3], b[1], t[q][r]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double s = t[q][r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[pp][wave][q * R + r] = s;
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < NW; ++u) s += red[pp][u][q * R + r];
        t[q][r] = s;
      }
    pp ^= 1;
    // The f32 -> f64 conversions of phase 1 must not stay live into phase 2 (the compiler would keep 2 x R x CH x 4
    // doubles beside the floats they came from and spill): an empty asm makes the K registers opaque here, so phase 2
    // converts again from the f32 registers (a quarter-rate VALU op on an HBM-bound kernel).
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) asm volatile("" : "+v"(kr[r][c]));
    // phase 2: column sums, and the next block's loads re-issued chunk by chunk
    const int64_t nxt = blk + gridDim.x < nblk ? blk + gridDim.x : blk;      // (unconditional loads, as in knm_pass_kernel)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q][c][e] = fma((double)kr[r][c][e], t[q][r], acc[q][c][e]);
      load_block(nxt, c);
    }
  }
  double* my = slab + (int64_t)blockIdx.x * slab_ld * 2;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = tid + c * NT;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) my[q * slab_ld + (int64_t)ch * 4 + e] = acc[q][c][e];
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
  part[wave][lane] = j < M ? slab_column_sum(slab + j, slab_ld, nslab, wave) : 0.0;
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
  part[wave][lane] = j < M ? slab_column_sum(base + j, 2 * slab_ld, nslab, wave) : 0.0;
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

// ---- class-batched pass (internal; the batched CG of solve.cpp drives it)
bool knm_pass_batch_cfg(int B, const int64_t* M, int* nt, int* ch, int* r) {
  PassCfg c0;
  for (int b = 0; b < B; ++b) {
    PassCfg c;
    if (M[b] <= 0 || !pick_cfg(M[b], &c)) return false;
    if (b == 0) c0 = c;
    else if (c.nt != c0.nt || c.ch != c0.ch || c.r != c0.r) return false;
  }
  if (nt) *nt = c0.nt;
  if (ch) *ch = c0.ch;
  if (r) *r = c0.r;
  return B > 0;
}

static void batch_geometry(int B, const int64_t* n, const int64_t* M, const PassCfg& cfg, int* gmax, int64_t* slab_ld) {
  int g = 1;
  int64_t mm = 1;
  for (int b = 0; b < B; ++b) {
    if (n[b] > 0) g = std::max(g, grid_for(cfg, n[b]));
    mm = std::max(mm, M[b]);
  }
  *gmax = g;
  *slab_ld = round_up(mm, 4);
}

int64_t knm_pass_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M) {
  PassCfg cfg;
  int nt, ch, r;
  if (!knm_pass_batch_cfg(B, M, &nt, &ch, &r) || !pick_cfg(M[0], &cfg)) return ODX_ERR_UNSUPPORTED;
  int gmax;
  int64_t slab_ld;
  batch_geometry(B, n, M, cfg, &gmax, &slab_ld);
  return (int64_t)B * gmax * slab_ld * (int64_t)sizeof(double);
}

#define ODX_PASSB_LAUNCH(NT_, CH_, R_)                                                                               \
  hipLaunchKernelGGL((knm_pass_batched_kernel<NT_, CH_, R_>), dim3(gmax, B), dim3(NT_), 0, s, pb, v, vstride, slab, \
                     slab_ld, slab_stride)

// out[b] = K_b' (K_b v[b]) for the B classes of a batch with ONE pass launch and ONE reduce launch; class b is handled
// exactly as odx_knm_fwd_bwd(K_b, ..) would handle it (same configuration, same workgroup count, same slab order).
int knm_pass_batched(int B, const float* const* K, const int64_t* ldk, const int64_t* n, const int64_t* M, const double* v,
                     int64_t vstride, double* out, int64_t ostride, void* workspace, int64_t workspace_bytes,
                     hipStream_t s) {
  ODX_REQUIRE(B >= 1 && B <= ODX_MAX_ZBATCH, "knm_pass_batched: 1..%d classes", ODX_MAX_ZBATCH);
  PassCfg cfg;
  ODX_REQUIRE(knm_pass_batch_cfg(B, M, nullptr, nullptr, nullptr) && pick_cfg(M[0], &cfg),
              "knm_pass_batched: the classes of a batch must share one pass configuration");
  int gmax;
  int64_t slab_ld;
  batch_geometry(B, n, M, cfg, &gmax, &slab_ld);
  const int64_t slab_stride = (int64_t)gmax * slab_ld;
  if (workspace == nullptr || workspace_bytes < (int64_t)B * slab_stride * (int64_t)sizeof(double)) {
    set_error("knm_pass_batched: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  PassBatch pb;
  for (int b = 0; b < ODX_MAX_ZBATCH; ++b) {
    const bool on = b < B;
    pb.K[b] = on ? K[b] : nullptr;
    pb.ldk[b] = on ? ldk[b] : 0;
    pb.n[b] = on ? n[b] : 0;
    pb.M[b] = on ? (int)M[b] : 0;
    pb.grid[b] = (on && n[b] > 0) ? grid_for(cfg, n[b]) : 0;
    if (on && n[b] > 0)
      ODX_REQUIRE(K[b] && ldk[b] % 4 == 0 && ldk[b] >= round_up(M[b], 4) && aligned16(K[b]),
                  "knm_pass_batched: class %d: K must be 16-byte aligned with ldk %% 4 == 0, ldk >= roundup(M, 4)", b);
  }
  double* slab = static_cast<double*>(workspace);
  if (cfg.nt == 256 && cfg.ch == 1) ODX_PASSB_LAUNCH(256, 1, 16);
  else if (cfg.nt == 256 && cfg.ch == 2) ODX_PASSB_LAUNCH(256, 2, 8);
  else if (cfg.nt == 256 && cfg.ch == 4) ODX_PASSB_LAUNCH(256, 4, 4);
  else if (cfg.nt == 512 && cfg.ch == 4) ODX_PASSB_LAUNCH(512, 4, 4);
  else if (cfg.nt == 512 && cfg.ch == 5) ODX_PASSB_LAUNCH(512, 5, 4);
  else if (cfg.nt == 512 && cfg.ch == 6) ODX_PASSB_LAUNCH(512, 6, 2);
  else ODX_PASSB_LAUNCH(1024, 5, 1);
  ODX_CHECK_LAUNCH("knm_pass_batched");
  int64_t mm = 1;
  for (int b = 0; b < B; ++b) mm = std::max(mm, M[b]);
  hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3((unsigned)ceil_div(mm, 64), B), dim3(256), 0, s, pb, slab, slab_ld,
                     slab_stride, out, ostride);
  ODX_CHECK_LAUNCH("knm_pass_batched(reduce)");
  return ODX_OK;
}

}  // namespace odx

using namespace odx;

// out[b][j] = sum over class b's nslab[b] slabs (slab_stride doubles from class to class), in the single-class reducer's order
int odx::slab_reduce_batched_f64(int B, const int64_t* M, const int* nslab, const double* slab, int64_t slab_ld, int64_t slab_stride,
                                 double* out, int64_t ostride, hipStream_t s) {
  PassBatch pb;
  int64_t mm = 1;
  for (int b = 0; b < ODX_MAX_ZBATCH; ++b) {
    pb.K[b] = nullptr; pb.ldk[b] = 0; pb.n[b] = 0;
    pb.M[b] = b < B ? (int)M[b] : 0;
    pb.grid[b] = b < B ? nslab[b] : 0;
    if (b < B && M[b] > mm) mm = M[b];
  }
  hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3((unsigned)ceil_div(mm, 64), B), dim3(256), 0, s, pb, slab, slab_ld, slab_stride, out,
                     ostride);
  ODX_CHECK_LAUNCH("slab_reduce_batched_f64");
  return ODX_OK;
}

int odx::slab_reduce_f64(const double* slab, int64_t slab_ld, int nslab, int64_t M, double* out, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(M, 64)), dim3(256), 0, s, slab, slab_ld, nslab, M, out);
  ODX_CHECK_LAUNCH("slab_reduce_f64");
  return ODX_OK;
}

int odx::slab_reduce2_f64(const double* slab, int64_t slab_ld, int nslab, int64_t M, double* out, double* out2, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3((unsigned)ceil_div(M, 64), 2), dim3(256), 0, s, slab, slab_ld, nslab, M, out, out2);
  ODX_CHECK_LAUNCH("slab_reduce2_f64");
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
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_pass2_kernel<NT_, CH_, R_>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                          \
    hipLaunchKernelGGL((knm_pass2_kernel<NT_, CH_, R_>), dim3(grid), dim3(NT_), lds2, s, K, ldk, n, M, v, v2, slab,   \
                       slab_ld);                                                                                        \
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
