// f64 dense linear algebra of the FALKON preconditioner (and the RLS solve) on gfx950:
// blocked Cholesky, triangular inverse, transposes, triangular matrix-vector products.
// The O(M^3) work runs as NT GEMMs on v_mfma_f64_16x16x4_f64 (gemm.hip); only the 128 x 128
// diagonal blocks are factored / inverted by a single workgroup inside LDS.
#include "odx_internal.h"
#include <vector>

namespace odx {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- diagonal block: chol + inverse
// One 512-thread workgroup owns a 128 x 128 diagonal block in LDS (rows padded to 129 f64)
// and works on it in 32-wide panels, so there are a dozen workgroup barriers instead of one
// per column:
//   chol32_wave     : wave 0 factors a 32 x 32 diagonal block, one row per lane, pivots and
//                     columns exchanged by v_readlane
//   inv32_wave      : wave 0 inverts a 32 x 32 lower-triangular block, one column per lane,
//                     L read from LDS as broadcasts
//   panel / trailing: every thread owns <= 6 panel outputs / a 6 x 3 register tile
// Rows/cols >= jb are padded with the identity so every loop is uniform.
constexpr int DB_NB = POTRF_NB;   // 128
constexpr int DB_LD = DB_NB + 1;  // LDS row stride (f64): odd => column walks are conflict-free
constexpr int DB_XLD = 33;
constexpr int DB_NT = 512;       // threads: 8 waves => 256 VGPRs per lane for the one-wave 32 x 32 steps
constexpr int DB_PQ = 3072 / DB_NT;  // panel outputs per thread (96 rows x 32 cols at most)

// value of v in lane `src` (compile-time constant) as a wave-uniform scalar: v_readlane_b32 x 2
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void chol32_wave(double* s, double* rd, int J, int jb, int32_t* info, int info_base,
                                            int lane) {
  const int l = lane & 31;
  double row[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) row[k] = s[(J + l) * DB_LD + J + k];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    double piv = readlane_f64(row[c], c);
    if (!(piv > 0.0)) {
      if (lane == 0 && J + c < jb && *info == 0) *info = info_base + J + c + 1;
      piv = 1.0;
    }
    // 1/sqrt by the hardware estimate + two Newton steps (the IEEE sqrt and divide sequences sit on
    // the serial critical path of the factorisation; |rel. err| after refinement < 2^-52)
    double inv = __builtin_amdgcn_rsq(piv);
    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
    const double d = piv * inv;
    if (l == c) rd[c] = inv;
    const double lc = (l > c) ? row[c] * inv : ((l == c) ? d : 0.0);
    row[c] = lc;
#pragma unroll
    for (int k = c + 1; k < 32; ++k) row[k] = fma(-lc, readlane_f64(lc, k), row[k]);
  }
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < 32; ++k) s[(J + l) * DB_LD + J + k] = (k <= l) ? row[k] : 0.0;
  }
}

// xi (32 x 33, lower, zeros above) = inverse of the lower-triangular block of s at (J, J)
__device__ __forceinline__ void inv32_wave(const double* s, const double* rd, int J, double* xi, int lane) {
  const int c = lane & 31;
  double x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) acc = fma(-s[(J + i) * DB_LD + J + k], x[k], acc);
    x[i] = acc * rd[i];
  }
  if (lane < 32) {
#pragma unroll
    for (int i = 0; i < 32; ++i) xi[i * DB_XLD + c] = x[i];
  }
}

// In-LDS blocked Cholesky of the 128 x 128 block (lower in, lower out, strict upper zero).
__device__ __forceinline__ void lds_chol_128(double* s, double* xi, double* rd, double* __restrict__ gx, int jb,
                                             int32_t* info, int info_base) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int J = 0; J < DB_NB; J += 32) {
    if (J >= jb) {
      // a 32-block of the identity padding behind a ragged last block (jb < 128; the RLS systems' 1025 = 8 x 128 + 1 rows end in
      // a block of ONE row): its factor and its inverse are the identity already — nothing to compute, 16 us per block saved
      for (int e = tid; e < 1024; e += DB_NT) gx[(J >> 5) * 1024 + e] = (e >> 5) == (e & 31) ? 1.0 : 0.0;
      continue;
    }
    if (wave == 0) {
      chol32_wave(s, rd, J, jb, info, info_base, lane);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the block's LDS writes before its reads
      inv32_wave(s, rd, J, xi, lane);
    }
    __syncthreads();
    // the inverse phase needs X_JJ again: park it in global scratch (L2) instead of re-deriving it
    for (int e = tid; e < 1024; e += DB_NT) gx[(J >> 5) * 1024 + e] = xi[(e >> 5) * DB_XLD + (e & 31)];
    const int nrem = DB_NB - J - 32;  // rows below the panel
    if (nrem > 0 && J + 32 < jb) {   // (rows of the identity padding below: their panel is zero and stays zero)
      // Both products run on v_mfma_f64_16x16x4_f64 (16 x 16 tiles dealt round-robin to the 8 waves); operands come
      // straight from LDS: lane l feeds row / column (l & 15), k = 4 ks + (l >> 4).
      const int tr = lane & 15, kq = lane >> 4, nt16 = nrem >> 4;
      // panel: L21[i][c] = sum_k A21[i][k] * X11[c][k]       (nrem x 32, k = 32)
      f64x4v pacc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + 8 * u;
        pacc[u] = f64x4v{0.0, 0.0, 0.0, 0.0};
        if (t < 2 * nt16) {
          const int tm = t >> 1, tn = t & 1;
          const double* pa = s + (J + 32 + tm * 16 + tr) * DB_LD + J + kq;
          const double* pb = xi + (tn * 16 + tr) * DB_XLD + kq;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) pacc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], pacc[u], 0, 0, 0);
        }
      }
      __syncthreads();      // every tile has read its A21 rows before any of them is overwritten
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = wave + 8 * u;
        if (t < 2 * nt16) {
          const int tm = t >> 1, tn = t & 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) s[(J + 32 + tm * 16 + kq + 4 * q) * DB_LD + J + tn * 16 + tr] = pacc[u][q];
        }
      }
      __syncthreads();
      // trailing update, lower tiles (a >= b): s[i][k2] -= sum_c L21[i][c] L21[k2][c]      (k = 32)
      const int ntl = nt16 * (nt16 + 1) / 2;
#pragma unroll 1
      for (int t = wave; t < ntl; t += 8) {
        int ta = 0, tb = t;
        while (tb > ta) { tb -= ta + 1; ++ta; }          // t -> (ta, tb), tb <= ta
        const double* pa = s + (J + 32 + ta * 16 + tr) * DB_LD + J + kq;
        const double* pb = s + (J + 32 + tb * 16 + tr) * DB_LD + J + kq;
        f64x4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[(J + 32 + ta * 16 + kq + 4 * q) * DB_LD + J + 32 + tb * 16 + tr] -= acc[q];
      }
      __syncthreads();
    }
  }
}

// In-place inverse of the lower-triangular 128 x 128 block in LDS, block column by block column
// from the right:  X_JJ = L_JJ^-1;  X_[below,J] = -X_[below,below] (L_[below,J] X_JJ).
template <bool CACHED>
__device__ __forceinline__ void lds_trinv_128(double* s, double* xi, double* rd, const double* __restrict__ gx) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int J = DB_NB - 32; J >= 0; J -= 32) {
    __syncthreads();
    if (CACHED) {
      for (int e = tid; e < 1024; e += DB_NT) xi[(e >> 5) * DB_XLD + (e & 31)] = gx[(J >> 5) * 1024 + e];
    } else {
      if (tid < 32) rd[tid] = 1.0 / s[(J + tid) * DB_LD + J + tid];
      __syncthreads();
      if (wave == 0) inv32_wave(s, rd, J, xi, lane);
    }
    __syncthreads();
    const int nrem = DB_NB - J - 32;
    if (nrem > 0) {
      double w[DB_PQ];
      // W[i][c] = sum_k L[i][J+k] X_JJ[k][c]
#pragma unroll
      for (int q = 0; q < DB_PQ; ++q) {
        const int e = tid + DB_NT * q;
        const int i = J + 32 + (e >> 5), c = e & 31;
        double a = 0.0;
        if (i < DB_NB) {
#pragma unroll 8
          for (int k = 0; k < 32; ++k) a = fma(s[i * DB_LD + J + k], xi[k * DB_XLD + c], a);
        }
        w[q] = a;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < DB_PQ; ++q) {
        const int e = tid + DB_NT * q;
        const int i = J + 32 + (e >> 5), c = e & 31;
        if (i < DB_NB) s[i * DB_LD + J + c] = w[q];
      }
      __syncthreads();
      // Y[i][c] = -sum_{k = J+32 .. i} X[i][k] W[k][c]
#pragma unroll
      for (int q = 0; q < DB_PQ; ++q) {
        const int e = tid + DB_NT * q;
        const int i = J + 32 + (e >> 5), c = e & 31;
        double a = 0.0;
        if (i < DB_NB) {
          for (int k = J + 32; k <= i; ++k) a = fma(s[i * DB_LD + k], s[k * DB_LD + J + c], a);
        }
        w[q] = -a;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < DB_PQ; ++q) {
        const int e = tid + DB_NT * q;
        const int i = J + 32 + (e >> 5), c = e & 31;
        if (i < DB_NB) s[i * DB_LD + J + c] = w[q];
      }
    }
    for (int e = tid; e < 1024; e += DB_NT) {  // the diagonal block itself
      const int r = e >> 5, c = e & 31;
      s[(J + r) * DB_LD + J + c] = xi[r * DB_XLD + c];
    }
  }
  __syncthreads();
}

// (eight loads in flight per thread, none behind a branch: an absent element reads A[0] and is replaced afterwards — a load
// under a condition is waited for at the join, one memory latency per element, 32 of them in a row: a quarter of the kernel)
__device__ __forceinline__ void lds_load_lower_128(double* s, const double* __restrict__ A, int64_t lda, int jb) {
#pragma unroll 1
  for (int e0 = threadIdx.x; e0 < DB_NB * DB_NB; e0 += 8 * DB_NT) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + q * DB_NT, i = e >> 7, j = e & 127;
      const bool ok = i < jb && j <= i;
      v[q] = A[ok ? (int64_t)i * lda + j : 0];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + q * DB_NT, i = e >> 7, j = e & 127;
      const bool ok = i < jb && j <= i;
      s[i * DB_LD + j] = ok ? v[q] : (i == j ? 1.0 : 0.0);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void lds_store_dinv_128(const double* s, double* __restrict__ Dinv, int jb) {
  for (int e = threadIdx.x; e < DB_NB * DB_NB; e += DB_NT) {
    const int i = e >> 7, j = e & 127;
    Dinv[e] = (i < jb && j <= i) ? s[i * DB_LD + j] : 0.0;
  }
}

// A (jb x jb, lower part read) -> L in place (strict upper zeroed); Dinv (128 x 128, ld 128) = L^-1.
// WITH_INV = false stops after the factorisation: the four 32 x 32 diagonal inverses stay parked in the tail of the
// block's Dinv slot (where trsm128_kernel reads them) and the 128 x 128 inverse is left to a batched pass.
template <int NB, bool WITH_INV>
__global__ __launch_bounds__(DB_NT) void potrf_diag_kernel(double* __restrict__ A, int64_t lda, int jb,
                                                          double* __restrict__ Dinv, int32_t* __restrict__ info,
                                                          int info_base, int64_t zstrideA, int64_t zstrideD) {
  static_assert(NB == DB_NB, "diagonal-block kernels are built for NB = 128");
  A += (int64_t)blockIdx.x * zstrideA;        // one workgroup per matrix of the class batch
  Dinv += (int64_t)blockIdx.x * zstrideD;
  info += blockIdx.x;
  __shared__ double s[DB_NB * DB_LD];
  __shared__ double xi[32 * DB_XLD];
  __shared__ double rd[32];
  // Dinv doubles as the scratch for the four 32 x 32 diagonal inverses until it is overwritten
  // with the full inverse at the end (4096 doubles at its tail are not touched in between)
  double* gx = Dinv + DB_NB * DB_NB - 4096;
  lds_load_lower_128(s, A, lda, jb);
  lds_chol_128(s, xi, rd, gx, jb, info, info_base);
  if (jb == DB_NB) {
#pragma unroll 4
    for (int e = threadIdx.x; e < DB_NB * DB_NB; e += DB_NT) {
      const int i = e >> 7, j = e & 127;
      A[(int64_t)i * lda + j] = (j <= i) ? s[i * DB_LD + j] : 0.0;
    }
  } else {
    for (int e = threadIdx.x; e < jb * jb; e += DB_NT) {
      const int i = e / jb, j = e % jb;
      A[(int64_t)i * lda + j] = (j <= i) ? s[i * DB_LD + j] : 0.0;
    }
  }
  if (!WITH_INV) return;
  __threadfence_block();
  lds_trinv_128<true>(s, xi, rd, gx);
  lds_store_dinv_128(s, Dinv, jb);
}

// L21 <- A21 L11^-T for a 128-wide panel by block forward substitution over four 32-column blocks, using the
// 32 x 32 diagonal inverses X_JJ left by potrf_diag_kernel<.., false>:
//     L21[:, J] = (A21[:, J] - sum_{k < 32 J} L21[:, k] L11[J, k]') X_JJ'
// One workgroup owns 32 rows; everything it needs sits in LDS (33 + 25 + 8 + 8 KB).  This replaces a generic in-place
// GEMM against the explicit 128 x 128 inverse, which put that inverse (45 us on one CU) on the critical path of
// every block step.
constexpr int TS_R = 64;          // rows per workgroup
constexpr int TS_TLD = 130;       // LDS row strides (f64): 4 r + 2 kq banks over a 32-lane half => conflict-free ds_read_b64
constexpr int TS_BLD = 34;
__global__ __launch_bounds__(256) void trsm128_kernel(double* __restrict__ A21, int64_t lda, int64_t m,
                                                      const double* __restrict__ L11, const double* __restrict__ gx,
                                                      int64_t zstrideA, int64_t zstrideD) {
  extern __shared__ __attribute__((aligned(16))) double ts_lds[];
  A21 += (int64_t)blockIdx.y * zstrideA;      // blockIdx.y = matrix of the class batch
  L11 += (int64_t)blockIdx.y * zstrideA;
  gx += (int64_t)blockIdx.y * zstrideD;
  double* T = ts_lds;                                  // TS_R x TS_TLD; columns of block J double as S
  double* Lb = T + TS_R * TS_TLD;                      // 6 blocks L11[J, I] (I < J), each 32 x TS_BLD, [c][k]
  double* Xb = Lb + 6 * 32 * TS_BLD;                   // 4 blocks X_JJ, each 32 x TS_BLD, [c][k], zero above the diagonal
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * TS_R;
  const int rows = (int)(m - r0 < TS_R ? m - r0 : TS_R);
  // everything this workgroup needs, in one round of loads: all the 16-byte loads of a thread are issued before the
  // first of them is consumed (a load -> LDS store loop would wait out the L2 latency once per element)
  {
    f64x2 vt[16], vl[12], vx[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) {                  // T: 64 rows x 64 pairs
      const int e = tid + 256 * q, r = e >> 6, c = (e & 63) * 2;
      vt[q] = (r < rows) ? *reinterpret_cast<const f64x2*>(A21 + (r0 + r) * lda + c) : f64x2{0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {                  // 6 blocks x 32 rows x 16 pairs
      const int e = tid + 256 * q, blk = e >> 9, i = (e >> 4) & 31, k = (e & 15) * 2;
      const int J = blk < 1 ? 1 : (blk < 3 ? 2 : 3), I = blk - (J * (J - 1)) / 2;
      vl[q] = *reinterpret_cast<const f64x2*>(L11 + (int64_t)(32 * J + i) * lda + 32 * I + k);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) vx[q] = *reinterpret_cast<const f64x2*>(gx + (tid + 256 * q) * 2);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + 256 * q, r = e >> 6, c = (e & 63) * 2;
      *reinterpret_cast<f64x2*>(&T[r * TS_TLD + c]) = vt[q];
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const int e = tid + 256 * q, blk = e >> 9, i = (e >> 4) & 31, k = (e & 15) * 2;
      *reinterpret_cast<f64x2*>(&Lb[(blk * 32 + i) * TS_BLD + k]) = vl[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = (tid + 256 * q) * 2;
      *reinterpret_cast<f64x2*>(&Xb[((e >> 10) * 32 + ((e >> 5) & 31)) * TS_BLD + (e & 31)]) = vx[q];
    }
  }
  __syncthreads();
  // 64 x 32 outputs per column block = 4 x 2 MFMA tiles of 16 x 16; wave w owns row tile w, both column tiles.
  const int ar = wave * 16 + (lane & 15), kq = lane >> 4, bc = lane & 15;
#pragma unroll
  for (int J = 0; J < 4; ++J) {
    f64x4v s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < J; ++I) {
      const double* lb = Lb + (((J * (J - 1)) / 2 + I) * 32) * TS_BLD;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const double a = T[ar * TS_TLD + 32 * I + 4 * ks + kq];
        s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[bc * TS_BLD + 4 * ks + kq], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[(16 + bc) * TS_BLD + 4 * ks + kq], s1, 0, 0, 0);
      }
    }
    // S = A21[:, J] - (L21[:, < J] L11[J, < J]'), written over the J columns of T (only this wave's rows)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave * 16 + (lane >> 4) + 4 * q;
      T[r * TS_TLD + 32 * J + bc] -= s0[q];
      T[r * TS_TLD + 32 * J + 16 + bc] -= s1[q];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): this wave's own LDS writes before it reads them back
    __builtin_amdgcn_wave_barrier();
    f64x4v y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
    const double* xb = Xb + (J * 32) * TS_BLD;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double a = T[ar * TS_TLD + 32 * J + 4 * ks + kq];
      y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[bc * TS_BLD + 4 * ks + kq], y0, 0, 0, 0);
      y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[(16 + bc) * TS_BLD + 4 * ks + kq], y1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();       // every lane of the wave has read S before it is overwritten
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave * 16 + (lane >> 4) + 4 * q;
      T[r * TS_TLD + 32 * J + bc] = y0[q];
      T[r * TS_TLD + 32 * J + 16 + bc] = y1[q];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = tid + 256 * q, r = e >> 6, c = (e & 63) * 2;
    if (r < rows) *reinterpret_cast<f64x2*>(A21 + (r0 + r) * lda + c) = *reinterpret_cast<const f64x2*>(&T[r * TS_TLD + c]);
  }
}
constexpr int TS_LDS_BYTES = (TS_R * TS_TLD + 10 * 32 * TS_BLD) * (int)sizeof(double);   // 153,600 B

// ---------------------------------------------------------------- small utility kernels
__global__ __launch_bounds__(256) void transpose_f64_kernel(const double* __restrict__ src, int64_t lds_,
                                                            double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                            int64_t cols, int64_t zstride_src, int64_t zstride_dst) {
  __shared__ double tile[32][33];
  src += (int64_t)blockIdx.z * zstride_src;
  dst += (int64_t)blockIdx.z * zstride_dst;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = r0 + ty + q * 8, c = c0 + tx;
    tile[ty + q * 8][tx] = (r < rows && c < cols) ? src[r * lds_ + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = c0 + ty + q * 8, c = r0 + tx;  // dst is cols x rows
    if (r < cols && c < rows) dst[r * ldd + c] = tile[tx][ty + q * 8];
  }
}

__global__ __launch_bounds__(256) void add_diag_f64_kernel(double* A, int64_t lda, int64_t M, double value,
                                                           int64_t zstride) {
  A += (int64_t)blockIdx.y * zstride;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < M) A[i * lda + i] += value;
}

__global__ __launch_bounds__(256) void fill_f64_kernel(double* A, int64_t lda, int64_t rows, int64_t cols,
                                                       double value) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c < cols && r < rows) A[r * lda + c] = value;
}

template <typename S, typename Dt>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ src, int64_t lds_, Dt* __restrict__ dst,
                                                      int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c < cols && r < rows) dst[r * ldd + c] = (Dt)src[r * lds_ + c];
}

// copies the NB x NB diagonal inverses onto the diagonals of Li (as is) and Lit (transposed)
__global__ __launch_bounds__(256) void place_diag_inverses_kernel(const double* __restrict__ Dinv, int nb,
                                                                  int64_t M, double* __restrict__ Li,
                                                                  double* __restrict__ Lit, int64_t ld,
                                                                  int64_t zstrideD, int64_t zstrideO) {
  Dinv += (int64_t)blockIdx.z * zstrideD;
  Li += (int64_t)blockIdx.z * zstrideO;
  Lit += (int64_t)blockIdx.z * zstrideO;
  const int b = blockIdx.y;
  const int64_t r0 = (int64_t)b * nb;
  const double* D = Dinv + (int64_t)b * nb * nb;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nb * nb; e += gridDim.x * 256) {
    const int i = e / nb, j = e % nb;
    if (r0 + i < M && r0 + j < M && j <= i) {
      const double v = D[e];
      Li[(r0 + i) * ld + r0 + j] = v;
      Lit[(r0 + j) * ld + r0 + i] = v;
    }
  }
}

// y[i] = alpha * sum_{j in tri range} Tri[i][j] x[j] + beta * z[i]; one wave per row, 16-byte loads, four of them in
// flight per lane (four partial sums) over the aligned pairs of the row's range; the odd elements at its ends go to lane 0.
__device__ __forceinline__ void trmv_row(const double* __restrict__ Tri, int64_t ld, int64_t M, int uplo,
                                         const double* __restrict__ x, double alpha, double beta,
                                         const double* __restrict__ z, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= M) return;
  if (!uplo) i = M - 1 - i;      // longest rows first: a lower factor's long rows, dispatched last, were the launch's tail
  const int64_t jlo = uplo ? i : 0, jhi = uplo ? M : i + 1;  // [jlo, jhi)
  const double* row = Tri + i * ld;
  const int64_t q0 = (jlo + 1) >> 1, q1 = jhi >> 1;           // whole pairs [2 q0, 2 q1); ld is even so pairs are 16-B aligned
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (lane == 0) {
    if (jlo & 1) s0 = row[jlo] * x[jlo];
    if (jhi & 1) s1 = row[jhi - 1] * x[jhi - 1];
  }
  int64_t p = q0 + lane;
  for (; p + 192 < q1; p += 256) {
    const f64x2 a0 = *reinterpret_cast<const f64x2*>(row + 2 * p), a1 = *reinterpret_cast<const f64x2*>(row + 2 * (p + 64));
    const f64x2 a2 = *reinterpret_cast<const f64x2*>(row + 2 * (p + 128)), a3 = *reinterpret_cast<const f64x2*>(row + 2 * (p + 192));
    const f64x2 b0 = *reinterpret_cast<const f64x2*>(x + 2 * p), b1 = *reinterpret_cast<const f64x2*>(x + 2 * (p + 64));
    const f64x2 b2 = *reinterpret_cast<const f64x2*>(x + 2 * (p + 128)), b3 = *reinterpret_cast<const f64x2*>(x + 2 * (p + 192));
    s0 = fma(a0[1], b0[1], fma(a0[0], b0[0], s0));
    s1 = fma(a1[1], b1[1], fma(a1[0], b1[0], s1));
    s2 = fma(a2[1], b2[1], fma(a2[0], b2[0], s2));
    s3 = fma(a3[1], b3[1], fma(a3[0], b3[0], s3));
  }
  for (; p < q1; p += 64) {
    const f64x2 a = *reinterpret_cast<const f64x2*>(row + 2 * p);
    const f64x2 b = *reinterpret_cast<const f64x2*>(x + 2 * p);
    s0 = fma(a[1], b[1], fma(a[0], b[0], s0));
  }
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    double r = alpha * s;
    if (beta != 0.0) r += beta * z[i];
    y[i] = r;
  }
}

__global__ __launch_bounds__(256) void trmv_f64_kernel(const double* __restrict__ Tri, int64_t ld, int64_t M,
                                                       int uplo, const double* __restrict__ x, double alpha,
                                                       double beta, const double* __restrict__ z,
                                                       double* __restrict__ y) {
  trmv_row(Tri, ld, M, uplo, x, alpha, beta, z, y);
}

// The same product for the classes of a batch (blockIdx.y = class): factors tri_stride apart, vectors vstride apart,
// each class with its own size and (when `scaled`) its own alpha.  Per class the arithmetic of trmv_f64_kernel.
__global__ __launch_bounds__(256) void trmv_batched_kernel(const double* __restrict__ Tri, int64_t ld, int64_t tri_stride,
                                                           int uplo, VecBatch vb, const double* __restrict__ x,
                                                           int64_t xstride, int scaled, double beta,
                                                           const double* __restrict__ z, int64_t zstride,
                                                           double* __restrict__ y, int64_t ystride) {
  const int b = blockIdx.y;
  trmv_row(Tri + (int64_t)b * tri_stride, ld, vb.M[b], uplo, x + (int64_t)b * xstride, scaled ? vb.scale[b] : 1.0, beta,
           z ? z + (int64_t)b * zstride : nullptr, y + (int64_t)b * ystride);
}

// ---------------------------------------------------------------- host drivers
int transpose_f64(const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                  hipStream_t stream, int zcount, int64_t zstride_src, int64_t zstride_dst) {
  if (rows <= 0 || cols <= 0 || zcount <= 0) return ODX_OK;
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32), (unsigned)zcount);
  ODX_REQUIRE(grid.y < 65536 && zcount < 65536, "transpose_f64: too many rows");
  hipLaunchKernelGGL(transpose_f64_kernel, grid, dim3(256), 0, stream, src, lds_, dst, ldd, rows, cols, zstride_src,
                     zstride_dst);
  ODX_CHECK_LAUNCH("transpose_f64");
  return ODX_OK;
}

int add_diag_f64(double* A, int64_t lda, int64_t M, double value, hipStream_t stream, int zcount, int64_t zstride) {
  if (M <= 0 || zcount <= 0) return ODX_OK;
  hipLaunchKernelGGL(add_diag_f64_kernel, dim3((unsigned)ceil_div(M, 256), (unsigned)zcount), dim3(256), 0, stream, A,
                     lda, M, value, zstride);
  ODX_CHECK_LAUNCH("add_diag_f64");
  return ODX_OK;
}

int fill_f64(double* A, int64_t lda, int64_t rows, int64_t cols, double value, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return ODX_OK;
  ODX_REQUIRE(rows < 65536 * 32768ll, "fill_f64: too many rows");
  // rows go on grid.y in slabs of <= 65535
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)ceil_div(cols, 256), (unsigned)nr), dim3(256), 0, stream,
                       A + r0 * lda, lda, nr, cols, value);
  }
  ODX_CHECK_LAUNCH("fill_f64");
  return ODX_OK;
}

// 128 x 128 inverses of all diagonal blocks of a lower-triangular L, one workgroup per block.
__global__ __launch_bounds__(DB_NT) void trtri_diag_kernel(const double* __restrict__ L, int64_t ldl, int64_t M,
                                                          double* __restrict__ Dinv, int64_t zstrideA,
                                                          int64_t zstrideD) {
  __shared__ double s[DB_NB * DB_LD];
  L += (int64_t)blockIdx.y * zstrideA;
  Dinv += (int64_t)blockIdx.y * zstrideD;
  __shared__ double xi[32 * DB_XLD];
  const int64_t r0 = (int64_t)blockIdx.x * DB_NB;
  const int jb = (int)(M - r0 < DB_NB ? M - r0 : DB_NB);
  __shared__ double rd[32];
  lds_load_lower_128(s, L + r0 * ldl + r0, ldl, jb);
  lds_trinv_128<false>(s, xi, rd, nullptr);
  lds_store_dinv_128(s, Dinv + (int64_t)blockIdx.x * DB_NB * DB_NB, jb);
}

struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int device = -1;
  bool owned = true;           // made here (destroyed on release) — or handed in by the caller (odx_set_helper_streams)
};

// Internal helper streams, a pair per (host thread, device, calling stream): slot 0 forks the inverse of L_T inside the
// preconditioner, slot 1 carries the look-ahead trailing updates of potrf_f64.  Keyed by the calling stream so that
// factorisations issued on different streams (classes trained side by side) do not queue behind each other on one
// helper.  Everything is ordered with events; the host never waits.
// CU mask the helper streams are created with (odx_set_side_stream_cu_mask: a factorisation chain confined to a part of
// the chip takes its look-ahead and fork streams along); empty = the whole device.  Streams that exist keep their mask.
static std::vector<uint32_t> g_side_mask;
// Helper streams handed in by the host side (odx_set_helper_streams): which hardware queue a stream created HERE lands on depends
// on how many streams the process has created before, and a look-ahead helper that shares the main stream's queue serialises
// the chain's GEMMs behind every build (round 6: 6.83 instead of 6.68 s per headline step after three more streams had been
// probed at start-up).  The host measures which streams sit on queues of their own (odx/streams.py) and gives two of them.
static hipStream_t g_given_helper[2] = {nullptr, nullptr};

// Helper streams pay where the chain is GEMM-bound (the headline's M = 1e4: the look-ahead update beside the next panel, the
// inverse of L_T beside T T').  Below CHAIN_HELPER_MIN_M centres a chain is latency-bound and is run several at a time by its
// callers (fit_batch's half chains, the classes of a Minibootstrap round): every extra stream then competes for the runtime's
// few hardware queues (4 by default) with the streams of the other chains — measured on a Minibootstrap round: 0.48 s with
// the small chains in order on their caller's stream, 0.52-0.57 s with helpers, depending on which streams collided.
// The option chain_helpers (odx_set_option: -1 automatic, 0 never, 1 always) overrides the rule.
constexpr int64_t CHAIN_HELPER_MIN_M = 4096;

static bool chain_helpers(int64_t M) {
  const int o = lib_option(OPT_CHAIN_HELPERS);
  if (o == 0 || o == 1) return o == 1;
  return M >= CHAIN_HELPER_MIN_M;
}

// The helper streams stay until the caller releases them (odx_release_helper_streams: the host side does at the end of a fit /
// a training step; releasing inside every chain cost 7 % on a 0.2 s training step and is gone).
struct SideEntry {
  int device;
  hipStream_t caller;
  SideStream ss[2];
};
static thread_local std::vector<SideEntry*> g_side_pool;

// Destroys the calling thread's helper streams (they come back on demand).  A process that goes on to latency-bound work after
// its factorisations should call this: on this runtime the helper streams of a class-batched chain, idle but alive, cost every
// later small launch of the process (the bench line's one-image forwards ran 4.7 -> 7.6 ms behind the headline job until its
// extras released them; idle streams that never carried a chain do not have that effect — tools/stream_footprint_probe.py).
static int release_side_streams(bool wait = true) {
  for (SideEntry* e : g_side_pool) {
    for (SideStream& s : e->ss) {
      if (s.stream != nullptr) {
        // (without the wait: the runtime keeps a destroyed stream and its events until the work queued on them has completed)
        if (wait) ODX_CHECK_HIP(hipStreamSynchronize(s.stream));
        if (s.owned) ODX_CHECK_HIP(hipStreamDestroy(s.stream));
      }
      if (s.fork != nullptr) ODX_CHECK_HIP(hipEventDestroy(s.fork));
      if (s.join != nullptr) ODX_CHECK_HIP(hipEventDestroy(s.join));
    }
    delete e;
  }
  g_side_pool.clear();
  return ODX_OK;
}

static int side_stream(SideStream** out, int slot, hipStream_t caller, int64_t M) {
  typedef SideEntry Entry;
  std::vector<Entry*>& pool = g_side_pool;
  int dev = 0;
  ODX_CHECK_HIP(hipGetDevice(&dev));
  ODX_REQUIRE(slot >= 0 && slot < 2, "side_stream: bad slot");
  if (!chain_helpers(M)) {
    // no helper: the "side" work goes on the caller's stream, in order (the fork / join events become no-ops there)
    static thread_local SideStream inl[2];
    SideStream& s = inl[slot];
    if (s.fork == nullptr || s.device != dev) {
      ODX_CHECK_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
      ODX_CHECK_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
      s.device = dev;
    }
    s.stream = caller;
    *out = &s;
    return ODX_OK;
  }
  Entry* e = nullptr;
  for (Entry* c : pool)
    if (c->device == dev && c->caller == caller) e = c;
  if (e == nullptr) {
    e = new Entry();
    e->device = dev;
    e->caller = caller;
    pool.push_back(e);
  }
  SideStream& s = e->ss[slot];
  if (s.stream == nullptr) {
    s.owned = true;
    if (g_given_helper[slot] != nullptr && g_side_mask.empty()) {
      s.stream = g_given_helper[slot];
      s.owned = false;
    } else if (!g_side_mask.empty()) {
      ODX_CHECK_HIP(hipExtStreamCreateWithCUMask(&s.stream, (uint32_t)g_side_mask.size(), g_side_mask.data()));
    } else if (slot == 1) {
      // The look-ahead updates are bulk work that must not starve the latency-bound chain they overlap with: lowest
      // priority, so freed CU slots go to the chain's small kernels first.
      int lo = 0, hi = 0;
      ODX_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      ODX_CHECK_HIP(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, lo));
    } else {
      ODX_CHECK_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    }
    ODX_CHECK_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
    ODX_CHECK_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
    s.device = dev;
  }
  *out = &s;
  return ODX_OK;
}

// Two-level right-looking blocked Cholesky (lower).  Outer panels of 512 columns, inner blocks
// of NB = 128:
//   inner, per 128-block inside the panel:
//     L11 = chol(A11), D = L11^-1          (one workgroup, LDS)
//     L21 = A21 D'                         (NT GEMM, in place: one column tile per row panel)
//     A[below, rest of panel] -= L21 L21[rest of panel]'      (k = 128, only <= 384 columns wide)
//   outer, per panel:
//     A22 -= L21 L21'                      (k = 512: one read-modify-write of the trailing matrix
//                                           per 512 columns instead of per 128, MFMA-bound)
constexpr int POTRF_NBO = 512;

// The GEMM-shaped work of the A factor (T T' / M, the rank-512 updates of its Cholesky) on the split-f16 tile core: A only
// preconditions the system (the solution the CG converges to does not depend on it, and the reference's all-f32 falkon gives it
// f32 accuracy at best), so its products are formed at f32 accuracy at ~6 x the f64 MFMA rate; T, whose products define the
// regulariser, stays f64 throughout.  Option precond (odx_set_option): 1 = never ("f64"), 2 = always ("split"), 0 = from 4096
// centres on (below that the chain is latency-bound and nothing is gained).
static bool precond_split(int64_t M) {
  const int o = lib_option(OPT_PRECOND);
  if (o == 1) return false;
  if (o == 2) return true;
  return M >= 4096;
}
// scratch of the split path per matrix, in doubles: 2 M roundup(M, 64) 4-byte units for the packed T and, later, the four
// packed operands of a triangular-inverse merge level + two packed 512-column panels
static int64_t precond_pack_units(int64_t M) { return 2 * M * h2_f64_packed_ld(M); }     // packed T, later the packs of a merge level
static int64_t precond_split_doubles(int64_t M) { return (precond_pack_units(M) + 2 * M * POTRF_NBO) / 2 + 2; }
// power of two s with bound * s in [2^11, 2^12]: the entries' two f16 terms stay far from overflow (65504)
static float split_scale_for(double bound) {
  int e = 0;
  frexp(bound > 1e-300 ? bound : 1.0, &e);      // bound = f 2^e, f in [0.5, 1)
  return ldexpf(1.f, 12 - e);
}


// pk != nullptr: the rank-512 trailing updates (the bulk of the flops) run on the split-f16 tile core (gemm_h2_f64:
// f32-accurate products added into the f64 trailing matrix) — for a factor that only preconditions.  pk: two packed panel
// buffers per matrix, pk_buf 4-byte units apart, matrices pk_z apart; pk_scale: power of two with |L_ij| pk_scale << 65504.
int potrf_f64(double* A, int64_t lda, int64_t M, double* Dinv, int32_t* info, hipStream_t stream, const ZBatch& zb, uint32_t* pk,
              int64_t pk_buf, int64_t pk_z, float pk_scale) {
  constexpr int NB = POTRF_NB;
  ODX_REQUIRE(lda % 2 == 0 && aligned16(A) && aligned16(Dinv), "potrf_f64: A/Dinv must be 16-byte aligned, lda even");
  const int Z = zb.count;
  ODX_REQUIRE(Z >= 1 && Z <= ODX_MAX_ZBATCH && zb.strideA % 2 == 0 && zb.strideD % 2 == 0,
              "potrf_f64: class batch of 1..%d matrices, even strides", ODX_MAX_ZBATCH);
  auto zgemm = [&](GemmParams<double>& g) {      // the same product for every matrix of the class batch
    g.zbatches = Z; g.zstrideA = zb.strideA; g.zstrideB = zb.strideA; g.zstrideC = zb.strideA;
  };
  ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(trsm128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    TS_LDS_BYTES));
  SideStream* look = nullptr;
  ODX_PROPAGATE(side_stream(&look, 1, stream, M));
  bool pending = false;      // a trailing update is in flight on the helper stream
  for (int64_t K0 = 0; K0 < M; K0 += POTRF_NBO) {
    const int64_t kbo = M - K0 < POTRF_NBO ? M - K0 : POTRF_NBO;
    for (int64_t k0 = K0; k0 < K0 + kbo; k0 += NB) {
      const int jb = (int)(M - k0 < NB ? M - k0 : NB);
      double* D = Dinv + (k0 / NB) * NB * NB;
      hipLaunchKernelGGL((potrf_diag_kernel<NB, false>), dim3((unsigned)Z), dim3(DB_NT), 0, stream, A + k0 * lda + k0, lda,
                         jb, D, info, (int)k0, zb.strideA, zb.strideD);
      ODX_CHECK_LAUNCH("potrf_diag");
      const int64_t m = M - k0 - jb;
      if (m <= 0) break;
      double* A21 = A + (k0 + jb) * lda + k0;
      hipLaunchKernelGGL(trsm128_kernel, dim3((unsigned)ceil_div(m, TS_R), (unsigned)Z), dim3(256), TS_LDS_BYTES, stream,
                         A21, lda, m, A + k0 * lda + k0, D + NB * NB - 4096, zb.strideA, zb.strideD);
      ODX_CHECK_LAUNCH("trsm128");
      const int64_t pc = (K0 + kbo) - (k0 + jb);  // panel columns right of this block
      if (pc > 0) {
        GemmParams<double> u;
        u.A = A21; u.lda = lda; u.B = A21; u.ldb = lda; u.C = A + (k0 + jb) * lda + (k0 + jb); u.ldc = lda;
        u.m = m; u.n = pc; u.k = jb; u.alpha = -1.0; u.beta = 1.0; u.flags = ODX_GEMM_LOWER_ONLY;
        zgemm(u);
        ODX_PROPAGATE(launch_gemm_f64(u, stream));
      }
    }
    const int64_t mt = M - K0 - kbo;
    if (mt > 0) {
      // Trailing update with look-ahead: the next panel's columns are brought up to date on the main stream (a thin
      // GEMM), so its latency-bound chain of diagonal blocks can start at once; the rest of the trailing matrix (the
      // bulk of the flops) is updated on the helper stream meanwhile.  The next thin update touches columns the helper
      // is still writing, so it waits for the helper first.
      double* P = A + (K0 + kbo) * lda + K0;
      const int64_t nw = mt < POTRF_NBO ? mt : POTRF_NBO;      // width of the next panel
      if (pending) {
        ODX_CHECK_HIP(hipStreamWaitEvent(stream, look->join, 0));
        pending = false;
      }
      const int64_t mr = mt - nw;
      double minus1[ODX_MAX_ZBATCH];
      for (int z = 0; z < Z; ++z) minus1[z] = -1.0;
      uint32_t* Pk = pk ? pk + ((K0 / POTRF_NBO) & 1) * pk_buf : nullptr;      // (its last reader, two panels back, has been joined)
      if (Pk) {
        ODX_PROPAGATE(split_f64(P, lda, zb.strideA, mt, kbo, pk_scale, Pk, POTRF_NBO, pk_z, Z, stream));
        ODX_PROPAGATE(gemm_h2_f64(Pk, POTRF_NBO, pk_z, pk_scale, Pk, POTRF_NBO, pk_z, pk_scale, A + (K0 + kbo) * (lda + 1), lda,
                                  zb.strideA, mt, nw, kbo, minus1, 1.0, ODX_GEMM_LOWER_ONLY, Z, stream));
      } else {
        GemmParams<double> u;
        u.A = P; u.lda = lda; u.B = P; u.ldb = lda; u.C = A + (K0 + kbo) * (lda + 1); u.ldc = lda;
        u.m = mt; u.n = nw; u.k = kbo; u.alpha = -1.0; u.beta = 1.0; u.flags = ODX_GEMM_LOWER_ONLY;
        zgemm(u);
        ODX_PROPAGATE(launch_gemm_f64(u, stream));
      }
      if (mr > 0) {
        ODX_CHECK_HIP(hipEventRecord(look->fork, stream));
        ODX_CHECK_HIP(hipStreamWaitEvent(look->stream, look->fork, 0));
        if (Pk) {
          ODX_PROPAGATE(gemm_h2_f64(Pk + nw * POTRF_NBO, POTRF_NBO, pk_z, pk_scale, Pk + nw * POTRF_NBO, POTRF_NBO, pk_z, pk_scale,
                                    A + (K0 + kbo + nw) * (lda + 1), lda, zb.strideA, mr, mr, kbo, minus1, 1.0, ODX_GEMM_LOWER_ONLY, Z,
                                    look->stream));
        } else {
          double* P2 = P + nw * lda;
          GemmParams<double> r;
          r.A = P2; r.lda = lda; r.B = P2; r.ldb = lda; r.C = A + (K0 + kbo + nw) * (lda + 1); r.ldc = lda;
          r.m = mr; r.n = mr; r.k = kbo; r.alpha = -1.0; r.beta = 1.0; r.flags = ODX_GEMM_LOWER_ONLY;
          zgemm(r);
          ODX_PROPAGATE(launch_gemm_f64(r, look->stream));
        }
        ODX_CHECK_HIP(hipEventRecord(look->join, look->stream));
        pending = true;
      }
    }
  }
  if (pending) ODX_CHECK_HIP(hipStreamWaitEvent(stream, look->join, 0));
  // Dinv: the 128 x 128 inverses of all diagonal blocks in one batched launch (off the block-by-block critical path)
  hipLaunchKernelGGL(trtri_diag_kernel, dim3((unsigned)ceil_div(M, NB), (unsigned)Z), dim3(DB_NT), 0, stream, A, lda, M,
                     Dinv, zb.strideA, zb.strideD);
  ODX_CHECK_LAUNCH("trtri_diag");
  return ODX_OK;
}

int trmv_batched_f64(const double* Tri, int64_t ld, int64_t tri_stride, int uplo, const VecBatch& vb, const double* x,
                     int64_t xstride, bool scaled, double beta, const double* z, int64_t zstride, double* y,
                     int64_t ystride, hipStream_t stream) {
  int mm = 0;
  for (int b = 0; b < vb.B; ++b) mm = vb.M[b] > mm ? vb.M[b] : mm;
  if (mm <= 0) return ODX_OK;
  hipLaunchKernelGGL(trmv_batched_kernel, dim3((unsigned)ceil_div(mm, 4), (unsigned)vb.B), dim3(256), 0, stream, Tri, ld,
                     tri_stride, uplo, vb, x, xstride, scaled ? 1 : 0, beta, z, zstride, y, ystride);
  ODX_CHECK_LAUNCH("trmv_batched_f64");
  return ODX_OK;
}

// Li = L^-1, Lit = L^-T by pairwise merging of inverted diagonal blocks (s = NB, 2NB, ...):
//   [X11 0; X21 X22] with X21 = -X22 (L21 X11).  All pairs of one level form one batched launch.
//   GEMM 1:  WT = (L21 X11)'        A = L21 (m2 x s), B = X11' = Lit block (upper), stored transposed
//   GEMM 2:  X21 = -X22 WT'         A = X22 (lower), B = WT;  X21 -> Li, X21' -> Lit
// Li and Lit must be zero on entry outside what is written here.  WT: >= M*M doubles.
// pk != nullptr: the merge levels of 512 rows and more run on the split-f16 tile core (the inverse of a factor that only
// preconditions): pk holds pk_cap 4-byte units per matrix (matrices pk_z apart; a level needs at most 2 M roundup(M, 64));
// bound_l / bound_inv: bounds of |L_ij| and of |(L^-1)_ij| the operand scales are taken from.
int trtri_from_diag_f64(const double* L, int64_t ldl, int64_t M, const double* Dinv, double* Li, double* Lit,
                        int64_t ld, double* WT, hipStream_t stream, const ZBatch& zb, uint32_t* pk, int64_t pk_cap, int64_t pk_z,
                        double bound_l, double bound_inv) {
  constexpr int NB = POTRF_NB;
  ODX_REQUIRE(ld % 2 == 0 && ldl % 2 == 0, "trtri_f64: leading dimensions must be even");
  const int Z = zb.count;
  ODX_REQUIRE(Z >= 1 && Z <= ODX_MAX_ZBATCH && zb.strideA % 2 == 0 && zb.strideO % 2 == 0 && zb.strideW % 2 == 0,
              "trtri_f64: class batch of 1..%d matrices, even strides", ODX_MAX_ZBATCH);
  const int nblk = (int)ceil_div(M, NB);
  hipLaunchKernelGGL(place_diag_inverses_kernel, dim3(16, (unsigned)nblk, (unsigned)Z), dim3(256), 0, stream, Dinv, NB, M,
                     Li, Lit, ld, zb.strideD, zb.strideO);
  ODX_CHECK_LAUNCH("place_diag_inverses");
  for (int64_t s = NB; s < M; s *= 2) {
    const int nb = (int)ceil_div(M, 2 * s);  // pairs; the last may be ragged or empty
    const int nbe = (int)ceil_div(M - s, 2 * s);                       // pairs with rows below their first block
    const int64_t m2l = M - s - (int64_t)(nbe - 1) * 2 * s;            // rows of the last pair's second block (<= s)
    const int64_t szr = (int64_t)(nbe - 1) * s * s + (m2l < s ? m2l : s) * s;     // packs whose rows are ragged (ld = s)
    if (pk != nullptr && s >= 512 && 2 * szr + 2 * (int64_t)nbe * s * s <= pk_cap) {
      const float sl = split_scale_for(bound_l), si = split_scale_for(bound_inv), sw = split_scale_for(bound_l * bound_inv);
      uint32_t *pL21 = pk, *pX11 = pL21 + szr, *pX22 = pX11 + (int64_t)nbe * s * s, *pWT = pX22 + szr;
      double one[ODX_MAX_ZBATCH], mone[ODX_MAX_ZBATCH];
      for (int z = 0; z < Z; ++z) { one[z] = 1.0; mone[z] = -1.0; }
      SplitBlocksArgs k1;                    // L21 of every pair: rows ragged
      k1.X = L + s * ldl; k1.ldx = ldl; k1.bsx = 2 * s * (ldl + 1); k1.zsx = zb.strideA;
      k1.P = pL21; k1.ldp = s; k1.bsp = s * s; k1.zsp = pk_z; k1.rows = s; k1.cols = s;
      k1.rg_total = M; k1.rg_off = s; k1.rg_step = 2 * s; k1.rows_ragged = 1; k1.nb = nbe; k1.Z = Z; k1.scale = sl;
      ODX_PROPAGATE(split_f64_blocks(k1, stream));
      SplitBlocksArgs k2;                    // X11' of every pair (the Lit block: upper)
      k2.X = Lit; k2.ldx = ld; k2.bsx = 2 * s * (ld + 1); k2.zsx = zb.strideO;
      k2.P = pX11; k2.ldp = s; k2.bsp = s * s; k2.zsp = pk_z; k2.rows = s; k2.cols = s; k2.nb = nbe; k2.Z = Z; k2.scale = si;
      ODX_PROPAGATE(split_f64_blocks(k2, stream));
      H2F64Args a1;                          // WT = (L21 X11)'
      a1.PA = pL21; a1.ldpa = s; a1.zsa = pk_z; a1.bsa = s * s; a1.sa = sl;
      a1.PB = pX11; a1.ldpb = s; a1.zsb = pk_z; a1.bsb = s * s; a1.sb = si;
      a1.C = WT; a1.ldc = s; a1.zsc = zb.strideW; a1.bsc = s * s;
      a1.m = s; a1.n = s; a1.k = s; a1.rg_total = M; a1.rg_off = s; a1.rg_step = 2 * s;
      a1.flags = ODX_GEMM_B_UPPER | ODX_GEMM_STORE_T; a1.zcount = Z; a1.nb = nbe; a1.beta = 0.0;
      for (int z = 0; z < Z; ++z) a1.alpha[z] = one[z];
      ODX_PROPAGATE(gemm_h2_f64_ex(a1, stream));
      SplitBlocksArgs k3;                    // X22 of every pair: rows and columns ragged
      k3.X = Li + s * (ld + 1); k3.ldx = ld; k3.bsx = 2 * s * (ld + 1); k3.zsx = zb.strideO;
      k3.P = pX22; k3.ldp = s; k3.bsp = s * s; k3.zsp = pk_z; k3.rows = s; k3.cols = s;
      k3.rg_total = M; k3.rg_off = s; k3.rg_step = 2 * s; k3.rows_ragged = 1; k3.cols_ragged = 1; k3.nb = nbe; k3.Z = Z; k3.scale = si;
      ODX_PROPAGATE(split_f64_blocks(k3, stream));
      SplitBlocksArgs k4;                    // WT of every pair: s rows, columns ragged
      k4.X = WT; k4.ldx = s; k4.bsx = s * s; k4.zsx = zb.strideW;
      k4.P = pWT; k4.ldp = s; k4.bsp = s * s; k4.zsp = pk_z; k4.rows = s; k4.cols = s;
      k4.rg_total = M; k4.rg_off = s; k4.rg_step = 2 * s; k4.cols_ragged = 1; k4.nb = nbe; k4.Z = Z; k4.scale = sw;
      ODX_PROPAGATE(split_f64_blocks(k4, stream));
      H2F64Args a2;                          // X21 = -X22 WT'  -> Li, and transposed -> Lit
      a2.PA = pX22; a2.ldpa = s; a2.zsa = pk_z; a2.bsa = s * s; a2.sa = si;
      a2.PB = pWT; a2.ldpb = s; a2.zsb = pk_z; a2.bsb = s * s; a2.sb = sw;
      a2.C = Li + s * ld; a2.ldc = ld; a2.zsc = zb.strideO; a2.bsc = 2 * s * (ld + 1);
      a2.C2 = Lit + s; a2.ldc2 = ld; a2.zsc2 = zb.strideO; a2.bsc2 = 2 * s * (ld + 1);
      a2.m = s; a2.n = s; a2.k = s; a2.rg_total = M; a2.rg_off = s; a2.rg_step = 2 * s; a2.k_is_m = 1;
      a2.flags = ODX_GEMM_A_LOWER; a2.zcount = Z; a2.nb = nbe; a2.beta = 0.0;
      for (int z = 0; z < Z; ++z) a2.alpha[z] = mone[z];
      ODX_PROPAGATE(gemm_h2_f64_ex(a2, stream));
      continue;
    }
    GemmParams<double> g1;
    g1.A = L + s * ldl; g1.lda = ldl;                  // L21 of pair 0: rows s.., cols 0..
    g1.B = Lit; g1.ldb = ld;                           // X11' of pair 0
    g1.C = WT; g1.ldc = s;                             // WT_b (s x m2), transposed store
    g1.m = s; g1.n = s; g1.k = s; g1.alpha = 1.0; g1.beta = 0.0;
    g1.flags = ODX_GEMM_B_UPPER | ODX_GEMM_STORE_T;
    g1.batches = nb;
    g1.strideA = 2 * s * (ldl + 1); g1.strideB = 2 * s * (ld + 1); g1.strideC = s * s;
    g1.ragged_total = M; g1.ragged_off = s; g1.ragged_step = 2 * s;
    g1.zbatches = Z; g1.zstrideA = zb.strideA; g1.zstrideB = zb.strideO; g1.zstrideC = zb.strideW;
    ODX_PROPAGATE(launch_gemm_f64(g1, stream));
    GemmParams<double> g2;
    g2.A = Li + s * (ld + 1); g2.lda = ld;             // X22 of pair 0
    g2.B = WT; g2.ldb = s;                             // WT_b: rows j < s, k over m2
    g2.C = Li + s * ld; g2.ldc = ld;                   // X21 -> Li
    g2.C2 = Lit + s; g2.ldc2 = ld;                     // X21' -> Lit
    g2.m = s; g2.n = s; g2.k = s; g2.alpha = -1.0; g2.beta = 0.0;
    g2.flags = ODX_GEMM_A_LOWER;
    g2.batches = nb;
    g2.strideA = 2 * s * (ld + 1); g2.strideB = s * s; g2.strideC = 2 * s * (ld + 1); g2.strideC2 = 2 * s * (ld + 1);
    g2.ragged_total = M; g2.ragged_off = s; g2.ragged_step = 2 * s; g2.ragged_k_is_m = 1;
    g2.zbatches = Z; g2.zstrideA = zb.strideO; g2.zstrideB = zb.strideW; g2.zstrideC = zb.strideO; g2.zstrideC2 = zb.strideO;
    ODX_PROPAGATE(launch_gemm_f64(g2, stream));
  }
  return ODX_OK;
}

}  // namespace odx

using namespace odx;

extern "C" int odx_trmv_f64(const double* Tri, int64_t ld, int64_t M, int uplo, const double* x, double alpha,
                            double beta, const double* z, double* y, odx_stream_t stream) {
  if (M <= 0) return ODX_OK;
  ODX_REQUIRE(Tri && x && y && (beta == 0.0 || z), "odx_trmv_f64: null pointer");
  ODX_REQUIRE(ld % 2 == 0 && ld >= M && aligned16(Tri) && aligned16(x), "odx_trmv_f64: Tri/x must be 16-byte aligned, ld even, ld >= M");
  ODX_REQUIRE(x != y, "odx_trmv_f64: x and y must not alias");
  hipLaunchKernelGGL(trmv_f64_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, as_stream(stream), Tri, ld, M,
                     uplo, x, alpha, beta, z, y);
  ODX_CHECK_LAUNCH("odx_trmv_f64");
  return ODX_OK;
}

extern "C" int odx_convert_f32_f64(const float* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows,
                                   int64_t cols, odx_stream_t stream) {
  if (rows <= 0 || cols <= 0) return ODX_OK;
  ODX_REQUIRE(src && dst, "odx_convert_f32_f64: null pointer");
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL((convert_kernel<float, double>), dim3((unsigned)ceil_div(cols, 256), (unsigned)nr), dim3(256),
                       0, as_stream(stream), src + r0 * lds_, lds_, dst + r0 * ldd, ldd, nr, cols);
  }
  ODX_CHECK_LAUNCH("odx_convert_f32_f64");
  return ODX_OK;
}

extern "C" int odx_convert_f64_f32(const double* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows,
                                   int64_t cols, odx_stream_t stream) {
  if (rows <= 0 || cols <= 0) return ODX_OK;
  ODX_REQUIRE(src && dst, "odx_convert_f64_f32: null pointer");
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL((convert_kernel<double, float>), dim3((unsigned)ceil_div(cols, 256), (unsigned)nr), dim3(256),
                       0, as_stream(stream), src + r0 * lds_, lds_, dst + r0 * ldd, ldd, nr, cols);
  }
  ODX_CHECK_LAUNCH("odx_convert_f64_f32");
  return ODX_OK;
}

extern "C" int64_t odx_potrf_workspace_bytes(int64_t M) {
  if (M <= 0) return 0;
  return ceil_div(M, POTRF_NB) * POTRF_NB * POTRF_NB * (int64_t)sizeof(double);
}

extern "C" int odx_potrf_f64(double* A, int64_t lda, int64_t M, int32_t* info, void* workspace,
                             int64_t workspace_bytes, odx_stream_t stream) {
  if (M <= 0) return ODX_OK;
  ODX_REQUIRE(A && info && workspace, "odx_potrf_f64: null pointer");
  if (workspace_bytes < odx_potrf_workspace_bytes(M)) {
    set_error("odx_potrf_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  ODX_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), as_stream(stream)));
  const int rc = potrf_f64(A, lda, M, static_cast<double*>(workspace), info, as_stream(stream));
  return rc;
}

// workspace: Dinv | WT (M*M)
extern "C" int64_t odx_trtri_workspace_bytes(int64_t M) {
  if (M <= 0) return 0;
  return odx_potrf_workspace_bytes(M) + M * M * (int64_t)sizeof(double);
}

extern "C" int odx_trtri_f64(const double* L, int64_t ldl, int64_t M, double* Li, double* Lit, int64_t ld,
                             void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  if (M <= 0) return ODX_OK;
  ODX_REQUIRE(L && Li && Lit && workspace, "odx_trtri_f64: null pointer");
  ODX_REQUIRE(ld >= M && ldl >= M && aligned16(L) && aligned16(Li) && aligned16(Lit), "odx_trtri_f64: bad ld / alignment");
  if (workspace_bytes < odx_trtri_workspace_bytes(M)) {
    set_error("odx_trtri_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  double* Dinv = static_cast<double*>(workspace);
  double* WT = Dinv + ceil_div(M, POTRF_NB) * POTRF_NB * POTRF_NB;
  hipLaunchKernelGGL(trtri_diag_kernel, dim3((unsigned)ceil_div(M, POTRF_NB)), dim3(DB_NT), 0, s, L, ldl, M, Dinv,
                     (int64_t)0, (int64_t)0);
  ODX_CHECK_LAUNCH("trtri_diag");
  ODX_PROPAGATE(fill_f64(Li, ld, M, M, 0.0, s));
  ODX_PROPAGATE(fill_f64(Lit, ld, M, M, 0.0, s));
  return trtri_from_diag_f64(L, ldl, M, Dinv, Li, Lit, ld, WT, s);
}

// ---------------------------------------------------------------- FALKON preconditioner
// workspace: Zd (M x ldzd) | zsq (M, padded) | W0 | W1 | W2 | W3 (M x ld each) | DinvT | DinvA
//
// Two independent chains run side by side once L_T exists: the inverse of L_T (batched GEMMs that
// fill the chip) and  T T'/M + lam I -> L_A  (a latency-bound chain of small kernels).  The first
// goes to an internal side stream, forked and joined with events around it; everything stays
// asynchronous with respect to the host.
static int64_t precond_ld(int64_t M) { return round_up(M, 2); }
// row stride of the f64 copy of the centres: a multiple-of-128 feature count would put every row of a tile column on the same
// memory channels (K_MM's Gram at D = 1024: 48.9 TF with rows 8 KB apart, 56.1 TF with 64 doubles of padding)
static int64_t precond_ldz(int D) { const int64_t l = round_up(D, 2); return l % 128 == 0 ? l + 64 : l; }

extern "C" int64_t odx_falkon_precond_workspace_bytes(int64_t M, int D) {
  if (M <= 0 || D <= 0) return 0;
  const int64_t ld = precond_ld(M), ldzd = precond_ldz(D);
  int64_t dbl = M * ldzd + round_up(M, 2) + 4 * M * ld + 2 * ceil_div(M, POTRF_NB) * POTRF_NB * POTRF_NB + precond_split_doubles(M);
  return dbl * (int64_t)sizeof(double);
}

static int falkon_precond_f64_impl(const float* Z, int64_t ldz, int64_t M, int D, double sigma, double lam,
                                   double eps, double* LTi, double* LTit, double* LAi, double* LAit, int64_t ld,
                                   int32_t* info, void* workspace, int64_t workspace_bytes, odx_stream_t stream);

// (Left alive, the idle helper streams of a class-batched chain slow every later small launch of the process: the caller
// releases them when its fit / training step is queued, odx_release_helper_streams.)
extern "C" int odx_falkon_precond_f64(const float* Z, int64_t ldz, int64_t M, int D, double sigma, double lam,
                                      double eps, double* LTi, double* LTit, double* LAi, double* LAit, int64_t ld,
                                      int32_t* info, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  const int rc = falkon_precond_f64_impl(Z, ldz, M, D, sigma, lam, eps, LTi, LTit, LAi, LAit, ld, info, workspace, workspace_bytes, stream);
  return rc;
}

static int falkon_precond_f64_impl(const float* Z, int64_t ldz, int64_t M, int D, double sigma, double lam,
                                   double eps, double* LTi, double* LTit, double* LAi, double* LAit, int64_t ld,
                                   int32_t* info, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && D > 0 && sigma > 0, "odx_falkon_precond_f64: bad sizes");
  ODX_REQUIRE(Z && LTi && LTit && LAi && LAit && info && workspace, "odx_falkon_precond_f64: null pointer");
  ODX_REQUIRE(ld % 2 == 0 && ld >= M && aligned16(LTi) && aligned16(LTit) && aligned16(LAi) && aligned16(LAit),
              "odx_falkon_precond_f64: outputs must be 16-byte aligned with even ld >= M");
  ODX_REQUIRE(aligned16(workspace), "odx_falkon_precond_f64: workspace must be 16-byte aligned");
  if (workspace_bytes < odx_falkon_precond_workspace_bytes(M, D)) {
    set_error("odx_falkon_precond_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  SideStream* side = nullptr;
  ODX_PROPAGATE(side_stream(&side, 0, s, M));
  hipStream_t s2 = side->stream;
  const int64_t wld = precond_ld(M), ldzd = precond_ldz(D);
  const int64_t dsz = ceil_div(M, POTRF_NB) * POTRF_NB * POTRF_NB;
  double* Zd = static_cast<double*>(workspace);
  double* zsq = Zd + M * ldzd;
  double* W0 = zsq + round_up(M, 2);
  double* W1 = W0 + M * wld;
  double* W2 = W1 + M * wld;
  double* W3 = W2 + M * wld;
  double* DinvT = W3 + M * wld;
  double* DinvA = DinvT + dsz;
  const bool split = precond_split(M);
  const int64_t ldpt = h2_f64_packed_ld(M);
  uint32_t* Tk = reinterpret_cast<uint32_t*>(DinvA + dsz);             // packed T (later: merge-level packs), then the two packed panels
  uint32_t* Pk = Tk + precond_pack_units(M);

  ODX_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), s));
  ODX_CHECK_HIP(hipMemsetAsync(Zd, 0, (size_t)(M * ldzd) * sizeof(double), s));
  ODX_PROPAGATE(odx_convert_f32_f64(Z, ldz, Zd, ldzd, M, D, stream));
  // W0 = K_MM + eps*M*I (lower), then L_T in place
  ODX_CHECK_HIP(hipMemsetAsync(W0, 0, (size_t)(M * wld) * sizeof(double), s));
  ODX_PROPAGATE(gauss_kmm_f64(Zd, ldzd, M, D, sigma, eps * (double)M, W0, wld, zsq, s));
  ODX_PROPAGATE(potrf_f64(W0, wld, M, DinvT, info, s));
  // fork: inverses of L_T on the side stream (scratch W3)
  ODX_CHECK_HIP(hipEventRecord(side->fork, s));
  ODX_CHECK_HIP(hipStreamWaitEvent(s2, side->fork, 0));
  ODX_PROPAGATE(fill_f64(LTi, ld, M, M, 0.0, s2));
  ODX_PROPAGATE(fill_f64(LTit, ld, M, M, 0.0, s2));
  ODX_PROPAGATE(trtri_from_diag_f64(W0, wld, M, DinvT, LTi, LTit, ld, W3, s2));
  ODX_CHECK_HIP(hipEventRecord(side->join, s2));
  // main: W1 = L_T' = T (upper); W2 = T T' / M + lam I (lower tiles); L_A in place in W2
  ODX_CHECK_HIP(hipMemsetAsync(W1, 0, (size_t)(M * wld) * sizeof(double), s));
  ODX_PROPAGATE(transpose_f64(W0, wld, W1, wld, M, M, s));
  ODX_CHECK_HIP(hipMemsetAsync(W2, 0, (size_t)(M * wld) * sizeof(double), s));
  if (split) {
    // |T_ij| <= sqrt(max diagonal of T'T) = sqrt(1 + eps M);  |L_A ij| <= sqrt(max diagonal of T T' / M + lam) <= sqrt(1 + eps + lam)
    // (M < 65536 everywhere in this file: one bound for every call, so that the class-batched chain packs with the very
    // scales of the single-class one and stays bit-identical to it)
    const float st = split_scale_for(sqrt(1.0 + eps * 65536.0)), sa = split_scale_for(sqrt(1.0 + eps + lam));
    const double a1 = 1.0 / (double)M;
    ODX_PROPAGATE(split_f64(W1, wld, 0, M, M, st, Tk, ldpt, 0, 1, s));
    ODX_PROPAGATE(gemm_h2_f64(Tk, ldpt, 0, st, Tk, ldpt, 0, st, W2, wld, 0, M, M, M, &a1, 0.0,
                              ODX_GEMM_LOWER_ONLY | ODX_GEMM_A_UPPER | ODX_GEMM_B_UPPER, 1, s));
    ODX_PROPAGATE(add_diag_f64(W2, wld, M, lam, s));
    ODX_PROPAGATE(potrf_f64(W2, wld, M, DinvA, info, s, ZBatch(), Pk, M * POTRF_NBO, 0, sa));
  } else {
    GemmParams<double> g;
    g.A = W1; g.lda = wld; g.B = W1; g.ldb = wld; g.C = W2; g.ldc = wld;
    g.m = M; g.n = M; g.k = M; g.alpha = 1.0 / (double)M; g.beta = 0.0;
    g.flags = ODX_GEMM_LOWER_ONLY | ODX_GEMM_A_UPPER | ODX_GEMM_B_UPPER;
    ODX_PROPAGATE(launch_gemm_f64(g, s));
    ODX_PROPAGATE(add_diag_f64(W2, wld, M, lam, s));
    ODX_PROPAGATE(potrf_f64(W2, wld, M, DinvA, info, s));
  }
  // join, then the inverses of L_A (scratch W1: T is no longer needed)
  ODX_CHECK_HIP(hipStreamWaitEvent(s, side->join, 0));
  ODX_PROPAGATE(fill_f64(LAi, ld, M, M, 0.0, s));
  ODX_PROPAGATE(fill_f64(LAit, ld, M, M, 0.0, s));
  if (split && eps + lam > 0.0)
    return trtri_from_diag_f64(W2, wld, M, DinvA, LAi, LAit, ld, W1, s, ZBatch(), Tk, precond_pack_units(M), 0, sqrt(1.0 + eps + lam),
                               1.0 / sqrt(eps + lam));
  ODX_PROPAGATE(trtri_from_diag_f64(W2, wld, M, DinvA, LAi, LAit, ld, W1, s));
  return ODX_OK;
}

// ---------------------------------------------------------------- FALKON preconditioners of a class batch
// The same computation as odx_falkon_precond_f64 for B independent classes at once: every kernel of the two blocked
// Choleskys, the T T' product and the two triangular inverses takes the class as one more grid dimension, so the
// ~1500-launch dependent chain of ONE preconditioner advances all B of them (the chain is latency-bound: per 128-column
// block a one-workgroup diagonal factorisation, a panel solve and a rank-128 update — B classes fill B times as much of
// the chip per launch).  Classes may have different numbers of centres M_b <= Mmax: matrix b is K_MM_b bordered with an
// identity block up to Mmax, whose Cholesky factor / inverse is the bordered factor / inverse — the leading M_b x M_b
// blocks of the outputs are what the single-class call produces (same block boundaries, the padding only adds exact
// zeros to the sums), and the CG reads only those.
// workspace: per class  Zd (Mmax x ldzd) | zsq (Mmax, padded)   then   W0[B] | W1[B] | W2[B] | W3[B] | DinvT[B] | DinvA[B]
extern "C" int64_t odx_falkon_precond_batched_workspace_bytes(int64_t Mmax, int D, int B) {
  if (Mmax <= 0 || D <= 0 || B <= 0) return 0;
  const int64_t ld = precond_ld(Mmax), ldzd = precond_ldz(D);
  const int64_t per = Mmax * ldzd + round_up(Mmax, 2) + 4 * Mmax * ld + 2 * ceil_div(Mmax, POTRF_NB) * POTRF_NB * POTRF_NB +
                      precond_split_doubles(Mmax);
  return per * B * (int64_t)sizeof(double);
}

static int falkon_precond_batched_f64_impl(const float* const* Z, const int64_t* ldz, const int64_t* M, int B,
                                           int64_t Mmax, int D, double sigma, double lam, double eps, double* out,
                                           int64_t ld, int64_t out_stride, int32_t* info, void* workspace,
                                           int64_t workspace_bytes, odx_stream_t stream);

extern "C" int odx_falkon_precond_batched_f64(const float* const* Z, const int64_t* ldz, const int64_t* M, int B,
                                              int64_t Mmax, int D, double sigma, double lam, double eps, double* out,
                                              int64_t ld, int64_t out_stride, int32_t* info, void* workspace,
                                              int64_t workspace_bytes, odx_stream_t stream) {
  const int rc = falkon_precond_batched_f64_impl(Z, ldz, M, B, Mmax, D, sigma, lam, eps, out, ld, out_stride, info, workspace, workspace_bytes, stream);
  return rc;
}

static int falkon_precond_batched_f64_impl(const float* const* Z, const int64_t* ldz, const int64_t* M, int B,
                                           int64_t Mmax, int D, double sigma, double lam, double eps, double* out,
                                           int64_t ld, int64_t out_stride, int32_t* info, void* workspace,
                                           int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(B >= 1 && B <= ODX_MAX_ZBATCH, "odx_falkon_precond_batched_f64: 1 <= B <= %d classes per call", ODX_MAX_ZBATCH);
  ODX_REQUIRE(Z && ldz && M && out && info && workspace && Mmax > 0 && D > 0 && sigma > 0,
              "odx_falkon_precond_batched_f64: null pointer or bad size");
  ODX_REQUIRE(ld % 2 == 0 && ld >= Mmax && aligned16(out) && out_stride % 2 == 0 && out_stride >= 4 * Mmax * ld,
              "odx_falkon_precond_batched_f64: out must be 16-byte aligned, ld even >= Mmax, out_stride >= 4 Mmax ld");
  ODX_REQUIRE(aligned16(workspace), "odx_falkon_precond_batched_f64: workspace must be 16-byte aligned");
  for (int b = 0; b < B; ++b)
    ODX_REQUIRE(Z[b] && M[b] > 0 && M[b] <= Mmax, "odx_falkon_precond_batched_f64: class %d: need 0 < M <= Mmax", b);
  if (workspace_bytes < odx_falkon_precond_batched_workspace_bytes(Mmax, D, B)) {
    set_error("odx_falkon_precond_batched_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  SideStream* side = nullptr;
  ODX_PROPAGATE(side_stream(&side, 0, s, Mmax));
  hipStream_t s2 = side->stream;
  const int64_t wld = precond_ld(Mmax), ldzd = precond_ldz(D);
  const int64_t dsz = ceil_div(Mmax, POTRF_NB) * POTRF_NB * POTRF_NB;
  const int64_t zper = Mmax * ldzd + round_up(Mmax, 2), wsz = Mmax * wld;
  double* Zd0 = static_cast<double*>(workspace);
  double* W0 = Zd0 + (int64_t)B * zper;
  double* W1 = W0 + (int64_t)B * wsz;
  double* W2 = W1 + (int64_t)B * wsz;
  double* W3 = W2 + (int64_t)B * wsz;
  double* DinvT = W3 + (int64_t)B * wsz;
  double* DinvA = DinvT + (int64_t)B * dsz;
  double* LTi = out, *LTit = out + Mmax * ld, *LAi = out + 2 * Mmax * ld, *LAit = out + 3 * Mmax * ld;
  const bool split = precond_split(Mmax);
  const int64_t ldpt = h2_f64_packed_ld(Mmax);
  uint32_t* Tk = reinterpret_cast<uint32_t*>(DinvA + (int64_t)B * dsz);   // packed T of every class (later: merge-level packs), then the packed panels
  uint32_t* Pk = Tk + (int64_t)B * precond_pack_units(Mmax);

  ODX_CHECK_HIP(hipMemsetAsync(info, 0, (size_t)B * sizeof(int32_t), s));
  ODX_CHECK_HIP(hipMemsetAsync(Zd0, 0, (size_t)((int64_t)B * zper) * sizeof(double), s));
  ODX_CHECK_HIP(hipMemsetAsync(W0, 0, (size_t)(3 * (int64_t)B * wsz) * sizeof(double), s));      // W0, W1, W2
  // every output factor starts as zero (the inverses are written triangle by triangle): one fill for the whole block
  if (out_stride == 4 * Mmax * ld) {
    ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)((int64_t)B * out_stride) * sizeof(double), s));
  } else {
    for (int b = 0; b < B; ++b)
      ODX_CHECK_HIP(hipMemsetAsync(out + (int64_t)b * out_stride, 0, (size_t)(4 * Mmax * ld) * sizeof(double), s));
  }
  // W0_b = [K_MM_b + eps M_b I, 0; 0, I]  (lower): the K_MM of all classes by one launch
  VecBatch kb;
  kb.B = B;
  for (int b = 0; b < B; ++b) {
    ODX_PROPAGATE(odx_convert_f32_f64(Z[b], ldz[b], Zd0 + (int64_t)b * zper, ldzd, M[b], D, stream));
    kb.M[b] = (int)M[b];
    kb.scale[b] = eps * (double)M[b];
  }
  ODX_PROPAGATE(gauss_kmm_f64_batched(Zd0, ldzd, zper, Mmax * ldzd, kb, D, sigma, W0, wld, wsz, s));
  for (int b = 0; b < B; ++b)
    if (M[b] < Mmax) ODX_PROPAGATE(add_diag_f64(W0 + (int64_t)b * wsz + M[b] * (wld + 1), wld, Mmax - M[b], 1.0, s));
  ZBatch zt;
  zt.count = B; zt.strideA = wsz; zt.strideD = dsz; zt.strideO = out_stride; zt.strideW = wsz;
  ODX_PROPAGATE(potrf_f64(W0, wld, Mmax, DinvT, info, s, zt));
  // fork: inverses of all L_T on the side stream (scratch W3)
  ODX_CHECK_HIP(hipEventRecord(side->fork, s));
  ODX_CHECK_HIP(hipStreamWaitEvent(s2, side->fork, 0));
  ODX_PROPAGATE(trtri_from_diag_f64(W0, wld, Mmax, DinvT, LTi, LTit, ld, W3, s2, zt));
  ODX_CHECK_HIP(hipEventRecord(side->join, s2));
  // main: W1 = L_T' = T; W2 = T T' / M_b + lam I; L_A in place in W2
  ODX_PROPAGATE(transpose_f64(W0, wld, W1, wld, Mmax, Mmax, s, B, wsz, wsz));
  if (split) {
    // (bounds as in the single-class call; the identity border of a class with fewer centres is inside them)
    const float st = split_scale_for(sqrt(1.0 + eps * 65536.0)), sa = split_scale_for(sqrt(1.0 + eps + lam));
    double a1[ODX_MAX_ZBATCH];
    for (int b = 0; b < B; ++b) a1[b] = 1.0 / (double)M[b];
    ODX_PROPAGATE(split_f64(W1, wld, wsz, Mmax, Mmax, st, Tk, ldpt, precond_pack_units(Mmax), B, s));
    ODX_PROPAGATE(gemm_h2_f64(Tk, ldpt, precond_pack_units(Mmax), st, Tk, ldpt, precond_pack_units(Mmax), st, W2, wld, wsz, Mmax, Mmax, Mmax, a1, 0.0,
                              ODX_GEMM_LOWER_ONLY | ODX_GEMM_A_UPPER | ODX_GEMM_B_UPPER, B, s));
    ODX_PROPAGATE(add_diag_f64(W2, wld, Mmax, lam, s, B, wsz));
    ODX_PROPAGATE(potrf_f64(W2, wld, Mmax, DinvA, info, s, zt, Pk, (int64_t)B * Mmax * POTRF_NBO, Mmax * POTRF_NBO, sa));
  } else {
    GemmParams<double> g;
    g.A = W1; g.lda = wld; g.B = W1; g.ldb = wld; g.C = W2; g.ldc = wld;
    g.m = Mmax; g.n = Mmax; g.k = Mmax; g.beta = 0.0;
    g.flags = ODX_GEMM_LOWER_ONLY | ODX_GEMM_A_UPPER | ODX_GEMM_B_UPPER;
    g.zbatches = B; g.zstrideA = wsz; g.zstrideB = wsz; g.zstrideC = wsz;
    g.zalpha_on = 1;
    for (int b = 0; b < B; ++b) g.zalpha[b] = 1.0 / (double)M[b];
    ODX_PROPAGATE(launch_gemm_f64(g, s));
    ODX_PROPAGATE(add_diag_f64(W2, wld, Mmax, lam, s, B, wsz));
    ODX_PROPAGATE(potrf_f64(W2, wld, Mmax, DinvA, info, s, zt));
  }
  // join, then the inverses of L_A (scratch W1: T is no longer needed)
  ODX_CHECK_HIP(hipStreamWaitEvent(s, side->join, 0));
  // (every eigenvalue of T T' / M_b + lam I is at least eps + lam: |(L_A^-1)_ij| <= 1 / sqrt(eps + lam))
  if (split && eps + lam > 0.0)
    return trtri_from_diag_f64(W2, wld, Mmax, DinvA, LAi, LAit, ld, W1, s, zt, Tk, precond_pack_units(Mmax), precond_pack_units(Mmax),
                               sqrt(1.0 + eps + lam), 1.0 / sqrt(eps + lam));
  return trtri_from_diag_f64(W2, wld, Mmax, DinvA, LAi, LAit, ld, W1, s, zt);
}

extern "C" int odx_set_side_stream_cu_mask(const uint32_t* mask, int words) {
  ODX_REQUIRE(words >= 0 && words <= 64 && (words == 0 || mask != nullptr), "odx_set_side_stream_cu_mask: bad argument");
  g_side_mask.assign(mask, mask + words);
  return ODX_OK;
}

extern "C" int odx_set_helper_streams(odx_stream_t s0, odx_stream_t s1) {
  g_given_helper[0] = reinterpret_cast<hipStream_t>(s0);
  g_given_helper[1] = reinterpret_cast<hipStream_t>(s1);
  return ODX_OK;
}

extern "C" int odx_release_helper_streams(void) {
  return odx::release_side_streams(false);
}
