// Tile GEMM core for gfx950:  acc(i, j) = sum_k A[i, k] * B[j, k]   ("NT": both operands
// row-major with k contiguous).  One 256-thread workgroup (4 waves as 2 x 2) owns a
// 128 x 128 output tile; each wave a 64 x 64 sub-tile built from
//   f32: 2 x 2 tiles of v_mfma_f32_32x32x2_f32  (exact f32 fmaf chain, 64 cyc/SIMD)
//   f64: 4 x 4 tiles of v_mfma_f64_16x16x4_f64
// A k-tile is 128 B of every row (32 f32 / 16 f64).  Global -> registers (16 B per lane,
// 8 lanes per 128-B row segment => full-line coalescing) -> LDS rows padded to 144 B:
//   f32 fragments: one ds_read_b128 per lane and 8 k's; row stride 36 dwords makes the
//       16-lane groups of ds_read_b128 hit 16 distinct 16-B slots (36 r mod 64 covers all
//       multiples of 4) => conflict-free;
//   f64 fragments: one ds_read_b64 per lane and 4 k's; banks 36 r + 2 kq are distinct over a
//       32-lane half => conflict-free.
// The k order inside a k-tile is permuted identically for A and B (it is a sum over k).
// Register prefetch of the next k-tile overlaps the MFMA work of the current one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace odx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_ROW = 144;                                   // bytes, 128 data + 16 pad
constexpr int GEMM_LDS_BYTES = (GEMM_BM + GEMM_BN) * GEMM_LDS_ROW;  // 36,864 B

template <typename T>
struct GemmTraits;

template <>
struct GemmTraits<float> {
  static constexpr int BK = 32;   // elements per k-tile
  static constexpr int EPV = 4;   // elements per 16 B
  static constexpr int TM = 2, TN = 2, MT = 32, NREG = 16;  // TN for a 128-wide tile
  typedef f32x16 Acc;
};
template <>
struct GemmTraits<double> {
  static constexpr int BK = 16;
  static constexpr int EPV = 2;
  static constexpr int TM = 4, TN = 4, MT = 16, NREG = 4;
  typedef f64x4 Acc;
};
// MFMA tiles across a wave's share (BN / 2 columns) of a BN-wide block tile
template <typename T, int BN>
struct GemmTileN {
  static constexpr int TN = BN / 2 / GemmTraits<T>::MT;
  static_assert(TN >= 1, "block tile too narrow for this MFMA shape");
};

template <typename T, int BN = GEMM_BN>
struct GemmStage {
  u32x4 a[4];
  u32x4 b[BN / 32];   // BN rows x 8 segments / 256 threads
};

// One operand tile: 128 rows x 128 B.  Thread t fetches 16-B segments idx = t + 256 p:
// row = idx >> 3, seg = idx & 7.  Rows >= nrows and k >= K read as zero.
// CLEAN = the whole k-tile lies inside [0, K): no per-element masking, so the four loads issue
// back to back and are only waited for at the LDS store (a mask applied right after each load
// would serialise them behind s_waitcnt vmcnt(0)).
template <typename T, bool CLEAN, int NP>
__device__ __forceinline__ void gemm_load_operand(u32x4 (&r)[NP], const T* __restrict__ base, int64_t ld,
                                                  int64_t row0, int64_t nrows, int64_t k0, int64_t K) {
  constexpr int EPV = GemmTraits<T>::EPV;
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + GEMM_THREADS * p;
    const int row = idx >> 3;
    const int seg = idx & 7;
    const int64_t gr = row0 + row;
    const int64_t kk = k0 + seg * EPV;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (CLEAN) {
      if (gr < nrows) v = *reinterpret_cast<const u32x4*>(base + gr * ld + kk);
    } else if (gr < nrows && kk < K) {
      v = *reinterpret_cast<const u32x4*>(base + gr * ld + kk);
      if (kk + EPV > K) {  // ragged k tail inside this 16-B segment
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int q = 0; q < EPV; ++q)
          if (kk + q >= K) e[q] = T(0);
      }
    }
    r[p] = v;
  }
}

template <typename T, int BN>
__device__ __forceinline__ void gemm_load_stage(GemmStage<T, BN>& st, const T* __restrict__ A, int64_t lda, int64_t m,
                                                const T* __restrict__ B, int64_t ldb, int64_t n, int64_t i0,
                                                int64_t j0, int64_t kt, int64_t ke) {
  if (kt + GemmTraits<T>::BK <= ke) {
    gemm_load_operand<T, true, 4>(st.a, A, lda, i0, m, kt, ke);
    gemm_load_operand<T, true, BN / 32>(st.b, B, ldb, j0, n, kt, ke);
  } else {
    gemm_load_operand<T, false, 4>(st.a, A, lda, i0, m, kt, ke);
    gemm_load_operand<T, false, BN / 32>(st.b, B, ldb, j0, n, kt, ke);
  }
}

template <int NP>
__device__ __forceinline__ void gemm_store_operand(const u32x4 (&r)[NP], char* lds) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + GEMM_THREADS * p;
    *reinterpret_cast<u32x4*>(lds + (idx >> 3) * GEMM_LDS_ROW + (idx & 7) * 16) = r[p];
  }
}

// MFMA work of one k-tile held in LDS.  wr / wc = wave row / column (0..1).
template <int TN>
__device__ __forceinline__ void gemm_compute_ktile(f32x16 (&acc)[2][TN], const char* ldsA, const char* ldsB,
                                                   int wr, int wc, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    f32x4 a[2], b[TN];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      a[t] = *reinterpret_cast<const f32x4*>(ldsA + (wr * 64 + t * 32 + r) * GEMM_LDS_ROW + (ks * 8 + h * 4) * 4);
#pragma unroll
    for (int t = 0; t < TN; ++t)
      b[t] = *reinterpret_cast<const f32x4*>(ldsB + (wc * 32 * TN + t * 32 + r) * GEMM_LDS_ROW + (ks * 8 + h * 4) * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
  }
}

template <int TN>
__device__ __forceinline__ void gemm_compute_ktile(f64x4 (&acc)[4][TN], const char* ldsA, const char* ldsB,
                                                   int wr, int wc, int lane) {
  const int r = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    double a[4], b[TN];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      a[t] = *reinterpret_cast<const double*>(ldsA + (wr * 64 + t * 16 + r) * GEMM_LDS_ROW + (ks * 4 + kq) * 8);
#pragma unroll
    for (int t = 0; t < TN; ++t)
      b[t] = *reinterpret_cast<const double*>(ldsB + (wc * 16 * TN + t * 16 + r) * GEMM_LDS_ROW + (ks * 4 + kq) * 8);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
}

// Accumulator element -> (row, col) inside the wave's 64 x 64 sub-tile.
// f32 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
// f64 16x16: col = lane & 15, row = (lane >> 4) + 4 reg
template <typename T>
__device__ __forceinline__ int gemm_acc_row(int tm, int reg, int lane);
template <>
__device__ __forceinline__ int gemm_acc_row<float>(int tm, int reg, int lane) {
  return tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
template <>
__device__ __forceinline__ int gemm_acc_row<double>(int tm, int reg, int lane) {
  return tm * 16 + (lane >> 4) + 4 * reg;
}
template <typename T>
__device__ __forceinline__ int gemm_acc_col(int tn, int lane);
template <>
__device__ __forceinline__ int gemm_acc_col<float>(int tn, int lane) { return tn * 32 + (lane & 31); }
template <>
__device__ __forceinline__ int gemm_acc_col<double>(int tn, int lane) { return tn * 16 + (lane & 15); }

// acc += A[i0.., kb..ke) * B[j0.., kb..ke)'.  kb must be a multiple of BK.  lds: one k-tile,
// (GEMM_BM + BN) * GEMM_LDS_ROW bytes.  The next k-tile's global loads are issued before the
// MFMA work of the current one and land in registers; two barriers per k-tile order the LDS
// refill.  (Measured on MI355X: a second LDS buffer with one barrier per k-tile, and a two-deep
// register prefetch, both run at the same rate as this form — the 128 x 128 f32 tile sits at
// ~76 % MFMA-busy whatever the staging scheme — so the smallest LDS footprint is kept and spent
// on a third resident workgroup per CU instead.)
template <typename T, int TN>
__device__ __forceinline__ void gemm_mainloop(typename GemmTraits<T>::Acc (&acc)[GemmTraits<T>::TM][TN],
                                              const T* __restrict__ A, int64_t lda, int64_t m,
                                              const T* __restrict__ B, int64_t ldb, int64_t n, int64_t i0,
                                              int64_t j0, int64_t kb, int64_t ke, char* lds) {
  constexpr int BN = 2 * TN * GemmTraits<T>::MT;
  constexpr int BK = GemmTraits<T>::BK;
  char* ldsA = lds;
  char* ldsB = lds + GEMM_BM * GEMM_LDS_ROW;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (kb >= ke) return;
  GemmStage<T, BN> st;
  gemm_load_stage<T, BN>(st, A, lda, m, B, ldb, n, i0, j0, kb, ke);
  for (int64_t kt = kb; kt < ke; kt += BK) {
    __syncthreads();  // everyone finished reading the previous k-tile
    gemm_store_operand(st.a, ldsA);
    gemm_store_operand(st.b, ldsB);
    __syncthreads();
    if (kt + BK < ke) gemm_load_stage<T, BN>(st, A, lda, m, B, ldb, n, i0, j0, kt + BK, ke);
    gemm_compute_ktile(acc, ldsA, ldsB, wr, wc, lane);
  }
}

template <typename T, int TN>
__device__ __forceinline__ void gemm_zero_acc(typename GemmTraits<T>::Acc (&acc)[GemmTraits<T>::TM][TN]) {
#pragma unroll
  for (int tm = 0; tm < GemmTraits<T>::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < GemmTraits<T>::NREG; ++r) acc[tm][tn][r] = T(0);
}

// XCD-aware, bijective remap of a 1-D block id: blocks b and b+8 share an XCD (and its L2)
// under the observed round-robin dispatch, so give every XCD a contiguous run of tiles.
// Placement is a speed matter only.
__device__ __forceinline__ int64_t xcd_remap(int64_t id, int64_t nwg) {
  const int64_t q = nwg >> 3, r = nwg & 7;
  const int64_t xcd = id & 7, local = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

}  // namespace odx
