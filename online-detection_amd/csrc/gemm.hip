// Generic NT GEMM on the tile core: C = alpha * A B' + beta * C with triangular-operand
// k-range clipping, lower-only output, transposed / dual store and ragged batching.
// Used by the f64 preconditioner (potrf / trtri / T T'), the RLS Gram, and exported for tests.
#include "gemm_core.h"
#include "odx_internal.h"

namespace odx {

// One 32-row slab of a tile from LDS to its destination in whole 16-byte segments.  TR: the slab holds the tile transposed
// (slab[c][r]) and goes to dst[col][row].  The orientation is a template parameter so that the segment index splits into
// (slow, fast) by shifts — with a run-time orientation it was an integer division per segment — and the destination
// address advances by a constant number of rows per trip instead of a 64-bit row * ld product per segment: at k = 128 (the
// panel updates of the blocked Cholesky) the epilogue is as long as the main loop.
template <typename T, int BN, bool TR>
__device__ __forceinline__ void gemm_slab_out(const T* __restrict__ slab, T* __restrict__ dst, int64_t ldd, int64_t row0,
                                              int64_t col0, int64_t m, int64_t n, T alpha, T beta, bool rmw) {
  using Tr = GemmTraits<T>;
  constexpr int EPV = Tr::EPV;
  constexpr int SLAB = 32;
  constexpr int NA = TR ? BN : SLAB, NB = TR ? SLAB : BN;           // out(a, b): a = slow index, b = fast (contiguous) index
  constexpr int LDSLAB = (TR ? SLAB : BN) + EPV;
  constexpr int SEGS = NB / EPV;                                     // a power of two
  constexpr int STEP = GEMM_THREADS / SEGS;                          // slow-index rows per trip
  static_assert((SEGS & (SEGS - 1)) == 0 && GEMM_THREADS % SEGS == 0 && NA % STEP == 0, "gemm_slab_out: segment geometry");
  typedef T VecT __attribute__((ext_vector_type(EPV)));
  const int64_t a0 = TR ? col0 : row0, b0 = TR ? row0 : col0;
  const int64_t alim = TR ? n : m, blim = TR ? m : n;
  const int tid = threadIdx.x;
  const int b = (tid & (SEGS - 1)) * EPV;
  const int64_t gb = b0 + b;
  if (gb >= blim) return;
  const bool whole = gb + EPV <= blim;
  int a = tid / SEGS;
  T* g = dst + (a0 + a) * ldd + gb;
  const T* sp = slab + a * LDSLAB + b;
#pragma unroll
  for (int it = 0; it < NA / STEP; ++it, a += STEP, g += (int64_t)STEP * ldd, sp += STEP * LDSLAB) {
    if (a0 + a >= alim) break;
    const VecT v = *reinterpret_cast<const VecT*>(sp);
    if (whole) {
      VecT o;
      if (rmw) {
        const VecT c = *reinterpret_cast<const VecT*>(g);
#pragma unroll
        for (int q = 0; q < EPV; ++q) o[q] = alpha * v[q] + beta * c[q];
      } else {
#pragma unroll
        for (int q = 0; q < EPV; ++q) o[q] = alpha * v[q];
      }
      *reinterpret_cast<VecT*>(g) = o;
    } else {
      for (int q = 0; q < EPV && gb + q < blim; ++q) {
        T o = alpha * v[q];
        if (rmw) o += beta * g[q];
        g[q] = o;
      }
    }
  }
}

template <typename T, int BN>
__global__ __launch_bounds__(GEMM_THREADS, (sizeof(T) == 8 && BN == 64) ? 3 : 1) void gemm_nt_kernel(GemmParams<T> p) {
  using Tr = GemmTraits<T>;
  constexpr int TN = GemmTileN<T, BN>::TN;
  constexpr int LDS_BYTES = (GEMM_BM + BN) * GEMM_LDS_ROW;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const int b = blockIdx.y;
  int64_t m = p.m, k = p.zk_on ? p.zklen[blockIdx.z] : p.k;
  if (p.ragged_total > 0) {
    int64_t mb = p.ragged_total - p.ragged_off - (int64_t)b * p.ragged_step;
    if (mb > p.m) mb = p.m;
    if (mb <= 0) return;
    m = mb;
    if (p.ragged_k_is_m) k = mb;
  }
  const int64_t n = p.n;
  const int64_t tiles_n = (n + BN - 1) / BN;
  // Tile order: workgroups b and b + 8 share an XCD (round-robin dispatch), so XCD x walks the
  // tile rows x, x + 8, ... left to right: neighbours in time share the A row panel in that XCD's
  // L2, and triangular work (lower-only output, k-ranges clipped by a triangular operand) is
  // spread evenly over the XCDs instead of piling the long rows onto the last one.  gridDim.x is a multiple of 8, so
  // the physical XCD of a workgroup is blockIdx.x & 7 whatever its batch; the batch index rotates which tile rows that
  // XCD gets: a batch of many small products (m <= 128: ONE tile row, i.e. one XCD per product without the rotation —
  // the merge levels of the triangular inverse, the last trailing updates of a Cholesky) covers all eight XCDs, and
  // the uneven row weights of triangular work average out over the batch.
  const int64_t xcd = (blockIdx.x + blockIdx.y + blockIdx.z) & 7, local = blockIdx.x >> 3;
  // lower-only output of a plain product: row r carries r + 1 tiles, so XCD x with rows x, x + 8, ... ends up 9 % over
  // the mean at 79 tile rows; alternate groups of eight rows run backwards (x, 15 - x, 16 + x, ...), which evens the sums
  const int64_t grp = local / tiles_n;
  const bool serp = (p.flags & ODX_GEMM_LOWER_ONLY) && !(p.flags & (ODX_GEMM_A_UPPER | ODX_GEMM_B_UPPER | ODX_GEMM_A_LOWER | ODX_GEMM_B_LOWER));
  const int64_t bi = 8 * grp + ((serp && (grp & 1)) ? 7 - xcd : xcd), bj = local % tiles_n;
  const int64_t i0 = bi * GEMM_BM, j0 = bj * BN;
  if (i0 >= m) return;
  if ((p.flags & ODX_GEMM_LOWER_ONLY) && j0 > i0 + GEMM_BM - 1) return;

  int64_t kb = 0, ke = k;
  if (p.flags & ODX_GEMM_A_UPPER) kb = max(kb, i0);
  if (p.flags & ODX_GEMM_B_UPPER) kb = max(kb, j0);
  if (p.flags & ODX_GEMM_A_LOWER) ke = min(ke, i0 + GEMM_BM);
  if (p.flags & ODX_GEMM_B_LOWER) ke = min(ke, j0 + BN);
  kb = (kb / Tr::BK) * Tr::BK;

  const int z = blockIdx.z;
  const T* A = p.A + (int64_t)b * p.strideA + (int64_t)z * p.zstrideA;
  const T* B = p.B + (int64_t)b * p.strideB + (int64_t)z * p.zstrideB;
  if (p.zk_on) {                          // (read before the k-range clipping below uses k)
    A += p.zkoff[z];
    B += p.zkoff[z];
  }
  T* C = p.C + (int64_t)b * p.strideC + (int64_t)z * p.zstrideC;
  T* C2 = p.C2 ? p.C2 + (int64_t)b * p.strideC2 + (int64_t)z * p.zstrideC2 : nullptr;
  const T alpha = p.zalpha_on ? p.zalpha[z] : p.alpha;

  typename Tr::Acc acc[Tr::TM][TN];
  gemm_zero_acc<T>(acc);
  gemm_mainloop<T>(acc, A, p.lda, m, B, p.ldb, n, i0, j0, kb, ke, lds);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const bool st = (p.flags & ODX_GEMM_STORE_T) != 0;
  const bool has_beta = p.beta != T(0);

  if (p.vec_epilogue) {
    // Epilogue staged through LDS in four 32-row slabs: accumulators -> LDS (as stored, or
    // transposed), then every thread moves whole 16-byte segments, so C is read and written in
    // full 1-KB rows (the MFMA accumulator layout alone gives 8-byte pieces of 4 rows per
    // instruction) and the transposed / dual stores are as coalesced as the plain one.
    constexpr int EPV = Tr::EPV;
    constexpr int SLAB = 32;
    constexpr int LDN = BN + EPV;         // normal:      slab[r][c], r < 32, c < BN
    constexpr int LDT = SLAB + EPV;       // transposed:  slab[c][r], c < BN, r < 32
    static_assert(sizeof(T) * SLAB * LDN <= LDS_BYTES && sizeof(T) * BN * LDT <= LDS_BYTES, "slab");
    T* slab = reinterpret_cast<T*>(lds);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r0 = pass * SLAB;                       // tile rows [r0, r0 + 32)
#pragma unroll
      for (int variant = 0; variant < 2; ++variant) {   // 0: the C store, 1: the transposed copy C2
        const bool tr = variant == 0 ? st : true;
        if (variant == 1 && C2 == nullptr) continue;
        __syncthreads();
        if (wr == (pass >> 1)) {
#pragma unroll
          for (int tmh = 0; tmh < Tr::TM / 2; ++tmh) {
            const int tm = (pass & 1) * (Tr::TM / 2) + tmh;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int r = 0; r < Tr::NREG; ++r) {
                const int rr = wr * 64 + gemm_acc_row<T>(tm, r, lane) - r0;
                const int cc = wc * (BN / 2) + gemm_acc_col<T>(tn, lane);
                slab[tr ? cc * LDT + rr : rr * LDN + cc] = acc[tm][tn][r];
              }
          }
        }
        __syncthreads();
        T* dst = variant == 0 ? C : C2;
        const int64_t ldd = variant == 0 ? p.ldc : p.ldc2;
        const bool rmw = variant == 0 && has_beta;
        if (tr) gemm_slab_out<T, BN, true>(slab, dst, ldd, i0 + r0, j0, m, n, alpha, p.beta, rmw);
        else gemm_slab_out<T, BN, false>(slab, dst, ldd, i0 + r0, j0, m, n, alpha, p.beta, rmw);
      }
    }
    return;
  }

  // scalar epilogue (unaligned C): batches of one tile-row, all reads of C before the writes
#pragma unroll
  for (int tm = 0; tm < Tr::TM; ++tm) {
    T cv[TN][Tr::NREG];
    if (has_beta) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < Tr::NREG; ++r) {
          const int64_t row = i0 + wr * 64 + gemm_acc_row<T>(tm, r, lane);
          const int64_t col = j0 + wc * (BN / 2) + gemm_acc_col<T>(tn, lane);
          cv[tn][r] = (row < m && col < n) ? C[st ? col * p.ldc + row : row * p.ldc + col] : T(0);
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < Tr::NREG; ++r) {
        const int64_t row = i0 + wr * 64 + gemm_acc_row<T>(tm, r, lane);
        const int64_t col = j0 + wc * (BN / 2) + gemm_acc_col<T>(tn, lane);
        if (row < m && col < n) {
          T val = alpha * acc[tm][tn][r];
          if (has_beta) val += p.beta * cv[tn][r];
          C[st ? col * p.ldc + row : row * p.ldc + col] = val;
          if (C2) C2[col * p.ldc2 + row] = val;
        }
      }
  }
}

template <typename T>
static int launch_gemm(const GemmParams<T>& p, hipStream_t stream, const char* name) {
  constexpr int EPV = GemmTraits<T>::EPV;
  if (p.m <= 0 || p.n <= 0 || p.batches <= 0) return ODX_OK;
  ODX_REQUIRE(p.A && p.B && p.C, "%s: null operand", name);
  ODX_REQUIRE(p.lda % EPV == 0 && p.ldb % EPV == 0, "%s: lda/ldb must be multiples of %d elements", name, EPV);
  ODX_REQUIRE(aligned16(p.A) && aligned16(p.B), "%s: A/B must be 16-byte aligned", name);
  ODX_REQUIRE(p.strideA % EPV == 0 && p.strideB % EPV == 0, "%s: batch strides must keep 16-byte alignment", name);
  // 128 x 64 tiles (two or three workgroups per CU hide each other's barriers and load latency)
  // unless the product is computed in place over its own A operand: then one workgroup must own
  // every column of a row panel (n <= 128), which needs the 128-wide tile.
  const bool inplace = static_cast<const void*>(p.C) == static_cast<const void*>(p.A);
  if (inplace) ODX_REQUIRE(p.n <= GEMM_BN, "%s: in-place product needs n <= %d", name, GEMM_BN);
  const bool narrow = sizeof(T) == 8 && !inplace;   // f64: 128 x 64 tiles; f32 keeps 128 x 128
  const int bn = narrow ? 64 : GEMM_BN;
  const int64_t tiles = 8 * ceil_div(ceil_div(p.m, GEMM_BM), 8) * ceil_div(p.n, bn);
  ODX_REQUIRE(tiles < (1ll << 31) && p.batches < 65536, "%s: grid too large", name);
  ODX_REQUIRE(p.zbatches >= 1 && p.zbatches <= ODX_MAX_ZBATCH, "%s: 1 <= zbatches <= %d", name, ODX_MAX_ZBATCH);
  ODX_REQUIRE(p.zstrideA % EPV == 0 && p.zstrideB % EPV == 0, "%s: class strides must keep 16-byte alignment", name);
  if (p.zk_on)
    for (int z = 0; z < p.zbatches; ++z)
      ODX_REQUIRE(p.zkoff[z] % EPV == 0 && p.zklen[z] >= 0, "%s: per-class k windows must start 16-byte aligned", name);
  dim3 grid((unsigned)tiles, (unsigned)p.batches, (unsigned)p.zbatches);
  GemmParams<T> q = p;
  q.vec_epilogue = aligned16(p.C) && p.ldc % EPV == 0 && p.strideC % EPV == 0 && p.zstrideC % EPV == 0 &&
                   (p.C2 == nullptr || (aligned16(p.C2) && p.ldc2 % EPV == 0 && p.strideC2 % EPV == 0 && p.zstrideC2 % EPV == 0));
  if (!narrow) hipLaunchKernelGGL((gemm_nt_kernel<T, GEMM_BN>), grid, dim3(GEMM_THREADS), 0, stream, q);
  else hipLaunchKernelGGL((gemm_nt_kernel<T, 64>), grid, dim3(GEMM_THREADS), 0, stream, q);
  ODX_CHECK_LAUNCH(name);
  return ODX_OK;
}

int launch_gemm_f64(const GemmParams<double>& p, hipStream_t stream) { return launch_gemm<double>(p, stream, "gemm_nt_f64"); }
int launch_gemm_f32(const GemmParams<float>& p, hipStream_t stream) { return launch_gemm<float>(p, stream, "gemm_nt_f32"); }

}  // namespace odx

extern "C" int odx_gemm_nt_f64(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                               int64_t m, int64_t n, int64_t k, double alpha, double beta, int flags,
                               odx_stream_t stream) {
  odx::GemmParams<double> p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.m = m; p.n = n; p.k = k; p.alpha = alpha; p.beta = beta; p.flags = flags;
  return odx::launch_gemm_f64(p, odx::as_stream(stream));
}

extern "C" int odx_gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t m, int64_t n, int64_t k, float alpha, float beta, int flags,
                               odx_stream_t stream) {
  odx::GemmParams<float> p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.m = m; p.n = n; p.k = k; p.alpha = alpha; p.beta = beta; p.flags = flags;
  return odx::launch_gemm_f32(p, odx::as_stream(stream));
}
