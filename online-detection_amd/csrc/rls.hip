// Per-class regularised least squares of the box regressors (A7), f64 throughout:
//   G = [X 1]' [X 1] (+ lam I),  R = chol(G),  w_k = R^-T R^-1 [X 1]' y_k,  k = 0..3
// following RegionRefinerTrainer.solve (train_region_refiner.py:100-119).  The Gram is an
// NT GEMM on the f64 MFMA core over a gathered, transposed, bias-augmented f64 copy of the
// class's rows, formed chunk by chunk so that row shards / chunks simply accumulate.
#include <stdlib.h>
#include "odx_internal.h"

namespace odx {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4r __attribute__((ext_vector_type(4)));

// Xt[d][r] = X[idx[c0 + r]][d] (d < D), Xt[D][r] = 1, zero for r >= cn (pad up to ldt).
__global__ __launch_bounds__(256) void rls_gather_transpose_kernel(const float* __restrict__ X, int64_t ldx, int D,
                                                                   const int64_t* __restrict__ idx, int64_t c0,
                                                                   int64_t cn, double* __restrict__ Xt, int64_t ldt) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t r0 = (int64_t)blockIdx.x * 32;             // sample block
  const int64_t d0 = (int64_t)blockIdx.y * 32;             // feature block (covers D + 1)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = r0 + ty + q * 8, d = d0 + tx;
    double v = 0.0;
    if (r < cn) {
      if (d < D) v = (double)X[idx[c0 + r] * ldx + d];
      else if (d == D) v = 1.0;
    }
    tile[ty + q * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t d = d0 + ty + q * 8, r = r0 + tx;
    if (d <= D && r < ldt) Xt[d * ldt + r] = tile[tx][ty + q * 8];
  }
}

// P[i][k] = sum_d X[idx[i]][d] W[k][d] + W[k][D]; one wave per row, f64 accumulate.
__global__ __launch_bounds__(256) void rls_predict_rows_kernel(const float* __restrict__ X, int64_t ldx, int D,
                                                               const int64_t* __restrict__ idx, int64_t nc,
                                                               const double* __restrict__ W, int64_t ldw,
                                                               double* __restrict__ P, int64_t ldp) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nc) return;
  const float* x = X + (idx ? idx[i] : i) * ldx;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int d = lane; d < D; d += 64) {
    const double xv = (double)x[d];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = fma(xv, W[k * ldw + d], s[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
  }
  if (lane < 4) P[i * ldp + lane] = s[lane] + W[lane * ldw + D];
}

// The same gather for the rows of ALL classes at once: idx holds the row ids class after class, every class's segment
// padded to a multiple of 16 entries with -1 (a padded column is all zero, its bias entry too).
__global__ __launch_bounds__(256) void rls_gather_transpose_all_kernel(const float* __restrict__ X, int64_t ldx, int D,
                                                                       const int64_t* __restrict__ idx, int64_t npad,
                                                                       double* __restrict__ Xt, int64_t ldt) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * 32, d0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t r = r0 + ty + q * 8, d = d0 + tx;
    double v = 0.0;
    if (r < npad) {
      const int64_t row = idx[r];
      if (row >= 0) {
        if (d < D) v = (double)X[row * ldx + d];
        else if (d == D) v = 1.0;
      }
    }
    tile[ty + q * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t d = d0 + ty + q * 8, r = r0 + tx;
    if (d <= D && r < ldt) Xt[d * ldt + r] = tile[tx][ty + q * 8];
  }
}

// O5[c][j][d] = sum over class c's padded segment of Y5[j][r] * Xt[d][r], j = 0..4 (four whitened target rows and the ones
// row): the skinny 5 x (D + 1) products X'Y and [X 1]'1 of every class.  One wave per (class, feature row d): the row
// segment of Xt is contiguous (16-byte loads), the five Y5 segments are re-read from L2 by the D + 1 waves of the class.
// As a GEMM on 128 x 64 tiles this product used 5 of a tile's 128 rows and took 2.06 ms for 30 classes of 1e4 rows at
// D = 1024 (2.5 GB of Xt at 1.2 TB/s); here it is one HBM-bound sweep.
struct RlsSegs {
  int64_t off[ODX_MAX_ZBATCH];
  int64_t len[ODX_MAX_ZBATCH];
};

__global__ __launch_bounds__(256) void rls_xty_kernel(const double* __restrict__ Xt, int64_t ldt, const double* __restrict__ Y5,
                                                      RlsSegs sg, int D1, double* __restrict__ O5, int64_t ldo) {
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D1) return;
  const int64_t off = sg.off[c], len = sg.len[c];              // off % 16 == 0: 16-byte aligned f64 pairs
  const double* x = Xt + (int64_t)d * ldt + off;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t r = 2 * lane; r < len; r += 128) {
    const f64x2 xv = *reinterpret_cast<const f64x2*>(x + r);     // padded entries of Xt are zero: reading one past len is harmless
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f64x2 yv = *reinterpret_cast<const f64x2*>(Y5 + (int64_t)j * ldt + off + r);
      s[j] = fma(xv[0], yv[0], s[j]);
      s[j] = fma(xv[1], yv[1], s[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o);
  }
  if (lane < 5) O5[((int64_t)c * 5 + lane) * ldo + d] = s[lane];
}

// ---------------------------------------------------------------- the Grams of a class batch straight from the f32 rows
// G_c (D x D, lower 128 x 64 tiles) += X_c' X_c for the rows listed in class c's padded segment of idx (-1 = no row), on the
// f64 MFMA (v_mfma_f64_16x16x4_f64) WITHOUT the transposed f64 copy of the rows the NT GEMM needs (round 3: a 2.46 GB write
// and read-back at config 3's size): a k-tile is a block of ROWS of X — 128 (A side) + 64 (B side) consecutive floats of each,
// gathered by row id into LDS as [k][column] rows, the next k-tile's floats prefetched into registers under the MFMAs
// (rls_gram_rows32_kernel below).  One workgroup owns a tile and walks all of the class's rows: the sum over k is in one fixed
// order (bitwise reproducible; row shards add their Grams by all-reduce afterwards).
typedef float f32x2r __attribute__((ext_vector_type(2)));
constexpr int RG_BM = 128, RG_BN = 64;

// The tile (class c, tile row bi, tile column bj) of a workgroup of the Gram launches (grid: tiles x 1 x classes).
__device__ __forceinline__ void rls_gram_tile_of(int D, bool heavy_first, int& c, int& bi, int& bj) {
  c = blockIdx.z;
  const int tiles_n = (D + RG_BN - 1) / RG_BN;
  // gridDim.x = 8 x ceil(tile rows / 8) x tiles_n: workgroup x runs on XCD x & 7 (round-robin dispatch), and that XCD walks
  // the tile rows xcd, xcd + 8, ... left to right, so the tiles sharing an A panel (and the rows' B pieces next to each
  // other) share one L2; the class index rotates which rows an XCD gets (row r carries r + 1 tiles)
  if (!heavy_first) {
    const int xcd = (blockIdx.x + c) & 7, local = blockIdx.x >> 3;
    bi = 8 * (local / tiles_n) + xcd;
    bj = local % tiles_n;
  } else {
    // with the targets' products the tiles of the first column carry ~ 15 % more work: they go FIRST in dispatch order (all
    // classes'), so that none of them starts in the launch's last round and stretches its tail (measured: + 0.7 ms with the
    // tiles in their usual places).  L = the linear dispatch index; both regions are multiples of 8 long, so L & 7 still
    // names the XCD.
    const int64_t L = (int64_t)blockIdx.z * gridDim.x + blockIdx.x;
    const int nC = gridDim.z, nH = gridDim.x / tiles_n, nL = gridDim.x - nH;
    if (L < (int64_t)nC * nH) {
      c = (int)(L / nH);
      const int h = (int)(L % nH);
      bi = 8 * (h >> 3) + ((h + c) & 7);
      bj = 0;
    } else {
      const int64_t L2 = L - (int64_t)nC * nH;
      c = (int)(L2 / nL);
      const int t = (int)(L2 % nL), local = t >> 3;
      bi = 8 * (local / (tiles_n - 1)) + ((t + c) & 7);
      bj = 1 + local % (tiles_n - 1);
    }
  }
}

// The kernel: a k-tile of 32 rows held in LDS as the rows' own FLOATS (converted to f64 on the way from LDS into the matrix
// instructions' operands — six conversions per eight MFMAs, on the vector ALU beside them), two barriers per 64 matrix
// instructions.  (Round 4's form — 16-row k-tiles held as f64 in LDS — gave the same sums bit for bit and was 0.5-1.5 % slower
// in same-box A/B runs; it is gone.)  Above the diagonal the half tiles leave their unused block unwritten.
constexpr int RG32_BK = 32;
constexpr int RG32_LDA = RG_BM + 16, RG32_LDB = RG_BN + 16;   // floats per LDS row: 16 banks further per k-row (the four k-rows a
                                                               // 64-lane ds_read_b32 touches fall on four different bank quarters)

__global__ __launch_bounds__(256, 3) void rls_gram_rows32_kernel(const float* __restrict__ X, int64_t ldx, int D,
                                                                 const int64_t* __restrict__ idx, RlsSegs sg, double* __restrict__ G,
                                                                 int64_t ldg, int64_t g_stride, const float* __restrict__ Yraw,
                                                                 int64_t ldyr, double* __restrict__ O5, int64_t ldo) {
  __shared__ __attribute__((aligned(16))) float lds_a[RG32_BK * RG32_LDA];
  __shared__ __attribute__((aligned(16))) float lds_b[RG32_BK * RG32_LDB];
  __shared__ __attribute__((aligned(16))) float lds_y[RG32_BK * 4];
  int c, bi, bj;
  rls_gram_tile_of(D, Yraw != nullptr, c, bi, bj);
  const int i0 = bi * RG_BM, j0 = bj * RG_BN;
  if (i0 >= D || j0 > i0 + RG_BM - 1) return;                  // lower tiles only
  const int64_t off = sg.off[c], len = sg.len[c];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // a tile's four waves own 64 x 32 outputs each — except in the tile right of the diagonal's first half (columns i0 + 64 ..
  // against rows i0 ..): its upper 64 rows lie strictly above the diagonal, so the four waves share the LOWER 64 x 64 block,
  // 32 x 32 each, and issue half the matrix instructions (8 of a tile row's 2 bi + 2 tiles: 5 % of the launch's)
  const bool half = j0 == i0 + 64;
  const int arow = half ? 64 + wr * 32 : wr * 64;
  const int krow = tid >> 4, seg = tid & 15;                   // this thread stages k-rows krow and krow + 16 of a k-tile
  int cae[4], cbe[2];
  bool aok[4], bok[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = i0 + q * 32 + seg * 2;
    aok[q] = col < D;
    cae[q] = aok[q] ? col : 0;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int col = j0 + q * 32 + seg * 2;
    bok[q] = col < D;
    cbe[q] = bok[q] ? col : 0;
  }
  f64x4 acc[4][2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t nk = (len + RG32_BK - 1) / RG32_BK;
  const bool xty = Yraw != nullptr && j0 == 0;
  if (nk == 0) {
    if (xty && tid < 128 && i0 + tid < D)
      for (int j = 0; j < 5; ++j) O5[((int64_t)c * 5 + j) * ldo + i0 + tid] = 0.0;
    return;
  }
  // row ids one k-tile ahead of the rows (positions past the class's rows — its -1 padding, another class's segment, the end
  // of the array — count as no row and are not read)
  auto row_of = [&](int64_t kt, int h) -> int64_t {
    const int64_t q = kt * RG32_BK + krow + 16 * h;
    return q < len ? idx[off + q] : -1;
  };
  f32x2r ra[2][4], rb[2][2];
  f32x4r ry[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  bool valid[2];
  auto load = [&](int64_t row, int h) {
    valid[h] = row >= 0;
    const float* x = X + (valid[h] ? row : 0) * ldx;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[h][q] = *reinterpret_cast<const f32x2r*>(x + cae[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) rb[h][q] = *reinterpret_cast<const f32x2r*>(x + cbe[q]);
    if (xty && seg == 0) ry[h] = *reinterpret_cast<const f32x4r*>(Yraw + (valid[h] ? row : 0) * ldyr);
  };
  double ysum[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int ycol = tid & 127, yhalf = tid >> 7;               // the targets' products: column ycol of the A side, two k-rows of every step
  int64_t row_next[2] = {row_of(1, 0), row_of(1, 1)};
  load(row_of(0, 0), 0);
  load(row_of(0, 1), 1);
  const int r16 = lane & 15, kq = lane >> 4;
  for (int64_t kt = 0; kt < nk; ++kt) {
    __syncthreads();                                           // everyone finished reading the previous k-tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* da = lds_a + (krow + 16 * h) * RG32_LDA + seg * 2;
      float* db = lds_b + (krow + 16 * h) * RG32_LDB + seg * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float m = valid[h] && aok[q] ? 1.f : 0.f;
        *reinterpret_cast<f32x2r*>(da + q * 32) = f32x2r{m * ra[h][q][0], m * ra[h][q][1]};
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float m = valid[h] && bok[q] ? 1.f : 0.f;
        *reinterpret_cast<f32x2r*>(db + q * 32) = f32x2r{m * rb[h][q][0], m * rb[h][q][1]};
      }
      if (xty && seg == 0) *reinterpret_cast<f32x4r*>(lds_y + (krow + 16 * h) * 4) = ry[h];   // (a padded row's A entries are zero)
    }
    __syncthreads();
    load(row_next[0], 0);
    load(row_next[1], 1);
    row_next[0] = row_of(kt + 2, 0);
    row_next[1] = row_of(kt + 2, 1);
    __builtin_amdgcn_sched_barrier(0);                         // (the loads stay in front of the MFMAs they hide under)
    float a[2][4], b[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[0][t] = lds_a[kq * RG32_LDA + arow + t * 16 + r16];
#pragma unroll
    for (int t = 0; t < 2; ++t) b[0][t] = lds_b[kq * RG32_LDB + wc * 32 + t * 16 + r16];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 7) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[nxt][t] = lds_a[((ks + 1) * 4 + kq) * RG32_LDA + arow + t * 16 + r16];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[nxt][t] = lds_b[((ks + 1) * 4 + kq) * RG32_LDB + wc * 32 + t * 16 + r16];
        __builtin_amdgcn_sched_barrier(0);                     // (reads first: the scheduler sinks them below the multiplies otherwise)
      }
      double ad[4], bd[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) ad[t] = (double)a[cur][t];
#pragma unroll
      for (int t = 0; t < 2; ++t) bd[t] = (double)b[cur][t];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[tm], bd[tn], acc[tm][tn], 0, 0, 0);
      if (!half) {
#pragma unroll
        for (int tm = 2; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[tm], bd[tn], acc[tm][tn], 0, 0, 0);
      }
      if (xty) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int kr = ks * 4 + yhalf * 2 + r;
          const double av = (double)lds_a[kr * RG32_LDA + ycol];
          const f32x4r y = *reinterpret_cast<const f32x4r*>(lds_y + kr * 4);
          ysum[0] = fma(av, (double)y[0], ysum[0]);
          ysum[1] = fma(av, (double)y[1], ysum[1]);
          ysum[2] = fma(av, (double)y[2], ysum[2]);
          ysum[3] = fma(av, (double)y[3], ysum[3]);
          ysum[4] += av;
        }
      }
    }
  }
  if (xty) {
    __syncthreads();
    double* red = reinterpret_cast<double*>(lds_a);            // (5 x 128 doubles of the 32 x 144 floats)
    if (yhalf == 1) {
#pragma unroll
      for (int j = 0; j < 5; ++j) red[j * 128 + ycol] = ysum[j];
    }
    __syncthreads();
    if (yhalf == 0 && i0 + ycol < D) {
#pragma unroll
      for (int j = 0; j < 5; ++j) O5[((int64_t)c * 5 + j) * ldo + i0 + ycol] = ysum[j] + red[j * 128 + ycol];
    }
  }
  double* g = G + (int64_t)c * g_stride;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = i0 + arow + tm * 16 + kq + 4 * reg;
      if (row >= D || (half && tm >= 2)) continue;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = j0 + wc * 32 + tn * 16 + r16;
        if (col < D) g[(int64_t)row * ldg + col] += acc[tm][tn][reg];
      }
    }
}

// The skinny products of the same rows: P[c][chunk][j][d] = sum over the chunk's rows of Y5[j][r] [X 1][r][d], j = 0..4 (the four
// whitened target rows and the ones row), d = 0..D (d = D: the bias column).  A thread owns a column and walks the chunk's
// rows (coalesced 1-KB row pieces per workgroup, the Y5 values broadcast); RX_CH chunks per class give the launch its
// parallelism, rls_xty_reduce_kernel adds them in a fixed order.
constexpr int RX_CH = 32;

__global__ __launch_bounds__(256) void rls_xty_rows_kernel(const float* __restrict__ X, int64_t ldx, int D, const int64_t* __restrict__ idx,
                                                           RlsSegs sg, const double* __restrict__ Y4, int64_t ldy, double* __restrict__ P,
                                                           int64_t ldo) {
  const int c = blockIdx.z, chunk = blockIdx.y;
  const int d = blockIdx.x * 256 + threadIdx.x;
  const int64_t off = sg.off[c], len = sg.len[c];
  const int64_t per = ((len + RX_CH - 1) / RX_CH + 15) / 16 * 16;
  const int64_t r0 = (int64_t)chunk * per, r1 = r0 + per < len ? r0 + per : len;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (d <= D) {
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t row = idx[off + r];
      if (row < 0) continue;
      const double x = d < D ? (double)X[row * ldx + d] : 1.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = fma(Y4[(int64_t)j * ldy + off + r], x, s[j]);
      s[4] += x;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) P[(((int64_t)c * RX_CH + chunk) * 5 + j) * ldo + d] = s[j];
  }
}

__global__ __launch_bounds__(256) void rls_xty_reduce_kernel(const double* __restrict__ P, int64_t ldo, int D1, double* __restrict__ O5) {
  const int c = blockIdx.y, j = blockIdx.z;
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D1) return;
  double s = 0.0;
  for (int chunk = 0; chunk < RX_CH; ++chunk) s += P[(((int64_t)c * RX_CH + chunk) * 5 + j) * ldo + d];
  O5[((int64_t)c * 5 + j) * ldo + d] = s;
}

// P[i][k] for the rows of ALL classes with one launch: row i belongs to the class whose [start, start + len) holds it.  A wave
// takes RP_R consecutive rows of ONE class: the class's four weight rows are read once per column chunk for all of them (a row
// of its own per wave read 32 KB of weights from the caches for every 4 KB row from memory — the kernel ran at 2.2 TB/s).
// sg.off[c] = first row of class c, sg.len[c] = first row GROUP of class c.  Per row the sums run in the order of the one-row-per-wave
// form of this kernel (lane's chunks ascending, then the butterfly).
constexpr int RP_R = 4;

__global__ __launch_bounds__(256) void rls_predict_rows_batched_kernel(const float* __restrict__ X, int64_t ldx, int D,
                                                                       const int64_t* __restrict__ idx, RlsSegs sg, int C,
                                                                       const double* __restrict__ W, int64_t ldw,
                                                                       int64_t w_stride, double* __restrict__ P, int64_t ldp,
                                                                       int64_t total, int64_t groups) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= groups) return;
  int c = 0;
#pragma unroll 1
  for (int k = 1; k < C; ++k)
    if (g >= sg.len[k]) c = k;
  const int64_t r0 = sg.off[c] + (g - sg.len[c]) * RP_R;
  const int64_t rend = c + 1 < C ? sg.off[c + 1] : total;
  const int nr = (int)(rend - r0 < RP_R ? rend - r0 : RP_R);
  const double* Wc = W + (int64_t)c * w_stride;
  const float* x[RP_R];
#pragma unroll
  for (int u = 0; u < RP_R; ++u) x[u] = X + idx[r0 + (u < nr ? u : nr - 1)] * ldx;       // (a short group repeats its last row)
  double s[RP_R][4];
#pragma unroll
  for (int u = 0; u < RP_R; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[u][k] = 0.0;
  const int nvec = D / 4;                                     // ldx % 4 == 0 and X 16-byte aligned: whole float4s
#pragma unroll 2
  for (int cc = lane; cc < nvec; cc += 64) {
    f32x4r v[RP_R];
#pragma unroll
    for (int u = 0; u < RP_R; ++u) v[u] = *reinterpret_cast<const f32x4r*>(x[u] + cc * 4);
    double w[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) w[k][q] = Wc[k * ldw + cc * 4 + q];
#pragma unroll
    for (int u = 0; u < RP_R; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double xv = (double)v[u][q];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[u][k] = fma(xv, w[k][q], s[u][k]);
      }
  }
  for (int d = nvec * 4 + lane; d < D; d += 64) {
#pragma unroll
    for (int u = 0; u < RP_R; ++u) {
      const double xv = (double)x[u][d];
#pragma unroll
      for (int k = 0; k < 4; ++k) s[u][k] = fma(xv, Wc[k * ldw + d], s[u][k]);
    }
  }
#pragma unroll
  for (int u = 0; u < RP_R; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[u][k] += __shfl_xor(s[u][k], off);
    }
  if (lane < 4 * RP_R) {
    const int u = lane >> 2, k = lane & 3;
    double v = 0.0;
#pragma unroll
    for (int uu = 0; uu < RP_R; ++uu)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        if (uu == u && kk == k) v = s[uu][kk];
    if (u < nr) P[(r0 + u) * ldp + k] = v + Wc[k * ldw + D];
  }
}

static int64_t rls_chunk(int64_t workspace_bytes, int D) {
  const int64_t D1 = D + 1;
  int64_t chunk = workspace_bytes / (D1 * (int64_t)sizeof(double));
  chunk = (chunk / 16) * 16;
  return chunk;
}

}  // namespace odx

using namespace odx;

extern "C" int64_t odx_rls_gram_workspace_bytes(int64_t nc, int D) {
  if (nc <= 0 || D <= 0) return 0;
  int64_t chunk = round_up(nc, 16);
  if (chunk > 32768) chunk = 32768;
  return chunk * (int64_t)(D + 1) * (int64_t)sizeof(double);
}

extern "C" int odx_rls_gram_f64(const float* X, int64_t ldx, int D, const int64_t* idx, int64_t nc, const double* Yt,
                                int64_t ldy, double* G, int64_t ldg, double* XtY, int64_t ldxy, void* workspace,
                                int64_t workspace_bytes, odx_stream_t stream) {
  if (nc <= 0) return ODX_OK;
  ODX_REQUIRE(X && idx && Yt && G && XtY && workspace && D > 0, "odx_rls_gram_f64: bad argument");
  ODX_REQUIRE(ldy % 2 == 0 && aligned16(Yt) && aligned16(workspace), "odx_rls_gram_f64: Yt/workspace must be 16-byte aligned, ldy even");
  const int64_t D1 = D + 1;
  ODX_REQUIRE(ldg >= D1 && ldxy >= D1, "odx_rls_gram_f64: ldg/ldxy < D + 1");
  const int64_t chunk = rls_chunk(workspace_bytes, D);
  if (chunk < 16) {
    set_error("odx_rls_gram_f64: workspace too small for one 16-row chunk");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  double* Xt = static_cast<double*>(workspace);
  for (int64_t c0 = 0; c0 < nc; c0 += chunk) {
    const int64_t cn = nc - c0 < chunk ? nc - c0 : chunk;
    const int64_t ldt = round_up(cn, 16);
    dim3 grid((unsigned)ceil_div(ldt, 32), (unsigned)ceil_div(D1, 32));
    hipLaunchKernelGGL(rls_gather_transpose_kernel, grid, dim3(256), 0, s, X, ldx, D, idx, c0, cn, Xt, ldt);
    ODX_CHECK_LAUNCH("rls_gather_transpose");
    GemmParams<double> g;
    g.A = Xt; g.lda = ldt; g.B = Xt; g.ldb = ldt; g.C = G; g.ldc = ldg;
    g.m = D1; g.n = D1; g.k = cn; g.alpha = 1.0; g.beta = 1.0; g.flags = ODX_GEMM_LOWER_ONLY;
    ODX_PROPAGATE(launch_gemm_f64(g, s));
    GemmParams<double> h;
    h.A = Yt + c0; h.lda = ldy; h.B = Xt; h.ldb = ldt; h.C = XtY; h.ldc = ldxy;
    h.m = 4; h.n = D1; h.k = cn; h.alpha = 1.0; h.beta = 1.0;
    ODX_PROPAGATE(launch_gemm_f64(h, s));
  }
  return ODX_OK;
}

// workspace: Dinv | WT (D1*D1) | Li (D1 x ld) | Lit (D1 x ld) | z (ld)
extern "C" int64_t odx_rls_solve_workspace_bytes(int D) {
  if (D <= 0) return 0;
  const int64_t D1 = D + 1, ld = round_up(D1, 2);
  return odx_potrf_workspace_bytes(D1) + (round_up(D1 * D1, 2) + 2 * D1 * ld + ld) * (int64_t)sizeof(double);
}

extern "C" int odx_rls_solve_f64(double* G, int64_t ldg, int D, double lam, const double* XtY, int64_t ldxy, double* W,
                                 int64_t ldw, int32_t* info, void* workspace, int64_t workspace_bytes,
                                 odx_stream_t stream) {
  ODX_REQUIRE(G && XtY && W && info && workspace && D > 0, "odx_rls_solve_f64: bad argument");
  const int64_t D1 = D + 1, ld = round_up(D1, 2);
  ODX_REQUIRE(ldg % 2 == 0 && ldg >= D1 && aligned16(G), "odx_rls_solve_f64: G must be 16-byte aligned with even ldg >= D + 1");
  ODX_REQUIRE(ldxy % 2 == 0 && ldxy >= D1 && aligned16(XtY), "odx_rls_solve_f64: XtY must be 16-byte aligned with even ldxy");
  ODX_REQUIRE(ldw >= D1 && aligned16(workspace), "odx_rls_solve_f64: ldw < D + 1 or unaligned workspace");
  if (workspace_bytes < odx_rls_solve_workspace_bytes(D)) {
    set_error("odx_rls_solve_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  double* Dinv = static_cast<double*>(workspace);
  double* WT = Dinv + ceil_div(D1, POTRF_NB) * POTRF_NB * POTRF_NB;
  double* Li = WT + round_up(D1 * D1, 2);
  double* Lit = Li + D1 * ld;
  double* z = Lit + D1 * ld;
  ODX_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), s));
  ODX_PROPAGATE(add_diag_f64(G, ldg, D1, lam, s));
  ODX_PROPAGATE(potrf_f64(G, ldg, D1, Dinv, info, s));
  ODX_PROPAGATE(fill_f64(Li, ld, D1, D1, 0.0, s));
  ODX_PROPAGATE(fill_f64(Lit, ld, D1, D1, 0.0, s));
  ODX_PROPAGATE(trtri_from_diag_f64(G, ldg, D1, Dinv, Li, Lit, ld, WT, s));
  for (int k = 0; k < 4; ++k) {
    ODX_PROPAGATE(odx_trmv_f64(Li, ld, D1, 0, XtY + k * ldxy, 1.0, 0.0, nullptr, z, stream));
    ODX_PROPAGATE(odx_trmv_f64(Lit, ld, D1, 1, z, 1.0, 0.0, nullptr, W + k * ldw, stream));
  }
  return ODX_OK;
}

extern "C" int odx_rls_predict_rows_f64(const float* X, int64_t ldx, int D, const int64_t* idx, int64_t nc,
                                        const double* W, int64_t ldw, double* P, int64_t ldp, odx_stream_t stream) {
  if (nc <= 0) return ODX_OK;
  ODX_REQUIRE(X && W && P && D > 0 && ldw >= D + 1 && ldp >= 4, "odx_rls_predict_rows_f64: bad argument");
  hipLaunchKernelGGL(rls_predict_rows_kernel, dim3((unsigned)ceil_div(nc, 4)), dim3(256), 0, as_stream(stream), X, ldx,
                     D, idx, nc, W, ldw, P, ldp);
  ODX_CHECK_LAUNCH("odx_rls_predict_rows_f64");
  return ODX_OK;
}

extern "C" int odx_rls_predict_rows_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx, const int64_t* seg_start,
                                                int C, int64_t total, const double* W, int64_t ldw, int64_t w_stride, double* P,
                                                int64_t ldp, odx_stream_t stream) {
  if (total <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(C <= ODX_MAX_ZBATCH, "odx_rls_predict_rows_batched_f64: at most %d classes per call", ODX_MAX_ZBATCH);
  ODX_REQUIRE(X && idx && seg_start && W && P && D > 0 && ldw >= D + 1 && ldp >= 4 && w_stride >= 4 * ldw, "odx_rls_predict_rows_batched_f64: bad argument");
  ODX_REQUIRE(ldx % 4 == 0 && aligned16(X), "odx_rls_predict_rows_batched_f64: X must be 16-byte aligned with ldx %% 4 == 0");
  RlsSegs sg;
  for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = 0;
  for (int c = 0; c < C; ++c) {
    ODX_REQUIRE(seg_start[c] >= 0 && seg_start[c] <= total && (c == 0 ? seg_start[c] == 0 : seg_start[c] >= seg_start[c - 1]),
                "odx_rls_predict_rows_batched_f64: class starts must be ascending from 0");
    sg.off[c] = seg_start[c];
  }
  int64_t groups = 0;
  for (int c = 0; c < C; ++c) {
    sg.len[c] = groups;                                       // (the class's first row group)
    groups += ceil_div((c + 1 < C ? seg_start[c + 1] : total) - seg_start[c], (int64_t)RP_R);
  }
  hipLaunchKernelGGL(rls_predict_rows_batched_kernel, dim3((unsigned)ceil_div(groups, 4)), dim3(256), 0, as_stream(stream), X, ldx, D, idx,
                     sg, C, W, ldw, w_stride, P, ldp, total, groups);
  ODX_CHECK_LAUNCH("odx_rls_predict_rows_batched_f64");
  return ODX_OK;
}

// ---------------------------------------------------------------- the regressors of a class batch
// RegionRefinerTrainer trains its classes one after the other (train_region_refiner.py:27-98); they are independent, and
// at the reference's sizes (D + 1 = 1025 .. 2049, a few thousand rows per class) one class fills neither the f64 matrix
// cores (a Gram of 1025 x 1025 outputs is ~80 tiles for 256 CUs) nor the launch queue (its Cholesky is a chain of
// dependent small kernels).  Here every kernel of the per-class path takes the class as a grid dimension:
//   odx_rls_gram_batched_f64   one gather of all classes' rows (sorted by class, segments padded to 16) + ONE Gram GEMM and
//                              ONE X'Y GEMM whose class z contracts over its own column window
//   odx_rls_solve_batched_f64  + lam I, Cholesky, triangular inverses and the eight triangular products for all classes
// with the arithmetic of odx_rls_gram_f64 / odx_rls_solve_f64 per class.  The split in two calls leaves room for the
// all-reduce of the Grams when rows are sharded.
// workspace: Xt ((D + 1) x ldt) | Y5 (5 x ldt: the four target rows and the ones row) | O5 (32 classes x 5 x ldo)
extern "C" int64_t odx_rls_gram_batched_workspace_bytes(int64_t npad, int D) {
  if (npad <= 0 || D <= 0) return 0;
  const int64_t ldt = round_up(npad, 16), ldo = round_up((int64_t)D + 1, 2);
  const int64_t nt = ldt * (int64_t)(D + 1) + 5 * ldt + (int64_t)ODX_MAX_ZBATCH * 5 * ldo;          // the NT form's copy of the rows
  const int64_t rows = (int64_t)ODX_MAX_ZBATCH * (32 + 1) * 5 * ldo;                               // the rows form's partial sums
  return (nt > rows ? nt : rows) * (int64_t)sizeof(double);
}

// XtY[c] (4 x D1) += O5[c] rows 0..3;  G[c] row D (the bias row of the lower triangle) += O5[c] row 4
__global__ __launch_bounds__(256) void rls_fold_bias_kernel(const double* __restrict__ O5, int64_t ldo, int D1, double* __restrict__ XtY,
                                                            int64_t ldxy, int64_t xy_stride, double* __restrict__ G, int64_t ldg,
                                                            int64_t g_stride) {
  const int c = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= D1) return;
  const double* o = O5 + (int64_t)c * 5 * ldo;
#pragma unroll
  for (int r = 0; r < 4; ++r) XtY[(int64_t)c * xy_stride + r * ldxy + j] += o[r * ldo + j];
  G[(int64_t)c * g_stride + (int64_t)(D1 - 1) * ldg + j] += o[4 * ldo + j];
}

// The whitened targets' products from the raw ones (rls_gram_rows32_kernel's O5 = [Y 1]' X per class, columns 0 .. D - 1): with
// Yw = (Y - 1 mu') T,   X' Yw = (X' Y - (X' 1) mu') T,   1' Yw = 0,   and the Gram's bias row [X 1]' 1 = (X' 1, n).
// stats (C, 9, 4) f64 = [mu; T; T_inv] per class (the block the trainer keeps anyway), cnt (C) f64 = rows per class.
__global__ __launch_bounds__(256) void rls_fold_whitened_kernel(const double* __restrict__ O5, int64_t ldo, int D1,
                                                                const double* __restrict__ stats, const double* __restrict__ cnt,
                                                                double* __restrict__ XtY, int64_t ldxy, int64_t xy_stride,
                                                                double* __restrict__ G, int64_t ldg, int64_t g_stride) {
  const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D1) return;
  const double* st = stats + (int64_t)c * 36;
  double* xy = XtY + (int64_t)c * xy_stride;
  double* gb = G + (int64_t)c * g_stride + (int64_t)(D1 - 1) * ldg;
  if (d == D1 - 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) xy[j * ldxy + d] = 0.0;
    gb[d] += cnt[c];
    return;
  }
  const double* o = O5 + (int64_t)c * 5 * ldo;
  const double ones = o[4 * ldo + d];
  double v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = o[i * ldo + d] - st[i] * ones;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) t = fma(v[i], st[4 + i * 4 + j], t);
    xy[j * ldxy + d] = t;
  }
  gb[d] += ones;
}

// The Grams come straight from the f32 rows (rls_gram_rows32_kernel) when the rows allow 16-byte column groups; otherwise
// (D % 8 != 0, unaligned rows) a transposed f64 copy of the rows + the generic NT GEMM.  The option rls_force_nt_gram (a test
// hook, odx_set_option) sends every D down the second route so that both can be compared on the same rows.
static bool rls_rows_form(const float* X, int64_t ldx, int D) {
  return !lib_option(OPT_RLS_FORCE_NT_GRAM) && D % 8 == 0 && ldx % 4 == 0 && aligned16(X);
}

extern "C" int odx_rls_rows_form(const float* X, int64_t ldx, int D) { return rls_rows_form(X, ldx, D) ? 1 : 0; }

// The padded row-id array of a class batch and its inverse maps, in one launch: run (total) holds the row ids class after class
// (class k: len[k] of them), the padded array gives class k the positions seg_off[k] .. (a multiple of 16, -1 behind its rows).
//   idx_pad[p] = row id or -1;   for the i-th id of run: gid[i] = its class slot, pos[i] = its rank in the class, dest[i] = p;
//   lens[k] = len[k].
// (the trainer built these with a dozen tensor statements and three small host-to-device copies: 0.3 ms of launch latency in
// front of the Grams)
__global__ __launch_bounds__(256) void rls_pad_index_kernel(const int64_t* __restrict__ run, RlsSegs sg, RlsSegs st, int C, int64_t npad,
                                                            int64_t* __restrict__ idx_pad, int64_t* __restrict__ gid,
                                                            int64_t* __restrict__ pos, int64_t* __restrict__ dest,
                                                            int64_t* __restrict__ lens) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < C) lens[p] = sg.len[p];                              // (the class sizes as a device array: no copy of the host's list)
  if (p >= npad) return;
  int k = 0;
#pragma unroll 1
  for (int q = 1; q < C; ++q)
    if (p >= sg.off[q]) k = q;
  const int64_t r = p - sg.off[k];
  if (r < sg.len[k]) {
    const int64_t i = st.off[k] + r;
    idx_pad[p] = run[i];
    gid[i] = k;
    pos[i] = r;
    dest[i] = p;
  } else {
    idx_pad[p] = -1;
  }
}

extern "C" int odx_rls_pad_index(const int64_t* run, int64_t total, const int64_t* seg_off, const int64_t* seg_len, int C, int64_t npad,
                                 int64_t* idx_pad, int64_t* gid, int64_t* pos, int64_t* dest, int64_t* lens, odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_pad_index: 1..%d classes per call", ODX_MAX_ZBATCH);
  if (npad <= 0) return ODX_OK;
  ODX_REQUIRE(run && seg_off && seg_len && idx_pad && gid && pos && dest && lens, "odx_rls_pad_index: bad argument");
  RlsSegs sg, st;
  for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = st.off[c] = st.len[c] = 0;
  int64_t at = 0;
  for (int c = 0; c < C; ++c) {
    ODX_REQUIRE(seg_len[c] >= 0 && seg_off[c] >= (c ? seg_off[c - 1] + seg_len[c - 1] : 0) && seg_off[c] + seg_len[c] <= npad,
                "odx_rls_pad_index: class %d: segments must ascend without overlap inside the padded array", c);
    sg.off[c] = seg_off[c];
    sg.len[c] = seg_len[c];
    st.off[c] = at;
    at += seg_len[c];
  }
  ODX_REQUIRE(at == total, "odx_rls_pad_index: the segment lengths must add up to the number of row ids");
  hipLaunchKernelGGL(rls_pad_index_kernel, dim3((unsigned)ceil_div(npad > C ? npad : C, 256)), dim3(256), 0, as_stream(stream), run, sg, st,
                     C, npad, idx_pad, gid, pos, dest, lens);
  ODX_CHECK_LAUNCH("rls_pad_index");
  return ODX_OK;
}

// The Grams of a class batch AND the raw targets' products [Y 1]' X in one sweep over the rows (rls_gram_rows32_kernel, rows form
// only): Yraw (n, >= 4) f32 holds the UN-whitened targets by row id, O5 (C, 5, ldo) f64 receives Y' X (rows 0 .. 3) and 1' X (row 4)
// for columns 0 .. D - 1.  odx_rls_fold_whitened_f64 turns them into the whitened targets' X' Yw and the Gram's bias row once the
// statistics are known — the sweep does not wait for them.
extern "C" int odx_rls_gram_raw_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                                            const int64_t* seg_off, const int64_t* seg_len, int C, const float* Yraw, int64_t ldyr,
                                            double* G, int64_t ldg, int64_t g_stride, double* O5, int64_t ldo, odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_gram_raw_batched_f64: 1..%d classes per call", ODX_MAX_ZBATCH);
  ODX_REQUIRE(X && idx_pad && seg_off && seg_len && Yraw && G && O5 && D > 0 && npad > 0, "odx_rls_gram_raw_batched_f64: bad argument");
  ODX_REQUIRE(rls_rows_form(X, ldx, D), "odx_rls_gram_raw_batched_f64: needs the rows form (odx_rls_rows_form)");
  const int64_t D1 = D + 1;
  ODX_REQUIRE(ldyr >= 4 && ldyr % 4 == 0 && aligned16(Yraw), "odx_rls_gram_raw_batched_f64: Yraw rows of 4 floats, 16-byte aligned");
  ODX_REQUIRE(ldg >= D1 && g_stride >= D1 * ldg && ldo >= D, "odx_rls_gram_raw_batched_f64: output strides too small");
  RlsSegs sg;
  for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = 0;
  for (int c = 0; c < C; ++c) {
    ODX_REQUIRE(seg_off[c] % 16 == 0 && seg_len[c] >= 0 && round_up(seg_off[c] + seg_len[c], 16) <= npad,
                "odx_rls_gram_raw_batched_f64: class %d: segment must start at a multiple of 16 and end, padded to one, inside the index array", c);
    sg.off[c] = seg_off[c];
    sg.len[c] = seg_len[c];
  }
  const int tiles = (int)(8 * ceil_div(ceil_div(D, RG_BM), 8) * ceil_div(D, RG_BN));
  hipLaunchKernelGGL(rls_gram_rows32_kernel, dim3((unsigned)tiles, 1, (unsigned)C), dim3(256), 0,
                     as_stream(stream), X, ldx, D, idx_pad, sg, G, ldg, g_stride, Yraw, ldyr, O5, ldo);
  ODX_CHECK_LAUNCH("rls_gram_rows (raw targets)");
  return ODX_OK;
}

// XtY[c] (4 x (D + 1)) = the whitened targets' [X 1]' Yw and G[c]'s bias row += [X 1]' 1, from odx_rls_gram_raw_batched_f64's O5,
// the classes' statistics stats (C, 9, 4) f64 = [mu; T; T_inv] and row counts cnt (C) f64.
extern "C" int odx_rls_fold_whitened_f64(const double* O5, int64_t ldo, int D, int C, const double* stats, const double* cnt,
                                         double* G, int64_t ldg, int64_t g_stride, double* XtY, int64_t ldxy, int64_t xy_stride,
                                         odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_fold_whitened_f64: 1..%d classes per call", ODX_MAX_ZBATCH);
  ODX_REQUIRE(O5 && stats && cnt && G && XtY && D > 0, "odx_rls_fold_whitened_f64: bad argument");
  const int64_t D1 = D + 1;
  ODX_REQUIRE(ldo >= D && ldg >= D1 && g_stride >= D1 * ldg && ldxy >= D1 && xy_stride >= 4 * ldxy, "odx_rls_fold_whitened_f64: strides too small");
  hipLaunchKernelGGL(rls_fold_whitened_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C), dim3(256), 0, as_stream(stream), O5, ldo, (int)D1,
                     stats, cnt, XtY, ldxy, xy_stride, G, ldg, g_stride);
  ODX_CHECK_LAUNCH("rls_fold_whitened");
  return ODX_OK;
}

extern "C" int odx_rls_xty_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                                       const int64_t* seg_off, const int64_t* seg_len, int C, const double* Yt, int64_t ldy,
                                       double* G, int64_t ldg, int64_t g_stride, double* XtY, int64_t ldxy, int64_t xy_stride,
                                       void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_xty_batched_f64: 1..%d classes per call", ODX_MAX_ZBATCH);
  if (npad <= 0) return ODX_OK;
  ODX_REQUIRE(X && idx_pad && seg_off && seg_len && Yt && G && XtY && workspace && D > 0, "odx_rls_xty_batched_f64: bad argument");
  const int64_t D1 = D + 1, ldt = round_up(npad, 16);
  ODX_REQUIRE(ldy % 2 == 0 && ldy >= ldt && aligned16(Yt) && aligned16(workspace), "odx_rls_xty_batched_f64: Yt/workspace 16-byte aligned, ldy even >= padded rows");
  ODX_REQUIRE(ldg >= D1 && ldxy >= D1 && g_stride >= D1 * ldg && xy_stride >= 4 * ldxy, "odx_rls_xty_batched_f64: output strides too small");
  if (!rls_rows_form(X, ldx, D)) {
    set_error("odx_rls_xty_batched_f64: needs the rows form (D %% 8 == 0, ldx %% 4 == 0, X 16-byte aligned): odx_rls_rows_form");
    return ODX_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < odx_rls_gram_batched_workspace_bytes(npad, D)) {
    set_error("odx_rls_xty_batched_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  RlsSegs sg;
  for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = 0;
  for (int c = 0; c < C; ++c) {
    ODX_REQUIRE(seg_off[c] % 16 == 0 && seg_len[c] >= 0 && round_up(seg_off[c] + seg_len[c], 16) <= npad,
                "odx_rls_xty_batched_f64: class %d: segment must start at a multiple of 16 and end, padded to one, inside the index array", c);
    sg.off[c] = seg_off[c];
    sg.len[c] = seg_len[c];
  }
  const int64_t ldo = round_up(D1, 2);
  double* P = static_cast<double*>(workspace);                       // C x RX_CH x 5 x ldo partial sums, then O5
  double* O5 = P + (int64_t)C * RX_CH * 5 * ldo;
  hipLaunchKernelGGL(rls_xty_rows_kernel, dim3((unsigned)ceil_div(D1, 256), RX_CH, (unsigned)C), dim3(256), 0, s, X, ldx, D, idx_pad, sg, Yt,
                     ldy, P, ldo);
  ODX_CHECK_LAUNCH("rls_xty_rows");
  hipLaunchKernelGGL(rls_xty_reduce_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C, 5), dim3(256), 0, s, P, ldo, (int)D1, O5);
  ODX_CHECK_LAUNCH("rls_xty_reduce");
  hipLaunchKernelGGL(rls_fold_bias_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C), dim3(256), 0, s, O5, ldo, (int)D1, XtY, ldxy,
                     xy_stride, G, ldg, g_stride);
  ODX_CHECK_LAUNCH("rls_fold_bias");
  return ODX_OK;
}

extern "C" int odx_rls_gram_batched_f64(const float* X, int64_t ldx, int D, const int64_t* idx_pad, int64_t npad,
                                        const int64_t* seg_off, const int64_t* seg_len, int C, const double* Yt,
                                        int64_t ldy, double* G, int64_t ldg, int64_t g_stride, double* XtY,
                                        int64_t ldxy, int64_t xy_stride, void* workspace, int64_t workspace_bytes,
                                        odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_gram_batched_f64: 1..%d classes per call", ODX_MAX_ZBATCH);
  if (npad <= 0) return ODX_OK;
  const bool gram_only = Yt == nullptr && XtY == nullptr;          // (the targets follow with odx_rls_xty_batched_f64)
  ODX_REQUIRE(X && idx_pad && seg_off && seg_len && G && workspace && D > 0 && (gram_only || (Yt && XtY)),
              "odx_rls_gram_batched_f64: bad argument");
  const int64_t D1 = D + 1, ldt = round_up(npad, 16);
  ODX_REQUIRE(aligned16(workspace) && (gram_only || (ldy % 2 == 0 && ldy >= ldt && aligned16(Yt))),
              "odx_rls_gram_batched_f64: Yt/workspace 16-byte aligned, ldy even >= padded rows");
  ODX_REQUIRE(ldg >= D1 && g_stride >= D1 * ldg && (gram_only || (ldxy >= D1 && xy_stride >= 4 * ldxy)),
              "odx_rls_gram_batched_f64: output strides too small");
  if (workspace_bytes < odx_rls_gram_batched_workspace_bytes(npad, D)) {
    set_error("odx_rls_gram_batched_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  {
    // default: the Grams and the skinny products straight from the f32 rows (rls_gram_rows32_kernel, rls_xty_rows_kernel)
    if (rls_rows_form(X, ldx, D)) {
      RlsSegs sg;
      for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = 0;
      for (int c = 0; c < C; ++c) {
        // (the kernel reads whole 16-row k-tiles: the -1 padding behind a segment must lie inside the array too)
        ODX_REQUIRE(seg_off[c] % 16 == 0 && seg_len[c] >= 0 && round_up(seg_off[c] + seg_len[c], 16) <= npad,
                    "odx_rls_gram_batched_f64: class %d: segment must start at a multiple of 16 and end, padded to one, inside the index array", c);
        sg.off[c] = seg_off[c];
        sg.len[c] = seg_len[c];
      }
      const int tiles = (int)(8 * ceil_div(ceil_div(D, RG_BM), 8) * ceil_div(D, RG_BN));
      hipLaunchKernelGGL(rls_gram_rows32_kernel, dim3((unsigned)tiles, 1, (unsigned)C), dim3(256), 0, s,
                         X, ldx, D, idx_pad, sg, G, ldg, g_stride, (const float*)nullptr, (int64_t)0, (double*)nullptr, (int64_t)0);
      ODX_CHECK_LAUNCH("rls_gram_rows");
      if (gram_only) return ODX_OK;
      const int64_t ldo = round_up(D1, 2);
      double* P = static_cast<double*>(workspace);                       // C x RX_CH x 5 x ldo partial sums, then O5
      double* O5 = P + (int64_t)C * RX_CH * 5 * ldo;
      hipLaunchKernelGGL(rls_xty_rows_kernel, dim3((unsigned)ceil_div(D1, 256), RX_CH, (unsigned)C), dim3(256), 0, s, X, ldx, D, idx_pad, sg, Yt,
                         ldy, P, ldo);
      ODX_CHECK_LAUNCH("rls_xty_rows");
      hipLaunchKernelGGL(rls_xty_reduce_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C, 5), dim3(256), 0, s, P, ldo, (int)D1, O5);
      ODX_CHECK_LAUNCH("rls_xty_reduce");
      hipLaunchKernelGGL(rls_fold_bias_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C), dim3(256), 0, s, O5, ldo, (int)D1, XtY, ldxy,
                         xy_stride, G, ldg, g_stride);
      ODX_CHECK_LAUNCH("rls_fold_bias");
      return ODX_OK;
    }
  }
  if (gram_only) {
    set_error("odx_rls_gram_batched_f64: the Gram-only call needs the rows form (D %% 8 == 0, ldx %% 4 == 0, X 16-byte aligned): odx_rls_rows_form");
    return ODX_ERR_UNSUPPORTED;
  }
  double* Xt = static_cast<double*>(workspace);
  dim3 grid((unsigned)ceil_div(ldt, 32), (unsigned)ceil_div(D1, 32));
  hipLaunchKernelGGL(rls_gather_transpose_all_kernel, grid, dim3(256), 0, s, X, ldx, D, idx_pad, npad, Xt, ldt);
  ODX_CHECK_LAUNCH("rls_gather_transpose_all");
  // The bias column makes the Gram (D + 1) x (D + 1): at D = 1024 a ninth tile row holding ONE row, a fifth of the
  // workgroups of the lower triangle.  The D x D part is one GEMM over whole tile rows; the bias row [X 1]' 1 rides along
  // with the four target rows (a 5 x (D + 1) product, one tile row either way) and is folded into G afterwards.
  double* Y5 = Xt + ldt * D1;
  double* O5 = Y5 + 5 * ldt;
  const int64_t ldo = round_up(D1, 2);
  ODX_CHECK_HIP(hipMemcpy2DAsync(Y5, (size_t)ldt * sizeof(double), Yt, (size_t)ldy * sizeof(double), (size_t)ldt * sizeof(double), 4,
                                 hipMemcpyDeviceToDevice, s));
  ODX_CHECK_HIP(hipMemcpyAsync(Y5 + 4 * ldt, Xt + (int64_t)D * ldt, (size_t)ldt * sizeof(double), hipMemcpyDeviceToDevice, s));
  GemmParams<double> g;
  g.A = Xt; g.lda = ldt; g.B = Xt; g.ldb = ldt; g.C = G; g.ldc = ldg;
  g.m = D; g.n = D; g.k = 0; g.alpha = 1.0; g.beta = 1.0; g.flags = ODX_GEMM_LOWER_ONLY;
  g.zbatches = C; g.zstrideC = g_stride; g.zk_on = 1;
  RlsSegs sg;
  for (int c = 0; c < ODX_MAX_ZBATCH; ++c) sg.off[c] = sg.len[c] = 0;
  for (int c = 0; c < C; ++c) {
    ODX_REQUIRE(seg_off[c] % 16 == 0 && seg_len[c] >= 0 && seg_off[c] + seg_len[c] <= npad,
                "odx_rls_gram_batched_f64: class %d: segment must start at a multiple of 16 inside the padded index array", c);
    g.zkoff[c] = seg_off[c];
    g.zklen[c] = seg_len[c];
    sg.off[c] = seg_off[c];
    sg.len[c] = seg_len[c];
  }
  ODX_PROPAGATE(launch_gemm_f64(g, s));
  hipLaunchKernelGGL(rls_xty_kernel, dim3((unsigned)ceil_div(D1, 4), (unsigned)C), dim3(256), 0, s, Xt, ldt, Y5, sg, (int)D1, O5, ldo);
  ODX_CHECK_LAUNCH("rls_xty");
  hipLaunchKernelGGL(rls_fold_bias_kernel, dim3((unsigned)ceil_div(D1, 256), (unsigned)C), dim3(256), 0, s, O5, ldo, (int)D1, XtY, ldxy,
                     xy_stride, G, ldg, g_stride);
  ODX_CHECK_LAUNCH("rls_fold_bias");
  return ODX_OK;
}

// workspace per class: Dinv | WT (D1*D1) | Li (D1 x ld) | Lit (D1 x ld) | z (4 x ld)
extern "C" int64_t odx_rls_solve_batched_workspace_bytes(int D, int C) {
  if (D <= 0 || C <= 0) return 0;
  const int64_t D1 = D + 1, ld = round_up(D1, 2);
  const int64_t per = ceil_div(D1, POTRF_NB) * POTRF_NB * POTRF_NB + round_up(D1 * D1, 2) + 2 * D1 * ld + 4 * ld;
  return per * C * (int64_t)sizeof(double);
}

// ---------------------------------------------------------------- the four solves of a class by block substitution
// W_q = (L L')^-1 b_q, q = 0..3, from the Cholesky factor L (lower, in G) and the inverses of its 128 x 128 diagonal blocks
// (Dinv, what potrf_f64 leaves): forward L y = b block row by block row, then back L' w = y — ONE workgroup per class, the
// vectors in LDS.  Replaces the explicit inverse of L (six merge levels of small GEMMs + two zeroed D1 x D1 matrices per
// class: 0.7 ms of a dependent chain at D = 1024) and eight triangular products per class (0.2 ms); L is read twice (once by
// rows, once by columns of its block columns: 16-lane-coalesced either way), 8.4 MB per class by one CU.
constexpr int RS_NT = 512, RS_NB = 128, RS_MAXB = 24;          // up to 24 x 128 = 3072 unknowns (D + 1 <= 3072)

__global__ __launch_bounds__(RS_NT) void rls_substitute_kernel(const double* __restrict__ Lall, int64_t ldl, int64_t l_stride,
                                                               const double* __restrict__ Dall, int64_t d_stride, int D1,
                                                               const double* __restrict__ Ball, int64_t ldb, int64_t b_stride,
                                                               double* __restrict__ Wall, int64_t ldw, int64_t w_stride) {
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (D1 + RS_NB - 1) / RS_NB, npad = nb * RS_NB;
  const double* L = Lall + (int64_t)c * l_stride;
  const double* Dv = Dall + (int64_t)c * d_stride;
  const double* B = Ball + (int64_t)c * b_stride;
  double* W = Wall + (int64_t)c * w_stride;
  double* y = rs_lds;                       // [npad][4]
  double* t = y + (int64_t)npad * 4;        // [128][4]
  double* part = t + RS_NB * 4;             // [8 waves][128][4]
  for (int e = tid; e < npad * 4; e += RS_NT) {
    const int j = e >> 2, q = e & 3;
    y[e] = j < D1 ? B[(int64_t)q * ldb + j] : 0.0;
  }
  __syncthreads();
  // ---- forward: y_k = Dinv_k (b_k - L[k, < k] y[< k])
  for (int k = 0; k < nb; ++k) {
    const int r0 = k * RS_NB;
    {                                       // wave w: rows 16 w .. 16 w + 15 of the block row, all sixteen at once — sixteen loads
      const int r = wave * 16;              // in flight per step of the walk along the rows (a row at a time: one memory latency
      double a[16][4] = {};                 // per row and step, 128 rows x 16 steps of them in a row)
      const double* lr[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) lr[u] = L + (int64_t)(r0 + r + u < D1 ? r0 + r + u : 0) * ldl;
      for (int j = lane; j < r0; j += 64) {
        double l[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) l[u] = lr[u][j];
        const double y0 = y[j * 4 + 0], y1 = y[j * 4 + 1], y2 = y[j * 4 + 2], y3 = y[j * 4 + 3];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          a[u][0] = fma(l[u], y0, a[u][0]); a[u][1] = fma(l[u], y1, a[u][1]); a[u][2] = fma(l[u], y2, a[u][2]); a[u][3] = fma(l[u], y3, a[u][3]);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool ok = r0 + r + u < D1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double v = ok ? a[u][q] : 0.0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == q) t[(r + u) * 4 + q] = y[(r0 + r + u) * 4 + q] - v;
        }
      }
    }
    __syncthreads();
    // y_k = Dinv_k t: the block's rows pass through LDS 32 at a time (every thread loads 8 of the chunk's 4096 doubles at once:
    // one memory latency per chunk; a row or a column per thread straight from memory is 128 latencies in a row)
    const double* Dk = Dv + (int64_t)k * RS_NB * RS_NB;
    double dreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) dreg[u] = Dk[u * RS_NT + tid];
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u * RS_NT + tid] = dreg[u];
      __syncthreads();
      if (m < 3) {                          // the next chunk's loads fly under this chunk's products
#pragma unroll
        for (int u = 0; u < 8; ++u) dreg[u] = Dk[(m + 1) * 4096 + u * RS_NT + tid];
      }
      {
        const int rl = tid >> 4, q = (tid >> 2) & 3, sp = tid & 3;       // 32 rows x 4 right-hand sides x 4 column phases
        const int r = m * 32 + rl;
        double a = 0.0;
        for (int j = sp; j <= r; j += 4) a = fma(part[rl * RS_NB + j], t[j * 4 + q], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        if (sp == 0) y[(r0 + r) * 4 + q] = a;
      }
      __syncthreads();
    }
  }
  // ---- back: w_k = Dinv_k' (y_k - L[> k, k]' w[> k]); w overwrites y block by block from the bottom
  for (int k = nb - 1; k >= 0; --k) {
    const int c0 = k * RS_NB;
    double p[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};      // this lane's two columns c0 + lane, c0 + lane + 64
    for (int row = c0 + RS_NB + wave; row < D1; row += 64) {          // eight rows of this wave per step: sixteen loads in flight
      double l0[8], l1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = row + 8 * u;
        const double* lr = L + (int64_t)(rr < D1 ? rr : row) * ldl + c0;
        l0[u] = lr[lane];
        l1[u] = lr[lane + 64];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rr = row + 8 * u;
        if (rr < D1) {
          const double w0 = y[rr * 4 + 0], w1 = y[rr * 4 + 1], w2 = y[rr * 4 + 2], w3 = y[rr * 4 + 3];
          p[0][0] = fma(l0[u], w0, p[0][0]); p[0][1] = fma(l0[u], w1, p[0][1]); p[0][2] = fma(l0[u], w2, p[0][2]); p[0][3] = fma(l0[u], w3, p[0][3]);
          p[1][0] = fma(l1[u], w0, p[1][0]); p[1][1] = fma(l1[u], w1, p[1][1]); p[1][2] = fma(l1[u], w2, p[1][2]); p[1][3] = fma(l1[u], w3, p[1][3]);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[(wave * RS_NB + lane + 64 * h) * 4 + q] = p[h][q];
    __syncthreads();
    {
      const int j = tid >> 2, q = tid & 3;            // 128 columns x 4 right-hand sides = 512 threads
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += part[(w * RS_NB + j) * 4 + q];
      t[j * 4 + q] = y[(c0 + j) * 4 + q] - sum;
    }
    __syncthreads();
    {
      // w_k = Dinv_k' t: thread (column j, right-hand side q) walks the rows of the block, 32 at a time through LDS
      const double* Dk = Dv + (int64_t)k * RS_NB * RS_NB;
      const int j = tid >> 2, q = tid & 3;
      double sum = 0.0;
      double dreg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dreg[u] = Dk[u * RS_NT + tid];
      for (int m = 0; m < 4; ++m) {
        __syncthreads();                                // (the partial sums / the previous chunk have been read)
#pragma unroll
        for (int u = 0; u < 8; ++u) part[u * RS_NT + tid] = dreg[u];
        __syncthreads();
        if (m < 3) {
#pragma unroll
          for (int u = 0; u < 8; ++u) dreg[u] = Dk[(m + 1) * 4096 + u * RS_NT + tid];
        }
        for (int rl = 0; rl < 32; ++rl) {
          const int r = m * 32 + rl;
          if (r >= j) sum = fma(part[rl * RS_NB + j], t[r * 4 + q], sum);
        }
      }
      y[(c0 + j) * 4 + q] = sum;
    }
    __syncthreads();
  }
  for (int e = tid; e < D1 * 4; e += RS_NT) {
    const int j = e >> 2, q = e & 3;
    W[(int64_t)q * ldw + j] = y[e];
  }
}

extern "C" int odx_rls_solve_batched_f64(double* G, int64_t ldg, int64_t g_stride, int D, int C, double lam, const double* XtY,
                                         int64_t ldxy, int64_t xy_stride, double* W, int64_t ldw, int64_t w_stride,
                                         int32_t* info, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(C >= 1 && C <= ODX_MAX_ZBATCH, "odx_rls_solve_batched_f64: 1..%d classes per call", ODX_MAX_ZBATCH);
  ODX_REQUIRE(G && XtY && W && info && workspace && D > 0, "odx_rls_solve_batched_f64: bad argument");
  const int64_t D1 = D + 1, ld = round_up(D1, 2);
  ODX_REQUIRE(ldg % 2 == 0 && ldg >= D1 && g_stride % 2 == 0 && g_stride >= D1 * ldg && aligned16(G),
              "odx_rls_solve_batched_f64: G 16-byte aligned, even ldg >= D + 1, even class stride");
  ODX_REQUIRE(ldxy % 2 == 0 && ldxy >= D1 && xy_stride % 2 == 0 && aligned16(XtY) && ldw % 2 == 0 && ldw >= D1 && w_stride % 2 == 0 &&
                  aligned16(W) && aligned16(workspace),
              "odx_rls_solve_batched_f64: XtY / W / workspace 16-byte aligned with even strides");
  if (workspace_bytes < odx_rls_solve_batched_workspace_bytes(D, C)) {
    set_error("odx_rls_solve_batched_f64: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  const int64_t dsz = ceil_div(D1, POTRF_NB) * POTRF_NB * POTRF_NB, wtsz = round_up(D1 * D1, 2), lsz = D1 * ld;
  double* Dinv = static_cast<double*>(workspace);
  double* WT = Dinv + (int64_t)C * dsz;
  double* Li = WT + (int64_t)C * wtsz;
  double* Lit = Li + (int64_t)C * lsz;
  double* z = Lit + (int64_t)C * lsz;
  ODX_CHECK_HIP(hipMemsetAsync(info, 0, (size_t)C * sizeof(int32_t), s));
  ODX_PROPAGATE(add_diag_f64(G, ldg, D1, lam, s, C, g_stride));
  ZBatch zb;
  zb.count = C; zb.strideA = g_stride; zb.strideD = dsz; zb.strideO = lsz; zb.strideW = wtsz;
  ODX_PROPAGATE(potrf_f64(G, ldg, D1, Dinv, info, s, zb));
  if (!lib_option(OPT_RLS_FORCE_INVERSE_SOLVE) && ceil_div(D1, RS_NB) <= RS_MAXB) {
    // block substitution with the factor (rls_substitute_kernel); wider systems (and the test hook rls_force_inverse_solve)
    // take the explicit inverse + products below
    const int64_t npad = ceil_div(D1, RS_NB) * RS_NB;
    const size_t lds = (size_t)(npad * 4 + RS_NB * 4 + 8 * RS_NB * 4) * sizeof(double);
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rls_substitute_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rls_substitute_kernel, dim3((unsigned)C), dim3(RS_NT), lds, s, G, ldg, g_stride, Dinv, dsz, (int)D1, XtY, ldxy, xy_stride, W,
                       ldw, w_stride);
    ODX_CHECK_LAUNCH("rls_substitute");
    return ODX_OK;
  }
  ODX_CHECK_HIP(hipMemsetAsync(Li, 0, (size_t)(2 * (int64_t)C * lsz) * sizeof(double), s));       // Li and Lit
  ODX_PROPAGATE(trtri_from_diag_f64(G, ldg, D1, Dinv, Li, Lit, ld, WT, s, zb));
  VecBatch vb;
  vb.B = C;
  for (int c = 0; c < C; ++c) vb.M[c] = (int)D1;
  for (int k = 0; k < 4; ++k) {
    ODX_PROPAGATE(trmv_batched_f64(Li, ld, lsz, 0, vb, XtY + k * ldxy, xy_stride, false, 0.0, nullptr, 0, z + k * ld, 4 * ld, s));
    ODX_PROPAGATE(trmv_batched_f64(Lit, ld, lsz, 1, vb, z + k * ld, 4 * ld, false, 0.0, nullptr, 0, W + k * ldw, w_stride, s));
  }
  return ODX_OK;
}
