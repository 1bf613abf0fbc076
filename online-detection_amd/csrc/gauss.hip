// Gaussian-kernel blocks on the MFMA tile core (gfx950).
//   odx_gauss_knm_f32 : K_nM = exp(gamma * max(0, |x|^2 + |z|^2 - 2 x.z)), stored f32 (A3)
//   odx_gauss_mmv_f32 : out[:, c] = K(X, Z[range c]) V[range c, c], K never stored, f64 sums (A5/A9)
//   gauss_kmm_f64     : K_MM in f64 for the preconditioner (lower triangle; f64 GEMM + elementwise epilogue)
// The -2 X Z' contraction runs on v_mfma_f32_32x32x2_f32 (bit-exact f32 fmaf chain); the
// norm broadcast, clamp, scale and exp are fused into the accumulator epilogue, so the
// distance matrix never touches memory.
#include "gemm_core.h"
#include "odx_internal.h"

namespace odx {

// ---------------------------------------------------------------- row squared norms
template <typename T, typename V4>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const T* __restrict__ X, int64_t ldx, int64_t n, int D,
                                                         T* __restrict__ out) {
  constexpr int EPV = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const T* x = X + row * ldx;
  T s = T(0);
  const int nvec = D / EPV;
  for (int c = lane; c < nvec; c += 64) {
    const V4 v = *reinterpret_cast<const V4*>(x + c * EPV);
#pragma unroll
    for (int q = 0; q < EPV; ++q) s = fma(v[q], v[q], s);
  }
  for (int d = nvec * EPV + lane; d < D; d += 64) s = fma(x[d], x[d], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[row] = s;
}

// The same row norms (the same per-row fmaf chain and reduction, bit for bit) and, from the same read of the rows, the
// matrix's largest |entry| — what the f16 split scales by (odx_split_f16 otherwise makes a pass of its own over the matrix
// for it: a Minibootstrap round spent more time in those 180 absmax launches than in the 180 splits they serve).  maxbits
// holds the IEEE bits of max |x| (a non-negative float orders like its bits) and must be 0 on entry; one atomic per workgroup.
__global__ __launch_bounds__(256) void row_sqnorm_absmax_kernel(const float* __restrict__ X, int64_t ldx, int64_t n, int D,
                                                                float* __restrict__ out, unsigned int* __restrict__ maxbits) {
  __shared__ unsigned int wm[4];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned int m = 0u;
  if (row < n) {
    const float* x = X + row * ldx;
    float s = 0.f;
    const int nvec = D / 4;
    // four 16-byte loads of a lane in flight before the first is consumed (one load per trip: 1.4 TB/s at D = 1024); the
    // fmaf chain keeps the order c = lane, lane + 64, ...: the same bits as odx_row_sqnorm_f32
    for (int c0 = lane; c0 < nvec; c0 += 256) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 64 * u;
        v[u] = c < nvec ? *reinterpret_cast<const f32x4*>(x + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (c0 + 64 * u < nvec) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s = fmaf(v[u][q], v[u][q], s);
            m = max(m, __float_as_uint(v[u][q]) & 0x7fffffffu);
          }
        }
      }
    }
    for (int d = nvec * 4 + lane; d < D; d += 64) {
      s = fmaf(x[d], x[d], s);
      m = max(m, __float_as_uint(x[d]) & 0x7fffffffu);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_xor(s, off);
      m = max(m, (unsigned int)__shfl_xor((int)m, off));
    }
    if (lane == 0) out[row] = s;
  }
  if (lane == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    // one atomic per workgroup on ONE address serialises (75 000 workgroups at n = 3e5: 0.75 of the kernel's 0.88 ms): only a
    // workgroup that would raise the maximum it can see issues one (a stale read costs a redundant atomic, never a wrong result)
    if (m > __hip_atomic_load(maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxbits, m);
  }
}

// ---------------------------------------------------------------- K_nM (f32)
__global__ __launch_bounds__(GEMM_THREADS, 3) void gauss_knm_f32_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ xsq, int64_t n,
    const float* __restrict__ Z, int64_t ldz, const float* __restrict__ zsq, int64_t M, int D, float gamma,
    float* __restrict__ K, int64_t ldk) {
  __shared__ __attribute__((aligned(16))) char lds[GEMM_LDS_BYTES];
  // Tile order: every XCD gets a contiguous run of tiles (xcd_remap), and inside it tiles are
  // walked in bands of 8 row panels, column panel by column panel, so the ~100 tiles resident on
  // an XCD at one time form an 8 x 12 patch: each X panel is shared by 12 and each Z panel by 8
  // of them in that XCD's L2 (instead of 1-2 row panels x 79 column panels, where every Z slice
  // is fetched from the Infinity Cache once per tile).
  constexpr int64_t GR = 8;
  const int64_t tiles_n = (M + GEMM_BN - 1) / GEMM_BN;
  const int64_t wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t band = wg / (GR * tiles_n), within = wg % (GR * tiles_n);
  const int64_t i0 = (band * GR + within % GR) * GEMM_BM, j0 = (within / GR) * GEMM_BN;
  if (i0 >= n) return;

  f32x16 acc[2][2];
  gemm_zero_acc<float>(acc);
  gemm_mainloop<float>(acc, X, ldx, n, Z, ldz, M, i0, j0, 0, D, lds);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t mpad = (M + 3) & ~int64_t(3);  // the CG pass reads whole float4 chunks
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int64_t col = j0 + wc * 64 + gemm_acc_col<float>(tn, lane);
    const float zs = col < M ? zsq[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wr * 64 + gemm_acc_row<float>(tm, r, lane);
        if (row < n && col < mpad) {
          float v = 0.f;
          if (col < M) {
            float d2 = fmaf(-2.f, acc[tm][tn][r], xsq[row]) + zs;
            d2 = fmaxf(d2, 0.f);
            v = expf(d2 * gamma);
          }
          K[row * ldk + col] = v;
        }
      }
  }
}

// ---------------------------------------------------------------- fused scoring (f32 K, f64 sums)
__global__ __launch_bounds__(GEMM_THREADS, 2) void gauss_mmv_f32_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ xsq, int64_t n,
    const float* __restrict__ Z, int64_t ldz, const float* __restrict__ zsq, int D, float gamma,
    const double* __restrict__ V, int64_t ldv, const int32_t* __restrict__ ranges, float* __restrict__ out,
    int64_t ldo) {
  __shared__ __attribute__((aligned(16))) char lds[GEMM_LDS_BYTES];
  __shared__ double red[2][2][64];
  __shared__ float xs_s[GEMM_BM];
  const int c = blockIdx.y;
  const int64_t s0 = ranges[2 * c], s1 = ranges[2 * c + 1];
  const int64_t i0 = (int64_t)blockIdx.x * GEMM_BM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  if (threadIdx.x < GEMM_BM) xs_s[threadIdx.x] = (i0 + threadIdx.x < n) ? xsq[i0 + threadIdx.x] : 0.f;
  // Running row sums: the wave owns 64 rows x 64 columns of every K tile; lane l keeps the f64
  // total of row slot (l & 31) of its lane half, i.e. of row wr*64 + slot_row(l & 31, l >> 5).
  double tot = 0.0;

  for (int64_t j0 = s0; j0 < s1; j0 += GEMM_BN) {
    f32x16 acc[2][2];
    gemm_zero_acc<float>(acc);
    gemm_mainloop<float>(acc, X, ldx, n, Z + j0 * ldz, ldz, s1 - j0, i0, 0, 0, D, lds);
    float zs[2];
    double al[2];
    bool cv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int64_t col = j0 + wc * 64 + gemm_acc_col<float>(tn, lane);
      cv[tn] = col < s1;
      zs[tn] = cv[tn] ? zsq[col] : 0.f;
      al[tn] = cv[tn] ? V[col * ldv + c] : 0.0;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float xs = xs_s[wr * 64 + gemm_acc_row<float>(tm, r, lane)];
        double v = 0.0;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          float d2 = fmaf(-2.f, acc[tm][tn][r], xs) + zs[tn];
          d2 = fmaxf(d2, 0.f);
          const float kv = cv[tn] ? expf(d2 * gamma) : 0.f;
          v = fma((double)kv, al[tn], v);
        }
        // sum over the 32 lanes holding this row's columns (the two lane halves stay apart)
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((lane & 31) == tm * 16 + r) tot += v;
      }
  }
  {
    const int slot = lane & 31;
    red[wr][wc][gemm_acc_row<float>(slot >> 4, slot & 15, lane)] = tot;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int w = threadIdx.x >> 6, rr = threadIdx.x & 63;
    const int64_t row = i0 + w * 64 + rr;
    if (row < n) out[row * ldo + c] = (float)(red[w][0][rr] + red[w][1][rr]);
  }
}

// ---------------------------------------------------------------- K_MM (f64, lower triangle)
__global__ __launch_bounds__(256) void row_sqnorm_f64_batched_kernel(const double* __restrict__ Zd, int64_t ldz,
                                                                     int64_t z_stride, int64_t zsq_off, VecBatch vb, int D) {
  const int64_t b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= vb.M[b]) return;
  const double* x = Zd + b * z_stride + row * ldz;
  double s = 0.0;
  const int nvec = D / 2;
  for (int c = lane; c < nvec; c += 64) {
    const f64x2 v = *reinterpret_cast<const f64x2*>(x + c * 2);
    s = fma(v[0], v[0], s);
    s = fma(v[1], v[1], s);
  }
  for (int d = nvec * 2 + lane; d < D; d += 64) s = fma(x[d], x[d], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) (const_cast<double*>(Zd) + b * z_stride + zsq_off)[row] = s;
}

// K_MM = exp(gamma max(0, |z_i|^2 + |z_j|^2 - 2 z_i.z_j)) (+ diag) on the lower triangle, in two steps: the Gram block
// Z Z' by the f64 NT GEMM (128 x 64 tiles, three workgroups per CU: 60 TFLOP/s where the one-workgroup 128 x 128 tile
// kernel above reaches 20), then this elementwise pass over the lower triangle (0.3 ms at M = 1e4).
__global__ __launch_bounds__(256) void kmm_epilogue_kernel(double* __restrict__ Kmm, int64_t ldk, int64_t k_stride,
                                                           const double* __restrict__ zsq, int64_t zsq_stride, VecBatch vb,
                                                           double gamma) {
  const int64_t b = blockIdx.z, row = blockIdx.y;
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= vb.M[b] || col > row) return;
  const double* zs = zsq + b * zsq_stride;
  double* g = Kmm + b * k_stride + row * ldk + col;
  double d2 = fma(-2.0, *g, zs[row]) + zs[col];
  d2 = fmax(d2, 0.0);
  double v = exp(d2 * gamma);
  if (row == col) v += vb.scale[b];
  *g = v;
}

int gauss_kmm_f64_batched(const double* Zd, int64_t ldz, int64_t z_stride, int64_t zsq_off, const VecBatch& vb, int D,
                          double sigma, double* Kmm, int64_t ldk, int64_t k_stride, hipStream_t stream) {
  ODX_REQUIRE(ldz % 2 == 0 && z_stride % 2 == 0 && aligned16(Zd), "gauss_kmm_f64_batched: Zd must be 16-byte aligned, even strides");
  int mm = 0;
  for (int b = 0; b < vb.B; ++b) mm = vb.M[b] > mm ? vb.M[b] : mm;
  if (mm <= 0) return ODX_OK;
  hipLaunchKernelGGL(row_sqnorm_f64_batched_kernel, dim3((unsigned)ceil_div(mm, 4), (unsigned)vb.B), dim3(256), 0, stream, Zd,
                     ldz, z_stride, zsq_off, vb, D);
  ODX_CHECK_LAUNCH("row_sqnorm_f64_batched");
  GemmParams<double> g;                      // rows past a class's M are zero in Zd: their Gram entries come out zero
  g.A = Zd; g.lda = ldz; g.B = Zd; g.ldb = ldz; g.C = Kmm; g.ldc = ldk;
  g.m = mm; g.n = mm; g.k = D; g.alpha = 1.0; g.beta = 0.0; g.flags = ODX_GEMM_LOWER_ONLY;
  g.zbatches = vb.B; g.zstrideA = z_stride; g.zstrideB = z_stride; g.zstrideC = k_stride;
  ODX_PROPAGATE(launch_gemm_f64(g, stream));
  ODX_REQUIRE(mm < 65536, "gauss_kmm_f64: M must be below 65536");
  hipLaunchKernelGGL(kmm_epilogue_kernel, dim3((unsigned)ceil_div(mm, 256), (unsigned)mm, (unsigned)vb.B), dim3(256), 0, stream,
                     Kmm, ldk, k_stride, Zd + zsq_off, z_stride, vb, -0.5 / (sigma * sigma));
  ODX_CHECK_LAUNCH("kmm_epilogue");
  return ODX_OK;
}

// one class: Zd (M x ldz) with its squared norms written to zsq
int gauss_kmm_f64(const double* Zd, int64_t ldz, int64_t M, int D, double sigma, double diag_add, double* Kmm,
                  int64_t ldk, double* zsq, hipStream_t stream) {
  ODX_REQUIRE(ldz % 2 == 0 && aligned16(Zd), "gauss_kmm_f64: Zd must be 16-byte aligned with even ld");
  VecBatch vb;
  vb.B = 1;
  vb.M[0] = (int)M;
  vb.scale[0] = diag_add;
  return gauss_kmm_f64_batched(Zd, ldz, 0, zsq - Zd, vb, D, sigma, Kmm, ldk, 0, stream);
}

// ---------------------------------------------------------------- small blocks by direct differences
// K_ij = exp(gamma sum_d (x_id - z_jd)^2) with the differences formed directly and summed in f64: no ||x||^2 + ||z||^2 - 2 x.z
// cancellation, so the stored f32 entry is the exactly rounded one.  For blocks whose whole contraction is a few million
// multiply-adds (toy problems, unit fixtures: the matrix cores have nothing to win there) the f32-accurate MFMA forms leave
// 1e-6 relative in an entry — the norms' cancellation, which the reference's own f32 formula has four-fold —, and on tiny
// ill-conditioned fits that is what stands between alpha and the 1e-4 bar (a 30-centre problem on 8-dimensional mask pixels:
// 1.06e-4 with it, 3e-5 with exactly rounded entries).  One thread per entry.
__global__ __launch_bounds__(256) void gauss_knm_direct_kernel(const float* __restrict__ X, int64_t ldx, int64_t n,
                                                               const float* __restrict__ Z, int64_t ldz, int64_t M, int D, double gamma,
                                                               float* __restrict__ K, int64_t ldk) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = e / ldk, j = e - i * ldk;
  if (i >= n) return;
  if (j >= M) {
    K[i * ldk + j] = 0.f;                 // pad columns
    return;
  }
  const float* x = X + i * ldx;
  const float* z = Z + j * ldz;
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double t = (double)x[d] - (double)z[d];
    s = fma(t, t, s);
  }
  K[i * ldk + j] = (float)exp(gamma * s);
}

}  // namespace odx

using namespace odx;

extern "C" int odx_row_sqnorm_f32(const float* X, int64_t ldx, int64_t n, int D, float* out, odx_stream_t stream) {
  if (n <= 0) return ODX_OK;
  ODX_REQUIRE(X && out && D > 0, "odx_row_sqnorm_f32: null pointer or D <= 0");
  ODX_REQUIRE(ldx % 4 == 0 && ldx >= D && aligned16(X), "odx_row_sqnorm_f32: X must be 16-byte aligned, ldx %% 4 == 0, ldx >= D");
  hipLaunchKernelGGL((row_sqnorm_kernel<float, f32x4>), dim3((unsigned)ceil_div(n, 4)), dim3(256), 0,
                     as_stream(stream), X, ldx, n, D, out);
  ODX_CHECK_LAUNCH("odx_row_sqnorm_f32");
  return ODX_OK;
}

extern "C" int odx_row_sqnorm_absmax_f32(const float* X, int64_t ldx, int64_t n, int D, float* out, float* meta,
                                         odx_stream_t stream) {
  if (n <= 0) return ODX_OK;
  ODX_REQUIRE(X && out && meta && D > 0, "odx_row_sqnorm_absmax_f32: null pointer or D <= 0");
  ODX_REQUIRE(ldx % 4 == 0 && ldx >= D && aligned16(X), "odx_row_sqnorm_absmax_f32: X must be 16-byte aligned, ldx %% 4 == 0, ldx >= D");
  hipLaunchKernelGGL(row_sqnorm_absmax_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, as_stream(stream), X, ldx, n, D, out,
                     reinterpret_cast<unsigned int*>(meta + 1));
  ODX_CHECK_LAUNCH("odx_row_sqnorm_absmax_f32");
  return ODX_OK;
}

extern "C" int odx_gauss_knm_f32(const float* X, int64_t ldx, const float* xsq, int64_t n, const float* Z, int64_t ldz,
                                 const float* zsq, int64_t M, int D, double sigma, float* K, int64_t ldk,
                                 odx_stream_t stream) {
  if (n <= 0 || M <= 0) return ODX_OK;
  ODX_REQUIRE(X && xsq && Z && zsq && K && D > 0 && sigma > 0, "odx_gauss_knm_f32: bad argument");
  ODX_REQUIRE(ldx % 4 == 0 && ldz % 4 == 0 && ldx >= D && ldz >= D && aligned16(X) && aligned16(Z),
              "odx_gauss_knm_f32: X/Z must be 16-byte aligned with ld %% 4 == 0 and ld >= D");
  ODX_REQUIRE(ldk % 4 == 0 && ldk >= round_up(M, 4) && aligned16(K), "odx_gauss_knm_f32: K must be 16-byte aligned, ldk %% 4 == 0, ldk >= roundup(M, 4)");
  const int64_t tiles = round_up(ceil_div(n, GEMM_BM), 8) * ceil_div(M, GEMM_BN);
  ODX_REQUIRE(tiles < (1ll << 31), "odx_gauss_knm_f32: grid too large");
  hipLaunchKernelGGL(gauss_knm_f32_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, as_stream(stream), X, ldx,
                     xsq, n, Z, ldz, zsq, M, D, (float)(-0.5 / (sigma * sigma)), K, ldk);
  ODX_CHECK_LAUNCH("odx_gauss_knm_f32");
  return ODX_OK;
}

extern "C" int odx_gauss_knm_direct_f32(const float* X, int64_t ldx, int64_t n, const float* Z, int64_t ldz, int64_t M, int D,
                                        double sigma, float* K, int64_t ldk, odx_stream_t stream) {
  if (n <= 0 || M <= 0) return ODX_OK;
  ODX_REQUIRE(X && Z && K && D > 0 && sigma > 0 && ldx >= D && ldz >= D, "odx_gauss_knm_direct_f32: bad argument");
  ODX_REQUIRE(ldk % 4 == 0 && ldk >= round_up(M, 4) && aligned16(K), "odx_gauss_knm_direct_f32: K must be 16-byte aligned, ldk %% 4 == 0, ldk >= roundup(M, 4)");
  const int64_t blocks = ceil_div(n * ldk, 256);
  ODX_REQUIRE(blocks < (1ll << 31), "odx_gauss_knm_direct_f32: grid too large");
  hipLaunchKernelGGL(gauss_knm_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), X, ldx, n, Z, ldz, M, D,
                     -0.5 / (sigma * sigma), K, ldk);
  ODX_CHECK_LAUNCH("odx_gauss_knm_direct_f32");
  return ODX_OK;
}

extern "C" int odx_gauss_mmv_f32(const float* X, int64_t ldx, const float* xsq, int64_t n, const float* Z, int64_t ldz,
                                 const float* zsq, int D, double sigma, const double* V, int64_t ldv,
                                 const int32_t* ranges, int C, float* out, int64_t ldo, odx_stream_t stream) {
  if (n <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(X && xsq && Z && zsq && V && ranges && out && D > 0 && sigma > 0 && ldv >= C, "odx_gauss_mmv_f32: bad argument");
  ODX_REQUIRE(ldx % 4 == 0 && ldz % 4 == 0 && ldx >= D && ldz >= D && aligned16(X) && aligned16(Z),
              "odx_gauss_mmv_f32: X/Z must be 16-byte aligned with ld %% 4 == 0 and ld >= D");
  ODX_REQUIRE(ldo >= C && C < 65536, "odx_gauss_mmv_f32: ldo < C or too many classes");
  const int64_t rb = ceil_div(n, GEMM_BM);
  ODX_REQUIRE(rb < (1ll << 31), "odx_gauss_mmv_f32: grid too large");
  hipLaunchKernelGGL(gauss_mmv_f32_kernel, dim3((unsigned)rb, (unsigned)C), dim3(GEMM_THREADS), 0, as_stream(stream),
                     X, ldx, xsq, n, Z, ldz, zsq, D, (float)(-0.5 / (sigma * sigma)), V, ldv, ranges, out, ldo);
  ODX_CHECK_LAUNCH("odx_gauss_mmv_f32");
  return ODX_OK;
}
