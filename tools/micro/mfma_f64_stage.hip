// The Gram kernel's k-tile around a register-only stream of v_mfma_f64_16x16x4_f64 (tools/micro/mfma_f64_mix.hip holds 64 TFLOP/s
// with the k-step's LDS reads and conversions and a barrier per tile): per 64 matrix instructions a thread also (G) issues 12
// 8-byte global loads whose values it (S) stores to LDS one tile later between two barriers, as the kernel stages its operands.
// Three workgroups of 256 per CU.  Which of the two takes the stream towards the kernel's 49?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_stage.hip -o tools/micro/mfma_f64_stage.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool G, bool S, bool D2, bool DB>
__global__ __launch_bounds__(256) void stage(const float* __restrict__ X, int64_t ldx, int64_t rows, double* out, int tiles) {
  __shared__ float lds[(DB ? 2 : 1) * 32 * 208];              // DB: two tile buffers, ONE barrier per tile (stores go to the other one)
  const int t = threadIdx.x, krow = t >> 4, seg = t & 15;
  for (int e = t; e < 32 * 208; e += 256) lds[e] = 0.001f * (e & 63);
  __syncthreads();
  f64x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f64x4{0.0, 0.0, 0.0, 0.0};
  f32x2 r[12], r2[12];                                          // D2: the loads of a tile are stored TWO tiles later (two sets)
  for (int q = 0; q < 12; ++q) r[q] = r2[q] = f32x2{1.f, 2.f};
  int64_t row = (blockIdx.x * 977 + krow * 131) % rows;
  for (int kt = 0; kt < tiles; ++kt) {
    if (D2) {
#pragma unroll
      for (int q = 0; q < 12; ++q) { const f32x2 tmp = r[q]; r[q] = r2[q]; r2[q] = tmp; }      // (r: the older set; unrolled by two in a kernel)
    }
    const int cur = DB ? (kt & 1) * 32 * 208 : 0, nxt = DB ? ((kt + 1) & 1) * 32 * 208 : 0;
    if (S && !DB) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 12; ++q) *reinterpret_cast<f32x2*>(lds + ((krow + 16 * (q / 6)) * 208 + (q % 6) * 32 + seg * 2)) = r[q];
      __syncthreads();
    }
    if (S && DB) {
#pragma unroll
      for (int q = 0; q < 12; ++q) *reinterpret_cast<f32x2*>(lds + nxt + ((krow + 16 * (q / 6)) * 208 + (q % 6) * 32 + seg * 2)) = r[q];
    }
    if (G) {
      const float* x0 = X + row * ldx + seg * 2;
      const float* x1 = X + ((row + 7919) % rows) * ldx + seg * 2;
#pragma unroll
      for (int q = 0; q < 6; ++q) r[q] = *reinterpret_cast<const f32x2*>(x0 + q * 32);
#pragma unroll
      for (int q = 0; q < 6; ++q) r[6 + q] = *reinterpret_cast<const f32x2*>(x1 + q * 32);
      row = (row + 30011) % rows;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float f[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) f[q] = lds[cur + (ks * 4 + (t >> 4 & 3)) * 208 + q * 16 + (t & 15)];
      double d[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) d[q] = (double)f[q];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[k & 3], d[4 + (k & 1)], acc[k], 0, 0, 0);
    }
    if (S && DB) __syncthreads();
  }
  double s = 0.0;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][3];
  out[blockIdx.x * 256 + t] = s + r[0][0] + r[11][1];
}

template <bool G, bool S, bool D2, bool DB>
static void run(int cus, const float* X, int64_t rows, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int tiles = 2000, w = 3;
  hipLaunchKernelGGL((stage<G, S, D2, DB>), dim3(cus * w), dim3(256), 0, 0, X, (int64_t)1024, rows, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((stage<G, S, D2, DB>), dim3(cus * w), dim3(256), 0, 0, X, (int64_t)1024, rows, out, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("global loads %s, LDS stores between barriers %s, stored %s later%s: %.1f TFLOP/s\n", G ? "yes" : "no", S ? "yes" : "no", D2 ? "two tiles" : "one tile", DB ? ", two LDS buffers and one barrier per tile" : "",
         (double)w * tiles * 64 * 4.0 * cus * 2048.0 / ms / 1e9);
}

int main() {
  const int64_t rows = 300000;
  float* X;
  hipMalloc(&X, rows * 1024 * sizeof(float));
  hipMemset(X, 0, rows * 1024 * sizeof(float));
  double* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(double));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<false, false, false, false>(cus, X, rows, out);
  run<false, true, false, false>(cus, X, rows, out);
  run<true, false, false, false>(cus, X, rows, out);
  run<true, true, false, false>(cus, X, rows, out);
  run<true, true, true, false>(cus, X, rows, out);
  run<true, true, false, true>(cus, X, rows, out);
  run<false, true, false, true>(cus, X, rows, out);
  return 0;
}
