// v_mfma_f64_16x16x4_f64 with the Gram kernel's other per-k-step instructions beside it: per eight matrix instructions R
// ds_read_b32 of operand floats and C v_cvt_f64_f32 turning them into the operands (the Gram's mix: 6 and 6), three waves per
// SIMD, every CU, optionally a workgroup barrier every 32 / 64 of them (the Gram's k-tile) — which ingredient takes the stream from
// the 66 TFLOP/s it sustains alone to the kernel's 49?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_mix.hip -o tools/micro/mfma_f64_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int R, int C, int BAR>
__global__ __launch_bounds__(256) void mix(double* out, int iters) {
  __shared__ float buf[256 * 8];
  const int t = threadIdx.x;
  for (int s = 0; s < 8; ++s) buf[s * 256 + t] = 0.001f * (t & 31) + 0.01f * s;
  __syncthreads();
  f64x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f64x4{0.0, 0.0, 0.0, 0.0};
  float f[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  double d[6] = {1.0, 0.5, 0.25, 2.0, 3.0, 1.5};
  for (int i = 0; i < iters; i += 4) {
    if (BAR == 1 || (BAR == 2 && (i & 4))) __syncthreads();       // a workgroup barrier every 32 / 64 matrix instructions
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int r = 0; r < R; ++r) f[r] = buf[((r + i + u) & 7) * 256 + t];
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] = (double)f[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[k & 3], d[4 + (k & 1)], acc[k], 0, 0, 0);
    }
  }
  double s = 0.0;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][3];
  out[blockIdx.x * 256 + t] = s + f[0] + f[5];
}

template <int R, int C, int BAR>
static void run(int cus, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, w = 3;
  hipLaunchKernelGGL((mix<R, C, BAR>), dim3(cus * w), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix<R, C, BAR>), dim3(cus * w), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%d LDS reads, %d conversions per 8 matrix instructions, barrier %s: %.1f TFLOP/s\n", R, C, BAR == 0 ? "never" : (BAR == 1 ? "every 32" : "every 64"), (double)w * iters * 8 * 4.0 * cus * 2048.0 / ms / 1e9);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(double));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<0, 0, 0>(cus, out);
  run<6, 0, 0>(cus, out);
  run<0, 6, 0>(cus, out);
  run<6, 6, 0>(cus, out);
  run<6, 6, 2>(cus, out);
  run<6, 6, 1>(cus, out);
  run<0, 0, 2>(cus, out);
  return 0;
}
