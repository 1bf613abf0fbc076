// What the f64 matrix pipe sustains (gfx950): v_mfma_f64_16x16x4_f64 from registers only — eight independent accumulators per wave,
// the same operands every time — with 1, 2, 3, 4 waves per SIMD on every CU.  TFLOP/s against the 78.6 of the data sheet, and the
// cycles per instruction per SIMD it implies at the clock the run held.  The RLS Gram kernel (rls_gram_rows32_kernel) issues
// exactly this instruction mix plus LDS reads: its rate is judged against THIS number (32 instructions per loop trip: with eight,
// the loop's own instructions and its taken branch cost a quarter).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_rate.hip -o gpurun_out/mfma_f64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_f64(double* out, int iters) {
  f64x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0.0;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the 4 x 4 x 4 form (four blocks: 512 flop per instruction)
__global__ __launch_bounds__(256) void k_mfma_f64_4x4(double* out, int iters) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  for (int i = 0; i < iters; i += 4) {            // (32 instructions per trip: the loop's own instructions must not count)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0.0;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(double));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int w = 1; w <= 4; ++w) {
    // 256 threads = 4 waves = one per SIMD: w workgroups per CU give w waves per SIMD
    hipLaunchKernelGGL(k_mfma_f64, dim3(cus * w), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma_f64, dim3(cus * w), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)w * iters * 8;                       // per SIMD
    const double flop = insts_per_simd * 4.0 * cus * 2048.0;                   // 16 x 16 x 4 x 2 per instruction
    printf("%d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s, %.1f ns per instruction per SIMD (64 cycles at 2.4 GHz = 26.7 ns)\n", w, ms,
           flop / ms / 1e9, ms * 1e6 / insts_per_simd);
  }
  for (int w = 1; w <= 4; w += 3) {
    hipLaunchKernelGGL(k_mfma_f64_4x4, dim3(cus * w), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma_f64_4x4, dim3(cus * w), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)w * iters * 8;
    printf("4 x 4 x 4 (4 blocks), %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s, %.1f ns per instruction per SIMD\n", w, ms,
           insts_per_simd * 4.0 * cus * 512.0 / ms / 1e9, ms * 1e6 / insts_per_simd);
  }
  return 0;
}
