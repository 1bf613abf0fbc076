// What the f16 matrix pipe sustains from registers only (gfx950): v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16, eight / four
// independent accumulators per wave, 1 .. 4 waves per SIMD on every CU — against the 2.5 PFLOP/s dense peak the roofline of the
// Gaussian kernels is priced with.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16_rate.hip -o tools/micro/mfma_f16_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1.0f + threadIdx.x * 1e-3f + k); b[k] = (_Float16)(0.5f - threadIdx.x * 1e-3f); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1.0f + threadIdx.x * 1e-3f + k); b[k] = (_Float16)(0.5f - threadIdx.x * 1e-3f); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int per_iter, double flop_per_inst, int cus, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  for (int w = 1; w <= 4; ++w) {
    hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)w * iters * per_iter;
    printf("%s, %d wave(s) per SIMD: %.2f ms, %.0f TFLOP/s, %.2f ns per instruction per SIMD\n", name, w, ms,
           insts_per_simd * 4.0 * cus * flop_per_inst / ms / 1e9, ms * 1e6 / insts_per_simd);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  run("v_mfma_f32_16x16x32_f16", k16, 8, 16.0 * 16 * 32 * 2, p.multiProcessorCount, out);
  run("v_mfma_f32_32x32x16_f16", k32, 4, 32.0 * 32 * 16 * 2, p.multiProcessorCount, out);
  return 0;
}
