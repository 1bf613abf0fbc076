// What the f16 matrix pipe sustains from registers only (gfx950): v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16, eight / four
// independent accumulators per wave, 1 .. 4 waves per SIMD on every CU — against the 2.5 PFLOP/s dense peak the roofline of the
// Gaussian kernels is priced with.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16_rate.hip -o tools/micro/mfma_f16_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k16(float* out, int iters, int mode) {
  f32x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) {           // mode 0: zeros; 1: a regular pattern; 2: pseudo-random values in (-1, 1) (every mantissa bit busy)
    unsigned h = (threadIdx.x * 8 + k) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float r1 = (float)(h & 0xffff) / 32768.f - 1.f, r2 = (float)(h >> 16) / 32768.f - 1.f;
    a[k] = (_Float16)(mode == 0 ? 0.f : (mode == 1 ? 1.0f + threadIdx.x * 1e-3f + k : r1));
    b[k] = (_Float16)(mode == 0 ? 0.f : (mode == 1 ? 0.5f - threadIdx.x * 1e-3f : r2));
  }
  for (int i = 0; i < iters; i += 4) {            // (32 instructions per trip: with eight, the loop's own instructions cost a fifth)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters, int mode) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) {           // mode 0: zeros; 1: a regular pattern; 2: pseudo-random values in (-1, 1) (every mantissa bit busy)
    unsigned h = (threadIdx.x * 8 + k) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float r1 = (float)(h & 0xffff) / 32768.f - 1.f, r2 = (float)(h >> 16) / 32768.f - 1.f;
    a[k] = (_Float16)(mode == 0 ? 0.f : (mode == 1 ? 1.0f + threadIdx.x * 1e-3f + k : r1));
    b[k] = (_Float16)(mode == 0 ? 0.f : (mode == 1 ? 0.5f - threadIdx.x * 1e-3f : r2));
  }
  for (int i = 0; i < iters; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int per_iter, double flop_per_inst, int cus, float* out, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  for (int w = 2; w <= 4; w += 2) {
    hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, out, 100, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)w * iters * per_iter;
    printf("%s, operands %s, %d wave(s) per SIMD: %.2f ms, %.0f TFLOP/s, %.2f ns per instruction per SIMD\n", name, mode == 0 ? "zero" : (mode == 1 ? "regular" : "random"), w, ms,
           insts_per_simd * 4.0 * cus * flop_per_inst / ms / 1e9, ms * 1e6 / insts_per_simd);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  for (int mode = 0; mode < 3; ++mode) {
    run("v_mfma_f32_16x16x32_f16", k16, 8, 16.0 * 16 * 32 * 2, p.multiProcessorCount, out, mode);
    run("v_mfma_f32_32x32x16_f16", k32, 4, 32.0 * 32 * 16 * 2, p.multiProcessorCount, out, mode);
  }
  return 0;
}
