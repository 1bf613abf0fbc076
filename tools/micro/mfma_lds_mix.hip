// The f16 matrix instructions with LDS fragment reads beside them: per loop trip R ds_read_b128 (1 KB per wave each, conflict-free)
// feed the operands of M matrix instructions — the mix of the 256 x 256 tile core is 24 reads per 96 instructions of 16 x 16 x 32
// (0.25 per instruction) or per 48 of 32 x 32 x 16 (0.5) and k-step of 32.  Two waves per SIMD (the core's occupancy), every CU.
// Says how much of the register-only rate (mfma_f16_rate.hip) survives the operand traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds_mix.hip -o tools/micro/mfma_lds_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int M, bool BIG>
__global__ __launch_bounds__(512) void mix(float* out, int iters) {
  __shared__ f16x8 buf[512 * 4];                               // 32 KB: every lane its own 16 bytes, four slots
  const int t = threadIdx.x;
  for (int s = 0; s < 4; ++s)
    for (int k = 0; k < 8; ++k) buf[s * 512 + t][k] = (_Float16)(0.01f * (t & 7) + 0.001f * k);
  __syncthreads();
  f32x4 a4[8];
  f32x16 a16[4];
  for (int k = 0; k < 8; ++k) a4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 16; ++e) a16[k][e] = 0.f;
  f16x8 frag[R > 0 ? R : 1];
  for (int r = 0; r < (R > 0 ? R : 1); ++r) frag[r] = buf[r % 4 * 512 + t];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < R; ++r) frag[r] = buf[((r + i) & 3) * 512 + t];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const f16x8 x = frag[m % (R > 0 ? R : 1)], y = frag[(m + 1) % (R > 0 ? R : 1)];
      if (BIG) a16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a16[m & 3], 0, 0, 0);
      else a4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a4[m & 7], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += a4[k][0] + a4[k][3];
  for (int k = 0; k < 4; ++k) s += a16[k][0] + a16[k][15];
  out[blockIdx.x * 512 + t] = s;
}

template <int R, int M, bool BIG>
static void run(int cus, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((mix<R, M, BIG>), dim3(cus), dim3(512), 0, 0, out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix<R, M, BIG>), dim3(cus), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)iters * M * 8.0 * cus * (BIG ? 32768.0 : 16384.0);     // 8 waves per CU; flop per instruction
  printf("%s, %2d reads per %2d instructions (%.2f): %.0f TFLOP/s, LDS %.1f TB/s\n", BIG ? "32x32x16" : "16x16x32", R, M, (double)R / M,
         flop / ms / 1e9, (double)iters * R * 8.0 * cus * 1024.0 / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 512 * sizeof(float));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<0, 24, false>(cus, out);
  run<6, 24, false>(cus, out);      // the core's mix on 16 x 16 x 32
  run<12, 24, false>(cus, out);
  run<0, 12, true>(cus, out);
  run<3, 12, true>(cus, out);       // half the core's operand traffic per flop
  run<6, 12, true>(cus, out);       // the core's mix on 32 x 32 x 16
  run<12, 12, true>(cus, out);
  return 0;
}
