// Issue rate of the vector instructions the compact-format pass decodes with, one wave per SIMD and four (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM)                                                                    \
  __global__ __launch_bounds__(256) void NAME(double* out, int iters) {                    \
    double d0 = threadIdx.x, d1 = 1.5, d2 = 2.5, d3 = 3.5;                                  \
    unsigned u0 = threadIdx.x * 77u, u1 = 123457u, u2 = 9999u, u3 = 31337u;                \
    float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f;                                           \
    for (int i = 0; i < iters; ++i) {                                                      \
      REP8(asm volatile(ASM : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(f0), \
                        "+v"(f1), "+v"(f2), "+v"(f3));)                                     \
    }                                                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3 + u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3; \
  }

BODY(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7")
BODY(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11")
BODY(k_cvt_f32_u32, "v_cvt_f32_u32 %8, %4\n v_cvt_f32_u32 %9, %5\n v_cvt_f32_u32 %10, %6\n v_cvt_f32_u32 %11, %7")
BODY(k_fma_f64, "v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %2, %3, %0, %2\n v_fma_f64 %3, %0, %1, %3")
BODY(k_add_f64, "v_add_f64 %0, %1, %0\n v_add_f64 %1, %2, %1\n v_add_f64 %2, %3, %2\n v_add_f64 %3, %0, %3")
BODY(k_perm, "v_perm_b32 %4, %5, %6, %7\n v_perm_b32 %5, %6, %7, %4\n v_perm_b32 %6, %7, %4, %5\n v_perm_b32 %7, %4, %5, %6")
BODY(k_fma_f32, "v_fma_f32 %8, %9, %10, %8\n v_fma_f32 %9, %10, %11, %9\n v_fma_f32 %10, %11, %8, %10\n v_fma_f32 %11, %8, %9, %11")
BODY(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %2, %3, %0, %2\n v_pk_fma_f32 %3, %0, %1, %3")
BODY(k_and_or, "v_and_or_b32 %4, %5, %6, %7\n v_and_or_b32 %5, %6, %7, %4\n v_and_or_b32 %6, %7, %4, %5\n v_and_or_b32 %7, %4, %5, %6")

template <typename Kern>
void run(const char* name, Kern k, int waves_per_simd, double* out) {
  const int iters = 20000, blocks = 256 * waves_per_simd;      // 256-thread blocks = one wave per SIMD each
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double insts = (double)iters * 32;                     // per wave
  // cycles per wave-instruction per SIMD at an assumed 2.4 GHz (the loop is pure VALU: the chip holds its clock)
  printf("%-16s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd,
         ms * 1e6 / (insts * waves_per_simd), ms * 1e6 / (insts * waves_per_simd) * 2.4);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  for (int w : {1, 4}) {
    run("v_cvt_f64_u32", k_cvt_f64_u32, w, out);
    run("v_cvt_f64_f32", k_cvt_f64_f32, w, out);
    run("v_cvt_f32_u32", k_cvt_f32_u32, w, out);
    run("v_fma_f64", k_fma_f64, w, out);
    run("v_add_f64", k_add_f64, w, out);
    run("v_perm_b32", k_perm, w, out);
    run("v_fma_f32", k_fma_f32, w, out);
    run("v_pk_fma_f32", k_pk_fma_f32, w, out);
    run("v_and_or_b32", k_and_or, w, out);
  }
  return 0;
}
