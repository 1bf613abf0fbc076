// Operand / result layout of v_mfma_f64_4x4x4_4b_f64 by experiment: one lane of A and one lane of B set to 1, everything else 0 —
// the lanes of D that come out non-zero say which (block, row, k) an A lane holds, which (block, k, column) a B lane, and where
// D[block][row][column] lands.  Prints, for every A lane, the B lanes it meets and the D lane of each meeting.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_4x4_layout.hip -o tools/micro/mfma_f64_4x4_layout.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int* hit) {                  // hit[la * 64 + lb] = D lane that is non-zero (or -1, or -2 for several)
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) hit[la * 64 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __ffsll((long long)m) - 1 : -2);
    }
}

int main() {
  int* d;
  hipMalloc(&d, 4096 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[4096];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d meets B lanes:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] != -1) printf(" %d->D%d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
