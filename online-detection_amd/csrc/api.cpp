// Error text, version and device queries of the libodx C ABI (include/odx.h).
#include "odx_common.h"
#include <string.h>

namespace odx {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace odx

extern "C" const char* odx_last_error_string(void) { return odx::g_error; }

// ---------------------------------------------------------------- options
// One table instead of environment variables read inside launch paths (round-5 review, item 7).  The defaults ARE the benched
// configuration; tests/test_abi.py pins them.
#include "odx_internal.h"
namespace odx {
struct OptSpec {
  const char* name;
  int def, lo, hi;      // default, inclusive range (h2_tile: the three legal values are checked separately)
};
static const OptSpec g_opt_spec[OPT_COUNT] = {
    {"h2_tile", 0, 0, 256}, {"precond", 0, 0, 2}, {"chain_helpers", -1, -1, 1}, {"rls_force_nt_gram", 0, 0, 1},
    {"rls_force_inverse_solve", 0, 0, 1}};
static int g_opt[OPT_COUNT] = {0, 0, -1, 0, 0};
int lib_option(int which) { return g_opt[which]; }
static int opt_index(const char* name) {
  if (name)
    for (int i = 0; i < OPT_COUNT; ++i)
      if (strcmp(name, g_opt_spec[i].name) == 0) return i;
  return -1;
}
}  // namespace odx

extern "C" int odx_set_option(const char* name, int value) {
  const int i = odx::opt_index(name);
  ODX_REQUIRE(i >= 0, "odx_set_option: unknown option '%s'", name ? name : "(null)");
  const odx::OptSpec& sp = odx::g_opt_spec[i];
  ODX_REQUIRE(value >= sp.lo && value <= sp.hi && (i != odx::OPT_H2_TILE || value == 0 || value == 128 || value == 256),
              "odx_set_option: %s = %d out of range", name, value);
  odx::g_opt[i] = value;
  return ODX_OK;
}

extern "C" int odx_get_option(const char* name, int* value) {
  const int i = odx::opt_index(name);
  ODX_REQUIRE(i >= 0 && value, "odx_get_option: unknown option '%s'", name ? name : "(null)");
  *value = odx::g_opt[i];
  return ODX_OK;
}

extern "C" int odx_option_default(const char* name, int* value) {
  const int i = odx::opt_index(name);
  ODX_REQUIRE(i >= 0 && value, "odx_option_default: unknown option '%s'", name ? name : "(null)");
  *value = odx::g_opt_spec[i].def;
  return ODX_OK;
}

extern "C" int odx_version(void) { return 100; }

extern "C" int odx_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    odx::set_error("odx_device_cus: no HIP device");
    return ODX_ERR_HIP;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    odx::set_error("odx_device_cus: attribute query failed");
    return ODX_ERR_HIP;
  }
  return cus;
}

// ---------------------------------------------------------------- CU-partitioned streams
// A stream whose kernels run only on the compute units of a mask (hipExtStreamCreateWithCUMask): the HBM-bound CG passes
// and the MFMA-bound Gaussian kernels of DIFFERENT classes then run beside each other on disjoint parts of the chip
// instead of one after the other on all of it (odx/job.py).  `words` 32-bit mask words, bit i = logical CU i of the device.
extern "C" int odx_stream_create_cu_mask(const uint32_t* mask, int words, odx_stream_t* stream) {
  ODX_REQUIRE(mask && words > 0 && stream, "odx_stream_create_cu_mask: bad argument");
  hipStream_t s = nullptr;
  ODX_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream = reinterpret_cast<odx_stream_t>(s);
  return ODX_OK;
}

extern "C" int odx_stream_destroy(odx_stream_t stream) {
  if (stream) ODX_CHECK_HIP(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
  return ODX_OK;
}

namespace odx {
// where a workgroup runs: out[3 b] = XCC_ID, out[3 b + 1] = HW_ID (CU_ID bits 8..11, SH_ID bit 12, SE_ID bits 13..15),
// out[3 b + 2] = order of arrival.  Every workgroup holds its CU for `spin` clock ticks so that a grid of as many
// workgroups as CUs lands on all of them.
__global__ __launch_bounds__(64) void placement_kernel(int32_t* __restrict__ out, int32_t* __restrict__ counter, int spin) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID, 4 bits
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID
    out[3 * blockIdx.x] = (int32_t)xcc;
    out[3 * blockIdx.x + 1] = (int32_t)hw;
    out[3 * blockIdx.x + 2] = atomicAdd(counter, 1);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  }
}
}  // namespace odx

// diagnostic: the placement of `blocks` one-wave workgroups launched on `stream` (out: 3 * blocks + 1 int32, zeroed here)
extern "C" int odx_debug_placement(int32_t* out, int blocks, int spin, odx_stream_t stream) {
  ODX_REQUIRE(out && blocks > 0 && spin >= 0, "odx_debug_placement: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)(3 * blocks + 1) * sizeof(int32_t), s));
  hipLaunchKernelGGL(odx::placement_kernel, dim3((unsigned)blocks), dim3(64), 0, s, out, out + 3 * blocks, spin);
  ODX_CHECK_LAUNCH("odx_debug_placement");
  return ODX_OK;
}
