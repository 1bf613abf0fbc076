// Feature-forward helpers of the detection heads: RoIAlign forward and NMS, the two
// ops the reference reaches through maskrcnn_benchmark's CUDA extension
// (mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py:21-25,47 -> Pooler ->
//  ROIAlign; mrcnn_modified/modeling/rpn/inference.py:116-121 and
//  src/modules/accuracy-evaluator/OnlineDetectionPostProcessor.py:55-57 -> boxlist_nms).
// Semantics restated from the published operators (Mask R-CNN RoIAlign in its legacy
// "aligned = False" form with adaptive sampling when sampling_ratio == 0; greedy NMS with the
// +1 pixel-area convention); parity unpinned by the reference (the extension is not vendored).
#include <math.h>

#include "odx_common.h"

namespace odx {

constexpr int ODX_MAX_FPN_LEVELS = 4;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Buffers of these calls are cleared by a KERNEL, not by hipMemsetAsync: the calls are captured into HIP graphs (the group
// forward), and a graph's memset nodes were not reliably ordered against the kernel nodes around them on this runtime (see
// meta_zero_kernel in gauss_h2.hip; the all-zero RoI features of round 5's group-graph probe).  Any address, any byte count.
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ p, size_t nbytes) {
  const size_t stride = (size_t)gridDim.x * 256 * 16;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < nbytes; i += stride) {
    if (i + 16 <= nbytes && ((reinterpret_cast<uintptr_t>(p) + i) & 15u) == 0) {
      *reinterpret_cast<uint4*>(p + i) = uint4{0u, 0u, 0u, 0u};
    } else {
      for (size_t j = i; j < nbytes && j < i + 16; ++j) p[j] = 0;
    }
  }
}

static int zero_bytes(void* p, size_t nbytes, hipStream_t s) {
  if (nbytes == 0) return ODX_OK;
  const size_t blocks = (nbytes + 4095) / 4096;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, static_cast<unsigned char*>(p), nbytes);
  ODX_CHECK_LAUNCH("zero_bytes");
  return ODX_OK;
}

// ---------------------------------------------------------------- RoIAlign forward
// feat (N, C, H, W) f32, rois (R, 5) = (batch index, x1, y1, x2, y2), out (R, C, PH, PW).
// One 256-thread workgroup per (roi, chunk of CCH channels): thread t owns output bin t of the
// PH x PW grid (PH * PW <= 256); the sample positions and bilinear weights of a bin are the same
// for every channel, so they are formed once per sample and reused across the chunk's channels.
constexpr int ROI_CCH = 16;

// One sample's bilinear value with the fusing of multiplies into adds written out (what the compiler chose for the sum
// w1 p1 + w2 p2 + w3 p3 + w4 p4 differed between the kernels below; stated once, every layout of the map gives the same bits).
__device__ __forceinline__ float roi_bilinear(float w1, float p1, float w2, float p2, float w3, float p3, float w4, float p4) {
  return fmaf(w4, p4, fmaf(w3, p3, fmaf(w2, p2, w1 * p1)));
}

// step > 0 ("rows" form): only the bins (ph, pw) with ph % step == 0 and pw % step == 0 are formed — the positions a
// stride-`step` 1 x 1 convolution reads, a quarter of the 14 x 14 grid for the conv5 head — and written as rows of an
// (R * OH * OW, C) matrix (NHWC), the layout the head's GEMMs consume; step == 0: the full (R, C, PH, PW) grid.
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat, int N, int C, int H, int W,
                                                            const float* __restrict__ rois, int R, float scale, int PH,
                                                            int PW, int sampling_ratio, float* __restrict__ out, int step) {
  const int r = blockIdx.x;
  const int c0 = blockIdx.y * ROI_CCH;
  const int t = threadIdx.x;
  int ph, pw, OW = 0, OH = 0;
  if (step > 0) {
    OH = (PH + step - 1) / step;
    OW = (PW + step - 1) / step;
    if (t >= OH * OW) return;
    ph = (t / OW) * step;
    pw = (t % OW) * step;
  } else {
    if (t >= PH * PW) return;
    ph = t / PW;
    pw = t % PW;
  }
  const float* roi = rois + (int64_t)r * 5;
  const int b = (int)roi[0];
  const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
  const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);   // malformed RoIs become 1 x 1
  const float bw = rw / (float)PW, bh = rh / (float)PH;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  const float count = (float)(gh * gw);
  float acc[ROI_CCH];
#pragma unroll
  for (int k = 0; k < ROI_CCH; ++k) acc[k] = 0.f;
  const int nch = min(ROI_CCH, C - c0);
  const float* base = feat + ((int64_t)b * C + c0) * H * W;
  if (b >= 0 && b < N) {
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
        if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) continue;   // sample outside: contributes 0
        float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
        int yl = (int)yy, xl = (int)xx, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else { yh = yl + 1; }
        if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else { xh = xl + 1; }
        const float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int i1 = yl * W + xl, i2 = yl * W + xh, i3 = yh * W + xl, i4 = yh * W + xh;
#pragma unroll
        for (int k = 0; k < ROI_CCH; ++k) {
          if (k < nch) {
            const float* p = base + (int64_t)k * H * W;
            acc[k] += roi_bilinear(w1, p[i1], w2, p[i2], w3, p[i3], w4, p[i4]);
          }
        }
      }
    }
  }
  if (step > 0) {
    float* row = out + (((int64_t)r * OH + ph / step) * OW + pw / step) * C + c0;
#pragma unroll
    for (int k = 0; k < ROI_CCH; ++k)
      if (k < nch) row[k] = acc[k] / count;
    return;
  }
#pragma unroll
  for (int k = 0; k < ROI_CCH; ++k)
    if (k < nch) out[(((int64_t)r * C + c0 + k) * PH + ph) * PW + pw] = acc[k] / count;
}

// The rows form of RoIAlign over an NHWC map: feat is the (N * H * W, C) row matrix the trunk's GEMMs write (row stride ldf),
// out the (R * OH * OW, C) row matrix the head's GEMMs read.  One workgroup per (RoI, bin the head reads); a thread owns four
// adjacent channels, so every sample is four 16-byte-per-lane reads of contiguous rows (the NCHW kernel above strides by
// H * W between channels) and the output row is written 16 bytes per lane.  Same sample positions, weights and order of
// additions per channel as roi_align_fwd_kernel.
// one output bin (ph, pw) of one RoI from an NHWC map (rows of ldf floats), all channels: the lanes of the workgroup own four
// adjacent channels each
// four adjacent channels of an NHWC map as floats: f32 (16 bytes), bfloat16 / IEEE half (8 bytes; IS_BF16 picks the decoding)
struct RoiF32 {};
struct RoiBf16 {};
struct RoiF16 {};
template <typename TAG> struct RoiElem;
template <> struct RoiElem<RoiF32> {
  typedef float T;
  static __device__ __forceinline__ f32x4 load4(const T* p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <> struct RoiElem<RoiBf16> {
  typedef unsigned short T;
  static __device__ __forceinline__ f32x4 load4(const T* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
  }
};
template <> struct RoiElem<RoiF16> {
  typedef _Float16 T;
  static __device__ __forceinline__ f32x4 load4(const T* p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 v = *reinterpret_cast<const h4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
};

template <typename TAG = RoiF32>
__device__ __forceinline__ void roi_align_bin_nhwc(const typename RoiElem<TAG>::T* __restrict__ feat, int64_t ldf, int N, int C, int H, int W,
                                                   const float* __restrict__ roi, float scale, int PH, int PW, int sampling_ratio,
                                                   int ph, int pw, float* __restrict__ row) {
  const int b = (int)roi[0];
  const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
  const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
  const float bw = rw / (float)PW, bh = rh / (float)PH;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  const float count = (float)(gh * gw);
  const typename RoiElem<TAG>::T* base = feat + (int64_t)b * H * W * ldf;
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (b >= 0 && b < N) {
      for (int iy = 0; iy < gh; ++iy) {
        float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
          if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) continue;
          float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
          int yl = (int)yy, xl = (int)xx, yh, xh;
          if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else { yh = yl + 1; }
          if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else { xh = xl + 1; }
          const float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const f32x4 p1 = RoiElem<TAG>::load4(base + (int64_t)(yl * W + xl) * ldf + c);
          const f32x4 p2 = RoiElem<TAG>::load4(base + (int64_t)(yl * W + xh) * ldf + c);
          const f32x4 p3 = RoiElem<TAG>::load4(base + (int64_t)(yh * W + xl) * ldf + c);
          const f32x4 p4 = RoiElem<TAG>::load4(base + (int64_t)(yh * W + xh) * ldf + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += roi_bilinear(w1, p1[k], w2, p2[k], w3, p3[k], w4, p4[k]);
        }
      }
    }
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = acc[k] / count;
    *reinterpret_cast<f32x4*>(row + c) = v;
  }
}

__global__ __launch_bounds__(256) void roi_align_rows_nhwc_kernel(const float* __restrict__ feat, int64_t ldf, int N, int C, int H, int W,
                                                                  const float* __restrict__ rois, float scale, int PH, int PW,
                                                                  int sampling_ratio, int step, float* __restrict__ out) {
  const int r = blockIdx.x;
  const int OW = (PW + step - 1) / step;
  const int ph = ((int)blockIdx.y / OW) * step, pw = ((int)blockIdx.y % OW) * step;
  roi_align_bin_nhwc(feat, ldf, N, C, H, W, rois + (int64_t)r * 5, scale, PH, PW, sampling_ratio, ph, pw,
                     out + ((int64_t)r * gridDim.y + blockIdx.y) * C);
}

// ---------------------------------------------------------------- multi-level RoIAlign (FPN Pooler)
// maskrcnn_benchmark's Pooler over an FPN pyramid (mrcnn_modified/modeling/roi_heads/box_head/roi_box_feature_extractors.py
// :61-68,79 -> Pooler; levels by maskrcnn_benchmark.modeling.poolers.LevelMapper): every RoI is pooled from ONE level,
//     level = clamp(floor(4 + log2(sqrt(area) / 224 + 1e-6)), k_min, k_max) - k_min,   area = (x2 - x1 + 1)(y2 - y1 + 1),
// k_min / k_max = -log2 of the first / last level's scale, with that level's map and scale.  The reference runs one
// RoIAlign per level over a boolean selection of the RoIs and scatters the results back; here ONE launch serves all RoIs —
// the workgroup of a RoI picks its level's map from a table in the kernel arguments.  out (R, C, PH, PW): viewed as
// (R, C PH PW) it is the row matrix the fc6 GEMM of FPN2MLPFeatureExtractor consumes (x.view(x.size(0), -1), :80).
struct FpnLevels {
  const float* feat[ODX_MAX_FPN_LEVELS];
  int H[ODX_MAX_FPN_LEVELS], W[ODX_MAX_FPN_LEVELS];
  float scale[ODX_MAX_FPN_LEVELS];
  int levels, k_min, k_max;
};

__device__ __forceinline__ int fpn_level_of(const float* roi, const FpnLevels& L) {
  const float area = (roi[3] - roi[1] + 1.f) * (roi[4] - roi[2] + 1.f);
  const float s = sqrtf(area);
  float lvl = floorf(4.f + log2f(s / 224.f + 1e-6f));
  lvl = fminf(fmaxf(lvl, (float)L.k_min), (float)L.k_max);
  return (int)lvl - L.k_min;
}

__global__ __launch_bounds__(256) void roi_align_fpn_kernel(FpnLevels L, int N, int C, const float* __restrict__ rois, int R,
                                                            int PH, int PW, int sampling_ratio, float* __restrict__ out,
                                                            int* __restrict__ level_out) {
  const int r = blockIdx.x;
  const int c0 = blockIdx.y * ROI_CCH;
  const int t = threadIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  const int lv = fpn_level_of(roi, L);
  if (level_out != nullptr && blockIdx.y == 0 && t == 0) level_out[r] = lv;
  if (t >= PH * PW) return;
  const int ph = t / PW, pw = t % PW;
  const float* feat = L.feat[0];
  int H = L.H[0], W = L.W[0];
  float scale = L.scale[0];
#pragma unroll
  for (int k = 1; k < ODX_MAX_FPN_LEVELS; ++k)
    if (k == lv) { feat = L.feat[k]; H = L.H[k]; W = L.W[k]; scale = L.scale[k]; }
  const int b = (int)roi[0];
  const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
  const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
  const float bw = rw / (float)PW, bh = rh / (float)PH;
  const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  const float count = (float)(gh * gw);
  float acc[ROI_CCH];
#pragma unroll
  for (int k = 0; k < ROI_CCH; ++k) acc[k] = 0.f;
  const int nch = min(ROI_CCH, C - c0);
  const float* base = feat + ((int64_t)b * C + c0) * H * W;
  if (b >= 0 && b < N) {
    for (int iy = 0; iy < gh; ++iy) {
      const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
        if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) continue;
        float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
        int yl = (int)yy, xl = (int)xx, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else { yh = yl + 1; }
        if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else { xh = xl + 1; }
        const float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int i1 = yl * W + xl, i2 = yl * W + xh, i3 = yh * W + xl, i4 = yh * W + xh;
#pragma unroll
        for (int k = 0; k < ROI_CCH; ++k) {
          if (k < nch) {
            const float* p = base + (int64_t)k * H * W;
            acc[k] += roi_bilinear(w1, p[i1], w2, p[i2], w3, p[i3], w4, p[i4]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ROI_CCH; ++k)
    if (k < nch) out[(((int64_t)r * C + c0 + k) * PH + ph) * PW + pw] = acc[k] / count;
}

// The multi-level RoIAlign from NHWC maps (the pyramid as the row GEMMs write it: level l = N H_l W_l rows of C channels), written
// as the rows fc6 reads with its weight's columns in (ph, pw, c) order: out (R, PH PW C).  One workgroup per (RoI, bin), a lane
// owns four adjacent channels (roi_align_bin_nhwc); same level choice, sample positions and sums per channel as
// roi_align_fpn_kernel.
__global__ __launch_bounds__(256) void roi_align_fpn_nhwc_kernel(FpnLevels L, int N, int C, const float* __restrict__ rois, int PH, int PW,
                                                                 int sampling_ratio, float* __restrict__ out, int* __restrict__ level_out) {
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  const int lv = fpn_level_of(roi, L);
  if (level_out != nullptr && blockIdx.y == 0 && threadIdx.x == 0) level_out[r] = lv;
  const float* feat = L.feat[0];
  int H = L.H[0], W = L.W[0];
  float scale = L.scale[0];
#pragma unroll
  for (int k = 1; k < ODX_MAX_FPN_LEVELS; ++k)
    if (k == lv) { feat = L.feat[k]; H = L.H[k]; W = L.W[k]; scale = L.scale[k]; }
  const int ph = (int)blockIdx.y / PW, pw = (int)blockIdx.y % PW;
  roi_align_bin_nhwc(feat, C, N, C, H, W, roi, scale, PH, PW, sampling_ratio, ph, pw, out + ((int64_t)r * gridDim.y + blockIdx.y) * C);
}

// The same from 16-bit NHWC maps (a pyramid run natively in bf16 / f16): the samples are decoded to f32 and everything after is
// the f32 kernel's arithmetic — the crops leave as f32 rows.
template <typename TAG>
__global__ __launch_bounds__(256) void roi_align_fpn_nhwc16_kernel(FpnLevels L, int N, int C, const float* __restrict__ rois, int PH, int PW,
                                                                   int sampling_ratio, float* __restrict__ out, int* __restrict__ level_out) {
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  const int lv = fpn_level_of(roi, L);
  if (level_out != nullptr && blockIdx.y == 0 && threadIdx.x == 0) level_out[r] = lv;
  const float* feat = L.feat[0];
  int H = L.H[0], W = L.W[0];
  float scale = L.scale[0];
#pragma unroll
  for (int k = 1; k < ODX_MAX_FPN_LEVELS; ++k)
    if (k == lv) { feat = L.feat[k]; H = L.H[k]; W = L.W[k]; scale = L.scale[k]; }
  const int ph = (int)blockIdx.y / PW, pw = (int)blockIdx.y % PW;
  roi_align_bin_nhwc<TAG>(reinterpret_cast<const typename RoiElem<TAG>::T*>(feat), C, N, C, H, W, roi, scale, PH, PW, sampling_ratio, ph, pw,
                          out + ((int64_t)r * gridDim.y + blockIdx.y) * C);
}

// ---------------------------------------------------------------- NMS
// boxes (R, 4) xyxy sorted by descending score.  Pass 1: 64 x 64 blocks of the suppression
// relation as 64-bit masks (one wavefront ballot's worth per row: bit j of mask[i][cb] = box
// cb*64+j is suppressed by box i, j after i).  Pass 2: one wave walks the boxes in order; lane w
// holds word w of the "removed" set.
__device__ __forceinline__ float box_iou_plus1(const float* a, const float* b) {
  const float l = fmaxf(a[0], b[0]), r = fminf(a[2], b[2]);
  const float t = fmaxf(a[1], b[1]), bt = fminf(a[3], b[3]);
  const float w = fmaxf(r - l + 1.f, 0.f), h = fmaxf(bt - t + 1.f, 0.f);
  const float inter = w * h;
  const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  return inter / (sa + sb - inter);
}

// counts != nullptr: a batch of independent box sets (the classes of an image), set z = blockIdx.z holding counts[z]
// boxes in a slot of Rmax (boxes, mask and keep advance by whole slots); otherwise one set of R boxes.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int R, float thr,
                                                      unsigned long long* __restrict__ mask, int words,
                                                      const int* __restrict__ counts) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;  // only boxes after i can be suppressed by i
  if (counts != nullptr) {
    boxes += (int64_t)blockIdx.z * R * 4;
    mask += (int64_t)blockIdx.z * R * words;
    R = counts[blockIdx.z];
    if (rb * 64 >= R || cb * 64 >= R) return;
  }
  __shared__ float cbx[64 * 4];
  const int lane = threadIdx.x;
  const int cj = cb * 64 + lane;
  if (cj < R) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cbx[lane * 4 + q] = boxes[(int64_t)cj * 4 + q];
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  if (i >= R) return;
  float me[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) me[q] = boxes[(int64_t)i * 4 + q];
  unsigned long long bits = 0ull;
  const int ncol = min(64, R - cb * 64);
  const int start = (rb == cb) ? lane + 1 : 0;
  for (int j = start; j < ncol; ++j)
    if (box_iou_plus1(me, cbx + j * 4) > thr) bits |= 1ull << j;
  mask[(int64_t)i * words + cb] = bits;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// One wave walks the sorted boxes a block of 64 at a time.  The set of suppressed boxes is spread over the lanes (lane w
// owns words w, w + 64, ...).  Per block: the word of the set that covers the block is broadcast; the 64 x 64 diagonal
// mask block (one word per lane) resolves the suppression inside the block with scalar bit operations (64 steps, no
// memory); then the mask rows of the block's survivors are OR-ed into the set, eight rows' loads in flight at a time.
// (Walking the boxes one by one made every kept box a dependent global load: 2.8 ms at R = 6000, now 0.2 ms.)
// max_keep > 0: stop after that many survivors (keep must be zero on entry: the boxes behind the last survivor are not
// visited) — the RPN keeps the first 300 of up to 6000 candidates, reached after a few hundred of them.
__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long* __restrict__ mask, int R, int words,
                                                        unsigned char* __restrict__ keep, const int* __restrict__ counts,
                                                        int max_keep) {
  const int lane = threadIdx.x;
  int kept = 0;
  if (counts != nullptr) {                // batch: one wave per box set; rows of the mask keep the slot's stride `words`
    mask += (int64_t)blockIdx.x * R * words;
    keep += (int64_t)blockIdx.x * R;
    R = counts[blockIdx.x];
    if (R <= 0) return;
  }
  const int nwords = (R + 63) >> 6;       // words that hold boxes of this set (<= words)
  constexpr int NW = 4;      // R <= 64 * 64 * NW boxes
  unsigned long long removed[NW];
#pragma unroll
  for (int q = 0; q < NW; ++q) removed[q] = 0ull;
  for (int wi = 0; wi < nwords; ++wi) {
    unsigned long long word = 0ull;
#pragma unroll
    for (int q = 0; q < NW; ++q)
      if ((wi >> 6) == q) word = removed[q];
    const unsigned long long cur = readlane_u64(word, wi & 63);
    const int i = wi * 64 + lane;
    const unsigned long long d = (i < R) ? mask[(int64_t)i * words + wi] : 0ull;   // bits above the lane's own box only
    unsigned long long alive = ~cur;
    if (wi == nwords - 1 && (R & 63)) alive &= (1ull << (R & 63)) - 1ull;
#pragma unroll
    for (int b = 0; b < 64; ++b) {
      const unsigned long long db = readlane_u64(d, b);
      if ((alive >> b) & 1ull) alive &= ~db;
    }
    if (max_keep > 0) {
      int room = max_keep - kept;
      while (__builtin_popcountll(alive) > room) alive &= ~(1ull << (63 - __builtin_clzll(alive)));   // drop the last survivors
      kept += __builtin_popcountll(alive);
    }
    if (i < R) keep[i] = (unsigned char)((alive >> lane) & 1ull);
    if (max_keep > 0 && kept >= max_keep) return;
    unsigned long long a = alive;
    while (a) {
      int64_t row[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        row[u] = -1;
        if (a) {
          row[u] = (int64_t)(wi * 64 + __builtin_ctzll(a)) * words;
          a &= a - 1ull;
        }
      }
      unsigned long long v[8][NW];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          const int col = lane + 64 * q;
          v[u][q] = (row[u] >= 0 && col < nwords && col > wi) ? mask[row[u] + col] : 0ull;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < NW; ++q) removed[q] |= v[u][q];
    }
  }
}

// ---------------------------------------------------------------- RPN candidates: top-k, decode, clip — one kernel
// RPNPostProcessor.forward_for_single_feature_map up to the suppression (rpn/inference.py:76-115) for a batch of images of one
// size: the pre_nms_top_n best anchors of an image by objectness, in descending order (ties: the lower anchor index first),
// their deltas decoded against the anchors (BoxCoder weights 1, +1 widths, dw / dh clamped) and clipped to the image.
// One 1024-thread workgroup per image: a radix select over the 56-bit keys (ordered logit bits, inverted anchor index) finds the
// k-th key in seven 8-bit passes over the image's logits (L2-resident: 28 500 floats at 600 x 800), the k keys at or above it
// are compacted into LDS and sorted there (bitonic, <= 8192 entries), and each thread then decodes its ranks.  Replaces a
// sigmoid, a top-k, a gather, an advanced-index gather and ~25 elementwise launches — and the library top-k whose replay from
// a captured HIP graph faulted (docs/HISTORY.md 7, round 5).  logits (B, A, H, W), deltas (B, 4 A, H, W), anchors (H W A, 4) in the
// (location, anchor type) order of grid_anchors; flat candidate index i = (h W + w) A + a.
constexpr int TOPK_NT = 1024, TOPK_MAX = 8192;

__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending as unsigned
}

__device__ __forceinline__ float from_ordered_bits(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(TOPK_NT) void rpn_topk_decode_kernel(const float* __restrict__ logits, const float* __restrict__ deltas,
                                                                  const float* __restrict__ anchors, int A, int HW, int k, int kpad,
                                                                  float xmax, float ymax, float dclamp, float* __restrict__ boxes,
                                                                  float* __restrict__ scores, int32_t* __restrict__ index) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long buf[];      // kpad entries (<= 64 KiB, asked for at launch)
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix, s_mask;
  __shared__ unsigned s_need, s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = A * HW;
  const float* lg = logits + (int64_t)b * N;
  auto key_of = [&](int j) -> unsigned long long {           // j walks memory order (a, hw); the candidate index is (hw, a)
    const int a = j / HW, hw = j - a * HW;
    const unsigned i = (unsigned)(hw * A + a);
    return ((unsigned long long)ordered_bits(lg[j]) << 24) | (unsigned long long)(0xffffffu - i);
  };
  if (tid == 0) { s_prefix = 0ull; s_mask = 0ull; s_need = (unsigned)k; s_cnt = 0u; }
  __syncthreads();
  for (int pass = 6; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_prefix, mask = s_mask;
    for (int j = tid; j < N; j += TOPK_NT) {
      const unsigned long long key = key_of(j);
      if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, acc = 0u;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= need) break;
        acc += hist[d];
      }
      s_need = need - acc;                                     // how many of digit d's keys are still wanted
      s_prefix = prefix | ((unsigned long long)d << (8 * pass));
      s_mask = mask | (255ull << (8 * pass));
    }
    __syncthreads();
  }
  const unsigned long long thresh = s_prefix;                  // the k-th largest key (keys are unique: they carry the index)
  for (int j = tid; j < kpad; j += TOPK_NT) buf[j] = 0ull;      // pads sort to the end
  __syncthreads();
  for (int j = tid; j < N; j += TOPK_NT) {
    const unsigned long long key = key_of(j);
    if (key >= thresh) {
      const unsigned pos = atomicAdd(&s_cnt, 1u);
      if (pos < (unsigned)kpad) buf[pos] = key;
    }
  }
  __syncthreads();
  // bitonic sort, descending, kpad = a power of two
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (kpad >> 1); t += TOPK_NT) {
        const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = buf[lo], y = buf[hi];
        if ((x < y) == desc) { buf[lo] = y; buf[hi] = x; }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < k; r += TOPK_NT) {                     // (separate multiplies and adds, as the reference's tensor ops round)
#pragma clang fp contract(off)
    const unsigned long long key = buf[r];
    const int i = (int)(0xffffffu - (unsigned)(key & 0xffffffull));
    const float v = from_ordered_bits((unsigned)(key >> 24));
    const int a = i % A, hw = i / A;
    const float* dl = deltas + ((int64_t)b * 4 * A + 4 * a) * HW + hw;
    const float dx = dl[0], dy = dl[HW];
    const float dw = fminf(dl[2 * (int64_t)HW], dclamp), dh = fminf(dl[3 * (int64_t)HW], dclamp);
    const float* an = anchors + (int64_t)i * 4;
    const float w = an[2] - an[0] + 1.f, h = an[3] - an[1] + 1.f;
    const float cx = an[0] + 0.5f * w, cy = an[1] + 0.5f * h;
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float* o = boxes + ((int64_t)b * k + r) * 4;
    o[0] = fminf(fmaxf(pcx - 0.5f * pw, 0.f), xmax);
    o[1] = fminf(fmaxf(pcy - 0.5f * ph, 0.f), ymax);
    o[2] = fminf(fmaxf(pcx + 0.5f * pw - 1.f, 0.f), xmax);
    o[3] = fminf(fmaxf(pcy + 0.5f * ph - 1.f, 0.f), ymax);
    scores[(int64_t)b * k + r] = 1.f / (1.f + expf(-v));
    if (index != nullptr) index[(int64_t)b * k + r] = i;
  }
}

extern "C" int odx_rpn_topk_decode_f32(const float* logits, const float* deltas, const float* anchors, int B, int A, int H, int W, int k,
                                       float img_w, float img_h, float delta_clamp, float* boxes, float* scores, int32_t* index,
                                       odx_stream_t stream) {
  if (B <= 0 || k <= 0) return ODX_OK;
  ODX_REQUIRE(logits && deltas && anchors && boxes && scores && A > 0 && H > 0 && W > 0, "odx_rpn_topk_decode_f32: bad argument");
  const int64_t N = (int64_t)A * H * W;
  ODX_REQUIRE(k <= TOPK_MAX && k <= N && N < (1 << 24), "odx_rpn_topk_decode_f32: k <= min(8192, A H W) and A H W < 2^24 required");
  int kpad = 2;
  while (kpad < k) kpad <<= 1;
  const size_t lds = (size_t)kpad * sizeof(unsigned long long);
  ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_topk_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(TOPK_MAX * sizeof(unsigned long long))));
  hipLaunchKernelGGL(rpn_topk_decode_kernel, dim3((unsigned)B), dim3(TOPK_NT), lds, as_stream(stream), logits, deltas, anchors, A, H * W, k,
                     kpad, img_w - 1.f, img_h - 1.f, delta_clamp, boxes, scores, index);
  ODX_CHECK_LAUNCH("odx_rpn_topk_decode_f32");
  return ODX_OK;
}

// The first `P` survivors of each set's suppression, in order, as a dense (B, P, 4) block + their count: one wave per set walks
// the keep flags 64 at a time (ballot + prefix count); slots past the count get a harmless 16 x 16 box.  With the kernel above
// and odx_nms_batched_first_f32 the proposal stage of a batch has no data-dependent shape and no library sort left.
__global__ __launch_bounds__(64) void nms_compact_kernel(const float* __restrict__ boxes, const unsigned char* __restrict__ keep,
                                                         const int* __restrict__ counts, int Rmax, int P, float* __restrict__ out,
                                                         int32_t* __restrict__ nkept) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int R = counts != nullptr ? counts[b] : Rmax;
  const float* bx = boxes + (int64_t)b * Rmax * 4;
  const unsigned char* kp = keep + (int64_t)b * Rmax;
  float* o = out + (int64_t)b * P * 4;
  int n = 0;
  for (int r0 = 0; r0 < R && n < P; r0 += 64) {
    const int r = r0 + lane;
    const bool k1 = r < R && kp[r] != 0;
    const unsigned long long m = __ballot(k1);
    const int at = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (k1 && at < P) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[(int64_t)at * 4 + q] = bx[(int64_t)r * 4 + q];
    }
    n += __builtin_popcountll(m);
  }
  n = n < P ? n : P;
  for (int s = n + lane; s < P; s += 64) {
    o[(int64_t)s * 4 + 0] = 0.f; o[(int64_t)s * 4 + 1] = 0.f; o[(int64_t)s * 4 + 2] = 15.f; o[(int64_t)s * 4 + 3] = 15.f;
  }
  if (lane == 0) nkept[b] = n;
}

extern "C" int odx_nms_compact_f32(const float* boxes, const unsigned char* keep, const int32_t* counts, int Rmax, int B, int P,
                                   float* out, int32_t* nkept, odx_stream_t stream) {
  if (B <= 0 || P <= 0) return ODX_OK;
  ODX_REQUIRE(boxes && keep && out && nkept && Rmax > 0, "odx_nms_compact_f32: bad argument");
  hipLaunchKernelGGL(nms_compact_kernel, dim3((unsigned)B), dim3(64), 0, as_stream(stream), boxes, keep, counts, Rmax, P, out, nkept);
  ODX_CHECK_LAUNCH("odx_nms_compact_f32");
  return ODX_OK;
}

// ---------------------------------------------------------------- mask pasting (Masker / paste_mask_in_image)
// One thread per image pixel and detection: zero-pad the S x S mask by `padding`, grow the box by the same ratio,
// truncate it to integers, resize the padded mask to the box bilinearly (align_corners = False) and threshold.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                          int R, int S, int im_h, int im_w, float thresh, int padding,
                                                          unsigned char* __restrict__ out) {
  const int r = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (int64_t)im_h * im_w) return;
  const int y = (int)(pix / im_w), x = (int)(pix - (int64_t)y * im_w);
  const int Sp = S + 2 * padding;
  const float scale = (float)Sp / (float)S;
  const float* b = boxes + 4 * r;
  float w_half = (b[2] - b[0]) * 0.5f, h_half = (b[3] - b[1]) * 0.5f;
  const float x_c = (b[2] + b[0]) * 0.5f, y_c = (b[3] + b[1]) * 0.5f;
  w_half *= scale;
  h_half *= scale;
  const int bx0 = (int)(x_c - w_half), bx2 = (int)(x_c + w_half), by0 = (int)(y_c - h_half), by2 = (int)(y_c + h_half);
  const int w = max(bx2 - bx0 + 1, 1), h = max(by2 - by0 + 1, 1);
  const int x_0 = max(bx0, 0), x_1 = min(bx2 + 1, im_w), y_0 = max(by0, 0), y_1 = min(by2 + 1, im_h);
  unsigned char v = 0;
  if (x >= x_0 && x < x_1 && y >= y_0 && y < y_1) {
    const float sx = (float)Sp / (float)w, sy = (float)Sp / (float)h;
    float fx = sx * ((float)(x - bx0) + 0.5f) - 0.5f, fy = sy * ((float)(y - by0) + 0.5f) - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    fy = fy < 0.f ? 0.f : fy;
    const int ix0 = (int)fx, iy0 = (int)fy;
    const int ix1 = ix0 + (ix0 < Sp - 1 ? 1 : 0), iy1 = iy0 + (iy0 < Sp - 1 ? 1 : 0);
    const float lx1 = fx - (float)ix0, ly1 = fy - (float)iy0, lx0 = 1.f - lx1, ly0 = 1.f - ly1;
    const float* m = masks + (int64_t)r * S * S;
    auto at = [&](int yy, int xx) -> float {
      yy -= padding;
      xx -= padding;
      return (yy >= 0 && yy < S && xx >= 0 && xx < S) ? m[yy * S + xx] : 0.f;
    };
    const float val = ly0 * (lx0 * at(iy0, ix0) + lx1 * at(iy0, ix1)) + ly1 * (lx0 * at(iy1, ix0) + lx1 * at(iy1, ix1));
    v = val > thresh ? 1 : 0;
  }
  out[((int64_t)r * im_h + y) * im_w + x] = v;
}


// ---------------------------------------------------------------- harvest labelling (the device work in front of a harvester's host read)
// The on-line RPN and detector harvesters (odx/harvest.py: RPNHarvester.prepare, DetectorHarvester.prepare; the reference's
// rpn_getProposals.py:265-449 and box_head_getProposals.py:117-226) label every anchor / proposal of an image against its
// ground-truth boxes before the host decides what to sample: overlaps, the associated box, the flags and their counts.  As tensor
// operations that is ~45 launches per image and harvester; here it is one launch (two for the anchors: the best overlap per box
// is a reduction over all of them).  The arithmetic is the tensor form's, operation by operation in f32 with no fused
// multiply-add (the flags compare overlaps with thresholds: they have to be the same bits), maxima keep the FIRST maximum.
__device__ __forceinline__ float iou_plus1(const float* g, const float* p) {
#pragma clang fp contract(off)
  const float xmin = fmaxf(g[0], p[0]), ymin = fmaxf(g[1], p[1]), xmax = fminf(g[2], p[2]), ymax = fminf(g[3], p[3]);
  const float w = xmax - xmin + 1.f, h = ymax - ymin + 1.f;
  const float inter = w * h;
  const float ga = (g[2] - g[0] + 1.f) * (g[3] - g[1] + 1.f);
  const float pa = (p[2] - p[0] + 1.f) * (p[3] - p[1] + 1.f);
  const float ov = inter / (ga + pa - inter);
  return (w > 0.f && h > 0.f) ? ov : 0.f;
}

constexpr int LABEL_MAX_GT = 64;

// counters (int32, zero on entry): [A candidates per type][G hit][G has_mine][A over per type][G A extras per (box, type)]
__global__ __launch_bounds__(256) void rpn_label_kernel(const float* __restrict__ gt, int G, const float* __restrict__ anchors,
                                                        const int64_t* __restrict__ cls, int n, int A, float neg_thr, float pos_thr,
                                                        float* __restrict__ ious, float* __restrict__ assoc, unsigned char* __restrict__ neg_mask,
                                                        unsigned char* __restrict__ over, unsigned int* __restrict__ best_bits,
                                                        int* __restrict__ counters) {
  __shared__ float gs[LABEL_MAX_GT * 4];
  for (int k = threadIdx.x; k < G * 4; k += 256) gs[k] = gt[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = anchors[(int64_t)i * 4 + q];
  float best = iou_plus1(gs, a);
  int arg = 0;
  for (int j = 1; j < G; ++j) {
    const float v = iou_plus1(gs + 4 * j, a);
    if (v > best) { best = v; arg = j; }
  }
  float as[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { as[q] = gs[4 * arg + q]; assoc[(int64_t)i * 4 + q] = as[q]; }
  ious[i] = best;
  const bool ng = best < neg_thr, ov = best > pos_thr;
  neg_mask[i] = ng;
  over[i] = ov;
  const int t = (int)cls[i];
  if (ng) atomicAdd(counters + t, 1);
  if (ov) atomicAdd(counters + A + 2 * G + t, 1);
  for (int j = 0; j < G; ++j) {
    const bool e0 = as[0] == gs[4 * j], e1 = as[1] == gs[4 * j + 1], e2 = as[2] == gs[4 * j + 2], e3 = as[3] == gs[4 * j + 3];
    if (e0 && e1 && e2 && e3) {
      atomicMax(best_bits + j, __float_as_uint(best));       // overlaps are >= 0: their bits order like the numbers
      counters[A + G + j] = 1;                                // (every writer stores the same value)
    }
    if ((e0 || e1 || e2 || e3) && ov) atomicAdd(counters + A + j, 1);
  }
}

__global__ __launch_bounds__(256) void rpn_label_extra_kernel(const float* __restrict__ gt, int G, const int64_t* __restrict__ cls, int n, int A,
                                                              const float* __restrict__ ious, const float* __restrict__ assoc,
                                                              const unsigned int* __restrict__ best_bits, unsigned char* __restrict__ extra,
                                                              int* __restrict__ counters) {
  __shared__ float gs[LABEL_MAX_GT * 4];
  for (int k = threadIdx.x; k < G * 4; k += 256) gs[k] = gt[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = ious[i];
  float as[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) as[q] = assoc[(int64_t)i * 4 + q];
  const int t = (int)cls[i];
  for (int j = 0; j < G; ++j) {
    const bool mine = as[0] == gs[4 * j] && as[1] == gs[4 * j + 1] && as[2] == gs[4 * j + 2] && as[3] == gs[4 * j + 3];
    const bool ex = mine && v == __uint_as_float(best_bits[j]);
    extra[(int64_t)j * n + i] = ex;
    if (ex) atomicAdd(counters + 2 * A + 2 * G + j * A + t, 1);
  }
}

// counters (int32, zero on entry): [G pairs per box][n_in candidates per listed class]
__global__ __launch_bounds__(256) void det_label_kernel(const float* __restrict__ gt_raw, const int* __restrict__ labels0, int G,
                                                        const float* __restrict__ prop_raw, int R, int C, float img_w, float img_h,
                                                        float reg_min, float neg_thr, const int* __restrict__ in_image, int n_in,
                                                        float* __restrict__ prop, float* __restrict__ overlap, unsigned char* __restrict__ sel,
                                                        unsigned char* __restrict__ cmask, int* __restrict__ counters) {
  __shared__ float gs[LABEL_MAX_GT * 4];
  __shared__ int ls[LABEL_MAX_GT];
  for (int k = threadIdx.x; k < G * 4; k += 256) {
    const float hi = (k & 1) ? img_h - 1.f : img_w - 1.f;         // clamp_boxes_: x to [0, w - 1], y to [0, h - 1]
    gs[k] = fminf(fmaxf(gt_raw[k], 0.f), hi);
  }
  for (int k = threadIdx.x; k < G; k += 256) ls[k] = labels0[k];
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float p[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float hi = (q & 1) ? img_h - 1.f : img_w - 1.f;
    p[q] = fminf(fmaxf(prop_raw[(int64_t)r * 4 + q], 0.f), hi);
    prop[(int64_t)r * 4 + q] = p[q];
  }
  float* orow = overlap + (int64_t)r * C;
  for (int c = 0; c < C; ++c) orow[c] = 0.f;
  float best = -1.f;
  int first = -1;
  for (int j = 0; j < G; ++j) {
    const float v = iou_plus1(gs + 4 * j, p);
    const int c = ls[j];
    if (c >= 0 && c < C) orow[c] = fmaxf(orow[c], v);
    if (v > best) { best = v; first = j; }                      // the FIRST maximum
  }
  const int assoc = best > 0.f ? first : -1;
  for (int j = 0; j < G; ++j) {
    const int c = ls[j];
    const bool s = c >= 0 && c < C && orow[c] > reg_min && assoc == j;
    sel[(int64_t)j * R + r] = s;
    if (s) atomicAdd(counters + j, 1);
  }
  for (int k = 0; k < n_in; ++k) {
    const bool m = orow[in_image[k]] < neg_thr;
    cmask[(int64_t)r * n_in + k] = m;
    if (m) atomicAdd(counters + G + k, 1);
  }
}

// regression targets of n (example, target) box pairs, the reference's formula operation by operation
// (box_head_getProposals.py:170-182, rpn_getProposals.py:420-432): ((gx - sx) / sw, (gy - sy) / sh, log(gw / sw), log(gh / sh))
__global__ __launch_bounds__(256) void box_targets_kernel(const float* __restrict__ ex, const float* __restrict__ tg, int64_t n,
                                                          float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* e = ex + i * 4;
  const float* g = tg + i * 4;
  const float sw = e[2] - e[0] + 1.f, sh = e[3] - e[1] + 1.f;
  const float sx = e[0] + 0.5f * sw, sy = e[1] + 0.5f * sh;
  const float gw = g[2] - g[0] + 1.f, gh = g[3] - g[1] + 1.f;
  const float gx = g[0] + 0.5f * gw, gy = g[1] + 0.5f * gh;
  out[i * 4 + 0] = (gx - sx) / sw;
  out[i * 4 + 1] = (gy - sy) / sh;
  out[i * 4 + 2] = logf(gw / sw);
  out[i * 4 + 3] = logf(gh / sh);
}

}  // namespace odx

using namespace odx;

extern "C" int odx_roi_align_fwd_f32(const float* feat, int N, int C, int H, int W, const float* rois, int R,
                                     float spatial_scale, int PH, int PW, int sampling_ratio, float* out,
                                     odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feat && rois && out && N > 0 && H > 0 && W > 0, "odx_roi_align_fwd_f32: bad argument");
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 256, "odx_roi_align_fwd_f32: PH * PW must be in 1..256");
  ODX_REQUIRE((int64_t)H * W < (1ll << 31) && ceil_div(C, ROI_CCH) < 65536, "odx_roi_align_fwd_f32: map too large");
  dim3 grid((unsigned)R, (unsigned)ceil_div(C, ROI_CCH));
  hipLaunchKernelGGL(roi_align_fwd_kernel, grid, dim3(256), 0, as_stream(stream), feat, N, C, H, W, rois, R,
                     spatial_scale, PH, PW, sampling_ratio, out, 0);
  ODX_CHECK_LAUNCH("odx_roi_align_fwd_f32");
  return ODX_OK;
}

// RoIAlign for a consumer that starts with a stride-`step` 1 x 1 convolution (the conv5 head, STRIDE_IN_1X1): the bins
// that convolution reads only, as rows of an (R * ceil(PH / step) * ceil(PW / step), C) matrix.
extern "C" int odx_roi_align_rows_f32(const float* feat, int N, int C, int H, int W, const float* rois, int R,
                                      float spatial_scale, int PH, int PW, int sampling_ratio, int step, float* out_rows,
                                      odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feat && rois && out_rows && N > 0 && H > 0 && W > 0 && step > 0, "odx_roi_align_rows_f32: bad argument");
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 256, "odx_roi_align_rows_f32: PH * PW must be in 1..256");
  ODX_REQUIRE((int64_t)H * W < (1ll << 31) && ceil_div(C, ROI_CCH) < 65536, "odx_roi_align_rows_f32: map too large");
  dim3 grid((unsigned)R, (unsigned)ceil_div(C, ROI_CCH));
  hipLaunchKernelGGL(roi_align_fwd_kernel, grid, dim3(256), 0, as_stream(stream), feat, N, C, H, W, rois, R,
                     spatial_scale, PH, PW, sampling_ratio, out_rows, step);
  ODX_CHECK_LAUNCH("odx_roi_align_rows_f32");
  return ODX_OK;
}

// The same bins from an NHWC map handed over as its row matrix (N * H * W rows of C channels, row stride ldf floats) — what the
// trunk run as row GEMMs writes: no NCHW copy of the map between the two.  C % 4 == 0, ldf % 4 == 0, 16-byte aligned.
extern "C" int odx_roi_align_rows_nhwc_f32(const float* feat_rows, int64_t ldf, int N, int C, int H, int W, const float* rois, int R,
                                           float spatial_scale, int PH, int PW, int sampling_ratio, int step, float* out_rows,
                                           odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feat_rows && rois && out_rows && N > 0 && H > 0 && W > 0 && step > 0, "odx_roi_align_rows_nhwc_f32: bad argument");
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 65535, "odx_roi_align_rows_nhwc_f32: PH * PW must be in 1..65535");
  ODX_REQUIRE(C % 4 == 0 && ldf % 4 == 0 && ldf >= C && (reinterpret_cast<uintptr_t>(feat_rows) & 15u) == 0 &&
              (reinterpret_cast<uintptr_t>(out_rows) & 15u) == 0, "odx_roi_align_rows_nhwc_f32: C %% 4, ldf %% 4 == 0 and 16-byte aligned rows expected");
  ODX_REQUIRE((int64_t)H * W < (1ll << 31), "odx_roi_align_rows_nhwc_f32: map too large");
  const int OH = (PH + step - 1) / step, OW = (PW + step - 1) / step;
  dim3 grid((unsigned)R, (unsigned)(OH * OW));
  hipLaunchKernelGGL(roi_align_rows_nhwc_kernel, grid, dim3(256), 0, as_stream(stream), feat_rows, ldf, N, C, H, W, rois,
                     spatial_scale, PH, PW, sampling_ratio, step, out_rows);
  ODX_CHECK_LAUNCH("odx_roi_align_rows_nhwc_f32");
  return ODX_OK;
}

extern "C" int odx_roi_align_fpn_f32(const float* const* feats, const int* H, const int* W, const float* scales, int levels,
                                     int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out,
                                     int* level_out, odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feats && H && W && scales && rois && out && N > 0, "odx_roi_align_fpn_f32: null pointer");
  ODX_REQUIRE(levels >= 1 && levels <= ODX_MAX_FPN_LEVELS, "odx_roi_align_fpn_f32: 1..%d pyramid levels", ODX_MAX_FPN_LEVELS);
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 256, "odx_roi_align_fpn_f32: PH * PW must be in 1..256");
  ODX_REQUIRE(ceil_div(C, ROI_CCH) < 65536, "odx_roi_align_fpn_f32: too many channels");
  FpnLevels L;
  for (int k = 0; k < ODX_MAX_FPN_LEVELS; ++k) {
    const int j = k < levels ? k : levels - 1;
    ODX_REQUIRE(feats[j] && H[j] > 0 && W[j] > 0 && scales[j] > 0.f && (int64_t)H[j] * W[j] < (1ll << 31),
                "odx_roi_align_fpn_f32: bad level %d", j);
    L.feat[k] = feats[j];
    L.H[k] = H[j];
    L.W[k] = W[j];
    L.scale[k] = scales[j];
  }
  L.levels = levels;
  // LevelMapper: k_min = -log2(scales[0]), k_max = -log2(scales[-1]) (scales are powers of two: 1/4 .. 1/32)
  L.k_min = (int)lrintf(-log2f(scales[0]));
  L.k_max = (int)lrintf(-log2f(scales[levels - 1]));
  ODX_REQUIRE(L.k_max - L.k_min == levels - 1, "odx_roi_align_fpn_f32: the level scales must halve from level to level");
  dim3 grid((unsigned)R, (unsigned)ceil_div(C, ROI_CCH));
  hipLaunchKernelGGL(roi_align_fpn_kernel, grid, dim3(256), 0, as_stream(stream), L, N, C, rois, R, PH, PW, sampling_ratio, out,
                     level_out);
  ODX_CHECK_LAUNCH("odx_roi_align_fpn_f32");
  return ODX_OK;
}

// odx_roi_align_fpn_f32 from NHWC maps (level l: N H_l W_l rows of C channels, C % 4 == 0, 16-byte aligned): out (R, PH PW C) —
// the rows fc6 reads when its weight's columns are ordered (ph, pw, c).
extern "C" int odx_roi_align_fpn_nhwc_f32(const float* const* feats, const int* H, const int* W, const float* scales, int levels,
                                          int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out_rows,
                                          int* level_out, odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feats && H && W && scales && rois && out_rows && N > 0, "odx_roi_align_fpn_nhwc_f32: null pointer");
  ODX_REQUIRE(levels >= 1 && levels <= ODX_MAX_FPN_LEVELS, "odx_roi_align_fpn_nhwc_f32: 1..%d pyramid levels", ODX_MAX_FPN_LEVELS);
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 65535 && C % 4 == 0 && (reinterpret_cast<uintptr_t>(out_rows) & 15u) == 0,
              "odx_roi_align_fpn_nhwc_f32: PH * PW in 1..65535, C %% 4 == 0, 16-byte aligned rows expected");
  FpnLevels L;
  for (int k = 0; k < ODX_MAX_FPN_LEVELS; ++k) {
    const int j = k < levels ? k : levels - 1;
    ODX_REQUIRE(feats[j] && H[j] > 0 && W[j] > 0 && scales[j] > 0.f && (int64_t)H[j] * W[j] < (1ll << 31) &&
                (reinterpret_cast<uintptr_t>(feats[j]) & 15u) == 0, "odx_roi_align_fpn_nhwc_f32: bad level %d", j);
    L.feat[k] = feats[j];
    L.H[k] = H[j];
    L.W[k] = W[j];
    L.scale[k] = scales[j];
  }
  L.levels = levels;
  L.k_min = (int)lrintf(-log2f(scales[0]));
  L.k_max = (int)lrintf(-log2f(scales[levels - 1]));
  ODX_REQUIRE(L.k_max - L.k_min == levels - 1, "odx_roi_align_fpn_nhwc_f32: the level scales must halve from level to level");
  const int threads = C >= 1024 ? 256 : (C >= 512 ? 128 : 64);
  hipLaunchKernelGGL(roi_align_fpn_nhwc_kernel, dim3((unsigned)R, (unsigned)(PH * PW)), dim3(threads), 0, as_stream(stream), L, N, C, rois,
                     PH, PW, sampling_ratio, out_rows, level_out);
  ODX_CHECK_LAUNCH("odx_roi_align_fpn_nhwc_f32");
  return ODX_OK;
}

// odx_roi_align_fpn_nhwc_f32 over 16-bit maps (is_bf16 != 0: bfloat16, else IEEE half; 8-byte aligned rows, C % 4 == 0): the
// crops as f32 rows (R, PH PW C).
extern "C" int odx_roi_align_fpn_nhwc_16(const void* const* feats, int is_bf16, const int* H, const int* W, const float* scales, int levels,
                                         int N, int C, const float* rois, int R, int PH, int PW, int sampling_ratio, float* out_rows,
                                         int* level_out, odx_stream_t stream) {
  if (R <= 0 || C <= 0) return ODX_OK;
  ODX_REQUIRE(feats && H && W && scales && rois && out_rows && N > 0, "odx_roi_align_fpn_nhwc_16: null pointer");
  ODX_REQUIRE(levels >= 1 && levels <= ODX_MAX_FPN_LEVELS, "odx_roi_align_fpn_nhwc_16: 1..%d pyramid levels", ODX_MAX_FPN_LEVELS);
  ODX_REQUIRE(PH > 0 && PW > 0 && PH * PW <= 65535 && C % 4 == 0 && (reinterpret_cast<uintptr_t>(out_rows) & 15u) == 0,
              "odx_roi_align_fpn_nhwc_16: PH * PW in 1..65535, C %% 4 == 0, 16-byte aligned output rows expected");
  FpnLevels L;
  for (int k = 0; k < ODX_MAX_FPN_LEVELS; ++k) {
    const int j = k < levels ? k : levels - 1;
    ODX_REQUIRE(feats[j] && H[j] > 0 && W[j] > 0 && scales[j] > 0.f && (int64_t)H[j] * W[j] < (1ll << 31) &&
                (reinterpret_cast<uintptr_t>(feats[j]) & 7u) == 0, "odx_roi_align_fpn_nhwc_16: bad level %d", j);
    L.feat[k] = static_cast<const float*>(feats[j]);          // (the table's pointer type; the kernel reads 16-bit elements)
    L.H[k] = H[j];
    L.W[k] = W[j];
    L.scale[k] = scales[j];
  }
  L.levels = levels;
  L.k_min = (int)lrintf(-log2f(scales[0]));
  L.k_max = (int)lrintf(-log2f(scales[levels - 1]));
  ODX_REQUIRE(L.k_max - L.k_min == levels - 1, "odx_roi_align_fpn_nhwc_16: the level scales must halve from level to level");
  const int threads = C >= 1024 ? 256 : (C >= 512 ? 128 : 64);
  const dim3 grid((unsigned)R, (unsigned)(PH * PW));
  if (is_bf16)
    hipLaunchKernelGGL(roi_align_fpn_nhwc16_kernel<RoiBf16>, grid, dim3(threads), 0, as_stream(stream), L, N, C, rois, PH, PW, sampling_ratio,
                       out_rows, level_out);
  else
    hipLaunchKernelGGL(roi_align_fpn_nhwc16_kernel<RoiF16>, grid, dim3(threads), 0, as_stream(stream), L, N, C, rois, PH, PW, sampling_ratio,
                       out_rows, level_out);
  ODX_CHECK_LAUNCH("odx_roi_align_fpn_nhwc_16");
  return ODX_OK;
}

extern "C" int64_t odx_nms_workspace_bytes(int R) {
  if (R <= 0) return 0;
  const int64_t words = ceil_div(R, 64);
  return (int64_t)R * words * 8;
}

extern "C" int odx_nms_f32(const float* boxes_sorted, int R, float iou_threshold, unsigned char* keep, void* workspace,
                           int64_t workspace_bytes, odx_stream_t stream) {
  if (R <= 0) return ODX_OK;
  ODX_REQUIRE(boxes_sorted && keep && workspace, "odx_nms_f32: null pointer");
  ODX_REQUIRE(R <= 64 * 64 * 4, "odx_nms_f32: at most 16384 boxes");
  if (workspace_bytes < odx_nms_workspace_bytes(R)) {
    set_error("odx_nms_f32: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  const int words = (int)ceil_div(R, 64);
  hipStream_t s = as_stream(stream);
  ODX_PROPAGATE(zero_bytes(workspace, (size_t)odx_nms_workspace_bytes(R), s));
  auto* mask = static_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words), dim3(64), 0, s, boxes_sorted, R,
                     iou_threshold, mask, words, (const int*)nullptr);
  ODX_CHECK_LAUNCH("odx_nms_f32(mask)");
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, mask, R, words, keep, (const int*)nullptr, 0);
  ODX_CHECK_LAUNCH("odx_nms_f32(reduce)");
  return ODX_OK;
}

// The same with an upper bound on the survivors: keep[i] = 1 for the first max_keep survivors only (what
// `nms(...)[:post_nms_top_n]` selects, rpn/inference.py:116-121) without walking the candidates behind them.
extern "C" int odx_nms_first_f32(const float* boxes_sorted, int R, float iou_threshold, int max_keep, unsigned char* keep,
                                 void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  if (R <= 0) return ODX_OK;
  ODX_REQUIRE(boxes_sorted && keep && workspace && max_keep > 0, "odx_nms_first_f32: null pointer or max_keep <= 0");
  ODX_REQUIRE(R <= 64 * 64 * 4, "odx_nms_first_f32: at most 16384 boxes");
  if (workspace_bytes < odx_nms_workspace_bytes(R)) {
    set_error("odx_nms_first_f32: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  const int words = (int)ceil_div(R, 64);
  hipStream_t s = as_stream(stream);
  ODX_PROPAGATE(zero_bytes(workspace, (size_t)odx_nms_workspace_bytes(R), s));
  ODX_PROPAGATE(zero_bytes(keep, (size_t)R, s));
  auto* mask = static_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words), dim3(64), 0, s, boxes_sorted, R,
                     iou_threshold, mask, words, (const int*)nullptr);
  ODX_CHECK_LAUNCH("odx_nms_first_f32(mask)");
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, mask, R, words, keep, (const int*)nullptr, max_keep);
  ODX_CHECK_LAUNCH("odx_nms_first_f32(reduce)");
  return ODX_OK;
}

// B independent box sets (the classes of an image after the score threshold) with one launch pair: set b holds counts[b]
// <= Rmax boxes, sorted by descending score, in slot b of boxes_sorted (B, Rmax, 4); keep (B, Rmax) gets 1 for the
// survivors of set b's greedy NMS and 0 elsewhere (incl. the unused tail of a slot).  counts lives on the device: the
// caller needs no host read between thresholding and suppression.
extern "C" int64_t odx_nms_batched_workspace_bytes(int Rmax, int B) {
  if (Rmax <= 0 || B <= 0) return 0;
  return (int64_t)B * odx_nms_workspace_bytes(Rmax);
}

static int nms_batched(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold, int max_keep,
                       unsigned char* keep, void* workspace, int64_t workspace_bytes, odx_stream_t stream);

extern "C" int odx_nms_batched_f32(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold,
                                   unsigned char* keep, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  return nms_batched(boxes_sorted, counts, Rmax, B, iou_threshold, 0, keep, workspace, workspace_bytes, stream);
}

// The same with an upper bound on each set's survivors (odx_nms_first_f32 per set): the proposals of a BATCH of images — B
// sets of up to 6000 candidates of which the first post_nms_top_n survivors are used (rpn/inference.py:116-121) — with one
// launch pair and no walk behind the last survivor needed.
extern "C" int odx_nms_batched_first_f32(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold,
                                         int max_keep, unsigned char* keep, void* workspace, int64_t workspace_bytes,
                                         odx_stream_t stream) {
  ODX_REQUIRE(max_keep > 0, "odx_nms_batched_first_f32: max_keep <= 0");
  return nms_batched(boxes_sorted, counts, Rmax, B, iou_threshold, max_keep, keep, workspace, workspace_bytes, stream);
}

static int nms_batched(const float* boxes_sorted, const int32_t* counts, int Rmax, int B, float iou_threshold, int max_keep,
                       unsigned char* keep, void* workspace, int64_t workspace_bytes, odx_stream_t stream) {
  if (Rmax <= 0 || B <= 0) return ODX_OK;
  ODX_REQUIRE(boxes_sorted && counts && keep && workspace, "odx_nms_batched_f32: null pointer");
  ODX_REQUIRE(Rmax <= 64 * 64 * 4 && B < 65536, "odx_nms_batched_f32: at most 16384 boxes per set, 65535 sets");
  if (workspace_bytes < odx_nms_batched_workspace_bytes(Rmax, B)) {
    set_error("odx_nms_batched_f32: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  const int words = (int)ceil_div(Rmax, 64);
  hipStream_t s = as_stream(stream);
  ODX_PROPAGATE(zero_bytes(workspace, (size_t)odx_nms_batched_workspace_bytes(Rmax, B), s));
  ODX_PROPAGATE(zero_bytes(keep, (size_t)B * Rmax, s));
  auto* mask = static_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words, (unsigned)B), dim3(64), 0, s, boxes_sorted, Rmax,
                     iou_threshold, mask, words, counts);
  ODX_CHECK_LAUNCH("odx_nms_batched_f32(mask)");
  hipLaunchKernelGGL(nms_reduce_kernel, dim3((unsigned)B), dim3(64), 0, s, mask, Rmax, words, keep, counts, max_keep);
  ODX_CHECK_LAUNCH("odx_nms_batched_f32(reduce)");
  return ODX_OK;
}

extern "C" int odx_paste_masks_u8(const float* masks, const float* boxes, int R, int S, int im_h, int im_w, float thresh,
                                  int padding, unsigned char* out, odx_stream_t stream) {
  if (R <= 0) return ODX_OK;
  ODX_REQUIRE(masks && boxes && out && S > 0 && im_h > 0 && im_w > 0 && padding >= 0, "odx_paste_masks_u8: bad argument");
  ODX_REQUIRE(R < 65536 && (int64_t)im_h * im_w < (1ll << 31), "odx_paste_masks_u8: too many detections or pixels");
  hipLaunchKernelGGL(paste_masks_kernel, dim3((unsigned)ceil_div((int64_t)im_h * im_w, 256), (unsigned)R), dim3(256), 0,
                     as_stream(stream), masks, boxes, R, S, im_h, im_w, thresh, padding, out);
  ODX_CHECK_LAUNCH("odx_paste_masks_u8");
  return ODX_OK;
}

// ---------------------------------------------------------------- epilogue of a trunk convolution
// y[n][c][p] = act(y[n][c][p] + bias[c] (+ residual[n][c][p])) in place over an NCHW map: what follows every convolution of
// the frozen-batch-norm ResNet trunk (the norm is folded into the weights and this bias) as ONE pass instead of the library's
// bias kernel + an addition + a ReLU.  blockIdx.y = (n, c) plane; 16-byte accesses when the plane allows.  f32, or a 16-bit
// type rounded after every addition as the separate ops round (bit-identical to them); NaN passes through the ReLU.
struct EpiF32 {
  typedef float T;
  static constexpr int V = 4;
  static __device__ __forceinline__ float load(const T* p) { return *p; }
  static __device__ __forceinline__ void store(T* p, float v) { *p = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
struct EpiBF16 {
  typedef unsigned short T;
  static constexpr int V = 8;
  static __device__ __forceinline__ float load(const T* p) { return __uint_as_float((unsigned)*p << 16); }
  static __device__ __forceinline__ unsigned short bits(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);     // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                     // round to nearest even
  }
  static __device__ __forceinline__ void store(T* p, float v) { *p = bits(v); }
  static __device__ __forceinline__ float round(float v) { return __uint_as_float((unsigned)bits(v) << 16); }
};
struct EpiF16 {
  typedef _Float16 T;
  static constexpr int V = 8;
  static __device__ __forceinline__ float load(const T* p) { return (float)*p; }
  static __device__ __forceinline__ void store(T* p, float v) { *p = (_Float16)v; }
  static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
};

template <typename E>
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(typename E::T* __restrict__ y, const typename E::T* __restrict__ bias,
                                                            const typename E::T* __restrict__ res, int64_t planes, int C, int64_t HW, int relu,
                                                            int vec) {
  typedef typename E::T T;
  constexpr int V = E::V;                 // elements per 16 bytes
  typedef T VecT __attribute__((ext_vector_type(V)));
  // (a grid's y extent stops at 65535: 300 RoIs x 512 channels of the conv5 head are 153 600 planes — a workgroup row walks them)
  for (int64_t plane = blockIdx.y; plane < planes; plane += gridDim.y) {
  const float b = E::load(bias + plane % C);
  T* yp = y + plane * HW;
  const T* rp = res ? res + plane * HW : nullptr;
  auto one = [&](T* dst, const T* r) {
    float v = E::round(E::load(dst) + b);
    if (r) v = E::round(v + E::load(r));
    if (relu) v = v < 0.f ? 0.f : v;
    E::store(dst, v);
  };
  if (vec) {
    const int64_t nv = HW / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
      VecT v = reinterpret_cast<const VecT*>(yp)[i];
      VecT r;
      if (rp) r = reinterpret_cast<const VecT*>(rp)[i];
      T* e = reinterpret_cast<T*>(&v);
      const T* re = reinterpret_cast<const T*>(&r);
#pragma unroll
      for (int q = 0; q < V; ++q) one(e + q, rp ? re + q : nullptr);
      reinterpret_cast<VecT*>(yp)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) one(yp + i, rp ? rp + i : nullptr);
  }
  }
}

template <typename E>
static int bias_act_nchw(void* y, const void* bias, const void* residual, int64_t N, int C, int64_t HW, int relu, odx_stream_t stream,
                         const char* who) {
  if (N <= 0 || C <= 0 || HW <= 0) return ODX_OK;
  ODX_REQUIRE(y && bias, "%s: null pointer", who);
  const int64_t planes = N * C;
  const int vec = (HW * (int64_t)sizeof(typename E::T)) % 16 == 0 && aligned16(y) && (!residual || aligned16(residual));
  const int64_t per = vec ? HW / E::V : HW;
  int64_t gx = ceil_div(per, 256);
  if (gx > 64) gx = 64;
  typedef typename E::T T;
  hipLaunchKernelGGL((bias_act_nchw_kernel<E>), dim3((unsigned)gx, (unsigned)(planes < 65535 ? planes : 65535)), dim3(256), 0, as_stream(stream),
                     static_cast<T*>(y), static_cast<const T*>(bias), static_cast<const T*>(residual), planes, C, HW, relu, vec);
  ODX_CHECK_LAUNCH(who);
  return ODX_OK;
}

// The stem's tail in one pass: relu(x + bias) of the 7 x 7 convolution's NCHW output, the 3 x 3 / stride 2 / padding 1 max pooling
// of it, written as the NHWC ROWS the stages' row GEMMs read (rows (B Ho Wo, C), row stride ldr), with max |out| left for the first
// packing (amax: bits of a non-negative float, zero on entry; nullptr: not wanted).  Before: the epilogue pass in place (read +
// write of the 245-MB map of eight 600 x 800 images), the library's pooling (read it again, 177 us), a permuting copy of the pooled
// map and a pass for its maximum.  Per element the arithmetic of bias_act_nchw_kernel (16-bit types: the sum rounded once), and
// a maximum of rounded values is exact: the same numbers.  A workgroup owns 64 output columns of one output row of one image, 64
// channels at a time; the transposition runs through LDS (reads coalesced along w, writes along c).
template <typename E, bool PAIRS>
__global__ __launch_bounds__(256) void stem_pool_rows_kernel(const typename E::T* __restrict__ x, const typename E::T* __restrict__ bias, int C,
                                                             int H, int W, int Ho, int Wo, typename E::T* __restrict__ rows, int64_t ldr,
                                                             unsigned int* __restrict__ amax) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, oh = blockIdx.y, ow0 = blockIdx.x * 64, tid = threadIdx.x;
  const int owl = tid & 63, cb = tid >> 6;
  unsigned int mx = 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int ow = ow0 + owl;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int cl = cb + 4 * j, c = c0 + cl;
      float m = 0.f;                                            // (every window holds its centre, and relu(.) >= 0)
      if (PAIRS) {
        // even W, f32: a lane reads the aligned PAIR (2 ow, 2 ow + 1) of each of the three rows — whole lines per wave instead of
        // three stride-2 reads — and takes 2 ow - 1 from its left neighbour (the wave's first lane reads it itself)
        const bool ok = ow < Wo && c < C;
        const float bv = ok ? E::load(bias + c) : 0.f;
        const typename E::T* plane = x + ((int64_t)b * C + (c < C ? c : 0)) * H * W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int ih = 2 * oh - 1 + dy;
          const bool rok = ok && ih >= 0 && ih < H;
          float2 pr = {0.f, 0.f};
          if (rok) pr = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(plane) + (int64_t)ih * W + 2 * ow);
          float left = __shfl_up(pr.y, 1);
          if (owl == 0 && rok && ow > 0) left = reinterpret_cast<const float*>(plane)[(int64_t)ih * W + 2 * ow - 1];
          if (rok) {
            m = fmaxf(m, fmaxf(pr.x + bv, pr.y + bv));
            if (ow > 0) m = fmaxf(m, left + bv);
          }
        }
      } else if (ow < Wo && c < C) {
        const float bv = E::load(bias + c);
        const typename E::T* plane = x + ((int64_t)b * C + c) * H * W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int ih = 2 * oh - 1 + dy;
          if (ih < 0 || ih >= H) continue;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int iw = 2 * ow - 1 + dx;
            if (iw < 0 || iw >= W) continue;
            const float v = E::round(E::load(plane + (int64_t)ih * W + iw) + bv);
            m = fmaxf(m, v);                                    // (fmaxf(0, v) is the ReLU; a NaN input is not kept — nor by max_pool2d's >)
          }
        }
      }
      tile[owl][cl] = m;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int ol = cb + 4 * j, ow2 = ow0 + ol, c = c0 + owl;
      if (ow2 < Wo && c < C) {
        const float v = tile[ol][owl];
        E::store(rows + (((int64_t)b * Ho + oh) * Wo + ow2) * ldr + c, v);
        mx = max(mx, __float_as_uint(v));
      }
    }
    __syncthreads();
  }
  if (amax != nullptr) {
    __shared__ unsigned int wmx[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, off));
    if ((tid & 63) == 0) wmx[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      mx = max(max(wmx[0], wmx[1]), max(wmx[2], wmx[3]));
      if (mx) atomicMax(amax, mx);
    }
  }
}

template <typename E>
static int stem_pool_rows(const void* x, const void* bias, int B, int C, int H, int W, void* rows, int64_t ldr, float* meta,
                          odx_stream_t stream, const char* who) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return ODX_OK;
  ODX_REQUIRE(x && bias && rows && ldr >= C, "%s: null pointer or row stride below the channel count", who);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  ODX_REQUIRE(Ho <= 65535 && B <= 65535, "%s: at most 65535 output rows / images per call", who);
  typedef typename E::T T;
  const dim3 grid((unsigned)ceil_div(Wo, 64), (unsigned)Ho, (unsigned)B);
  unsigned int* amax = meta ? reinterpret_cast<unsigned int*>(meta + 1) : nullptr;
  if (sizeof(T) == 4 && W % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0)
    hipLaunchKernelGGL((stem_pool_rows_kernel<E, true>), grid, dim3(256), 0, as_stream(stream), static_cast<const T*>(x),
                       static_cast<const T*>(bias), C, H, W, Ho, Wo, static_cast<T*>(rows), ldr, amax);
  else
    hipLaunchKernelGGL((stem_pool_rows_kernel<E, false>), grid, dim3(256), 0, as_stream(stream), static_cast<const T*>(x),
                       static_cast<const T*>(bias), C, H, W, Ho, Wo, static_cast<T*>(rows), ldr, amax);
  ODX_CHECK_LAUNCH(who);
  return ODX_OK;
}

extern "C" int odx_stem_pool_rows_f32(const float* x, const float* bias, int B, int C, int H, int W, float* rows, int64_t ldr, float* meta,
                                      odx_stream_t stream) {
  return stem_pool_rows<EpiF32>(x, bias, B, C, H, W, rows, ldr, meta, stream, "odx_stem_pool_rows_f32");
}

extern "C" int odx_stem_pool_rows_16(const void* x, const void* bias, int is_bf16, int B, int C, int H, int W, void* rows, int64_t ldr,
                                     odx_stream_t stream) {
  return is_bf16 ? stem_pool_rows<EpiBF16>(x, bias, B, C, H, W, rows, ldr, nullptr, stream, "odx_stem_pool_rows_16")
                 : stem_pool_rows<EpiF16>(x, bias, B, C, H, W, rows, ldr, nullptr, stream, "odx_stem_pool_rows_16");
}

extern "C" int odx_bias_act_nchw_f32(float* y, const float* bias, const float* residual, int64_t N, int C, int64_t HW, int relu,
                                     odx_stream_t stream) {
  return bias_act_nchw<EpiF32>(y, bias, residual, N, C, HW, relu, stream, "odx_bias_act_nchw_f32");
}

extern "C" int odx_bias_act_nchw_16(void* y, const void* bias, const void* residual, int is_bf16, int64_t N, int C, int64_t HW, int relu,
                                    odx_stream_t stream) {
  return is_bf16 ? bias_act_nchw<EpiBF16>(y, bias, residual, N, C, HW, relu, stream, "odx_bias_act_nchw_16")
                 : bias_act_nchw<EpiF16>(y, bias, residual, N, C, HW, relu, stream, "odx_bias_act_nchw_16");
}

// Labels of the n visible anchors of an image against its G ground-truth boxes (RPNHarvester.prepare): per anchor the best overlap
// (ious), the box it is associated with (assoc (n, 4)), overlap below neg_thr / above pos_thr (neg_mask, over: bytes), per box the
// anchors that are its best among those associated with it (extra (G, n) bytes), and the counts the host decides from —
// counters (int32): [A: candidates per anchor type][G: over-threshold anchors sharing a coordinate with box j][G: box j has
// anchors][A: over-threshold anchors per type][G x A: extras of box j per type].  cls (n) int64: anchor types.  G <= 64.
// workspace: G uint32 (scratch).
extern "C" int odx_rpn_label_f32(const float* gt, int G, const float* anchors, const int64_t* cls, int n, int A, float neg_thr,
                                 float pos_thr, float* ious, float* assoc, unsigned char* neg_mask, unsigned char* over,
                                 unsigned char* extra, int32_t* counters, void* workspace, odx_stream_t stream) {
  ODX_REQUIRE(G >= 1 && G <= LABEL_MAX_GT && A >= 1, "odx_rpn_label_f32: 1..%d ground-truth boxes per image", LABEL_MAX_GT);
  hipStream_t s = as_stream(stream);
  const size_t nc = (size_t)(2 * A + 2 * G + G * A);
  ODX_PROPAGATE(zero_bytes(counters, nc * sizeof(int32_t), s));
  if (n <= 0) return ODX_OK;
  ODX_REQUIRE(gt && anchors && cls && ious && assoc && neg_mask && over && extra && counters && workspace, "odx_rpn_label_f32: null pointer");
  ODX_PROPAGATE(zero_bytes(workspace, (size_t)G * sizeof(unsigned int), s));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(rpn_label_kernel, dim3(blocks), dim3(256), 0, s, gt, G, anchors, cls, n, A, neg_thr, pos_thr, ious, assoc, neg_mask, over,
                     static_cast<unsigned int*>(workspace), counters);
  ODX_CHECK_LAUNCH("odx_rpn_label_f32");
  hipLaunchKernelGGL(rpn_label_extra_kernel, dim3(blocks), dim3(256), 0, s, gt, G, cls, n, A, ious, assoc,
                     static_cast<const unsigned int*>(workspace), extra, counters);
  ODX_CHECK_LAUNCH("odx_rpn_label_f32(extra)");
  return ODX_OK;
}

// Labels of the R proposals of an image against its G ground-truth boxes (DetectorHarvester.prepare): boxes clamped to the image
// (prop (R, 4) out), per-class maximum overlap (overlap (R, C)), the regression pairs (sel (G, R) bytes: proposal r regresses onto
// box j — its class's overlap above reg_min and box j the FIRST box of largest overlap, if that is positive), the negatives'
// candidate flags of the n_in listed classes (cmask (R, n_in) bytes: overlap below neg_thr) and counters (int32): [G: pairs per
// box][n_in: candidates per listed class].  labels0 (G) int32: class of box j, 0-based; in_image (n_in) int32.  G <= 64.
extern "C" int odx_det_label_f32(const float* gt, const int32_t* labels0, int G, const float* proposals, int R, int C, float img_w,
                                 float img_h, float reg_min, float neg_thr, const int32_t* in_image, int n_in, float* prop, float* overlap,
                                 unsigned char* sel, unsigned char* cmask, int32_t* counters, odx_stream_t stream) {
  ODX_REQUIRE(G >= 1 && G <= LABEL_MAX_GT && C >= 1 && n_in >= 0, "odx_det_label_f32: 1..%d ground-truth boxes per image", LABEL_MAX_GT);
  hipStream_t s = as_stream(stream);
  ODX_PROPAGATE(zero_bytes(counters, (size_t)(G + n_in) * sizeof(int32_t), s));
  if (R <= 0) return ODX_OK;
  ODX_REQUIRE(gt && labels0 && proposals && prop && overlap && sel && counters && (n_in == 0 || (in_image && cmask)), "odx_det_label_f32: null pointer");
  hipLaunchKernelGGL(det_label_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, gt, labels0, G, proposals, R, C, img_w, img_h, reg_min,
                     neg_thr, in_image, n_in, prop, overlap, sel, cmask, counters);
  ODX_CHECK_LAUNCH("odx_det_label_f32");
  return ODX_OK;
}

// out (n, 4) = the box-regression targets of n (example, target) pairs (x1, y1, x2, y2 rows).
extern "C" int odx_box_targets_f32(const float* examples, const float* targets, int64_t n, float* out, odx_stream_t stream) {
  if (n <= 0) return ODX_OK;
  ODX_REQUIRE(examples && targets && out, "odx_box_targets_f32: null pointer");
  hipLaunchKernelGGL(box_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), examples, targets, n, out);
  ODX_CHECK_LAUNCH("odx_box_targets_f32");
  return ODX_OK;
}
