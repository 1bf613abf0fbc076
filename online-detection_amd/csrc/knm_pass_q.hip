// The conjugate-gradient pass  out = K' (K v + w)  over a K_nM shard stored in one of the COMPACT formats of
// gauss_h2.hip (odx_gauss_knm_h2_store): 24-bit fixed point (a u16 plane + a u8 plane, 3 bytes per entry) or bf16
// (2 bytes per entry).  The pass is HBM-bound and already streams at the chip's copy rate (knm_pass.hip: 6.2 TB/s of the
// 6.3 it delivers), so the only lever left on its time is the number of bytes per entry.
//
// Same persistent scheme as knm_pass_kernel / knm_pass2_kernel: a workgroup streams blocks of R rows; thread t owns the
// 4-column chunks t, t + NT, ... (CH of them) of every row — 8 bytes of the u16 plane and, for the 24-bit format, 4 bytes
// of the u8 plane per chunk: the same columns-per-thread as the f32 kernel, so the same (NT, CH) cover a row with the same
// few idle lanes, at 3 instead of 4 registers per chunk (twice the rows per block in the same budget) —, keeps the
// running column sums of K' t in registers as f64 and the R x CH chunks of the current block in registers as loaded
// (never as doubles); v sits in LDS as f64.  Phase 1 forms the R row dots and reduces
// them over the workgroup, phase 2 adds K[r, cols] t_r into the column sums and re-issues the next block's loads chunk by
// chunk.  K is read through buffer descriptors (wave-uniform base and length per (row, chunk column) window, one 32-bit
// per-lane offset): chunks past a row's end and rows past n read as zero by the hardware range check — no branches in
// the streaming loop.  Slab per workgroup + fixed-order reduce: bitwise reproducible.
//
// Decoding: a 24-bit entry is assembled from its three bytes by ONE v_perm_b32 and converted by v_cvt_f64_u32 (exact);
// the 2^-24 of the fixed-point scale is folded into v (phase 1) and into the slab write (phase 2).  A bf16 entry is a
// shift / mask and v_cvt_f64_f32.
#include <algorithm>
#include <stdlib.h>

#include "odx_internal.h"

namespace odx {

typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2q __attribute__((ext_vector_type(2)));
typedef double f64x2q __attribute__((ext_vector_type(2)));

enum { QF_U24 = 1, QF_BF16 = 2 };      // = ODX_KNM_U24 / ODX_KNM_BF16

// A thread's chunk: CW = 4 consecutive columns of a row — two dwords of the u16 plane, one of the u8 plane (the f32 kernel's
// columns per thread, so the same (NT, CH) cover a row; 8-column chunks with 16-byte loads measured no faster).
constexpr int QCW = 4;
template <int FMT, int CW>
struct QChunk {
  unsigned hi[CW / 2];
  unsigned lo[CW / 4];
};

// entry e (0 .. CW - 1) of a chunk as a double: the integer q for QF_U24 (value = q 2^-24), the value itself for QF_BF16
template <int FMT, int CW>
__device__ __forceinline__ double q_entry(const QChunk<FMT, CW>& k, int e) {
  const unsigned h = k.hi[e >> 1];
  if (FMT == QF_U24) {
    // v_perm_b32: selector bytes 0..3 pick bytes of the second source (the low-byte dword), 4..7 bytes of the first (the
    // u16 pair), 0x0c a zero byte: result = [low byte (e & 3) | u16 << 8]
    const unsigned sel = ((e & 1) ? 0x0c070600u : 0x0c050400u) | (unsigned)(e & 3);
    return (double)__builtin_amdgcn_perm(h, k.lo[e >> 2], sel);
  }
  return (double)__uint_as_float((e & 1) ? (h & 0xffff0000u) : (h << 16));
}

template <int NT, int CH, int R, int NV, int FMT, int WPE>
__device__ __forceinline__ void knm_passq_body(const unsigned short* __restrict__ Khi, int64_t ldk,
                                               const unsigned char* __restrict__ Klo, int64_t ldlo, int64_t n,
                                               int64_t M, const double* __restrict__ v1, const double* __restrict__ v2,
                                               const double* __restrict__ w, double* __restrict__ slab, int64_t slab_ld,
                                               int wg, int nwg) {
  constexpr int NW = NT / 64;
  constexpr int CW = QCW;
  extern __shared__ __attribute__((aligned(16))) double vsq[];       // [NV][vcap]
  __shared__ double red[2][NW][NV * R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = (int)((M + CW - 1) / CW);
  const int vcap = (nchunk + 1) * CW;      // one chunk of zeros behind the row: where the chunks past the row's end point
  const int64_t nblk = (n + R - 1) / R;
  const double vscale = FMT == QF_U24 ? 5.9604644775390625e-08 : 1.0;      // 2^-24 (exact)
  for (int i = tid; i < vcap; i += NT) {
    vsq[i] = (v1 != nullptr && i < M) ? v1[i] * vscale : 0.0;
    if (NV == 2) vsq[vcap + i] = i < M ? v2[i] * vscale : 0.0;
  }
  double acc[NV][CH][CW];
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < CW; ++e) acc[q][c][e] = 0.0;

  // Addressing.  ONE buffer descriptor per plane and row block, built from scalars once per block: base = the block's
  // first row, length = the bytes of its rows that exist.  A load takes [the lane's place inside a chunk column] as its
  // vector offset and [row inside the block] x [row stride] + [chunk column] as its scalar offset; the hardware checks
  // their SUM against the length (measured: with a one-row length every row but the first read as zero), so nothing is
  // read past the block's last row.  (A descriptor per (row, chunk column) window — the first form of this kernel — cost
  // ~8 scalar instructions per load: ~1000 per wave and block against ~700 vector ones, and the scalar unit is shared by
  // the CU's four SIMDs.)  Rows past n are read as the block's last existing row and their row dots are set to zero
  // before phase 2, so they add nothing.  A lane whose chunk lies past the row's end reads the start of the next row
  // instead of zeros: its v entries are forced to zero in phase 1 and its column sums are never stored.
  QChunk<FMT, CW> kr[R][CH];
  const int rowb_hi = (int)ldk * 2, rowb_lo = (int)ldlo;
  const int voff_hi = tid * (2 * CW), voff_lo = tid * CW;      // the only per-lane part of an address: the chunk inside a
                                                               // chunk column; row and chunk column ride on the scalar offset
  __amdgpu_buffer_rsrc_t rs_hi, rs_lo;
  int rows_open = R;
  auto open_block = [&](int64_t blk) {
    const int64_t row0 = blk * R;
    rows_open = (int)(n - row0 < R ? n - row0 : R);
    rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Khi + row0 * ldk), (short)0, rows_open * rowb_hi, 0x00020000);
    if (FMT == QF_U24)
      rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Klo + row0 * ldlo), (short)0, rows_open * rowb_lo, 0x00020000);
  };
  auto load_block = [&](int c) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rr = r < rows_open ? r : rows_open - 1;
      const u32x2q th = __builtin_amdgcn_raw_buffer_load_b64(rs_hi, voff_hi, rr * rowb_hi + c * (NT * 2 * CW), 0);
      kr[r][c].hi[0] = th[0], kr[r][c].hi[1] = th[1];
      if (FMT == QF_U24) kr[r][c].lo[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_lo, voff_lo, rr * rowb_lo + c * (NT * CW), 0);
    }
  };

  int64_t blk = wg;
  if (blk < nblk) {
    open_block(blk);
#pragma unroll
    for (int c = 0; c < CH; ++c) load_block(c);
  }
  __syncthreads();  // vsq is complete
  int pp = 0;
  for (; blk < nblk; blk += nwg) {
    double t[NV][R];
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) t[q][r] = 0.0;
    if (v1 != nullptr) {
      // phase 1: row dots.  An opaque zero in the LDS index keeps the (loop-invariant) reads of v inside the loop — hoisted
      // they would hold NV x CH x CW doubles for good (knm_pass2_kernel, same reason).
      int zofs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zofs));
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          // a chunk past the row's end holds the next row's entries, not zeros: it is multiplied by the zero chunk behind v
          // (one v_min per chunk; a select per loaded double of v cost 8 v_cndmask per chunk and vector)
          const int ch = tid + c * NT + zofs;
          const int vi = (ch < nchunk ? ch : nchunk) * CW;
          double vv[CW];
#pragma unroll
          for (int u = 0; u < CW / 2; ++u) {
            const f64x2q a = *reinterpret_cast<const f64x2q*>(&vsq[q * vcap + vi + 2 * u]);
            vv[2 * u] = a[0];
            vv[2 * u + 1] = a[1];
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < CW; ++e) t[q][r] = fma(q_entry<FMT, CW>(kr[r][c], e), vv[e], t[q][r]);
        }
        __builtin_amdgcn_sched_barrier(0);      // one chunk column's decoded entries at a time (register pressure)
      }
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double s = t[q][r];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
          if (lane == 0) red[pp][wave][q * R + r] = s;
        }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          double s = 0.0;
#pragma unroll
          for (int u = 0; u < NW; ++u) s += red[pp][u][q * R + r];
          t[q][r] = s;
        }
      pp ^= 1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = blk * R + r;
      if (row >= n) {                        // (the last block only) a repeated row, not a zero one, was read for it
#pragma unroll
        for (int q = 0; q < NV; ++q) t[q][r] = 0.0;
      } else if (NV == 1 && w != nullptr) {
        t[0][r] += w[row];
      }
    }
    // the decoded doubles of phase 1 must not stay live into phase 2 (R x CH x CW doubles: spills): make the raw registers
    // opaque here, phase 2 decodes again
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < CW / 2; ++u) asm volatile("" : "+v"(kr[r][c].hi[u]));
        if (FMT == QF_U24) {
#pragma unroll
          for (int u = 0; u < CW / 4; ++u) asm volatile("" : "+v"(kr[r][c].lo[u]));
        }
      }
    // phase 2: column sums, and the next block's loads re-issued chunk by chunk
    const int64_t nxt = blk + nwg;
    if (nxt < nblk) open_block(nxt);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const double kd = q_entry<FMT, CW>(kr[r][c], e);
#pragma unroll
          for (int q = 0; q < NV; ++q) acc[q][c][e] = fma(kd, t[q][r], acc[q][c][e]);
        }
      // (unconditional — behind the last block the loads re-read it and nobody waits for them: a branch around the loads
      // turns every register of the block into a loop-carried select, ~2 register copies per register and trip)
      load_block(c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* my = slab + (int64_t)wg * slab_ld * NV;
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = tid + c * NT;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < CW; ++e)
          if ((int64_t)ch * CW + e < slab_ld) my[q * slab_ld + (int64_t)ch * CW + e] = acc[q][c][e] * vscale;
      }
    }
}

template <int NT, int CH, int R, int NV, int FMT, int WPE>
__global__ __launch_bounds__(NT, WPE) void knm_passq_kernel(const unsigned short* __restrict__ Khi, int64_t ldk,
                                                       const unsigned char* __restrict__ Klo, int64_t ldlo, int64_t n,
                                                       int64_t M, const double* __restrict__ v1, const double* __restrict__ v2,
                                                       const double* __restrict__ w, double* __restrict__ slab, int64_t slab_ld) {
  knm_passq_body<NT, CH, R, NV, FMT, WPE>(Khi, ldk, Klo, ldlo, n, M, v1, v2, w, slab, slab_ld, (int)blockIdx.x, (int)gridDim.x);
}

// The classes of a batch by one launch (blockIdx.y = class): class b's block is walked by grid[b] workgroups exactly as its
// own odx_knm_fwd_bwd_q launch would walk it (same workgroup -> rows assignment, same slab order): its sums are the
// single-class call's bit for bit.  Vectors lie vstride apart, a class's slabs slab_stride apart.
struct PassBatchQ {
  const unsigned short* Khi[ODX_MAX_ZBATCH];
  const unsigned char* Klo[ODX_MAX_ZBATCH];
  int64_t ldk[ODX_MAX_ZBATCH];
  int64_t ldlo[ODX_MAX_ZBATCH];
  int64_t n[ODX_MAX_ZBATCH];
  int M[ODX_MAX_ZBATCH];
  int grid[ODX_MAX_ZBATCH];
};

template <int NT, int CH, int R, int FMT, int WPE = (NT >= 1024 ? 4 : 2)>
__global__ __launch_bounds__(NT, WPE) void knm_passq_batched_kernel(PassBatchQ pb, const double* __restrict__ v, int64_t vstride,
                                                                    double* __restrict__ slab, int64_t slab_ld, int64_t slab_stride) {
  const int b = blockIdx.y;
  if ((int)blockIdx.x >= pb.grid[b]) return;
  knm_passq_body<NT, CH, R, 1, FMT, WPE>(pb.Khi[b], pb.ldk[b], pb.Klo[b], pb.ldlo[b], pb.n[b], pb.M[b], v + (int64_t)b * vstride, nullptr,
                                         nullptr, slab + (int64_t)b * slab_stride, slab_ld, (int)blockIdx.x, pb.grid[b]);
}

// ---------------------------------------------------------------- two free-running halves (one vector, 8192 < M <= 10240)
// ONE 512-thread workgroup per CU whose two halves (waves 0-3, waves 4-7) each stream their own row blocks over ALL
// columns (256 threads x CH chunks), share one copy of v in LDS and never wait for each other: a half's four waves meet
// at a counter in LDS (one ds_add per wave and row block, a short spin), not at the workgroup's barrier.  The halves
// drift apart, so one multiplies while the other waits on memory — what two independent 256-thread workgroups per CU
// gain (5.7 against 5.2 TB/s alone for one 512-thread workgroup that waits and computes in step) without needing the
// CU's whole LDS for two copies of v: beside the preconditioner stream of the headline job a CU that holds one foreign
// workgroup still takes the pass's ONE workgroup, where it would take only one of two and push the other behind the
// whole persistent grid (5.10 TB/s in bench.py).  (The same halves kept in step by workgroup barriers, half a block
// apart, ran at 3.3 TB/s: only one half's loads are then in flight at a time.)  Every wave of a half makes the same
// number of trips, all eight waves are resident by construction, and the spin is bounded.  Bitwise reproducible: a
// half's column sums depend only on its own rows, the slab holds one vector per half, the fixed-order reduce adds
// 2 x grid of them.
template <int CH, int R, int FMT>
__device__ __forceinline__ void knm_passq_stag_body(const unsigned short* __restrict__ Khi, int64_t ldk,
                                                    const unsigned char* __restrict__ Klo, int64_t ldlo, int64_t n,
                                                    int64_t M, const double* __restrict__ v1,
                                                    const double* __restrict__ w, double* __restrict__ slab,
                                                    int64_t slab_ld, int wg, int nwg) {
  constexpr int NT = 256, CW = QCW;
  extern __shared__ __attribute__((aligned(16))) double vsq[];
  __shared__ double red[2][2][4][R];                 // [half][ping-pong][wave of the half][row]
  const int tid = threadIdx.x, lane = tid & 63;
  // (wave-uniform by construction; readfirstlane tells the compiler so — otherwise every load of a half's block is wrapped
  // in a waterfall loop over "possibly different" descriptors and the loop drains all loads at its back edge)
  const int h = __builtin_amdgcn_readfirstlane(tid >> 8), hw = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int ht = tid & 255;
  const int nchunk = (int)((M + CW - 1) / CW);
  // v in LDS is zero-filled up to the CH * NT chunks the threads walk: a chunk past the row's end multiplies whatever the
  // load returned (finite integers of the next row, or the range check's zeros) by 0 — no select, no index clamp per chunk
  constexpr int vcap = CH * NT * CW;
  const int64_t nblk = (n + R - 1) / R;
  const double vscale = FMT == QF_U24 ? 5.9604644775390625e-08 : 1.0;
  for (int i = tid; i < vcap; i += 512) vsq[i] = (v1 != nullptr && i < M) ? v1[i] * vscale : 0.0;
  const int voff_hi = ht * (2 * CW), voff_lo = ht * CW;
  double acc[CH][CW];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < CW; ++e) acc[c][e] = 0.0;
  QChunk<FMT, CW> kr[R][CH];
  const int rowb_hi = (int)ldk * 2, rowb_lo = (int)ldlo;
  __amdgpu_buffer_rsrc_t rs_hi, rs_lo;
  int rows_open = R;
  auto open_block = [&](int64_t blk) {
    const int64_t row0 = blk * R;
    rows_open = (int)(n - row0 < R ? n - row0 : R);
    rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Khi + row0 * ldk), (short)0, rows_open * rowb_hi, 0x00020000);
    if (FMT == QF_U24)
      rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Klo + row0 * ldlo), (short)0, rows_open * rowb_lo, 0x00020000);
  };
  auto load_block = [&](int c) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rr = r < rows_open ? r : rows_open - 1;
      const u32x2q th = __builtin_amdgcn_raw_buffer_load_b64(rs_hi, voff_hi, rr * rowb_hi + c * (NT * 2 * CW), 0);
      kr[r][c].hi[0] = th[0], kr[r][c].hi[1] = th[1];
      if (FMT == QF_U24) kr[r][c].lo[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_lo, voff_lo, rr * rowb_lo + c * (NT * CW), 0);
    }
  };
  // half h walks blocks first, first + step, ...: the same trip count for every wave of the half
  __shared__ unsigned int arrived[2];
  if (tid < 2) arrived[tid] = 0u;
  const int64_t first = 2 * (int64_t)wg + h, step = 2 * (int64_t)nwg;
  const int64_t mine = first < nblk ? (nblk - first + step - 1) / step : 0;
  if (mine > 0) {
    open_block(first);
#pragma unroll
    for (int c = 0; c < CH; ++c) load_block(c);
  }
  __syncthreads();  // vsq and the counters are complete; the only workgroup-wide barrier
  int pp = 0;
  for (int64_t k = 0; k < mine; ++k) {
    const int64_t blk = first + k * step;
    double t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = 0.0;
    if (v1 != nullptr) {
      // ---- phase 1: this half's row dots over all columns, one partial per wave
      int zofs;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zofs));
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int vi = (ht + c * NT + zofs) * CW;
        const f64x2q a = *reinterpret_cast<const f64x2q*>(&vsq[vi]);
        const f64x2q b = *reinterpret_cast<const f64x2q*>(&vsq[vi + 2]);
        const double v0 = a[0], v1_ = a[1], v2_ = b[0], v3 = b[1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          t[r] = fma(q_entry<FMT, CW>(kr[r][c], 0), v0, t[r]);
          t[r] = fma(q_entry<FMT, CW>(kr[r][c], 1), v1_, t[r]);
          t[r] = fma(q_entry<FMT, CW>(kr[r][c], 2), v2_, t[r]);
          t[r] = fma(q_entry<FMT, CW>(kr[r][c], 3), v3, t[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double sum = t[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) red[h][pp][hw][r] = sum;
      }
      // the half's four waves meet here: arrive (release: the partial sums above are in LDS first), then wait until the
      // counter shows all four arrivals of this trip (acquire).  red[] is double-buffered: a wave can be at most one trip
      // ahead of the slowest wave of its half, which is then still reading the other buffer.
      // (relaxed LDS atomics between compiler barriers: a wave's LDS operations execute in order, which is all the
      // ordering needed here; release / acquire semantics would also wait for the wave's outstanding GLOBAL loads — the
      // next block's prefetch — at every rendezvous: 4.2 TB/s)
      asm volatile("" ::: "memory");
      if (lane == 0) {
        __hip_atomic_fetch_add(&arrived[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int want = 4u * (unsigned int)(k + 1);
        // (all eight waves of the workgroup are co-resident by construction, so the wait ends; it is bounded only so that a
        // broken build cannot hang the device — and a bound that is hit must not fall through to sums the other waves have
        // not written yet: the wave traps, the launch fails, the host sees an error instead of a silently wrong K'(K v))
        bool met = false;
        for (int spin = 0; spin < (1 << 26); ++spin) {
          if (__hip_atomic_load(&arrived[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) { met = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!met) __builtin_trap();
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r) t[r] = ((red[h][pp][0][r] + red[h][pp][1][r]) + red[h][pp][2][r]) + red[h][pp][3][r];
      pp ^= 1;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          asm volatile("" : "+v"(kr[r][c].hi[0]));
          asm volatile("" : "+v"(kr[r][c].hi[1]));
          if (FMT == QF_U24) asm volatile("" : "+v"(kr[r][c].lo[0]));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = blk * R + r;
      if (row >= n) t[r] = 0.0;                      // a repeated row, not a zero one, was read for it
      else if (w != nullptr) t[r] += w[row];
    }
    // ---- phase 2: the column sums, and the next block's loads chunk by chunk
    const bool more = k + 1 < mine;
    if (more) open_block(blk + step);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[c][e] = fma(q_entry<FMT, CW>(kr[r][c], e), t[r], acc[c][e]);
      // (unconditional: a branch around the loads makes every register of the block a loop-carried select — 120 register
      // copies per trip; after the last block the loads re-read it and nobody waits for them)
      load_block(c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* my = slab + (2 * (int64_t)wg + h) * slab_ld;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = ht + c * NT;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < CW; ++e)
        if ((int64_t)ch * CW + e < slab_ld) my[(int64_t)ch * CW + e] = acc[c][e] * vscale;
    }
  }
}

template <int CH, int R, int FMT>
__global__ __launch_bounds__(512, 2) void knm_passq_stag_kernel(const unsigned short* __restrict__ Khi, int64_t ldk,
                                                                const unsigned char* __restrict__ Klo, int64_t ldlo, int64_t n,
                                                                int64_t M, const double* __restrict__ v1,
                                                                const double* __restrict__ w, double* __restrict__ slab,
                                                                int64_t slab_ld) {
  knm_passq_stag_body<CH, R, FMT>(Khi, ldk, Klo, ldlo, n, M, v1, w, slab, slab_ld, (int)blockIdx.x, (int)gridDim.x);
}

template <int CH, int R, int FMT>
__global__ __launch_bounds__(512, 2) void knm_passq_stag_batched_kernel(PassBatchQ pb, const double* __restrict__ v, int64_t vstride,
                                                                        double* __restrict__ slab, int64_t slab_ld,
                                                                        int64_t slab_stride) {
  const int b = blockIdx.y;
  if ((int)blockIdx.x >= pb.grid[b]) return;
  knm_passq_stag_body<CH, R, FMT>(pb.Khi[b], pb.ldk[b], pb.Klo[b], pb.ldlo[b], pb.n[b], pb.M[b], v + (int64_t)b * vstride, nullptr,
                                  slab + (int64_t)b * slab_stride, slab_ld, (int)blockIdx.x, pb.grid[b]);
}

struct QCfg {
  int nt, ch, r, wg_per_cu;      // nt == 0: the two-halves kernel (512 threads, ch chunks per thread of a half)
};

// One-vector configurations.  Up to 8192 columns the pass runs as TWO 256-thread workgroups per CU, each with its own
// copy of v in LDS: independent workgroups drift apart, so one multiplies while the other waits for memory — a single
// 512-thread workgroup meets at its barrier every row block, its eight waves wait and compute together, and the decode's
// extra vector work (a third more instructions per byte than the f32 kernel) then adds to the memory time instead of
// hiding under it.  Measured at n = 5e5, M = 1e4, 24-bit format (round 3, per-configuration timings):
// 512 x 5 x 6 rows 5.2 TB/s; two register sets alternating across the barrier 5.0-5.1; 8-column chunks (16-byte loads)
// 4.9-5.2; descriptors per (row, chunk) 5.2; 2 x (256 x 10 x 2 rows) 5.73 TB/s.  f32 kernel: 5.96.
static bool pick_qcfg(int64_t M, int nv, int fmt, QCfg* cfg) {
  const int64_t chunks = (M + 3) / 4;
  if (nv == 1) {
    if (chunks <= 256) { *cfg = {256, 1, 16, 2}; return true; }
    // (the two-halves form, two workgroups per CU: 4.9 / 5.0 TB/s at M = 2000 / 3000 against 4.1 / 4.5 for the barrier form
    // below; at 8 chunks per thread it needs the CU to itself and loses to it — 5.1 against 5.3 TB/s at M = 6000)
    if (chunks <= 512) { *cfg = {0, 2, 8, 2}; return true; }
    if (chunks <= 1024) { *cfg = {0, 4, 4, 2}; return true; }
    if (chunks <= 2048) { *cfg = {256, 8, 3, 2}; return true; }
    // (2 x (256 x 10 x 2 rows) is 10 % faster alone at M = 1e4 but needs the CU's whole LDS for its two copies of v: beside
    // the preconditioner stream of the headline job a CU holding one small workgroup of another kernel takes only ONE of
    // the two, the displaced workgroup of the persistent grid runs after the others, and the pass loses more than it won
    // — 5.10 against 5.30 TB/s inside bench.py.)
    if (chunks <= 2560) { *cfg = {0, 10, 2, 1}; return true; }      // two free-running halves
    if (chunks <= 3072) { *cfg = {512, 6, 4, 1}; return true; }
    if (chunks <= 5110) { *cfg = {1024, 5, 1, 1}; return true; }      // v + its zero chunk + the reduction scratch in 160 KB of LDS
    return false;
  }
  // two vectors: both in LDS (2 x roundup(M, 4) x 8 B beside the reduction scratch), a second set of column sums in registers
  if (chunks <= 1024 || chunks > 2560) return false;
  if (chunks <= 2048) *cfg = {512, 4, 2, 1};
  else *cfg = {512, 5, 2, 1};
  const int64_t lds = 2 * (chunks + 1) * 4 * 8 + 2 * (cfg->nt / 64) * cfg->r * 2 * 8 + 64;
  return lds <= 163840;
}

// compute units the persistent grids of this file are sized for: the device's, or what odx_set_pass_cus() says (the size
// of the CU partition the passes' stream is confined to, odx_stream_create_cu_mask)
static int g_pass_cus = 0;

static int pass_cus() {
  if (g_pass_cus > 0) return g_pass_cus;
  const int cus = odx_device_cus();
  return cus > 0 ? cus : 256;
}

static int qgrid_for(const QCfg& cfg, int64_t n) {
  const int cus = pass_cus();
  const int64_t nblk = cfg.nt == 0 ? ceil_div(ceil_div(n, cfg.r), 2) : ceil_div(n, cfg.r);     // staggered: two blocks per tick pair
  int64_t g = (int64_t)cus * cfg.wg_per_cu;
  if (g > nblk) g = nblk;
  if (g < 1) g = 1;
  return (int)g;
}

template <int NT, int CH, int R, int NV, int FMT, int WPE = (NT >= 1024 ? 4 : 2)>
static int launch_passq(int grid, size_t lds, hipStream_t s, const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int64_t n,
                        int64_t M, const double* v, const double* v2, const double* w, double* slab, int64_t slab_ld) {
  ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_passq_kernel<NT, CH, R, NV, FMT, WPE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((knm_passq_kernel<NT, CH, R, NV, FMT, WPE>), dim3(grid), dim3(NT), lds, s, static_cast<const unsigned short*>(K),
                     ldk, static_cast<const unsigned char*>(Klo), ldlo, n, M, v, v2, w, slab, slab_ld);
  return ODX_OK;
}

template <int NV, int FMT>
static int dispatch_passq(const QCfg& cfg, int grid, size_t lds, hipStream_t s, const void* K, int64_t ldk, const void* Klo,
                          int64_t ldlo, int64_t n, int64_t M, const double* v, const double* v2, const double* w, double* slab,
                          int64_t slab_ld) {
#define ODX_Q(NT_, CH_, R_) return launch_passq<NT_, CH_, R_, NV, FMT>(grid, lds, s, K, ldk, Klo, ldlo, n, M, v, v2, w, slab, slab_ld)
  if constexpr (NV == 1) {
    if (cfg.nt == 0) {
#define ODX_QH(CH_, R_)                                                                                                         \
  do {                                                                                                                          \
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_passq_stag_kernel<CH_, R_, FMT>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                   \
    hipLaunchKernelGGL((knm_passq_stag_kernel<CH_, R_, FMT>), dim3(grid), dim3(512), lds, s,                                    \
                       static_cast<const unsigned short*>(K), ldk, static_cast<const unsigned char*>(Klo), ldlo, n, M, v, w,    \
                       slab, slab_ld);                                                                                          \
    return ODX_OK;                                                                                                              \
  } while (0)
      if (cfg.ch == 2) ODX_QH(2, 8);
      if (cfg.ch == 4) ODX_QH(4, 4);
      ODX_QH(10, 2);
#undef ODX_QH
    }
    // (exactly the configurations pick_qcfg hands out)
    if (cfg.nt == 256 && cfg.ch == 1) ODX_Q(256, 1, 16);
    if (cfg.nt == 256 && cfg.ch == 8) ODX_Q(256, 8, 3);
    if (cfg.nt == 512 && cfg.ch == 6) ODX_Q(512, 6, 4);
    ODX_Q(1024, 5, 1);
  } else {
    if (cfg.ch == 4) ODX_Q(512, 4, 2);
    ODX_Q(512, 5, 2);
  }
#undef ODX_Q
}

static int check_q_block(const char* who, const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t M) {
  ODX_REQUIRE(fmt == ODX_KNM_U24 || fmt == ODX_KNM_BF16, "%s: storage format must be ODX_KNM_U24 or ODX_KNM_BF16 (got %d)", who, fmt);
  // (row sub-blocks of a stored shard are valid arguments: the planes need the alignment of one chunk load only)
  ODX_REQUIRE(K && ldk % 8 == 0 && ldk >= round_up(M, 8) && (reinterpret_cast<uintptr_t>(K) & 7u) == 0,
              "%s: K must be 8-byte aligned with ldk %% 8 == 0, ldk >= roundup(M, 8)", who);
  if (fmt == ODX_KNM_U24)
    ODX_REQUIRE(Klo && ldlo % 8 == 0 && ldlo >= round_up(M, 8) && (reinterpret_cast<uintptr_t>(Klo) & 3u) == 0,
                "%s: the low-byte plane must be 4-byte aligned with ldlo %% 8 == 0, ldlo >= roundup(M, 8)", who);
  return ODX_OK;
}

// ---- class-batched pass over compact blocks (internal; the batched CG of solve.cpp drives it)
static bool passq_batch_cfg(int B, const int64_t* M, int fmt, QCfg* out) {
  QCfg c0 = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    QCfg c;
    if (M[b] <= 0 || !pick_qcfg(M[b], 1, fmt, &c)) return false;
    if (b == 0) c0 = c;
    else if (c.nt != c0.nt || c.ch != c0.ch || c.r != c0.r) return false;
  }
  // (the configurations launch_passq_batched instantiates)
  const bool built = (c0.nt == 0 && (c0.ch == 2 || c0.ch == 4 || c0.ch == 10)) ||
                     (c0.nt == 256 && (c0.ch == 1 || c0.ch == 8)) || (c0.nt == 512 && c0.ch == 6);
  if (out) *out = c0;
  return B > 0 && built;
}

static void passq_batch_geometry(int B, const int64_t* n, const int64_t* M, const QCfg& cfg, int* gmax, int64_t* slab_ld) {
  int g = 1;
  int64_t mm = 1;
  for (int b = 0; b < B; ++b) {
    if (n[b] > 0) g = std::max(g, qgrid_for(cfg, n[b]));
    mm = std::max(mm, M[b]);
  }
  *gmax = g;
  *slab_ld = round_up(mm, 4);
}

int64_t knm_passq_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M, int fmt) {
  QCfg cfg;
  if ((fmt != ODX_KNM_U24 && fmt != ODX_KNM_BF16) || !passq_batch_cfg(B, M, fmt, &cfg)) return ODX_ERR_UNSUPPORTED;
  int gmax;
  int64_t slab_ld;
  passq_batch_geometry(B, n, M, cfg, &gmax, &slab_ld);
  return (int64_t)B * (cfg.nt == 0 ? 2 : 1) * gmax * slab_ld * (int64_t)sizeof(double);
}

template <int FMT>
static int launch_passq_batched(const QCfg& cfg, const PassBatchQ& pb, int gmax, int B, size_t lds, hipStream_t s, const double* v,
                                int64_t vstride, double* slab, int64_t slab_ld, int64_t slab_stride) {
#define ODX_QB(NT_, CH_, R_)                                                                                                        \
  do {                                                                                                                              \
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_passq_batched_kernel<NT_, CH_, R_, FMT>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                        \
    hipLaunchKernelGGL((knm_passq_batched_kernel<NT_, CH_, R_, FMT>), dim3(gmax, B), dim3(NT_), lds, s, pb, v, vstride, slab, slab_ld, \
                       slab_stride);                                                                                                \
    return ODX_OK;                                                                                                                  \
  } while (0)
#define ODX_QBH(CH_, R_)                                                                                                            \
  do {                                                                                                                              \
    ODX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knm_passq_stag_batched_kernel<CH_, R_, FMT>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                        \
    hipLaunchKernelGGL((knm_passq_stag_batched_kernel<CH_, R_, FMT>), dim3(gmax, B), dim3(512), lds, s, pb, v, vstride, slab, slab_ld, \
                       slab_stride);                                                                                                \
    return ODX_OK;                                                                                                                  \
  } while (0)
  if (cfg.nt == 0 && cfg.ch == 2) ODX_QBH(2, 8);
  if (cfg.nt == 0 && cfg.ch == 4) ODX_QBH(4, 4);
  if (cfg.nt == 0 && cfg.ch == 10) ODX_QBH(10, 2);
  if (cfg.nt == 256 && cfg.ch == 1) ODX_QB(256, 1, 16);
  if (cfg.nt == 256 && cfg.ch == 8) ODX_QB(256, 8, 3);
  if (cfg.nt == 512 && cfg.ch == 6) ODX_QB(512, 6, 4);
#undef ODX_QB
#undef ODX_QBH
  set_error("knm_passq_batched: no class-batched kernel for this pass configuration");
  return ODX_ERR_UNSUPPORTED;
}

// out[b] = K_b' (K_b v[b]) for the B classes of a batch (compact-format blocks) with ONE pass launch and ONE reduce launch;
// class b is handled exactly as odx_knm_fwd_bwd_q(K_b, ..) would handle it (same configuration, workgroup count, slab order)
int knm_passq_batched(int B, const void* const* Khi, const int64_t* ldk, const void* const* Klo, const int64_t* ldlo, int fmt,
                      const int64_t* n, const int64_t* M, const double* v, int64_t vstride, double* out, int64_t ostride,
                      void* workspace, int64_t workspace_bytes, hipStream_t s) {
  ODX_REQUIRE(B >= 1 && B <= ODX_MAX_ZBATCH, "knm_passq_batched: 1..%d classes", ODX_MAX_ZBATCH);
  QCfg cfg;
  ODX_REQUIRE((fmt == ODX_KNM_U24 || fmt == ODX_KNM_BF16) && passq_batch_cfg(B, M, fmt, &cfg),
              "knm_passq_batched: the classes of a batch must share one pass configuration");
  int gmax;
  int64_t slab_ld;
  passq_batch_geometry(B, n, M, cfg, &gmax, &slab_ld);
  const int per = cfg.nt == 0 ? 2 : 1;                            // the halves kernel leaves one vector per half
  const int64_t slab_stride = (int64_t)per * gmax * slab_ld;
  if (workspace == nullptr || workspace_bytes < (int64_t)B * slab_stride * (int64_t)sizeof(double)) {
    set_error("knm_passq_batched: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  PassBatchQ pb;
  int nslab[ODX_MAX_ZBATCH];
  for (int b = 0; b < ODX_MAX_ZBATCH; ++b) {
    const bool on = b < B;
    pb.Khi[b] = on ? static_cast<const unsigned short*>(Khi[b]) : nullptr;
    pb.Klo[b] = (on && fmt == ODX_KNM_U24) ? static_cast<const unsigned char*>(Klo[b]) : nullptr;
    pb.ldk[b] = on ? ldk[b] : 0;
    pb.ldlo[b] = (on && fmt == ODX_KNM_U24) ? ldlo[b] : 0;
    pb.n[b] = on ? n[b] : 0;
    pb.M[b] = on ? (int)M[b] : 0;
    pb.grid[b] = (on && n[b] > 0) ? qgrid_for(cfg, n[b]) : 0;
    nslab[b] = per * pb.grid[b];
    if (on && n[b] > 0) ODX_PROPAGATE(check_q_block("knm_passq_batched", Khi[b], ldk[b], Klo ? Klo[b] : nullptr, ldlo ? ldlo[b] : 0, fmt, M[b]));
  }
  double* slab = static_cast<double*>(workspace);
  const size_t lds = (cfg.nt == 0 ? (size_t)(cfg.ch * 256 * 4) : (size_t)(slab_ld + 4)) * sizeof(double);
  if (fmt == ODX_KNM_U24) ODX_PROPAGATE((launch_passq_batched<QF_U24>(cfg, pb, gmax, B, lds, s, v, vstride, slab, slab_ld, slab_stride)));
  else ODX_PROPAGATE((launch_passq_batched<QF_BF16>(cfg, pb, gmax, B, lds, s, v, vstride, slab, slab_ld, slab_stride)));
  ODX_CHECK_LAUNCH("knm_passq_batched");
  return slab_reduce_batched_f64(B, M, nslab, slab, slab_ld, slab_stride, out, ostride, s);
}

}  // namespace odx

using namespace odx;

// The kernel a pass over a block of M columns stored as `fmt` runs as (nv = 1: one vector, 2: two vectors from one read): the
// name rocprofv3 lists it under, from the SAME rule the launches use (pick_qcfg) — for tools that label measurements
// (bench.py's result line) without restating that rule.  "" when the shape has no configuration.
extern "C" const char* odx_knm_pass_kernel_name(int64_t M, int fmt, int nv) {
  static thread_local char name[96];
  name[0] = 0;
  if (fmt == 0) {                        // ODX_KNM_F32: knm_pass.hip's kernels (not templates)
    snprintf(name, sizeof(name), nv == 2 ? "knm_pass2_kernel" : "knm_pass_kernel");
    return name;
  }
  QCfg cfg;
  if ((fmt != QF_U24 && fmt != QF_BF16) || (nv != 1 && nv != 2) || M <= 0 || !pick_qcfg(M, nv, fmt, &cfg)) return name;
  if (cfg.nt == 0) snprintf(name, sizeof(name), "knm_passq_stag_kernel<%d,%d,%d>", cfg.ch, cfg.r, fmt);
  else snprintf(name, sizeof(name), "knm_passq_kernel<%d,%d,%d,%d,%d,%d>", cfg.nt, cfg.ch, cfg.r, nv, fmt, cfg.nt >= 1024 ? 4 : 2);
  return name;
}

extern "C" int odx_set_pass_cus(int cus) {
  ODX_REQUIRE(cus >= 0 && cus <= 4096, "odx_set_pass_cus: 0 (the device's) .. 4096");
  g_pass_cus = cus;
  return ODX_OK;
}

static int check_q(const char* who, const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t M) {
  return check_q_block(who, K, ldk, Klo, ldlo, fmt, M);
}

extern "C" int64_t odx_knm_fwd_bwd_q_workspace_bytes(int64_t n, int64_t M, int fmt) {
  QCfg cfg;
  if (n <= 0 || M <= 0) return 0;
  if ((fmt != ODX_KNM_U24 && fmt != ODX_KNM_BF16) || !pick_qcfg(M, 1, fmt, &cfg)) return ODX_ERR_UNSUPPORTED;
  int cus = odx_device_cus();       // (never less than what a partitioned launch needs)
  if (cus <= 0) cus = 256;
  return (int64_t)cus * (cfg.nt == 0 ? 2 * cfg.wg_per_cu : cfg.wg_per_cu) * round_up(M, 4) * (int64_t)sizeof(double);
}

extern "C" int odx_knm_fwd_bwd_q(const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t n, int64_t M,
                                 const double* v, const double* w, double* out, void* workspace, int64_t workspace_bytes,
                                 odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && out, "odx_knm_fwd_bwd_q: M <= 0 or null out");
  hipStream_t s = as_stream(stream);
  if (n <= 0) {
    ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)M * sizeof(double), s));
    return ODX_OK;
  }
  ODX_REQUIRE(v || w, "odx_knm_fwd_bwd_q: both v and w null");
  ODX_PROPAGATE(check_q("odx_knm_fwd_bwd_q", K, ldk, Klo, ldlo, fmt, M));
  QCfg cfg;
  if (!pick_qcfg(M, 1, fmt, &cfg)) {
    set_error("odx_knm_fwd_bwd_q: M = %lld exceeds the 20440 columns the compact-format pass kernels are built for", (long long)M);
    return ODX_ERR_UNSUPPORTED;
  }
  const int grid = qgrid_for(cfg, n);
  const int64_t slab_ld = round_up(M, 4);
  const int nslab = cfg.nt == 0 ? 2 * grid : grid;       // the staggered kernel leaves one vector per half
  if (workspace == nullptr || workspace_bytes < (int64_t)nslab * slab_ld * (int64_t)sizeof(double)) {
    set_error("odx_knm_fwd_bwd_q: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  double* slab = static_cast<double*>(workspace);
  // (the halves kernel keeps v zero-filled up to the 10 x 256 chunks of four its threads walk)
  const size_t lds = (cfg.nt == 0 ? (size_t)(cfg.ch * 256 * 4) : (size_t)(slab_ld + 4)) * sizeof(double);      // + the zero chunk
  if (fmt == ODX_KNM_U24) ODX_PROPAGATE((dispatch_passq<1, QF_U24>(cfg, grid, lds, s, K, ldk, Klo, ldlo, n, M, v, nullptr, w, slab, slab_ld)));
  else ODX_PROPAGATE((dispatch_passq<1, QF_BF16>(cfg, grid, lds, s, K, ldk, nullptr, 0, n, M, v, nullptr, w, slab, slab_ld)));
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd_q");
  return slab_reduce_f64(slab, slab_ld, nslab, M, out, s);
}

extern "C" int64_t odx_knm_fwd_bwd2_q_workspace_bytes(int64_t n, int64_t M, int fmt) {
  QCfg cfg;
  if (n <= 0 || M <= 0) return 0;
  if ((fmt != ODX_KNM_U24 && fmt != ODX_KNM_BF16) || !pick_qcfg(M, 2, fmt, &cfg)) return ODX_ERR_UNSUPPORTED;
  int cus = odx_device_cus();
  if (cus <= 0) cus = 256;
  return 2 * (int64_t)cus * cfg.wg_per_cu * round_up(M, 4) * (int64_t)sizeof(double);
}

extern "C" int odx_knm_fwd_bwd2_q(const void* K, int64_t ldk, const void* Klo, int64_t ldlo, int fmt, int64_t n, int64_t M,
                                  const double* v, const double* v2, double* out, double* out2, void* workspace,
                                  int64_t workspace_bytes, odx_stream_t stream) {
  ODX_REQUIRE(M > 0 && out && out2, "odx_knm_fwd_bwd2_q: M <= 0 or null out");
  hipStream_t s = as_stream(stream);
  if (n <= 0) {
    ODX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)M * sizeof(double), s));
    ODX_CHECK_HIP(hipMemsetAsync(out2, 0, (size_t)M * sizeof(double), s));
    return ODX_OK;
  }
  ODX_REQUIRE(v && v2, "odx_knm_fwd_bwd2_q: null v or v2");
  ODX_PROPAGATE(check_q("odx_knm_fwd_bwd2_q", K, ldk, Klo, ldlo, fmt, M));
  QCfg cfg;
  if (!pick_qcfg(M, 2, fmt, &cfg)) {
    set_error("odx_knm_fwd_bwd2_q: M = %lld is outside the two-vector configurations (use two single passes)", (long long)M);
    return ODX_ERR_UNSUPPORTED;
  }
  const int grid = qgrid_for(cfg, n);
  const int64_t slab_ld = round_up(M, 4);
  if (workspace == nullptr || workspace_bytes < 2 * (int64_t)grid * slab_ld * (int64_t)sizeof(double)) {
    set_error("odx_knm_fwd_bwd2_q: workspace too small");
    return ODX_ERR_WORKSPACE;
  }
  double* slab = static_cast<double*>(workspace);
  const size_t lds = (size_t)(2 * (slab_ld + 4) * sizeof(double));      // two vectors, each with its zero chunk
  if (fmt == ODX_KNM_U24) ODX_PROPAGATE((dispatch_passq<2, QF_U24>(cfg, grid, lds, s, K, ldk, Klo, ldlo, n, M, v, v2, nullptr, slab, slab_ld)));
  else ODX_PROPAGATE((dispatch_passq<2, QF_BF16>(cfg, grid, lds, s, K, ldk, nullptr, 0, n, M, v, v2, nullptr, slab, slab_ld)));
  ODX_CHECK_LAUNCH("odx_knm_fwd_bwd2_q");
  return slab_reduce2_f64(slab, slab_ld, grid, M, out, out2, s);
}
