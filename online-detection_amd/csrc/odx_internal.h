// Internal (non-ABI) interfaces shared between the translation units of libodx.
#pragma once
#include "odx_common.h"

namespace odx {
// The library's process-wide options (api.cpp: odx_set_option / odx_get_option; include/odx.h has the table).  Set once by the
// host side when it loads the library (odx/options.py); no kernel or launch path reads the environment.
enum LibOpt {
  OPT_H2_TILE = 0,                 // 0 automatic | 128 | 256: the split-f16 tile core of the Gaussian kernels and row GEMMs
  OPT_PRECOND,                     // 0 automatic (A factor on the split core from 4096 centres on) | 1 all-f64 chain | 2 split always
  OPT_CHAIN_HELPERS,               // -1 automatic (helper streams from 4096 centres on) | 0 never | 1 always
  OPT_RLS_FORCE_NT_GRAM,           // test hook: 1 = the transposed-copy + NT-GEMM Gram (the route of D % 8 != 0) for every D
  OPT_RLS_FORCE_INVERSE_SOLVE,     // test hook: 1 = the explicit-inverse solve (the route of D + 1 > 16 x 128) for every D
  OPT_COUNT
};
int lib_option(int which);


constexpr int ODX_MAX_ZBATCH = 32;    // classes per batched launch (kernel-argument arrays are sized by it)

template <typename T>
struct GemmParams {
  const T* A = nullptr; int64_t lda = 0;
  const T* B = nullptr; int64_t ldb = 0;
  T* C = nullptr; int64_t ldc = 0;
  T* C2 = nullptr; int64_t ldc2 = 0;   // optional transposed copy of the result: C2[j, i]
  int64_t m = 0, n = 0, k = 0;
  T alpha = T(1), beta = T(0);
  int flags = 0;
  // batching over blockIdx.y
  int batches = 1;
  int64_t strideA = 0, strideB = 0, strideC = 0, strideC2 = 0;
  // ragged batches (trtri levels): m_b = clamp(ragged_total - ragged_off - b * ragged_step, 0, m)
  int64_t ragged_total = 0, ragged_off = 0, ragged_step = 0;
  int ragged_k_is_m = 0;               // k_b = m_b (GEMM 2 of a trtri level)
  int vec_epilogue = 0;                // set by the launcher: C (and C2) allow 16-byte accesses
  // second batch dimension over blockIdx.z: the same product for `zbatches` independent matrices (the classes of a
  // batched preconditioner), operands zstride* elements apart, each with its own alpha (zalpha[z], when zalpha_on)
  int zbatches = 1;
  int64_t zstrideA = 0, zstrideB = 0, zstrideC = 0, zstrideC2 = 0;
  int zalpha_on = 0;
  T zalpha[ODX_MAX_ZBATCH] = {};
  // per-class window of the contraction axis (the RLS Grams of a class batch: class z owns columns
  // [zkoff[z], zkoff[z] + zklen[z]) of the shared operands): A and B advance by zkoff[z], k = zklen[z]
  int zk_on = 0;
  int64_t zkoff[ODX_MAX_ZBATCH] = {};
  int64_t zklen[ODX_MAX_ZBATCH] = {};
};

int launch_gemm_f64(const GemmParams<double>& p, hipStream_t stream);
int launch_gemm_f32(const GemmParams<float>& p, hipStream_t stream);

// dense_f64.hip
constexpr int POTRF_NB = 128;
// A class batch: `count` matrices of identical shape, `stride*` elements apart (count = 1: the plain single-matrix call).
// Every kernel of the factorisation takes the batch as one more grid dimension, so one launch advances all classes.
struct ZBatch {
  int count = 1;
  int64_t strideA = 0;      // between the matrices being factored / inverted
  int64_t strideD = 0;      // between their Dinv blocks
  int64_t strideO = 0;      // between their outputs (Li / Lit)
  int64_t strideW = 0;      // between their scratch (WT)
};
int potrf_f64(double* A, int64_t lda, int64_t M, double* Dinv /* nblk x NB x NB */, int32_t* info /* one per matrix */,
              hipStream_t stream, const ZBatch& zb = ZBatch(), uint32_t* pk = nullptr, int64_t pk_buf = 0, int64_t pk_z = 0,
              float pk_scale = 1.f);
int trtri_from_diag_f64(const double* L, int64_t ldl, int64_t M, const double* Dinv, double* Li, double* Lit,
                        int64_t ld, double* WT, hipStream_t stream, const ZBatch& zb = ZBatch(), uint32_t* pk = nullptr,
                        int64_t pk_cap = 0, int64_t pk_z = 0, double bound_l = 1.0, double bound_inv = 1.0);
int transpose_f64(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                  hipStream_t stream, int zcount = 1, int64_t zstride_src = 0, int64_t zstride_dst = 0);
int add_diag_f64(double* A, int64_t lda, int64_t M, double value, hipStream_t stream, int zcount = 1, int64_t zstride = 0);
int fill_f64(double* A, int64_t lda, int64_t rows, int64_t cols, double value, hipStream_t stream);

// knm_pass.hip: out[j] = sum_g slab[g][j], g = 0 .. nslab - 1, in a fixed order
int slab_reduce_f64(const double* slab, int64_t slab_ld, int nslab, int64_t M, double* out, hipStream_t stream);
// the same for the two vectors a two-product pass leaves per workgroup (slab g = [sums of v | sums of v2], 2 slab_ld apart)
int slab_reduce2_f64(const double* slab, int64_t slab_ld, int nslab, int64_t M, double* out, double* out2, hipStream_t stream);

int slab_reduce_batched_f64(int B, const int64_t* M, const int* nslab, const double* slab, int64_t slab_ld, int64_t slab_stride,
                            double* out, int64_t ostride, hipStream_t stream);
// knm_pass_q.hip: the CG pass over compact-format blocks (24-bit fixed point / bf16) for a batch of classes
int64_t knm_passq_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M, int fmt);
int knm_passq_batched(int B, const void* const* Khi, const int64_t* ldk, const void* const* Klo, const int64_t* ldlo, int fmt,
                      const int64_t* n, const int64_t* M, const double* v, int64_t vstride, double* out, int64_t ostride,
                      void* workspace, int64_t workspace_bytes, hipStream_t stream);

// knm_pass.hip: the CG pass for a batch of classes (one launch; per class the arithmetic of odx_knm_fwd_bwd)
bool knm_pass_batch_cfg(int B, const int64_t* M, int* nt, int* ch, int* r);
int64_t knm_pass_batched_workspace_bytes(int B, const int64_t* n, const int64_t* M);
int knm_pass_batched(int B, const float* const* K, const int64_t* ldk, const int64_t* n, const int64_t* M, const double* v,
                     int64_t vstride, double* out, int64_t ostride, void* workspace, int64_t workspace_bytes,
                     hipStream_t stream);

// dense_f64.hip / cg.hip: class-batched small algebra of the CG (blockIdx.y = class; vectors vstride apart)
struct VecBatch {
  int B = 1;
  int M[ODX_MAX_ZBATCH] = {};
  double scale[ODX_MAX_ZBATCH] = {};     // per-class alpha of a triangular product (1 / n_total)
};
int trmv_batched_f64(const double* Tri, int64_t ld, int64_t tri_stride, int uplo, const VecBatch& vb, const double* x,
                     int64_t xstride, bool scaled, double beta, const double* z, int64_t zstride, double* y,
                     int64_t ystride, hipStream_t stream);
int cg_init_batched(const VecBatch& vb, const double* Bv, double* X, double* R, double* P, double* state, int64_t vstride,
                    hipStream_t stream);
int cg_step_batched(const VecBatch& vb, double* X, double* R, const double* P, const double* AP, double* state,
                    double cg_eps, int full_grad, int64_t vstride, hipStream_t stream);
int cg_finish_batched(const VecBatch& vb, const double* R, double* P, double* state, double cg_eps, double tol,
                      int64_t vstride, hipStream_t stream);
int cg_full_residual_batched(const VecBatch& vb, const double* Bv, const double* AX, double* R, int64_t vstride,
                             hipStream_t stream);

// gauss_h2.hip: f64 matrices through the split-f16 tile core (f32-accurate products, f64 scale / accumulate into C)
int64_t h2_f64_packed_ld(int64_t cols);            // 4-byte units per packed row
int split_f64(const double* X, int64_t ldx, int64_t zsx, int64_t rows, int64_t cols, float scale, uint32_t* P, int64_t ldp,
              int64_t zsp, int z, hipStream_t stream);
struct SplitBlocksArgs {          // a batch of nb x Z blocks of f64 matrices -> packed splits (split_f64_blocks_kernel)
  const double* X = nullptr; int64_t ldx = 0, bsx = 0, zsx = 0;
  uint32_t* P = nullptr; int64_t ldp = 0, bsp = 0, zsp = 0;
  int64_t rows = 0, cols = 0;
  int64_t rg_total = 0, rg_off = 0, rg_step = 0;
  int rows_ragged = 0, cols_ragged = 0, nb = 1, Z = 1;
  float scale = 1.f;
};
int split_f64_blocks(const SplitBlocksArgs& a, hipStream_t stream);
struct H2F64Args {                // C = alpha (A B') / (sa sb) + beta C over nb blocks x zcount classes (gemm_h2w256_f64_kernel)
  const uint32_t* PA = nullptr; int64_t ldpa = 0, zsa = 0, bsa = 0; float sa = 1.f;
  const uint32_t* PB = nullptr; int64_t ldpb = 0, zsb = 0, bsb = 0; float sb = 1.f;
  double* C = nullptr; int64_t ldc = 0, zsc = 0, bsc = 0;
  double* C2 = nullptr; int64_t ldc2 = 0, zsc2 = 0, bsc2 = 0;
  int64_t m = 0, n = 0, k = 0;
  int64_t rg_total = 0, rg_off = 0, rg_step = 0;
  int k_is_m = 0, flags = 0, zcount = 1, nb = 1;
  double beta = 0.0;
  double alpha[ODX_MAX_ZBATCH] = {};
};
int gemm_h2_f64_ex(const H2F64Args& a, hipStream_t stream);
int gemm_h2_f64(const uint32_t* PA, int64_t ldpa, int64_t zsa, float sa, const uint32_t* PB, int64_t ldpb, int64_t zsb, float sb,
                double* C, int64_t ldc, int64_t zsc, int64_t m, int64_t n, int64_t k, const double* alpha, double beta, int flags,
                int zcount, hipStream_t stream);

// gauss.hip
int gauss_kmm_f64(const double* Zd, int64_t ldz, int64_t M, int D, double sigma, double diag_add, double* Kmm,
                  int64_t ldk, double* zsq /* M */, hipStream_t stream);

int gauss_kmm_f64_batched(const double* Zd, int64_t ldz, int64_t z_stride, int64_t zsq_off, const VecBatch& vb, int D,
                          double sigma, double* Kmm, int64_t ldk, int64_t k_stride, hipStream_t stream);

}  // namespace odx
