// mfgp_internal.h -- shared declarations of the HIP engine (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/mfgp.h"

namespace mfgp {

constexpr int NB = 128;  // leaf block = padding granule = largest GEMM tile edge
constexpr int BK = 32;   // K-step of the tile GEMM (doubles)
constexpr int MFGP_MAX_DEVICES = 16;   // (power of two) per-device one-time launch setup slots

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// Tile-GEMM task: one workgroup computes
//     C[i0+r][j0+c] = beta * C[..] + alpha * sum_{k<klen} A[i0+r][k0+k] * B[j0+c][k0'+k]
// (both operands K-contiguous, "NT").  All matrices share the leading dimension ld.
// Triangular operands are expressed by trimming [k0, k0+klen) to the non-zero range and masking
// the one diagonal window that remains (the buffers hold mirrored data in the other triangle).
// ---------------------------------------------------------------------------------------------
enum : int32_t {
    TF_A_LOWER = 1,  // A rows are rows of a lower-triangular matrix; K range ends on the diagonal:
                     //   zero where k > r + klen - BM
    TF_A_UPPER = 2,  // A rows are rows of an upper-triangular matrix; K range starts on the diagonal:
                     //   zero where k < r
    TF_B_LOWER = 4,  //   zero where k > c + klen - BN
    TF_B_UPPER = 8,  //   zero where k < c
};

struct GemmTask {
    int64_t a_off;   // element offset of A[i0][k0]
    int64_t b_off;   // element offset of B[j0][k0']
    int64_t c_off;   // element offset of C[i0][j0]
    int64_t c2_off;  // element offset of the mirrored copy C2[j0][i0] (written transposed), or -1
    int32_t klen;    // multiple of BK
    int32_t flags;
    double alpha;
    double beta;
    int64_t pad_;
};
static_assert(sizeof(GemmTask) == 64, "GemmTask layout");

// kernel-structure descriptor passed by value to the covariance kernels
struct KernSpecDev {
    double theta[2 * MFGP_MAX_PARTS + 2];   // [var_f, len_f]*, then noise, jitter: travels with the kernel arguments
                                            // (no parameter upload per evaluation)
    int32_t nf;                       // number of factors
    int32_t D;                        // columns of X
    int32_t type[MFGP_MAX_PARTS];
    int32_t c0[MFGP_MAX_PARTS];
    int32_t c1[MFGP_MAX_PARTS];
    int32_t term[MFGP_MAX_PARTS];
    int32_t gidx[MFGP_MAX_PARTS];     // distance group of the factor (factors with equal column ranges share r^2)
    int32_t ng;                       // number of distance groups (<= MFGP_MAX_GROUPS)
    int32_t gc0[3];                   // column range of each group
    int32_t gc1[3];
};
constexpr int MFGP_MAX_GROUPS = 3;

// ---- launchers (implemented in the .hip files) -------------------------------------------------
// tile: 128 or 64.  tasks = device pointer to ntasks GemmTask.
// role: 0 = recursion GEMMs, 1 = the K^-1 SYRK launch, 2 = predictive-variance product (distinct kernel symbols),
//       3 = a step on the serial Cholesky chain (64-tile only: mfgp_gemm_nt_f64_chain)
void launch_gemm(hipStream_t s, int tile, const GemmTask* tasks, int ntasks, const double* A,
                 const double* B, double* C, double* C2, int ld, int role = 0);
size_t gemm_lds_bytes(int tile);
// skinny variance product for <= 64 test rows: V[0 .. 16*rows16) = W X^T (X = L^-1 from the mirrored S); rows16 in {1, 2, 4}
void launch_predv_skinny(hipStream_t s, int rows16, const double* W, const double* S, double* V, int ld, int Np);

// leaf: Cholesky + inverse of the 128x128 diagonal block `blk` of A (ld), in LDS.
//   L block (zeros above diag) -> Lout[blk,blk];  X = L^-1 -> S[blk,blk] stored mirrored (X + X^T - diag)
//   half log-det partial -> logdet_part[blk];  first failing pivot (1-based global index) -> info (atomicMin style)
void launch_leaf(hipStream_t s, const double* A, double* Lout, double* S, int ld, int blk,
                 double* logdet_part, int* info, unsigned long long* stamps = nullptr);

// covariance builders
//   tri: lower-triangle 64x64 tiles of Ky = K + (noise+jitter) I over padded Np (identity padding)
void launch_kbuild_tri(hipStream_t s, const KernSpecDev& spec, const double* X,
                       int N, int Np, double* A, int ld);
//   panel: Kx[r][c] = k(Xs[r], X[c]) for r < Nsp, c < Np (0 for padded columns c >= N)
void launch_kbuild_panel(hipStream_t s, const KernSpecDev& spec, const double* Xs, int Nsp,
                         const double* X, int N, int Np, double* Kx, int ld);
//   rows [row_begin, row_end) (multiples of 64) of Ky, all Np columns, written at their place in A
void launch_kbuild_rows(hipStream_t s, const KernSpecDev& spec, const double* X, int N, int Np,
                        double* A, int ld, int row_begin, int row_end);
//   full symmetric K without noise into out (N x N, ld = N) for parity read-back
void launch_kbuild_full(hipStream_t s, const KernSpecDev& spec, const double* X,
                        int N, int Np, double* out, int ld);

// vector ops
//   y[i] = sum_{k in range(i)} M[i][k] x[k];  mode 0: k <= i (lower), 1: k >= i (upper), 2: all k < ncols
void launch_rowdot(hipStream_t s, const double* M, int ld, const double* x, double* y, int nrows,
                   int ncols, int mode);
void launch_append_finish(hipStream_t s, double* L, double* S, int ld, int n, const double* l, const double* w, double* z,
                          double kdiag, double y_new, double* out, double* X, const double* xs_new, int D, double* Y);
//   rowsumsq[i] = sum_{k < ncols} M[i][k]^2
void launch_rowsumsq(hipStream_t s, const double* M, int ld, double* out, int nrows, int ncols);
//   scalars[0] = sum z^2 ; scalars[1] = 2*sum logdet_part ; (single small block)
void launch_finish_solve(hipStream_t s, const double* z, int Np, const double* logdet_part, int nblk,
                         double* scalars);
//   gradient: partial sums over lower-triangle 64x64 tiles; out[2*nf+1] (natural-parameter gradient of NLML)
void launch_grad(hipStream_t s, const KernSpecDev& spec, const double* X,
                 const double* Kinv, int ld, const double* alpha, int N, int Np, double* partials,
                 double* out);
int grad_num_partials(int Np);
//   var[i] = max(kss - ss[i], 1e-15) + add ; kss from spec.theta
void launch_finish_var(hipStream_t s, const KernSpecDev& spec, const double* ss,
                       double* var, int n, double add);

// level chaining: stencil stack rows and the augmented-row assembly (vecops.hip)
void launch_stencil_rows(hipStream_t s, const double* Xc, const double* offs, int d, int c, int64_t t0, int n, int n_p,
                         double* T);
void launch_assemble_aug(hipStream_t s, const double* Xc, const double* m, int rows, int rows_p, int d, int c,
                         double* out, int ld);

}  // namespace mfgp
