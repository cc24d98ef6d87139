// mfgp_internal.h -- shared declarations of the HIP engine (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/mfgp.h"
#include "plan.h"

namespace mfgp {

constexpr int MFGP_MAX_DEVICES = 16;   // (power of two) per-device one-time launch setup slots

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// kernel-structure descriptor passed by value to the covariance kernels
struct KernSpecDev {
    double theta[MFGP_MAX_THETA + 2]; // per factor: variance, lengthscale(s); then noise, jitter at [np], [np + 1]: travels
                                      // with the kernel arguments (no parameter upload per evaluation)
    int32_t nf;                       // number of factors
    int32_t np;                       // P = number of kernel parameters (include/mfgp.h: layout)
    int32_t toff[MFGP_MAX_PARTS];     // index of variance_f in theta; its lengthscales follow
    int32_t nl[MFGP_MAX_PARTS];       // lengthscales of factor f: 1 (isotropic) or c1 - c0 (MFGP_KERN_ARD)
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
//       3 = a step on the serial Cholesky chain (64-tile only: mfgp_gemm_nt_f64_chain), 5 = its 32-tile form
// nbatch / bstride: the same task list over nbatch matrix sets lying bstride elements apart (mfgp_eval_batch); 1 / 0 otherwise
// -> 0, or -1 for a (tile, role) pair no kernel exists for
// flag / epoch: the device-resident failure marks of the handle's evaluations (one int per matrix set; leaf_f64.hip sets
// flag[set] = epoch when a diagonal block is not positive definite) -- a launch whose set is marked returns at once; nullptr: no check
int launch_gemm(hipStream_t s, int tile, const GemmTask* tasks, int ntasks, const double* A,
                const double* B, double* C, double* C2, int ld, int role = 0, int nbatch = 1, long long bstride = 0,
                const int* flag = nullptr, int epoch = 0);
size_t gemm_lds_bytes(int tile);

// <= 8 test rows (trimv_f64.hip): V[i][j] = sum_{k <= j} W[i][k] X[j][k] on the VALU behind ONE coalesced read of the triangle;
// R = the row count rounded up to 1, 2, 4 or 8 (the panel W holds at least that many rows); the same launch forms the means
// W[i] . alpha of the `rows` real test rows
void launch_predv_rows(hipStream_t s, int R, const double* W, const double* S, double* V, int ld, int Np, const double* alpha,
                       double* mean, int rows);
// 9 .. 64 test rows (trimv_f64.hip): the same product on v_mfma_f64_16x16x4, RT = ceil(rows / 16) row tiles, S streamed through LDS by
// LDS-DMA in the coalesced shape; the same launch forms the means of the `rows` real test rows
//   Wt: 64 Np doubles of scratch for the panel in MFMA fragment order (written by the launch)
void launch_predv_mfma(hipStream_t s, int RT, const double* W, double* Wt, const double* S, double* V, int ld, int Np,
                       const double* alpha, double* mean, int rows);
//   and their finish, one workgroup per test row: var[i] = max(kss - |V[i]|^2, 1e-15) + add
void launch_predv_finish(hipStream_t s, int rows, const double* V, int ld, int Np, double kss, double add, double* var);
void launch_predv_mfma2(hipStream_t s, int RT, const double* W, const double* S, double* Vp, int ld, int Np,
                        const double* alpha, double* mean, int rows);
bool predv_mfma2_pays(int rows, int Np);
void launch_predv_finish_planes(hipStream_t s, int rows, const double* Vp, int ld, int Np, double kss, double add, double* var);

// leaf: Cholesky + inverse of the 128x128 diagonal block `blk` of A (ld), in LDS.
//   L block (zeros above diag) -> Lout[blk,blk];  X = L^-1 -> S[blk,blk] stored mirrored (X + X^T - diag)
//   half log-det partial -> logdet_part[blk];  first failing pivot (1-based global index) -> info (atomicMin style)
//   batched: nbatch workgroups, set b at A + b * bstride (and Lout, S), logdet_part + b * ldstride, info + b * istride
//   flag / epoch: see launch_gemm (the leaf both checks and sets the mark)
void launch_leaf(hipStream_t s, const double* A, double* Lout, double* S, int ld, int blk,
                 double* logdet_part, int* info, int nbatch = 1, long long bstride = 0, int ldstride = 0, int istride = 0,
                 int* flag = nullptr, int epoch = 0);

// covariance builders
//   tri: lower-triangle 64x64 tiles of Ky = K + (noise+jitter) I over padded Np (identity padding)
void launch_kbuild_tri(hipStream_t s, const KernSpecDev& spec, const double* X,
                       int N, int Np, double* A, int ld);
//   the same for nbatch parameter sets (specs[b] -> A + b * bstride): ONE launch where every set takes the RBF fast path
//   (mfgp_kbuild_rbf2_batch_f64), one launch per set otherwise
constexpr int MFGP_BATCH_MAX = 16;
void launch_kbuild_tri_batch(hipStream_t s, const KernSpecDev* specs, int nbatch, const double* X, int N, int Np, double* A,
                             int ld, long long bstride);
//   panel: Kx[r][c] = k(Xs[r], X[c]) for r < Nsp, c < Np (0 for padded columns c >= N)
void launch_kbuild_panel(hipStream_t s, const KernSpecDev& spec, const double* Xs, int Nsp,
                         const double* X, int N, int Np, double* Kx, int ld);
// 1, 2 or 4 test rows (R; rows R .. of the panel are NOT written), read where they are: row r < n is
//     [ a[(t0 + r) / c][0 .. da) + offs[(t0 + r) % c][0 .. da)  |  m[(t0 + r)][0 .. dm) ]      (offs, m optional; c >= 1)
// and a zero row from n on.  The fast-path kernel descriptions only -> false otherwise (kbuild_panel_few_ok says which).
struct FewRows {
    const double* a;
    const double* offs;
    const double* m;
    int da, c, dm, n;
    long long t0;
};
inline FewRows few_rows_packed(const double* Xs, int D, int n) { return FewRows{Xs, nullptr, nullptr, D, 1, 0, n, 0}; }
bool kbuild_panel_few_ok(const KernSpecDev& spec);
bool launch_kbuild_panel_few(hipStream_t s, const KernSpecDev& spec, const FewRows& q, int R, const double* X, int N, int Np,
                             double* Kx, int ld);
//   rows [row_begin, row_end) (multiples of 64) of Ky, all Np columns, written at their place in A
void launch_kbuild_rows(hipStream_t s, const KernSpecDev& spec, const double* X, int N, int Np,
                        double* A, int ld, int row_begin, int row_end);
//   full symmetric K without noise into out (N x N, ld = N) for parity read-back
void launch_kbuild_full(hipStream_t s, const KernSpecDev& spec, const double* X,
                        int N, int Np, double* out, int ld);

// vector ops
//   y[i] = sum_{k in range(i)} M[i][k] x[k];  mode 0: k <= i (lower), 1: k >= i (upper), 2: all k < ncols
//   nbatch sets: M, x, y of set b lie b * (mstride, xstride, ystride) elements further on (blockIdx.y = b)
void launch_rowdot(hipStream_t s, const double* M, int ld, const double* x, double* y, int nrows,
                   int ncols, int mode, int nbatch = 1, long long mstride = 0, long long xstride = 0, long long ystride = 0);
void launch_append_finish(hipStream_t s, double* L, double* S, int ld, int n, const double* l, const double* w, double* z,
                          double* alpha, double kdiag, double y_new, double* out, double* X, const double* xs_new, int D, double* Y);
//   rowsumsq[i] = sum_{k < ncols} M[i][k]^2
void launch_rowsumsq(hipStream_t s, const double* M, int ld, double* out, int nrows, int ncols);
//   scalars[0] = sum z^2 ; scalars[1] = 2*sum logdet_part ; (single small block)
void launch_alpha_finish(hipStream_t s, const double* S, int ld, const double* z, double* alpha, int Np,
                         const double* logdet_part, int nblk, double* scalars,   // alpha = X^T z + the scalars, one launch
                         int nbatch = 1, long long sstride = 0, long long vstride = 0, int ldstride = 0, int scstride = 0);
void launch_finish_solve(hipStream_t s, const double* z, int Np, const double* logdet_part, int nblk,
                         double* scalars);
//   gradient: partial sums over lower-triangle 64x64 tiles; out[2*nf+1] (natural-parameter gradient of NLML)
void launch_grad(hipStream_t s, const KernSpecDev& spec, const double* X,
                 const double* Kinv, int ld, const double* alpha, int N, int Np, double* partials,
                 double* out);
int grad_num_partials(int Np);
//   the two halves of launch_grad: the tile partials (shard_size > 1: only the tiles whose rows of K^-1 rank shard_rank holds;
//   the caller zeroes `partials` first) and the fixed-order finish
void launch_grad_tiles(hipStream_t s, const KernSpecDev& spec, const double* X, const double* Kinv, int ld,
                       const double* alpha, int N, int Np, double* partials, int shard_rank, int shard_size);
void launch_grad_finish(hipStream_t s, const KernSpecDev& spec, const double* partials, int Np, double* out);
//   lower part of the mirrored inverse S <- transpose of its upper part (sharded evaluation, after the exchange of the rows of X^T)
void launch_mirror_lower(hipStream_t s, double* S, int ld, int Np);
//   pack this rank's blocks of the upper part of S into / unpack the other ranks' blocks out of the exchange's staging buffer
//   (lower: the part of each 128-row block up to and including its diagonal block instead -- the row blocks of Ky, mfgp_allgather_rows)
void launch_shard_rows_copy(hipStream_t s, double* S, int ld, int nblk, double* stage, const long long* off, long long chunk,
                            int rank, int size, bool unpack, bool lower = false);
//   exchange steps of a distributed Cholesky (plan.h Shard::dist): the diagonal blocks + the leaf's words to / from one message; the
//   blocks of block column c of L below the diagonal to / from the all-gather's buffer (chunk doubles per rank)
void launch_dist_diag_copy(hipStream_t s, double* L, double* S, int ld, int c, double* stage, double* logdet, int* info, bool unpack,
                           int* flag = nullptr, int epoch = 0);
void launch_dist_panel_copy(hipStream_t s, double* L, int ld, int nblk, int c, double* stage, long long chunk, int rank, int size,
                            bool unpack);
//   nbatch sets: K^-1 of set b at Kinv + b * kstride, alpha + b * astride, partials + b * pstride, out + b * ostride; thetas =
//   device-readable copy of the sets' parameter vectors, tstride apart (the finishing kernel divides by them)
void launch_grad_batch(hipStream_t s, const KernSpecDev* specs, int nbatch, const double* X, const double* Kinv, long long kstride,
                       int ld, const double* alpha, long long astride, int N, int Np, double* partials, long long pstride,
                       double* out, int ostride, const double* thetas, int tstride);
//   var[i] = max(kss - ss[i], 1e-15) + add ; kss from spec.theta
void launch_finish_var(hipStream_t s, const KernSpecDev& spec, const double* ss,
                       double* var, int n, double add);

// level chaining: stencil stack rows and the augmented-row assembly (vecops.hip)
void launch_stencil_rows(hipStream_t s, const double* Xc, const double* offs, int d, int c, int64_t t0, int n, int n_p,
                         double* T);
void launch_assemble_aug(hipStream_t s, const double* Xc, const double* m, int rows, int rows_p, int d, int c,
                         double* out, int ld);

// ---- host-side state ---------------------------------------------------------------------------------
extern thread_local std::string g_err;   // error text of calls that have no handle to hang it on

}  // namespace mfgp

struct mfgp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // bulk trailing updates of the look-ahead Cholesky
    std::vector<hipEvent_t> evpool;     // cross-stream dependencies of the plan
    std::string err, info_str;
    int64_t N = 0, Np = 0, cap = 0;  // cap: allocated padded size
    int D = 0, nblk = 0;
    double* slab = nullptr;          // ONE allocation holding the four Np^2 matrices
    double* buf[4] = {nullptr, nullptr, nullptr, nullptr};   // A | L | S | W inside the slab (stride cap^2)
    double *dX = nullptr, *dXs = nullptr, *dY = nullptr, *dz = nullptr, *dalpha = nullptr;
    double *dlogdet = nullptr, *dres = nullptr, *dpart = nullptr, *dvec = nullptr,
           *dvec2 = nullptr;
    int* dinfo = nullptr;
    int* dflag = nullptr;            // device-resident failure marks: [0] the handle's own evaluation, [1 + b] set b of a batch
    int epoch = 0;                   // number of the evaluation in flight: a mark equal to it says "this one has failed"
    mfgp::GemmTask* dtasks = nullptr;
    size_t tasks_cap = 0;
    int xs_cap_rows = 0, xs_cap_D = 0;
    double *dXc = nullptr, *dm = nullptr, *doffs = nullptr, *dAug = nullptr;  // level chaining scratch
    int64_t ch_rows = 0;
    int ch_c = 0, ch_D = 0;             // the chain scratch is sized for (ch_rows, ch_c) at input width ch_D
    double* hres = nullptr;  // pinned
    // small predictive batches (the DIRECT callback, acquisition batches, single-point serving) travel through pinned,
    // device-mapped memory instead of pageable copies: [IO_IN doubles of test rows | IO_OUT means | IO_OUT variances].
    // A hipMemcpyAsync to / from pageable memory is staged and blocks the calling thread (~15-20 us each, three per call).
    static constexpr int IO_IN = 65536, IO_OUT = 8192;
    double* hio = nullptr;   // host view
    double* dio = nullptr;   // device view
    int* hinfo = nullptr;    // pinned
    bool stage_timing = true; // per-stage event stamps inside an evaluation (off below Np = 4096 unless MFGP_STAGE_TIMING=1; mfgp_timings.timed says what a call measured)
    bool timing = true;          // any timing events at all (start / end of an evaluation, of a predict)
    bool timing_small = false;   // ... also around a predict of <= 64 test rows (MFGP_TIMING=1 only: the records are ~8 % of such a call)
    mfgp::Plan pl;                  // factorisation / inverse / K^-1 / predictive-variance launch lists (plan.cpp)
    mfgp::KernSpecDev spec{};
    bool have_kernel = false, have_data = false, factorized = false, kinv_valid = false, grad_valid = false,
         params_set = false;
    double theta[MFGP_MAX_THETA] = {0};
    double noise = 0, jitter = 0;
    double quad = 0, logdet = 0;
    double grad[MFGP_MAX_THETA + 1] = {0};
    hipEvent_t ev[10] = {};
    mfgp_timings tm{};
    mfgp_counters cum{};
    int64_t launches = 0;
    // multi-GPU (comm_rccl.hip): one RCCL communicator per handle, created by mfgp_comm_init; opaque here
    void* comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    double coll_us = 0.0;                // measured cost of one small collective of this communicator (mfgp_comm_calibrate; 0: not measured)
    double calib[6] = {0, 0, 0, 0, 0, 0};  //   {bcast_us, gather_us, gather bytes per rank, gather GB/s, repetitions, this rank's own worst median us}
    bool comm_aborted = false;           // the communicator was torn down after a failed / unmatched collective (comm_abort): no further one is issued
    int dbg_fail_collective_in = 0;      // test hook (mfgp_dbg_fail_collective_after): the n-th all-gather from now returns an RCCL error without being issued
    int dbg_fail_sharded_in = 0;         // test hook (mfgp_dbg_fail_sharded_after): the n-th sharded pass from now fails after the control exchange
    double* dstage = nullptr;            // device staging of mfgp_allgather_host
    size_t stage_cap = 0;
    // batched evaluation (mfgp_eval_batch): bsets matrix sets A | L | S | W of cap^2 each (the plan's task offsets apply to
    // every set: set b lies b * 4 cap^2 elements further on), their solve vectors, log-det partials, gradient partials and
    // 128-double result blocks in pinned, device-mapped memory (as hres for a single evaluation).  Separate from the handle's
    // own slab: the factorisation mfgp_predict works from survives a batch.
    int bsets = 0;
    int64_t bsets_cap = 0;               // the padded capacity (h->cap) the sets were allocated for
    double* bslab = nullptr;
    double *bz = nullptr, *balpha = nullptr, *blogdet = nullptr, *bpart = nullptr;
    double *bhres = nullptr, *bdres = nullptr;   // BRES doubles per set: [0,1] scalars, [30] pivot status, [64..] gradient, [128..] theta
    static constexpr int BRES = 256;
    long long* drow_off = nullptr;       // mfgp_allgather_rows: offset of every 128-row block's LOWER part inside its owner's chunk
    long long row_chunk = 0;             //   doubles per rank in that all-gather
    int row_off_cap = 0, row_off_nblk = 0, row_off_size = 0;   //   (capacity; the block count and communicator size the table was built for)
    double* ddist = nullptr;             // staging of a distributed Cholesky's exchange steps (one panel column, padded per rank)
    size_t dist_cap = 0;
    long long* dshard_off = nullptr;     // offset of every 128-row block inside its owner's chunk of the exchange (device copy)
    long long shard_chunk = 0;           // doubles per rank in the exchange's all-gather
    int shard_off_cap = 0;
    double *dctl = nullptr, *hctl = nullptr;   // control block of the leader / follower form of a sharded evaluation (device, pinned host)
    mfgp::Plan pls;                      // this rank's plan of a sharded evaluation (mfgp_eval_sharded), cached per (rank, size)
    mfgp::GemmTask* dtasks_s = nullptr;
    size_t tasks_s_cap = 0;
    mfgp::Plan plb;                      // the batch's own plan: same macro panels (same arithmetic), 128-tiles from fewer tiles per set
    int plb_div = 0;                     // the t128 divisor plb was built for (0: none yet)
    mfgp::GemmTask* dtasks_b = nullptr;
    size_t tasks_b_cap = 0;
};

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            if (h) (h)->err = m_; else mfgp::g_err = m_;                                        \
            return -2;                                                                          \
        }                                                                                       \
    } while (0)

namespace mfgp {
inline int fail(mfgp_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_err = msg;
    return code;
}
// releases the communicator of a handle (no-op without one); defined in comm_rccl.hip
void comm_release(mfgp_handle* h);
// tears it down without the peers (ncclCommAbort) and poisons the handle's collective calls
void comm_abort(mfgp_handle* h);
// wait for a stream that carries a collective: a poll with a deadline (MFGP_SHARD_TIMEOUT_S), never hipStreamSynchronize; past the
// deadline the communicator is aborted and the call fails with -4
int comm_stream_wait(mfgp_handle* h, hipStream_t s, const char* what);
// collectives of a sharded evaluation on the handle's communicator and stream (no-ops for a communicator of one / none):
//   -> 0 or a negative status (h->err set)
//   in-place all-gather of equal chunks: rank r's `chunk` doubles already sit at base + r * chunk
int comm_allgather_chunks(mfgp_handle* h, double* base, size_t chunk, hipStream_t s);
//   element-wise sum of `buf` over the ranks, in place
int comm_allreduce_sum(mfgp_handle* h, double* buf, size_t count, hipStream_t s);
//   `count` doubles at device address `dev` from rank `root` to every rank, in place
int comm_bcast_words(mfgp_handle* h, double* dev, size_t count, int root, hipStream_t s);

}  // namespace mfgp
