// plan.h -- the factorisation planner's types: PURE HOST C++ (no HIP), so that the planner also compiles with g++ into
// the CPU plan checker of tests/host_plan (sequential executor + cross-stream race check of every plan it emits).
#pragma once
#include <stdint.h>
#include <vector>

namespace mfgp {

constexpr int NB = 128;  // leaf block = padding granule = largest GEMM tile edge
constexpr int BK = 32;   // K-step of the tile GEMM (doubles)

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

enum Buf { BUF_A = 0, BUF_L = 1, BUF_S = 2, BUF_W = 3 };

// one launch of a plan
struct Step {
    int kind;  // StepKind: 0 = leaf, 1 = gemm, 2 = join (no launch: the stream only waits for wait_ev), 3 / 4 = the exchange steps
               // of a distributed Cholesky (Shard::dist; blk = the block column)
    int role;  // gemm kernel symbol: 0 recursion, 1 K^-1, 2 predictive variance, 3 serial-chain step (slim workgroups)
    int strm;  // 0 = main stream (the serial chain), 1 = bulk-update stream (look-ahead)
    int wait_ev, rec_ev;  // 1-based event indices (0 = none): wait before / record after the launch
    int blk;   // leaf block
    int tile, first, count, a, b, c, c2;  // gemm: tile edge, task range, operand / result buffers
    int gfirst, gcount; // gcount > 0: the task range to run INSTEAD of [first, first + count) when the gradient is wanted
                        // (the same tasks plus a chunk of the K^-1 accumulation, ordered longest first as a whole)
    int rec_ev_final;   // a second event recorded after the launch (the plan's final join), 0 = none
    int carry;          // STEP_COMM_PANEL only: 1 = the all-gather of block column blk also carries L, X and the leaf's words of diagonal
                        // block blk + 1 from its owner (the one-exchange-per-column form: no STEP_COMM_DIAG for column blk + 1)
};

// The planner's switches (DESIGN.md "Planner switches"): read from the environment ONCE when a handle plans for a size
// (plan_opts_from_env) and kept with the handle, so that the plan of its batched evaluations -- built later, when the first batch
// arrives -- is the same plan.  -1 / 0 = the size-dependent default (the measured best).
struct PlanOpts {
    int kind = 0;          // MFGP_PLAN: 0 sweep (default), 1 levels, 2 recursive
    int macro = 0;         // MFGP_MACRO: block columns per macro panel
    int shift = -1;        // MFGP_SHIFT
    int kinv_stream = -1;  // MFGP_KINV_STREAM
    int chain_slim = -1;   // MFGP_CHAIN_SLIM
    int t128_min = 0;      // MFGP_T128_MIN: tiles per launch from which 128-tiles are used
    int dist_chol = -1;    // MFGP_DIST_CHOL: a sharded evaluation's Cholesky distributed over the group too (Shard::dist): 1 / 0 force
                           // it on / off; default (-1): decided from the group's MEASURED collective latency (Shard::coll_us,
                           // dist_cholesky_pays) -- and off while nothing has been measured
    int dist_fuse = -1;    // MFGP_DIST_FUSE: the distributed Cholesky's two exchanges per block column fused into one wherever the next
                           // column belongs to the same macro panel (default 1; 0: round 5's 2 nblk - 1 collectives)
};
// Row ownership of a sharded evaluation (mfgp_eval_sharded): the work on the image of the identity / the rows of X^T and the rows
// of K^-1 -- 2 N^3 / 3 of an evaluation's N^3 flops -- splits by 128-row block with no dependency between blocks; block b belongs
// to rank shard_owner(b, size): block-cyclic in serpentine order (0 1 .. G-1 G-1 .. 1 0 ...), because the work of a block row
// falls with its index (rows of X^T: ~ (nb - b)^2) or peaks in the middle (rows of K^-1: ~ b (nb - b)).
// dist (round 5; SURVEY 8(e) "Cholesky": the 1-D block-cyclic multi-GPU factorisation for large N): the Cholesky's own work
// splits by the same ownership too.  Rank r runs the leaf of the diagonal blocks it owns, the rows of every panel column and of
// every trailing update of A that lie in its blocks -- N^3 / (3 G) instead of N^3 / 3 flops -- and the plan carries two exchange
// steps per block column c on the chain: COMM_DIAG(c) (L_cc, X_cc and the leaf's log-det / pivot words from owner(c) to
// everybody: 2 x 128 KB) before the panel, COMM_PANEL(c) (block column c of L below the diagonal, every 128-row block from its
// owner to everybody: 8 (N - 128 c) 128 bytes in all) after it.  Every rank ends with the complete L; the tasks it runs are the
// single evaluation's own tasks, so no result bit changes.
// One exchange per column (round 6; DESIGN section 8.1 of round 5): owner(c + 1) needs only ITS OWN row of panel c to bring
// A[c+1, c+1] up to date and run leaf(c + 1) -- it does so right after panel(c), before the all-gather of panel c, which then
// carries L, X and the leaf's words of block c + 1 along (Step::carry): COMM_DIAG(c + 1) disappears.  Possible wherever column
// c + 1 belongs to the same macro panel as c (its diagonal block is then complete once column c's K = 128 update has landed; the
// first column of a macro panel still waits for the previous macro's bulk update and keeps its broadcast):
// nblk + ceil(nblk / MB) - 1 collectives on the chain instead of 2 nblk - 1.
struct Shard {
    int rank = 0, size = 1;
    bool dist = false;     // (set by build_plan from PlanOpts::dist_chol, or from the measurement below)
    double coll_us = 0.0;  // microseconds one small collective of THIS group costs on the chain, measured on its own stream when the
                           // communicator was formed (mfgp_comm_calibrate: the slower of the medians of ncclBroadcast of a diagonal
                           // message and ncclAllGather of a panel chunk, rank 0's figures on every rank); 0: never measured
};
// Does distributing the Cholesky pay for a group of `size` ranks at `nblk` block columns, given what one collective costs?  The
// distributed plan saves (1 - 1/G) of the Cholesky's N^3 / 3 flops per rank -- priced at 60 TFLOP/s, what the one-GPU projections
// of round 5 gave for the saving at N = 8192 / 16384 / 32768 on 8 ranks (59 / 68 / 68: profiles/r05_dist_projection.txt) -- and
// puts 2 nblk - 1 collectives (each with its pack and unpack launch) on the serial chain.  It is taken when the saving exceeds
// 1.25 x that cost; never without a measurement (coll_us <= 0).  (With the exchanges of a column fused -- the default from round 6 --
// the count is nblk + ceil(nblk / MB) - 1: dist_collectives.)  Pure arithmetic on its arguments: every rank of a group holds
// the same coll_us and decides alike.
struct DistDecision {
    bool dist;
    double saving_ms, cost_ms;
    int collectives;
};
DistDecision dist_cholesky_pays(int nblk, int size, double coll_us, const PlanOpts& opts = PlanOpts());
// block columns per macro panel of the sweep at nblk block columns / exchange steps on the chain of a distributed Cholesky
int sweep_macro_columns(int nblk, const PlanOpts& opts);
int dist_collectives(int nblk, const PlanOpts& opts);
enum StepKind { STEP_LEAF = 0, STEP_GEMM = 1, STEP_JOIN = 2, STEP_COMM_DIAG = 3, STEP_COMM_PANEL = 4 };
inline int shard_owner(int blk, int size) {
    if (size <= 1) return 0;
    const int x = blk % (2 * size);
    return x < size ? x : 2 * size - 1 - x;
}

struct Plan {
    int nblk = 0;
    int64_t ld = 0;                 // = padded size Np
    int64_t stride = 0;             // elements between the four matrices A, L, S, W in the handle's slab
    size_t n_fixed_tasks = 0;       // tasks of the factorisation plan + the stand-alone K^-1 launch (predict plans follow)
    std::vector<GemmTask> tasks;
    std::vector<Step> steps;        // Cholesky + inverse (+ streamed K^-1)
    Step kinv_step{};               // stand-alone K^-1 = X^T X launch (lazy gradient after a gradient-free factorisation)
    Step predv_step{};              // predictive-variance product for predv_rows panel rows
    int predv_rows = 0;
    int n_events = 0;               // events the steps refer to (1-based ids 1..n_events)
    int t128_min = 300;             // tiles per launch from which 128-tiles are used (64-tiles below)
    int batch_div = 1;              // matrix sets per launch this plan was made for (1: a single evaluation)
    bool kinv_streamed = false;     // the steps accumulate K^-1 behind the chain
    PlanOpts opts;                  // the switches it was planned under
    Shard shard;                    // size > 1: this rank's share of a sharded evaluation (see Shard)
};

PlanOpts plan_opts_from_env();
// t128_div > 1: the plan of a BATCHED evaluation of about that many matrix sets per launch -- a launch carries t128_div times
// the tiles, so the 128-tile threshold is reached that much earlier (tile sizes do not change any result bit: a tile's
// elements accumulate over k in the same order in both kernels)
// shard.size > 1: the plan of ONE RANK of a sharded evaluation -- the Cholesky (chain, trailing updates of A) in full, the bulk
// work on B / X^T and the stand-alone K^-1 launch for the rank's own block rows only, K^-1 never streamed (it needs every rank's
// rows of X^T: it follows the exchange).  Tile sizes are chosen from the UNSHARDED tile counts; no result bit depends on them.
void build_plan(Plan& p, int nblk, int64_t ld, int64_t stride, const PlanOpts& opts = PlanOpts(), int t128_div = 1,
                const Shard& shard = Shard());
// (re)plan the predictive-variance product for a panel of rows_p rows; keeps everything planned before it
void plan_predv(Plan& p, int rows_p);

}  // namespace mfgp
