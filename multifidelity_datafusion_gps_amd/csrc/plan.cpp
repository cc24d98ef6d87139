// plan.cpp -- the planner: turns "factorise / invert / accumulate K^-1 / predictive variance" into lists of tile-GEMM
// tasks (gemm_f64.hip) and leaf launches (leaf_f64.hip) on two streams.  PURE HOST C++ (see plan.h): compiled into
// libmfgp_hip.so by hipcc and into the CPU plan checker (tests/host_plan) by g++.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "plan.h"

namespace mfgp {

// The planner's switches (environment, read when a handle plans; DESIGN.md "Planner switches").  Only those a GPU test runs
// (tests/test_gpu_plans.py VARIANTS) and the CPU plan checker covers: MFGP_PLAN, MFGP_MACRO, MFGP_SHIFT, MFGP_KINV_STREAM,
// MFGP_CHAIN_SLIM, MFGP_T128_MIN, and -- a sharded evaluation's plan only -- MFGP_DIST_CHOL (the Cholesky itself distributed over the rank
// group: tests/test_gpu_multirank.py, tests/test_plan_host.py).  Everything else the earlier rounds measured is a constant here (the measured best); the
// variants that lost their A/B live on in tools/gemm_lab/RETIRED.md.
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
PlanOpts plan_opts_from_env() {
    PlanOpts o;
    const char* pm = getenv("MFGP_PLAN");
    o.kind = (pm && strcmp(pm, "levels") == 0) ? 1 : ((pm && strcmp(pm, "recursive") == 0) ? 2 : 0);
    o.macro = std::max(0, env_int("MFGP_MACRO", 0));
    o.shift = env_int("MFGP_SHIFT", -1);
    o.kinv_stream = env_int("MFGP_KINV_STREAM", -1);
    o.chain_slim = env_int("MFGP_CHAIN_SLIM", -1);
    o.t128_min = std::max(0, env_int("MFGP_T128_MIN", 0));
    o.dist_chol = env_int("MFGP_DIST_CHOL", -1);
    o.dist_fuse = env_int("MFGP_DIST_FUSE", -1);
    return o;
}
static int opt(int v, int dflt) { return v >= 0 ? v : dflt; }

// ------------------------------------------------------------------------------------------------
// planner
// ------------------------------------------------------------------------------------------------
static void xcd_interleave(std::vector<GemmTask>& tasks, int first, int group);
static int pick_tile(const Plan& p, int ntiles128) {
    // 128-tiles run the MFMA pipe better, but a launch of few tiles is bound by its LONGEST tile (one tile per CU, 256
    // CUs): below ~300 tiles four times as many 64-tiles balance better (top inverse level at N = 4096: 2 x 330 -> 2 x 220 us).
    // With macro panels of K >= 768 (N >= 7168) a 128-tile lasts >= 130 us and the threshold moves to 600 (N = 8192: the
    // 364..448-tile column launches ran at 31 TFLOP/s as 128-tiles; whole evaluation 13.74 -> 13.36 ms).
    return ntiles128 >= p.t128_min ? 128 : 64;
}

static void add_gemm(Plan& p, std::vector<Step>& plan, int tile, int first, int a, int b, int c, int c2) {
    Step s{};
    s.kind = 1;
    s.tile = tile;
    s.first = first;
    s.count = (int)p.tasks.size() - first;
    s.a = a; s.b = b; s.c = c; s.c2 = c2;
    if (s.count > 0) plan.push_back(s);
}

// recursive Cholesky + inverse over leaf blocks [b0, b1)
static void plan_cholinv(Plan& p, int b0, int b1) {
    const int64_t ld = p.ld;
    if (b1 - b0 == 1) {
        Step s{};
        s.kind = 0;
        s.blk = b0;
        p.steps.push_back(s);
        return;
    }
    const int bm = b0 + (b1 - b0 + 1) / 2;
    plan_cholinv(p, b0, bm);
    const int n1 = bm - b0, n2 = b1 - bm;
    const int T = pick_tile(p, n1 * n2);
    const int sc = NB / T;
    const int64_t k0 = (int64_t)b0 * NB, km = (int64_t)bm * NB;
    // L21 = A21 * X11^T      (A: A, B: S lower rows j, C: L)
    {
        const int first = (int)p.tasks.size();
        for (int j = b0 * sc; j < bm * sc; ++j)       // long K first
            for (int i = bm * sc; i < b1 * sc; ++i) {
                GemmTask t{};
                t.a_off = (int64_t)i * T * ld + k0;
                t.b_off = (int64_t)j * T * ld + k0;
                t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                t.c2_off = -1;
                t.klen = (int)((int64_t)(j + 1) * T - k0);
                t.flags = TF_B_LOWER;
                t.alpha = 1.0; t.beta = 0.0;
                p.tasks.push_back(t);
            }
        // descending K length for load balance
        std::stable_sort(p.tasks.begin() + first, p.tasks.end(),
                         [](const GemmTask& x, const GemmTask& y) { return x.klen > y.klen; });
        add_gemm(p, p.steps, T, first, BUF_A, BUF_S, BUF_L, -1);
    }
    // A22 -= L21 L21^T       (A: L, B: L, C: A), lower tiles only
    {
        const int T2 = pick_tile(p, n2 * (n2 + 1) / 2);
        const int s2 = NB / T2;
        const int first = (int)p.tasks.size();
        for (int i = bm * s2; i < b1 * s2; ++i)
            for (int j = bm * s2; j <= i; ++j) {
                GemmTask t{};
                t.a_off = (int64_t)i * T2 * ld + k0;
                t.b_off = (int64_t)j * T2 * ld + k0;
                t.c_off = (int64_t)i * T2 * ld + (int64_t)j * T2;
                t.c2_off = -1;
                t.klen = n1 * NB;
                t.flags = 0;
                t.alpha = -1.0; t.beta = 1.0;
                p.tasks.push_back(t);
            }
        add_gemm(p, p.steps, T2, first, BUF_L, BUF_L, BUF_A, -1);
    }
    plan_cholinv(p, bm, b1);
    // P^T[j][i] = sum_{k>=j} X11^T[j][k] L21[i][k]     (A: S upper rows j, B: L rows i, C: W[j][i])
    {
        const int first = (int)p.tasks.size();
        for (int j = b0 * sc; j < bm * sc; ++j)
            for (int i = bm * sc; i < b1 * sc; ++i) {
                GemmTask t{};
                t.a_off = (int64_t)j * T * ld + (int64_t)j * T;
                t.b_off = (int64_t)i * T * ld + (int64_t)j * T;
                t.c_off = (int64_t)j * T * ld + (int64_t)i * T;
                t.c2_off = -1;
                t.klen = (int)(km - (int64_t)j * T);
                t.flags = TF_A_UPPER;
                t.alpha = 1.0; t.beta = 0.0;
                p.tasks.push_back(t);
            }
        add_gemm(p, p.steps, T, first, BUF_S, BUF_L, BUF_W, -1);
    }
    // X21[i][j] = - sum_{k<=i} X22[i][k] P^T[j][k]     (A: S lower rows i, B: W rows j, C: S lower + mirror)
    {
        const int first = (int)p.tasks.size();
        for (int i = b1 * sc - 1; i >= bm * sc; --i)  // long K first
            for (int j = b0 * sc; j < bm * sc; ++j) {
                GemmTask t{};
                t.a_off = (int64_t)i * T * ld + km;
                t.b_off = (int64_t)j * T * ld + km;
                t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                t.c2_off = (int64_t)j * T * ld + (int64_t)i * T;
                t.klen = (int)((int64_t)(i + 1) * T - km);
                t.flags = TF_A_LOWER;
                t.alpha = -1.0; t.beta = 0.0;
                p.tasks.push_back(t);
            }
        add_gemm(p, p.steps, T, first, BUF_S, BUF_W, BUF_S, BUF_S);
    }
}

// ------------------------------------------------------------------------------------------------
// plan B (default): right-looking blocked Cholesky (NB = 128) followed by a level-batched recursive
// triangular inverse.  Per block column k: leaf(k) factorises AND inverts the diagonal block, the
// panel below becomes L[i,k] = A[i,k] X_kk^T (tile GEMMs, K = 128), then the trailing SYRK update.
// The inverse X = L^-1 is then assembled bottom-up: every node of one tree level is independent, so a
// level is TWO launches (P^T = X11^T L21^T ; X21 = -X22 P) however many nodes it has.
// ------------------------------------------------------------------------------------------------
static int new_event(Plan& p) { return ++p.n_events; }   // 1-based

static void plan_potrf_rl(Plan& p) {
    // Blocked Cholesky with macro panels of MB leaf blocks and look-ahead over two streams (round 1's schedule: MFGP_PLAN=levels).
    //   main stream (the serial chain), for every block column c of a macro panel [M0, M1):
    //       leaf(c)     : L_cc, X_cc = L_cc^-1
    //       panel(c)    : L[i,c] = A[i,c] X_cc^T, i > c
    //       inner(c)    : A[i,j] -= L[i,c] L[j,c]^T for the macro's later columns j in (c, M1), K = 128: right-looking inside the
    //                     macro (a step on the chain costs launch + K-depth, and K = 128 three times beats 128 + 256 + 384:
    //                     N = 8192: 11.28 -> 11.04 ms, 4096: 3.57 -> 3.43, 2048: 1.46 -> 1.36 against the left-looking form)
    //   bulk stream, after the macro's chain: A[i,j] -= L[i,M0:M1] L[j,M0:M1]^T (K = MB*128), first the block
    //       columns of the NEXT macro panel in one launch (it releases the chain steps that need them), then the rest,
    //       which overlaps the next macro's chain.
    const int64_t ld = p.ld;
    const int nb = p.nblk;
    const int MB = p.opts.macro > 0 ? p.opts.macro : 4;   // 2 is ~1.5 % faster for one evaluation alone, 4 is ~5 % faster with evaluations in flight
    // `shift`: the chain's K = 128 inner updates also cover the NEXT macro panel's first column, so that no K = MB*128
    // step (and no wait for the previous macro's bulk update) gates its first leaf.  Pays where the factorisation is
    // chain-bound throughout (N = 4096: 3.42 -> 3.28 ms, 2048: 1.37 -> 1.28); neutral at N = 8192, where the first half
    // is bound by the bulk updates and the gating step's slack is worth as much as its latency.
    const bool shift = opt(p.opts.shift, nb < 48) != 0;
    // slim chain workgroups (role 3) co-reside with the bulk update's workgroups; alone they are ~30 % slower than the
    // double-buffered 64-tile kernel, so they only pay where bulk updates are long enough to overlap the chain
    const int chain_role = opt(p.opts.chain_slim, nb >= 48) ? 3 : 0;
    auto syrk_tasks = [&](int T, int jlo, int jhi, int klo, int khi) {
        // A[i,j] -= sum_{k in [klo,khi) blocks} L[i,k] L[j,k]^T for block columns j in [jlo,jhi), rows i >= j
        const int sc = NB / T;
        for (int j = jlo * sc; j < jhi * sc; ++j)
            for (int i = j; i < nb * sc; ++i) {
                GemmTask t{};
                t.a_off = (int64_t)i * T * ld + (int64_t)klo * NB;
                t.b_off = (int64_t)j * T * ld + (int64_t)klo * NB;
                t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                t.c2_off = -1;
                t.klen = (khi - klo) * NB;
                t.flags = 0;
                t.alpha = -1.0; t.beta = 1.0;
                p.tasks.push_back(t);
            }
    };
    auto ntiles_cols = [&](int jlo, int jhi) { int n = 0; for (int j = jlo; j < jhi; ++j) n += nb - j; return n; };
    std::vector<int> ev_col(nb, 0);  // event after the previous macro's bulk update reached block column c
    int ev_rest_prev = 0;            // event after the previous macro's bulk update of the REST (bulk stream)
    for (int M0 = 0; M0 < nb; M0 += MB) {
        const int M1 = std::min(M0 + MB, nb);
        const int M2 = std::min(M1 + MB, nb);
        int main_waited_ev = 0;   // the merged column launch signals ONE event for several columns: wait for it once
        for (int c = M0; c < M1; ++c) {
            Step s{};
            s.kind = 0;
            s.blk = c;
            if (ev_col[c] > 0 && ev_col[c] != main_waited_ev) s.wait_ev = main_waited_ev = ev_col[c];   // (an event wait costs ~6 us on the chain even when already signalled)
            p.steps.push_back(s);
            const int rem = nb - 1 - c;
            if (rem == 0) break;
            const int64_t kc = (int64_t)c * NB;
            {   // panel: L[i,c] = A[i,c] * X_cc^T
                const int T = pick_tile(p, rem);
                const int sc = NB / T;
                const int first = (int)p.tasks.size();
                for (int i = (c + 1) * sc; i < nb * sc; ++i)
                    for (int j = c * sc; j < (c + 1) * sc; ++j) {
                        GemmTask t{};
                        t.a_off = (int64_t)i * T * ld + kc;
                        t.b_off = (int64_t)j * T * ld + kc;
                        t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                        t.c2_off = -1;
                        t.klen = (int)((int64_t)(j + 1) * T - kc);
                        t.flags = TF_B_LOWER;
                        t.alpha = 1.0; t.beta = 0.0;
                        p.tasks.push_back(t);
                    }
                add_gemm(p, p.steps, T, first, BUF_A, BUF_S, BUF_L, -1);
                if (T == 64) p.steps.back().role = chain_role;
            }
            // right-looking inside the macro: column c -> the macro's later columns, K = 128.  `shift`: also -> the first
            // column of the NEXT macro panel, so that no K = MB*128 step gates its first leaf
            const int inner_hi = shift ? std::min(M1 + 1, nb) : M1;
            if (c + 1 < inner_hi) {
                const int first = (int)p.tasks.size();
                syrk_tasks(64, c + 1, inner_hi, c, c + 1);
                add_gemm(p, p.steps, 64, first, BUF_L, BUF_L, BUF_A, -1);
                p.steps.back().role = chain_role;
                const int e = ev_col[c + 1];
                if (e > 0 && e != main_waited_ev) p.steps.back().wait_ev = main_waited_ev = e;
            }
        }
        if (M1 >= nb) break;
        const int ev_chain = new_event(p);
        p.steps.back().rec_ev = ev_chain;   // chain(M) complete: every L[:, M0:M1] panel is final
        bool first_bulk = true;
        if (shift) {
            // the chain has already brought the next macro panel's first column up to date; the bulk stream takes the
            // columns (M1, M1+MB] in one launch (the next chain waits for its event once) and everything beyond in another
            const int lo = M1 + 1, hi = std::min(M1 + MB, nb - 1);
            if (lo <= hi) {
                const int T = pick_tile(p, ntiles_cols(lo, hi + 1));
                const int first = (int)p.tasks.size();
                syrk_tasks(T, lo, hi + 1, M0, M1);
                add_gemm(p, p.steps, T, first, BUF_L, BUF_L, BUF_A, -1);
                Step& st = p.steps.back();
                st.strm = 1;
                st.wait_ev = ev_chain;
                first_bulk = false;
                const int ev = new_event(p);
                st.rec_ev = ev;
                for (int cc = lo; cc <= hi; ++cc) ev_col[cc] = ev;
            }
            if (hi + 1 < nb) {
                const int T = pick_tile(p, ntiles_cols(hi + 1, nb));
                const int first = (int)p.tasks.size();
                syrk_tasks(T, hi + 1, nb, M0, M1);
                add_gemm(p, p.steps, T, first, BUF_L, BUF_L, BUF_A, -1);
                p.steps.back().strm = 1;
                if (first_bulk) p.steps.back().wait_ev = ev_chain;
            }
            continue;
        }
        {
            // the column that gates the next leaf stays on the MAIN stream: no event round trip on the chain.
            // It must still come after the previous macro's rest-update, which covers this column too and
            // runs on the bulk stream (normally long finished: the wait is on an already signalled event).
            const int T = pick_tile(p, nb - M1);
            const int first = (int)p.tasks.size();
            syrk_tasks(T, M1, M1 + 1, M0, M1);
            add_gemm(p, p.steps, T, first, BUF_L, BUF_L, BUF_A, -1);
            Step& st = p.steps.back();
            st.strm = 0;
            if (T == 64) st.role = chain_role;
            st.wait_ev = ev_rest_prev;
            ev_col[M1] = 0;
        }
        if (M1 + 1 < M2) {
            // the other block columns of the next macro panel: ONE launch on the bulk stream (each is needed one chain
            // step later than the previous; a launch per column left the GPU at ~140 workgroups three times in a row)
            const int cols = M2 - (M1 + 1);
            const int T = pick_tile(p, cols * (nb - M1 - 1));
            const int first = (int)p.tasks.size();
            syrk_tasks(T, M1 + 1, M2, M0, M1);
            add_gemm(p, p.steps, T, first, BUF_L, BUF_L, BUF_A, -1);
            Step& st = p.steps.back();
            st.strm = 1;
            st.wait_ev = ev_chain;
            first_bulk = false;
            const int ev = new_event(p);
            st.rec_ev = ev;
            for (int cc = M1 + 1; cc < M2; ++cc) ev_col[cc] = ev;
        }
        if (M2 < nb) {   // the rest of the trailing matrix: overlaps the next macro panel's chain
            const int T = pick_tile(p, ntiles_cols(M2, nb));
            const int first = (int)p.tasks.size();
            syrk_tasks(T, M2, nb, M0, M1);
            add_gemm(p, p.steps, T, first, BUF_L, BUF_L, BUF_A, -1);
            p.steps.back().strm = 1;
            if (first_bulk) p.steps.back().wait_ev = ev_chain;   // MB = 1: nothing else waited on the chain yet
            ev_rest_prev = new_event(p);
            p.steps.back().rec_ev = ev_rest_prev;
        } else {
            ev_rest_prev = 0;
        }
    }
}


struct TriNode { int b0, bm, b1, level; };
static int collect_nodes(int b0, int b1, std::vector<TriNode>& out) {
    if (b1 - b0 <= 1) return 0;
    const int bm = b0 + (b1 - b0 + 1) / 2;
    const int l = std::max(collect_nodes(b0, bm, out), collect_nodes(bm, b1, out)) + 1;
    out.push_back(TriNode{b0, bm, b1, l});
    return l;
}

static void plan_trtri_levels(Plan& p) {
    const int64_t ld = p.ld;
    std::vector<TriNode> nodes;
    const int top = collect_nodes(0, p.nblk, nodes);
    for (int lev = 1; lev <= top; ++lev) {
        int ntiles = 0;
        for (const TriNode& n : nodes)
            if (n.level == lev) ntiles += (n.bm - n.b0) * (n.b1 - n.bm);
        const int T = pick_tile(p, ntiles);
        const int sc = NB / T;
        {   // P^T[j][i] = sum_{k>=j} X11^T[j][k] L21[i][k]
            const int first = (int)p.tasks.size();
            for (const TriNode& n : nodes) {
                if (n.level != lev) continue;
                const int64_t km = (int64_t)n.bm * NB;
                for (int j = n.b0 * sc; j < n.bm * sc; ++j)
                    for (int i = n.bm * sc; i < n.b1 * sc; ++i) {
                        GemmTask t{};
                        t.a_off = (int64_t)j * T * ld + (int64_t)j * T;
                        t.b_off = (int64_t)i * T * ld + (int64_t)j * T;
                        t.c_off = (int64_t)j * T * ld + (int64_t)i * T;
                        t.c2_off = -1;
                        t.klen = (int)(km - (int64_t)j * T);
                        t.flags = TF_A_UPPER;
                        t.alpha = 1.0; t.beta = 0.0;
                        p.tasks.push_back(t);
                    }
            }
            std::stable_sort(p.tasks.begin() + first, p.tasks.end(),
                             [](const GemmTask& x, const GemmTask& y) { return x.klen > y.klen; });
            add_gemm(p, p.steps, T, first, BUF_S, BUF_L, BUF_W, -1);
        }
        {   // X21[i][j] = - sum_{k<=i} X22[i][k] P^T[j][k]
            const int first = (int)p.tasks.size();
            for (const TriNode& n : nodes) {
                if (n.level != lev) continue;
                const int64_t km = (int64_t)n.bm * NB;
                for (int i = n.bm * sc; i < n.b1 * sc; ++i)
                    for (int j = n.b0 * sc; j < n.bm * sc; ++j) {
                        GemmTask t{};
                        t.a_off = (int64_t)i * T * ld + km;
                        t.b_off = (int64_t)j * T * ld + km;
                        t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                        t.c2_off = (int64_t)j * T * ld + (int64_t)i * T;
                        t.klen = (int)((int64_t)(i + 1) * T - km);
                        t.flags = TF_A_LOWER;
                        t.alpha = -1.0; t.beta = 0.0;
                        p.tasks.push_back(t);
                    }
            }
            std::stable_sort(p.tasks.begin() + first, p.tasks.end(),
                             [](const GemmTask& x, const GemmTask& y) { return x.klen > y.klen; });
            add_gemm(p, p.steps, T, first, BUF_S, BUF_W, BUF_S, BUF_S);
        }
    }
}

// XCD-aware task placement.  Workgroup p of a launch lands on XCD p mod 8 (each XCD has its own L2); a task list in
// row-major tile order therefore hands every XCD tiles that share almost no operand panel.  `tasks[first..)` arrives
// in LOCALITY order (runs of `group` consecutive tasks = one compact block of output tiles sharing operand panels);
// the runs are dealt to the 8 XCDs in serpentine order (0..7, 7..0: the work per run decreases along the list, a plain
// round-robin would give XCD 0 the longest run of every round) and the per-XCD sequences interleaved, so that XCD x executes whole
// runs back to back.  Pure reordering: every tile's arithmetic (and the result bits) is unchanged.
static void xcd_interleave(std::vector<GemmTask>& tasks, int first, int group) {
    const int n = (int)tasks.size() - first;
    if (n < 8 * group) return;
    std::vector<GemmTask> lists[8];
    int g = 0;
    for (int t0 = 0; t0 < n; t0 += group, ++g)
        for (int t = t0; t < std::min(n, t0 + group); ++t)   // serpentine deal: the runs come in descending work
            lists[(g & 8) ? 7 - (g & 7) : (g & 7)].push_back(tasks[first + t]);
    int out = first;
    for (size_t m = 0; out < first + n; ++m)
        for (int x = 0; x < 8; ++x)
            if (m < lists[x].size()) tasks[out++] = lists[x][m];
}

static void plan_kinv(Plan& p) {
    const int64_t ld = p.ld;
    const int nb = p.nblk;
    const int kinv_t128_min = 600;   // N = 4096: 0.65 -> 0.54 ms, N = 3072: 0.48 -> 0.25 ms as 64-tiles
    const int T = nb * (nb + 1) / 2 >= kinv_t128_min ? 128 : 64;
    const int sc = NB / T;
    const int first = (int)p.tasks.size();
    // locality order: super-blocks of BI x BJ output tiles (BI row panels + BJ column panels feed BI*BJ tiles);
    // small i (= long K range) first.  Measured at N = 8192 (tools/sweep_kinv_order.sh): 1x8 3.73 ms / 3.05 GB fetched,
    // 4x8 3.85 ms / 2.28 GB, row-major without XCD placement 3.9 ms / 3.69 GB -- the launch is FMA-bound, so the
    // finer run (better balance over the XCDs) wins over the larger one (fewer panel re-reads).
    const int BI = 1, BJ = 8;
    const int nt = nb * sc;
    for (int i0 = 0; i0 < nt; i0 += BI)
        for (int j0 = 0; j0 <= std::min(nt - 1, i0 + BI - 1); j0 += BJ)
            for (int i = i0; i < std::min(nt, i0 + BI); ++i)
                for (int j = j0; j < std::min(j0 + BJ, i + 1); ++j) {
                    if (shard_owner(i * T / NB, p.shard.size) != p.shard.rank) continue;   // a sharded evaluation: own block rows only
                    GemmTask t{};
                    t.a_off = (int64_t)i * T * ld + (int64_t)i * T;
                    t.b_off = (int64_t)j * T * ld + (int64_t)i * T;
                    t.c_off = (int64_t)i * T * ld + (int64_t)j * T;
                    t.c2_off = -1;
                    t.klen = (int)(p.ld - (int64_t)i * T);
                    t.flags = TF_A_UPPER | (i == j ? TF_B_UPPER : 0);
                    t.alpha = 1.0; t.beta = 0.0;
                    p.tasks.push_back(t);
                }
    xcd_interleave(p.tasks, first, BI * BJ);
    std::vector<Step> tmp;
    add_gemm(p, tmp, T, first, BUF_S, BUF_S, BUF_A, -1);
    p.kinv_step = tmp.empty() ? Step{} : tmp[0];
    p.kinv_step.role = 1;
}

// V[r][i] = sum_{k<=i} Kx[r][k] X[i][k]   (A: W = Kx panel, B: S lower rows i, C: A)
void plan_predv(Plan& p, int rows_p) {
    p.tasks.resize(p.n_fixed_tasks);   // drop the product planned for another panel height, keep everything else
    const int64_t ld = p.ld;
    const int nb = p.nblk, rb = rows_p / NB;
    const int T = pick_tile(p, nb * rb);
    const int sc = NB / T;
    const int first = (int)p.tasks.size();
    // super-blocks: BI rows of X  x  BR panel rows.  Fabric reads (FETCH_SIZE x 2) of the launch at N = N* = 8192 by shape,
    // time unchanged throughout (round 4, gpurun_out/r04k): 8 x 4 14.2 GB, 16 x 4 14.3, 4 x 16 15.1, 4 x 8 13.0, 8 x 8 12.9
    const int BI = 8, BR = 8;
    const int ni = nb * sc, nr = rb * sc;
    for (int i0 = ni - 1; i0 >= 0; i0 -= BI)                 // large i (= long K range) first
        for (int r0 = 0; r0 < nr; r0 += BR)
            for (int i = i0; i > std::max(-1, i0 - BI); --i)
                for (int r = r0; r < std::min(nr, r0 + BR); ++r) {
                    GemmTask t{};
                    t.a_off = (int64_t)r * T * ld;
                    t.b_off = (int64_t)i * T * ld;
                    t.c_off = (int64_t)r * T * ld + (int64_t)i * T;
                    t.c2_off = -1;
                    t.klen = (int)((int64_t)(i + 1) * T);
                    t.flags = TF_B_LOWER;
                    t.alpha = 1.0; t.beta = 0.0;
                    p.tasks.push_back(t);
                }
    xcd_interleave(p.tasks, first, BI * BR);
    std::vector<Step> tmp;
    add_gemm(p, tmp, T, first, BUF_W, BUF_S, BUF_A, -1);
    p.predv_step = tmp.empty() ? Step{} : tmp[0];
    p.predv_step.role = 2;
    p.predv_rows = rows_p;
}


// ------------------------------------------------------------------------------------------------
// plan C (default, "sweep"): ONE right-looking sweep over the block columns produces L, X = L^-1 and (when the gradient
// is wanted) K^-1 = X^T X; everything except the Cholesky's own serial chain is bulk work that streams BEHIND the chain
// on the second stream instead of following it as separate phases.
//
// The inverse rides on the factorisation as the augmented system [A; I]: the column operations that turn A into L turn
// I into X^T (A L^-T = L, I L^-T = X^T).  With B the running image of I (B[i,j] = -sum_{k=i}^{j-1} X^T[i,k] L[j,k]^T,
// kept in the upper part of the workspace W):
//     X^T[i,c] = B[i,c] X_cc^T                       (a "panel" row above the diagonal, same product as L[i,c] = A[i,c] X_cc^T)
//     B[i,j]  -= X^T[i,c] L[j,c]^T ,  i <= c < j      (a "trailing update" above the diagonal, same product as the SYRK)
// and K^-1 accumulates as the columns of X^T complete:  K^-1[i,j] += X^T[i,c] X^T[j,c]^T , j <= i <= c, written over the
// dead part of A.  N^3/3 flops each, exactly those of dtrtri and dpotri/lauum -- as K = 512 macro-panel GEMMs.
//
// Per macro panel M = [M0, M1) of MB block columns:
//   main stream (serial chain), per column c:   leaf(c);  panel(c): L[i,c] for i > c  +  X^T[i,c] for M0 <= i < c;
//       inner(c): K = 128 updates of the macro's later columns of A  +  of B[M0..c, c+1..M1)
//   bulk stream, after chain(M):  the next macro's columns of A (release its chain);  X^T[0:M0, M] = B[0:M0, M] X_MM^T;
//       then ONE launch: the rest of A's trailing update  +  B[0:M1, M1:] update  (+ K^-1[0:M1, 0:M1] update, gradient only).
// Task offsets are ABSOLUTE within the handle's single slab of the four matrices (offset = buffer * stride + row * ld +
// col), so one launch mixes tasks whose operands live in different matrices.
// ------------------------------------------------------------------------------------------------
static void plan_sweep(Plan& p) {
    const int64_t ld = p.ld, bs = p.stride;
    const int nb = p.nblk;
    // block columns per macro panel = K / 128 of the bulk trailing update.  Chain-bound sizes: 4 (a short in-macro chain);
    // bulk-bound sizes: the longer K runs the matrix pipe better, but a 128-tile of K = 1024 holds its CU for 170 us and
    // the leaf (which needs a whole CU) waits for one to retire (one evaluation alone, N = 6144: 6.66 / 6.91 / 7.37 ms at
    // 4 / 6 / 8; N = 8192: 13.9 / 13.2 / 13.05 at 4 / 6 / 8 and, with three evaluations in flight (bench.py), 2074 / 2011 / 1999 /
    // 2049 ms per fit+predict at 4 / 6 / 8 / 12; N = 16384: 96.8 / 94.1 / 93.1 ms at 8 / 12 / 16)
    // Chain-bound sizes (one evaluation alone, ms at MB = 2 / 3 / 4): N = 2048 0.94 / 0.97 / 1.02, 3072 1.63 / 1.72 / 1.84; with the
    // X^T rows sharing the column launch: 4096 2.63 / 2.79 / 2.84, 5120 4.82 / 4.21 / 4.31, 5632 5.80 / 5.46 / 5.36.
    // Up to 13 block columns ONE macro panel (and then one stream and K^-1 on the chain, see the end): N = 1024 0.527 -> 0.42,
    // 1280 0.60 -> 0.55, 1536 0.72 -> 0.68 (1792: 0.825 either way, 2048: 0.94 / 0.99 in favour of two-column panels).
    // Between: 6 for 52 .. 63 block columns (N = 6656 7.96 -> 7.81 ms, 7040 9.35 -> 8.95, 7680 11.3 -> 11.0 against 4 / 8).
    // Round 3 (bulk kernels on the 4x4x4 MFMA, 52-62 TFLOP/s per launch instead of 47-50): the bulk stream is no longer the
    // longer side, the chain is -- and the in-macro K = 128 updates are chain work, so SHORTER macro panels win at every size
    // (profiles/r03_macro_sweep.txt, one evaluation alone, ms at MB = 2 / 3 / 4 / 5 / 6 / 8 / 16):
    //   N = 2048 0.88 / 0.83 / 0.85;  3072 1.55 / 1.42 / 1.56;  3584 1.82 / 2.11 / 2.15;  4096 2.32 / 2.50 / 2.65;
    //   6144 6.02 / 6.21 / 6.22 / 6.24 / 6.29;  8192 12.27 / 12.05 / 11.80 / 11.92 / 12.00 / 12.35 / 13.5;
    //   12288 - / - / 34.50 / 34.48 / 34.66 / 36.0 / 37.2;  16384 - / - / 75.9 / 75.7 / 77.6 / 77.9 / 81.7
    // (rounds 1-2 had 2 / 3 / 4 / 6 / 8 / 16 from 14 / 33 / 41 / 52 / 64 / 96 block columns up).
    const int MB = sweep_macro_columns(nb, p.opts);
    const bool shift = opt(p.opts.shift, nb < 48) != 0;   // the chain's K = 128 updates also cover the next macro's first column (see plan_potrf_rl)
    // slim (16 KB LDS) chain workgroups fit on a CU beside a 128 KB bulk workgroup; the 64 KB kernel would wait for one to
    // retire (N = 4096: 3.27 -> 3.03 ms).  Below ~24 blocks the bulk launches are 64-tiles themselves: no difference.
    const int chain_role = opt(p.opts.chain_slim, nb >= 24) ? 3 : 0;
    // tile edge of the chain's own launches (panel, inner).  They are latency-bound: a 64x64x128 tile is 128 dependent-ish
    // MFMAs per SIMD (5.3 us) behind 8 serial K-steps; as 32x32 tiles (4 waves, 16 KB of LDS, four times the workgroups, four
    // K-steps) the same work spreads over four times as many SIMDs (N = 1024 0.60 -> 0.53 ms, 2048 1.20 -> 1.03, 4096 2.93 -> 2.80).  Chain-bound sizes only: at N >= 6144 the chain hides behind the
    // bulk stream and fewer, larger workgroups disturb it less.
    // A BATCHED pass of three or more sets carries that many times the workgroups per chain launch: from N = 4096 the 64-tiles win
    // there too (B = 4: 7.17 -> 6.86 ms per pass, B = 6: 10.05 -> 9.71, B = 8: 12.87 -> 12.55; B = 3 level); below they lose (N = 2048,
    // B = 4: 1.47 -> 1.57; 1024: 0.51 -> 0.57).  The tile edge changes no result bit.
    const int CT = (nb < 48 && !(p.batch_div >= 3 && nb >= 32)) ? 32 : 64;
    const int chain_role_ct = CT == 32 ? 5 : chain_role;
    p.kinv_streamed = opt(p.opts.kinv_stream, 1) != 0 && p.shard.size <= 1;   // (sharded: K^-1 follows the exchange of the rows of X^T)
    const auto mine = [&](int64_t row) { return shard_owner((int)(row / NB), p.shard.size) == p.shard.rank; };
    const bool dist = p.shard.size > 1 && p.shard.dist;   // the Cholesky's own rows by owner too (plan.h Shard)
    int kinv_lo = 0;    // first block column whose contribution to K^-1 is still outstanding
    const bool merge_xpanel = nb > 24;    // fewer, fuller launches on the bulk stream (N = 3072: 1.67 without / 1.80 with; 3584: 2.38 / 2.17)
    // One macro panel (nothing runs beside the chain): K^-1 is accumulated column by column in the chain's own K = 128 launches
    // instead of one long-K launch at the end (N = 1024: 136 64-tiles of K <= 1024, 58 us on 136 CUs)
    const bool kinv_on_chain = p.kinv_streamed && nb <= MB;
    auto at = [&](int buf, int64_t row, int64_t col) { return (int64_t)buf * bs + row * ld + col; };
    auto push = [&](int64_t a, int64_t b, int64_t c, int64_t c2, int klen, int flags, double alpha, double beta) {
        GemmTask t{};
        t.a_off = a; t.b_off = b; t.c_off = c; t.c2_off = c2; t.klen = klen; t.flags = flags; t.alpha = alpha; t.beta = beta;
        p.tasks.push_back(t);
    };
    // A[i,j] -= L[i,klo:khi] L[j,klo:khi]^T for block columns j in [jlo,jhi), rows i >= j
    // Tiles are enumerated in SUPER-BLOCKS of bulk_bi x bulk_bj output tiles (bi row panels + bj column panels feed
    // bi*bj tiles): after the XCD-aware deal below the workgroups that run side by side on one XCD (own L2) share their
    // operand panels instead of each streaming its own pair from HBM / Infinity Cache.
    const int bulk_bi = 4, bulk_bj = 4;   // (round 4: 8 x 4 19.2 -> 17.2 GB of fabric reads per sweep at +0.8 % time, 8 x 8 18.1: not taken)  (sweep fetch 22.8 -> 17.9 -> 16.8 GB from row-major over 4x4 super-blocks to the XCD-aware deal)
    auto in_blocks = [&](int ilo, int ihi, int jlo, int jhi, auto&& fn) {   // fn(i, j) over [ilo,ihi) x [jlo,jhi)
        for (int i0 = ilo; i0 < ihi; i0 += bulk_bi)
            for (int j0 = jlo; j0 < jhi; j0 += bulk_bj)
                for (int i = i0; i < std::min(ihi, i0 + bulk_bi); ++i)
                    for (int j = j0; j < std::min(jhi, j0 + bulk_bj); ++j) fn(i, j);
    };
    // only_diag >= 0: only the tiles of diagonal block only_diag; skip_diag >= 0: everything but those (the one-exchange-per-column
    // form splits column c's K = 128 update of A: owner(c + 1) runs the diagonal block's tiles early, before the exchange)
    auto a_update = [&](int T, int jlo, int jhi, int klo, int khi, int only_diag = -1, int skip_diag = -1) {
        const int sc = NB / T;
        in_blocks(jlo * sc, nb * sc, jlo * sc, jhi * sc, [&](int i, int j) {
            if (i < j) return;
            if (dist && !mine((int64_t)i * T)) return;
            const bool on_diag_blk = i / sc == j / sc;
            if (only_diag >= 0 && !(on_diag_blk && i / sc == only_diag)) return;
            if (skip_diag >= 0 && on_diag_blk && i / sc == skip_diag) return;
            push(at(BUF_L, (int64_t)i * T, (int64_t)klo * NB), at(BUF_L, (int64_t)j * T, (int64_t)klo * NB),
                 at(BUF_A, (int64_t)i * T, (int64_t)j * T), -1, (khi - klo) * NB, 0, -1.0, 1.0);
        });
    };
    // L[i,c] = A[i,c] X_cc^T, block rows i in (c, nb)
    auto l_panel = [&](int T, int c) {
        const int sc = NB / T;
        const int64_t kc = (int64_t)c * NB;
        for (int i = (c + 1) * sc; i < nb * sc; ++i)
            for (int j = c * sc; j < (c + 1) * sc; ++j)
                if (!dist || mine((int64_t)i * T))
                push(at(BUF_A, (int64_t)i * T, kc), at(BUF_S, (int64_t)j * T, kc), at(BUF_L, (int64_t)i * T, (int64_t)j * T), -1,
                     (int)((int64_t)(j + 1) * T - kc), TF_B_LOWER, 1.0, 0.0);
    };
    // X^T[i, c] = B[i, klo*NB : (c+1)*NB) X[c, same]^T for block rows i in [ilo, ihi): result into S (upper) + mirror
    auto x_panel = [&](int T, int c, int klo, int ilo, int ihi, bool own_only = false) {
        const int sc = NB / T;
        const int64_t k0 = (int64_t)klo * NB;
        for (int i = ilo * sc; i < ihi * sc; ++i)
            for (int j = c * sc; j < (c + 1) * sc; ++j)
                if (!own_only || mine((int64_t)i * T))
                push(at(BUF_W, (int64_t)i * T, k0), at(BUF_S, (int64_t)j * T, k0), at(BUF_S, (int64_t)i * T, (int64_t)j * T),
                     at(BUF_S, (int64_t)j * T, (int64_t)i * T), (int)((int64_t)(j + 1) * T - k0), TF_B_LOWER, 1.0, 0.0);
    };
    // B[i,j] -= X^T[i, klo:khi] L[j, klo:khi]^T for block rows i in [ilo, ihi), block columns j in [jlo, jhi); a row that
    // lies inside [klo, khi) starts at its own diagonal block (X^T is upper triangular) and is the FIRST touch of its
    // B tiles (beta = 0).  Used K = 128 deep on the chain (inside the macro panel) and K = chunk deep on the bulk stream.
    auto b_update = [&](int T, int ilo, int ihi, int jlo, int jhi, int klo, int khi, bool own_only = false) {
        const int sc = NB / T;
        in_blocks(ilo * sc, ihi * sc, jlo * sc, jhi * sc, [&](int i, int j) {
            if (own_only && !mine((int64_t)i * T)) return;
            const bool inside = (int64_t)i * T >= (int64_t)klo * NB;
            const int64_t k0 = inside ? (int64_t)i * T : (int64_t)klo * NB;
            push(at(BUF_S, (int64_t)i * T, k0), at(BUF_L, (int64_t)j * T, k0), at(BUF_W, (int64_t)i * T, (int64_t)j * T), -1,
                 (int)((int64_t)khi * NB - k0), inside ? TF_A_UPPER : 0, -1.0, inside ? 0.0 : 1.0);
        });
    };
    // K^-1[i,j] (+)= X^T[i, klo:khi] X^T[j, klo:khi]^T for block rows i < khi, j <= i (lower tiles, over the dead part of A)
    auto kinv_update = [&](int T, int klo, int khi) {
        const int sc = NB / T;
        in_blocks(0, khi * sc, 0, khi * sc, [&](int i, int j) {
            if (j > i) return;
            const bool inside = (int64_t)i * T >= (int64_t)klo * NB;
            const int64_t k0 = inside ? (int64_t)i * T : (int64_t)klo * NB;
            // the gradient reduction reads K^-1 in 64x64 tiles and the diagonal ones in full: a 32-tile below the diagonal of
            // such a block also writes its mirror image
            const bool mirror = T < 64 && i != j && ((int64_t)i * T) / 64 == ((int64_t)j * T) / 64;
            push(at(BUF_S, (int64_t)i * T, k0), at(BUF_S, (int64_t)j * T, k0), at(BUF_A, (int64_t)i * T, (int64_t)j * T),
                 mirror ? at(BUF_A, (int64_t)j * T, (int64_t)i * T) : -1,
                 (int)((int64_t)khi * NB - k0), inside ? (TF_A_UPPER | (i == j ? TF_B_UPPER : 0)) : 0, 1.0,
                 inside ? 0.0 : 1.0);
        });
    };
    auto launch = [&](int T, int first, int strm, int role) -> Step* {
        Step s{};
        s.kind = 1; s.tile = T; s.first = first; s.count = (int)p.tasks.size() - first;
        s.a = s.b = s.c = s.c2 = BUF_A;   // absolute offsets: every operand is addressed from the slab base
        s.strm = strm; s.role = role;
        if (s.count <= 0) return nullptr;
        p.steps.push_back(s);
        return &p.steps.back();
    };
    auto ntiles_cols = [&](int jlo, int jhi) { int n = 0; for (int j = jlo; j < jhi; ++j) n += nb - j; return n; };

    std::vector<int> ev_col(nb, 0);  // event after the previous macro's bulk update reached block column c
    int ev_rest_prev = 0;            // event after the previous macro's last bulk launch
    bool bulk_used = false;
    // Events recorded on the bulk stream are ordered; once the main stream has waited for one, every earlier one is
    // implied (an event wait costs ~6 us on the chain even when the event signalled long ago: never wait twice)
    std::vector<int> bulk_order(1, 0);   // event id -> position among the bulk stream's records (0 = not a bulk event)
    int main_waited_order = 0, n_bulk_records = 0;
    auto bulk_event = [&]() {
        const int e = new_event(p);
        bulk_order.resize(e + 1, 0);
        bulk_order[e] = ++n_bulk_records;
        return e;
    };
    auto main_wait = [&](Step& st, int e) {   // make a main-stream step wait for bulk event e unless that is implied
        if (e <= 0 || bulk_order[e] <= main_waited_order) return;
        st.wait_ev = e;
        main_waited_order = bulk_order[e];
    };
    for (int M0 = 0; M0 < nb; M0 += MB) {
        const int M1 = std::min(M0 + MB, nb);
        const int M2 = std::min(M1 + MB, nb);
        // ---- the chain of this macro panel (main stream) ----
        for (int c = M0; c < M1; ++c) {
            // one exchange per column: column c's diagonal block came with the all-gather of panel c - 1 (its owner factorised it
            // early, see below) wherever c is not the first column of its macro panel
            const bool fused = dist && opt(p.opts.dist_fuse, 1) != 0;
            const bool came_early = fused && c > M0;
            if (!came_early && (!dist || mine((int64_t)c * NB))) {
                Step s{};
                s.kind = STEP_LEAF;
                s.blk = c;
                main_wait(s, ev_col[c]);
                p.steps.push_back(s);
            }
            if (dist && !came_early) {   // L_cc, X_cc from the rank that factorised them (main_wait: a rank without the leaf meets column c here first)
                Step s{};
                s.kind = STEP_COMM_DIAG;
                s.blk = c;
                main_wait(s, ev_col[c]);
                p.steps.push_back(s);
            }
            {   // panel(c): the column of L below the diagonal and the in-macro part of the column of X^T above it
                const int rows = (nb - 1 - c) + (c - M0);
                if (rows > 0) {
                    const int T = pick_tile(p, rows) == 128 ? 128 : CT;
                    const int first = (int)p.tasks.size();
                    l_panel(T, c);
                    x_panel(T, c, c, M0, c);
                    launch(T, first, 0, T == 128 ? 0 : chain_role_ct);
                }
            }
            const bool carry = fused && c + 1 < M1;     // the next column's diagonal block travels with this column's panel
            const bool early_here = carry && mine((int64_t)(c + 1) * NB);
            if (early_here) {
                // this rank owns block c + 1: its own row of panel c is all it needs -- the diagonal block's share of column c's
                // K = 128 update (the SAME tiles inner(c) would run: no result bit changes), then the leaf, before the exchange
                const int first = (int)p.tasks.size();
                a_update(CT, c + 1, c + 2, c, c + 1, c + 1);
                Step* st = launch(CT, first, 0, chain_role_ct);
                if (st) main_wait(*st, ev_col[c + 1]);
                Step s{};
                s.kind = STEP_LEAF;
                s.blk = c + 1;
                main_wait(s, ev_col[c + 1]);
                p.steps.push_back(s);
            }
            if (dist && c + 1 < nb) {   // block column c of L: every rank's rows to everybody (the updates read L[j, c] of every row j)
                Step s{};
                s.kind = STEP_COMM_PANEL;
                s.blk = c;
                s.carry = carry ? 1 : 0;
                if (carry) main_wait(s, ev_col[c + 1]);   // (a rank without the early leaf meets block c + 1 here first)
                p.steps.push_back(s);
            }
            {   // inner(c): right-looking K = 128 updates inside the macro (A: also the next macro's first column if `shift`)
                const int inner_hi = shift ? std::min(M1 + 1, nb) : M1;
                auto updates = [&]() {
                    if (c + 1 < inner_hi) a_update(CT, c + 1, inner_hi, c, c + 1, -1, early_here ? c + 1 : -1);
                    if (c + 1 < M1) b_update(CT, M0, c + 1, c + 1, M1, c, c + 1);
                };
                const int first = (int)p.tasks.size();
                updates();
                const int count = (int)p.tasks.size() - first;
                int gfirst = 0, gcount = 0;
                if (kinv_on_chain) {   // gradient variant: column c's contribution to K^-1[0:c+1, 0:c+1] rides along
                    gfirst = (int)p.tasks.size();
                    kinv_update(CT, c, c + 1);
                    updates();
                    gcount = (int)p.tasks.size() - gfirst;
                    kinv_lo = c + 1;
                }
                if (count > 0 || gcount > 0) {
                    Step st{};
                    st.kind = 1; st.tile = CT; st.first = first; st.count = count; st.gfirst = gfirst; st.gcount = gcount;
                    st.a = st.b = st.c = st.c2 = BUF_A;
                    st.strm = 0; st.role = chain_role_ct;
                    // the columns it touches were last written by the previous macro's bulk launches
                    for (int j = c + 1; j < inner_hi; ++j) main_wait(st, ev_col[j]);
                    p.steps.push_back(st);
                }
            }
        }
        // ---- bulk work released by this chain ----
        const bool last = (M1 >= nb);
        const size_t chain_last = p.steps.size() - 1;   // index of the chain's last step
        std::vector<size_t> bulk_steps;                 // bulk launches of this macro, in stream order
        if (!last && !shift) {
            // the column that gates the next leaf stays on the MAIN stream (no event round trip on the chain); it must
            // still follow the previous macro's last bulk launch, whose A-rest part covers this column too
            const int T = pick_tile(p, nb - M1);
            const int first = (int)p.tasks.size();
            a_update(T, M1, M1 + 1, M0, M1);
            Step* st = launch(T, first, 0, T == 64 ? chain_role : 0);
            if (st) main_wait(*st, ev_rest_prev);
            ev_col[M1] = 0;
        }
        // the other block columns of the next macro panel (the next chain waits for their event once) and
        // X^T[0:M0, M] = B[0:M0, M] X_MM^T (rows above the macro; the in-macro rows came with the chain).  Both need only
        // chain(M); where the bulk stream is the bottleneck (merge_xpanel) they share ONE launch, where the chain is
        // (small N) the column update goes first on its own so that the next chain is released as early as possible.
        {
            const int lo = M1 + 1, hi = last ? M1 : (shift ? std::min(M1 + MB, nb - 1) : M2 - 1);
            const bool have_cols = !last && lo <= hi, have_x = M0 > 0;
            const int n_cols = have_cols ? ntiles_cols(lo, hi + 1) : 0, n_x = have_x ? M0 * (M1 - M0) : 0;
            auto cols_launch = [&](int T, int first) {
                if (!launch(T, first, 1, 0)) return;
                bulk_steps.push_back(p.steps.size() - 1);
                if (have_cols) {
                    const int ev = bulk_event();
                    p.steps.back().rec_ev = ev;
                    for (int cc = lo; cc <= hi; ++cc) ev_col[cc] = ev;
                }
            };
            if (merge_xpanel && have_cols && have_x) {
                const int T = pick_tile(p, n_cols + n_x);
                const int first = (int)p.tasks.size();
                a_update(T, lo, hi + 1, M0, M1);
                for (int c = M0; c < M1; ++c) x_panel(T, c, M0, 0, M0, true);
                cols_launch(T, first);
            } else {
                if (have_cols) {
                    const int T = pick_tile(p, n_cols);
                    const int first = (int)p.tasks.size();
                    a_update(T, lo, hi + 1, M0, M1);
                    cols_launch(T, first);
                }
                if (have_x) {
                    const int T = pick_tile(p, n_x);
                    const int first = (int)p.tasks.size();
                    for (int c = M0; c < M1; ++c) x_panel(T, c, M0, 0, M0, true);
                    if (launch(T, first, 1, 0)) bulk_steps.push_back(p.steps.size() - 1);
                }
            }
        }
        {   // the rest, ONE launch: A's trailing update beyond the columns already done; B: the NEXT macro's columns are brought up
            // to date (they are consumed right after the next chain) and the columns beyond; and -- gradient only -- this macro
            // panel's contribution to K^-1.  (Chunks of several macro panels -- longer K per task -- were measured in rounds 2-3:
            // N = 8192, MB = 8: 14.1 / 14.6 ms at 1 / 2 panels per chunk; one panel at a time is what stayed.)
            const int a_lo = last ? nb : (shift ? std::min(M1 + MB, nb - 1) + 1 : M2);
            const bool kinv_now = p.kinv_streamed && kinv_lo < M1;
            const int n_a = a_lo < nb ? ntiles_cols(a_lo, nb) : 0;
            const int n_b = last ? 0 : M1 * (nb - M1);
            const int n_k = kinv_now ? M1 * (M1 + 1) / 2 : 0;
            const int T = pick_tile(p, n_a + n_b + n_k);
            auto by_length = [&](int first) {   // longest K first: the launch's tail is then made of its shortest tasks;
                std::stable_sort(p.tasks.begin() + first, p.tasks.end(),     // super-blocks stay together within a K class
                                 [](const GemmTask& x, const GemmTask& y) { return x.klen > y.klen; });
                xcd_interleave(p.tasks, first, bulk_bi * bulk_bj);            // deal the super-blocks to the 8 XCDs (workgroup p runs on XCD p mod 8)
            };
            auto common = [&]() {
                if (!last) b_update(T, 0, M1, M1, M2, M0, M1, true);           // catch-up of the next macro's columns
                if (!last && M2 < nb) b_update(T, 0, M1, M2, nb, M0, M1, true);   // the columns beyond
                if (a_lo < nb) a_update(T, a_lo, nb, M0, M1);
            };
            const int first = (int)p.tasks.size();
            common();
            by_length(first);
            const int count = (int)p.tasks.size() - first;
            int gfirst = 0, gcount = 0;
            if (kinv_now) {   // the gradient variant of this launch: the same tasks + the K^-1 chunk, ordered as a whole
                gfirst = (int)p.tasks.size();
                kinv_update(T, kinv_lo, M1);
                common();
                by_length(gfirst);
                gcount = (int)p.tasks.size() - gfirst;
                kinv_lo = M1;
            }
            if (count > 0 || gcount > 0) {
                Step st{};
                st.kind = 1; st.tile = T; st.first = first; st.count = count; st.gfirst = gfirst; st.gcount = gcount;
                st.a = st.b = st.c = st.c2 = BUF_A;
                st.strm = 1;
                p.steps.push_back(st);
                bulk_steps.push_back(p.steps.size() - 1);
            }
        }
        if (!bulk_steps.empty()) {
            bulk_used = true;
            const int ev_chain = new_event(p);
            bulk_order.resize(ev_chain + 1, 0);
            p.steps[chain_last].rec_ev = ev_chain;            // chain(M) complete: L[:, M], X_MM, B's in-macro part are final
            p.steps[bulk_steps.front()].wait_ev = ev_chain;   // the bulk stream's first step of this macro waits for the chain
            if (!last) {
                ev_rest_prev = bulk_event();
                Step& lastb = p.steps[bulk_steps.back()];
                if (lastb.rec_ev == 0) lastb.rec_ev = ev_rest_prev; else lastb.rec_ev_final = ev_rest_prev;
            }
        } else if (!last) {
            ev_rest_prev = 0;
        }
    }
    if (nb <= MB) {
        // a single macro panel: its bulk work depends on the whole chain and nothing runs beside it -- keep it on the main
        // stream and save the two event hops (chain -> bulk, bulk -> join: ~6 us each; N = 128: 85 -> 71 us per evaluation)
        for (Step& st : p.steps) { st.strm = 0; st.wait_ev = st.rec_ev = st.rec_ev_final = 0; }
        bulk_used = false;
    }
    if (bulk_used) {   // join: whatever follows on the main stream (solve, gradient) needs the bulk stream's results
        const int ev = new_event(p);
        // the last bulk launch in the list is the last on its stream
        for (size_t i = p.steps.size(); i-- > 0;)
            if (p.steps[i].strm == 1) {
                if (p.steps[i].rec_ev == 0) p.steps[i].rec_ev = ev; else p.steps[i].rec_ev_final = ev;
                break;
            }
        Step j{};
        j.kind = 2;
        j.wait_ev = ev;
        p.steps.push_back(j);
    }
}

int sweep_macro_columns(int nb, const PlanOpts& opts) {
    return opts.macro > 0 ? opts.macro : (nb >= 80 ? 5 : (nb >= 56 ? 4 : (nb >= 25 ? 2 : (nb > 13 ? 3 : nb))));
}
int dist_collectives(int nblk, const PlanOpts& opts) {
    if (opt(opts.dist_fuse, 1) == 0) return 2 * nblk - 1;
    const int MB = sweep_macro_columns(nblk, opts);
    return nblk - 1 + (nblk + MB - 1) / MB;     // one all-gather per column but the last + one broadcast per macro panel
}

DistDecision dist_cholesky_pays(int nblk, int size, double coll_us, const PlanOpts& opts) {
    DistDecision d{};
    d.collectives = dist_collectives(nblk, opts);
    const double n = 128.0 * nblk;
    d.saving_ms = size > 1 ? (1.0 - 1.0 / size) * (n * n * n / 3.0) / 60e12 * 1e3 : 0.0;
    d.cost_ms = d.collectives * coll_us * 1e-3;
    d.dist = size > 1 && coll_us > 0.0 && d.saving_ms > 1.25 * d.cost_ms;
    return d;
}

void build_plan(Plan& p, int nblk, int64_t ld, int64_t stride, const PlanOpts& opts, int t128_div, const Shard& shard) {
    p = Plan{};
    p.opts = opts;
    p.shard = shard;
    p.shard.dist = shard.size > 1 && (shard.dist || (opts.dist_chol >= 0 ? opts.dist_chol != 0
                                                                         : dist_cholesky_pays(nblk, shard.size, shard.coll_us, opts).dist));
    if (shard.size > 1) p.opts.kind = 0;      // a sharded evaluation always runs the sweep
    p.t128_min = std::max(1, (opts.t128_min > 0 ? opts.t128_min : (nblk >= 56 ? 600 : 300)) / std::max(1, t128_div));
    p.batch_div = std::max(1, t128_div);
    p.nblk = nblk;
    p.ld = ld;
    p.stride = stride;
    if (opts.kind == 2) {
        plan_cholinv(p, 0, nblk);
    } else if (opts.kind == 1) {
        plan_potrf_rl(p);
        plan_trtri_levels(p);
    } else {
        plan_sweep(p);
    }
    plan_kinv(p);
    p.n_fixed_tasks = p.tasks.size();
    p.predv_rows = 0;
}

}  // namespace mfgp
