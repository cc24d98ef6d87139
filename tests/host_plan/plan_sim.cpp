// plan_sim.cpp -- CPU checker of the factorisation planner (csrc/plan.cpp, compiled into this library by g++).
// TEST INFRASTRUCTURE: never linked into libmfgp_hip.so.
//
// For a plan built exactly as the engine builds it (same code, same environment switches) it
//   (1) executes the steps in enqueue order with a plain-C restatement of the tile-GEMM task semantics (masks, beta,
//       mirrored store) and of the leaf (Cholesky + inverse of a 128-block), on matrices pre-filled with NaN wherever
//       the K build does not write -- so any read of data that no earlier step produced poisons the result --
//       and checks  L L^T = A,  X L = I,  S mirrored,  K^-1 = X^T X;
//   (2) checks the two-stream schedule for DATA RACES: every pair of conflicting accesses (write/write, write/read) to
//       the same 32x32 cell of the same matrix must be ordered by stream order or by an event recorded before it is
//       waited for (vector clocks over the streams); tasks of one launch run concurrently and must not conflict.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <limits>
#include <string>
#include <vector>
#include "../../multifidelity_datafusion_gps_amd/csrc/plan.h"

using namespace mfgp;

namespace {

constexpr int CELL = 32;   // granularity of the race check = the smallest tile edge the planner emits

constexpr int NS = 3;   // streams: 0 main (the serial chain), 1 bulk, 2 columns
struct Clock { int t[NS]; };

struct Cell {
    int w_step = -1, w_task = -1;          // last writer
    std::vector<std::pair<int, int>> readers;   // (step, task) since the last write
};

struct Sim {
    Plan p;
    int64_t ld, stride;
    std::vector<double> mem;
    std::vector<Cell> cells;
    std::vector<Clock> step_clock;
    std::vector<int> step_strm;
    int cpr;   // cells per row of one matrix
    int races = 0;
    std::string first_race;

    bool hb(int a, int b) const {   // step a happens-before step b
        const int sa = step_strm[a];
        return step_clock[b].t[sa] >= step_clock[a].t[sa];
    }
    Cell& cell(int64_t off) {
        const int buf = (int)(off / stride);
        const int64_t r = (off % stride) / ld, c = off % ld;
        return cells[(size_t)buf * cpr * cpr + (size_t)(r / CELL) * cpr + (size_t)(c / CELL)];
    }
    void report(const char* kind, int64_t off, int s0, int t0, int s1, int t1) {
        ++races;
        if (first_race.empty()) {
            char tmp[256];
            const int buf = (int)(off / stride);
            snprintf(tmp, sizeof tmp, "%s on matrix %d cell (%lld,%lld): step %d task %d vs step %d task %d", kind, buf,
                     (long long)((off % stride) / ld / CELL), (long long)(off % ld / CELL), s0, t0, s1, t1);
            first_race = tmp;
        }
    }
    void read(int64_t off, int step, int task) {
        Cell& c = cell(off);
        if (c.w_step >= 0 && !(c.w_step == step && c.w_task == task)) {
            if (c.w_step == step || !hb(c.w_step, step)) report("read-after-write race", off, c.w_step, c.w_task, step, task);
        }
        if (c.readers.empty() || c.readers.back() != std::make_pair(step, task)) c.readers.push_back({step, task});
    }
    void write(int64_t off, int step, int task) {
        Cell& c = cell(off);
        if (c.w_step >= 0 && !(c.w_step == step && c.w_task == task)) {
            if (c.w_step == step || !hb(c.w_step, step)) report("write-after-write race", off, c.w_step, c.w_task, step, task);
        }
        for (auto& r : c.readers) {
            if (r.first == step && r.second == task) continue;
            if (r.first == step || !hb(r.first, step)) report("write-after-read race", off, r.first, r.second, step, task);
        }
        c.readers.clear();
        c.w_step = step;
        c.w_task = task;
    }
};

// operand window of a task: rows [0,T), k in [0,klen); cell (rb, kb) in CELL-units; is it entirely masked (read as zero)?
bool operand_cell_masked(int T, int klen, bool lower, bool upper, int rb, int kb) {
    const int r0 = rb * CELL, r1 = r0 + CELL - 1, k0 = kb * CELL, k1 = k0 + CELL - 1;
    if (lower && k0 > r1 + (klen - T)) return true;   // zero where k > r + klen - T
    if (upper && k1 < r0) return true;                // zero where k < r
    return false;
}

void task_accesses(Sim& s, const GemmTask& t, int T, int64_t base_a, int64_t base_b, int64_t base_c, int64_t base_c2, int step,
                   int task) {
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER, b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    for (int rb = 0; rb < T / CELL; ++rb)
        for (int kb = 0; kb < (t.klen + CELL - 1) / CELL; ++kb) {
            if (!operand_cell_masked(T, t.klen, a_lo, a_up, rb, kb)) s.read(base_a + t.a_off + (int64_t)rb * CELL * s.ld + kb * CELL, step, task);
            if (!operand_cell_masked(T, t.klen, b_lo, b_up, rb, kb)) s.read(base_b + t.b_off + (int64_t)rb * CELL * s.ld + kb * CELL, step, task);
        }
    for (int rb = 0; rb < T / CELL; ++rb)
        for (int cb = 0; cb < T / CELL; ++cb) {
            const int64_t off = base_c + t.c_off + (int64_t)rb * CELL * s.ld + cb * CELL;
            if (t.beta != 0.0) s.read(off, step, task);
            s.write(off, step, task);
            if (t.c2_off >= 0) s.write(base_c2 + t.c2_off + (int64_t)cb * CELL * s.ld + rb * CELL, step, task);
        }
}

void task_compute(Sim& s, const GemmTask& t, int T, int64_t base_a, int64_t base_b, int64_t base_c, int64_t base_c2) {
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER, b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const double* A = s.mem.data() + base_a + t.a_off;
    const double* B = s.mem.data() + base_b + t.b_off;
    double* C = s.mem.data() + base_c + t.c_off;
    double* C2 = t.c2_off >= 0 ? s.mem.data() + base_c2 + t.c2_off : nullptr;
    const int64_t ld = s.ld;
    std::vector<double> out((size_t)T * T);
    std::vector<double> arow(t.klen), brow((size_t)T * t.klen);
    for (int c = 0; c < T; ++c)
        for (int k = 0; k < t.klen; ++k) {
            double v = B[(int64_t)c * ld + k];
            if (b_lo && k > c + t.klen - T) v = 0.0;
            if (b_up && k < c) v = 0.0;
            brow[(size_t)c * t.klen + k] = v;
        }
    for (int r = 0; r < T; ++r) {
        for (int k = 0; k < t.klen; ++k) {
            double v = A[(int64_t)r * ld + k];
            if (a_lo && k > r + t.klen - T) v = 0.0;
            if (a_up && k < r) v = 0.0;
            arow[k] = v;
        }
        for (int c = 0; c < T; ++c) {
            const double* bp = &brow[(size_t)c * t.klen];
            double acc = 0.0;
            for (int k = 0; k < t.klen; ++k) acc += arow[k] * bp[k];
            out[(size_t)r * T + c] = acc;
        }
    }
    for (int r = 0; r < T; ++r)
        for (int c = 0; c < T; ++c) {
            double v = t.alpha * out[(size_t)r * T + c];
            double* pc = C + (int64_t)r * ld + c;
            if (t.beta != 0.0) v += t.beta * (*pc);
            *pc = v;
            if (C2) C2[(int64_t)c * ld + r] = v;
        }
}

// Cholesky + inverse of the 128-block `blk` of A (lower part read), L block (zeros above) and mirrored inverse out
int leaf_compute(Sim& s, int blk) {
    const int64_t ld = s.ld, g0 = (int64_t)blk * NB * ld + (int64_t)blk * NB;
    const double* A = s.mem.data() + (int64_t)BUF_A * s.stride + g0;
    double* Lo = s.mem.data() + (int64_t)BUF_L * s.stride + g0;
    double* So = s.mem.data() + (int64_t)BUF_S * s.stride + g0;
    std::vector<double> L((size_t)NB * NB, 0.0), X((size_t)NB * NB, 0.0);
    for (int j = 0; j < NB; ++j) {
        double d = A[(int64_t)j * ld + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * NB + k] * L[(size_t)j * NB + k];
        if (!(d > 0.0)) return blk * NB + j + 1;
        const double ljj = sqrt(d);
        L[(size_t)j * NB + j] = ljj;
        for (int i = j + 1; i < NB; ++i) {
            double v = A[(int64_t)i * ld + j];
            for (int k = 0; k < j; ++k) v -= L[(size_t)i * NB + k] * L[(size_t)j * NB + k];
            L[(size_t)i * NB + j] = v / ljj;
        }
    }
    for (int j = 0; j < NB; ++j) {   // X = L^-1 column by column
        X[(size_t)j * NB + j] = 1.0 / L[(size_t)j * NB + j];
        for (int i = j + 1; i < NB; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v += L[(size_t)i * NB + k] * X[(size_t)k * NB + j];
            X[(size_t)i * NB + j] = -v / L[(size_t)i * NB + i];
        }
    }
    for (int r = 0; r < NB; ++r)
        for (int c = 0; c < NB; ++c) {
            Lo[(int64_t)r * ld + c] = L[(size_t)r * NB + c];
            const int hi = std::max(r, c), lo = std::min(r, c);
            So[(int64_t)r * ld + c] = X[(size_t)hi * NB + lo];
        }
    return 0;
}


// matrices pre-filled with NaN; the SPD test matrix where the K build writes (64-tiles of the lower triangle); race-check cells
void prepare(Sim& s, bool numeric, std::vector<double>& A0, double* report) {
    const Plan& p = s.p;
    s.cpr = (int)(s.ld / CELL);
    s.cells.assign((size_t)4 * s.cpr * s.cpr, Cell());
    if (report) {
        for (int i = 0; i < 8; ++i) report[i] = 0.0;
        report[5] = (double)p.steps.size();
        report[6] = (double)p.n_fixed_tasks;
        report[7] = (double)p.n_events;
    }
    const int64_t N = s.ld;
    if (numeric) {
        s.mem.assign((size_t)4 * s.stride, std::numeric_limits<double>::quiet_NaN());
        A0.assign((size_t)N * N, 0.0);
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j <= i; ++j) {
                const double d = (double)(i - j) / 40.0;
                double v = exp(-0.5 * d * d) + 0.3 * exp(-d) * cos(0.05 * (double)(i - j));   // sum of two stationary PSD kernels
                if (i == j) v += 0.5 + 0.001 * (double)(i % 17);
                A0[(size_t)i * N + j] = A0[(size_t)j * N + i] = v;
            }
        double* A = s.mem.data() + (int64_t)BUF_A * s.stride;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j < N; ++j)
                if (j / 64 <= i / 64) A[i * s.ld + j] = A0[(size_t)i * N + j];
    }
}

// the stand-alone K^-1 launch of a plan (numeric)
void run_kinv(Sim& s) {
    const Plan& p = s.p;
    const Step& st = p.kinv_step;
    for (int k = 0; k < st.count; ++k)
        task_compute(s, p.tasks[st.first + k], st.tile, (int64_t)st.a * s.stride, (int64_t)st.b * s.stride,
                     (int64_t)st.c * s.stride, 0);
}

// The exchange steps of a distributed Cholesky (plan.h Shard::dist): which cells of L / S a rank reads (the blocks it owns and
// sends) and writes (the blocks it receives).  blk = the block column c.
void comm_accesses(Sim& s, const Step& st, int si) {
    const Plan& p = s.p;
    const int c = st.blk, size = p.shard.size, rank = p.shard.rank;
    auto touch = [&](int buf, int bi, int bj, bool wr) {
        const int64_t g0 = (int64_t)buf * s.stride + (int64_t)bi * NB * s.ld + (int64_t)bj * NB;
        for (int rb = 0; rb < NB / CELL; ++rb)
            for (int cb = 0; cb < NB / CELL; ++cb) {
                const int64_t off = g0 + (int64_t)rb * CELL * s.ld + cb * CELL;
                if (wr) s.write(off, si, 0); else s.read(off, si, 0);
            }
    };
    if (st.kind == STEP_COMM_DIAG) {
        const bool own = shard_owner(c, size) == rank;
        touch(BUF_L, c, c, !own);
        touch(BUF_S, c, c, !own);
    } else {
        for (int i = c + 1; i < p.nblk; ++i) touch(BUF_L, i, c, shard_owner(i, size) != rank);
        if (st.carry) {   // one exchange per column: diagonal block c + 1 travels along (read by its owner, written by the others)
            const bool own = shard_owner(c + 1, size) == rank;
            touch(BUF_L, c + 1, c + 1, !own);
            touch(BUF_S, c + 1, c + 1, !own);
        }
    }
}

// walk the steps in enqueue order: race check (+ execution when numeric).  Resumable: run() stops in front of the next exchange step
// of a distributed plan (its index is returned; the caller moves the data between the ranks' matrices and calls run() again) and
// returns the number of steps at the end; < 0 with msg set on an error.
struct Walker {
    Clock vc[NS] = {};
    std::vector<Clock> evclock;
    size_t si = 0;
    bool started = false;
    int run(Sim& s, int numeric, int want_grad, char* msg, int msglen) {
        const Plan& p = s.p;
        if (!started) {
            evclock.assign(p.n_events + 1, Clock{{-1, -1, -1}});
            s.step_clock.resize(p.steps.size());
            s.step_strm.resize(p.steps.size());
            started = true;
        }
        // the K build (one launch on the main stream, before every step) wrote the lower 64-tiles of A: model it as step -1
        // by leaving the cells without a writer -- every plan step is ordered after it by stream order / the chain events
        for (; si < p.steps.size(); ++si) {
            const Step& st = p.steps[si];
            const int strm = (st.strm >= 0 && st.strm < NS) ? st.strm : 0;
            const bool comm = st.kind == STEP_COMM_DIAG || st.kind == STEP_COMM_PANEL;
            if (comm && !resumed_at(si)) {
                pending = si;
                return (int)si;
            }
            if (st.wait_ev > 0) {
                if (evclock[st.wait_ev].t[0] < 0) {
                    snprintf(msg, msglen, "step %zu waits for event %d before it is recorded", si, st.wait_ev);
                    return -1;
                }
                for (int k = 0; k < NS; ++k) vc[strm].t[k] = std::max(vc[strm].t[k], evclock[st.wait_ev].t[k]);
            }
            vc[strm].t[strm] += 1;
            s.step_clock[si] = vc[strm];
            s.step_strm[si] = strm;
            if (comm) {
                if (strm != 0) {
                    snprintf(msg, msglen, "exchange step %zu is not on the main stream", si);
                    return -4;
                }
                comm_accesses(s, st, (int)si);
            } else if (st.kind == STEP_LEAF) {
                const int64_t g0 = (int64_t)st.blk * NB * s.ld + (int64_t)st.blk * NB;
                for (int rb = 0; rb < NB / CELL; ++rb)
                    for (int cb = 0; cb < NB / CELL; ++cb) {
                        // the leaf reads the 16-blocks on and below the diagonal: every CELL that holds one
                        if (cb <= rb) s.read((int64_t)BUF_A * s.stride + g0 + (int64_t)rb * CELL * s.ld + cb * CELL, (int)si, 0);
                        s.write((int64_t)BUF_L * s.stride + g0 + (int64_t)rb * CELL * s.ld + cb * CELL, (int)si, 0);
                        s.write((int64_t)BUF_S * s.stride + g0 + (int64_t)rb * CELL * s.ld + cb * CELL, (int)si, 0);
                    }
                if (numeric) {
                    const int info = leaf_compute(s, st.blk);
                    if (info) {
                        snprintf(msg, msglen, "leaf %d: non-positive pivot %d (or NaN input)", st.blk, info);
                        return -2;
                    }
                }
            } else if (st.kind == STEP_GEMM) {
                const bool g = want_grad && st.gcount > 0;
                const int n = g ? st.gcount : st.count, first = g ? st.gfirst : st.first;
                const int64_t ba = (int64_t)st.a * s.stride, bb = (int64_t)st.b * s.stride, bc = (int64_t)st.c * s.stride;
                const int64_t bc2 = st.c2 >= 0 ? (int64_t)st.c2 * s.stride : 0;
                for (int k = 0; k < n; ++k) task_accesses(s, p.tasks[first + k], st.tile, ba, bb, bc, bc2, (int)si, k);
                if (numeric)
                    for (int k = 0; k < n; ++k) task_compute(s, p.tasks[first + k], st.tile, ba, bb, bc, bc2);
            }
            if (st.rec_ev > 0) evclock[st.rec_ev] = vc[strm];
            if (st.rec_ev_final > 0) evclock[st.rec_ev_final] = vc[strm];
        }
        // everything must be visible to the main stream at the end (solve / gradient kernels follow there)
        for (size_t k = 0; k < p.steps.size(); ++k)
            if (s.step_strm[k] != 0 && vc[0].t[s.step_strm[k]] < s.step_clock[k].t[s.step_strm[k]]) {
                snprintf(msg, msglen, "step %zu of stream %d is not joined into the main stream at the end of the plan", k, s.step_strm[k]);
                return -3;
            }
        return (int)p.steps.size();
    }
    long long pending = -1;     // the exchange step run() stopped in front of; resume() lets the next run() pass it
    long long passed = -1;
    bool resumed_at(size_t k) const { return passed == (long long)k; }
    void resume() { passed = pending; }
};

// with_kinv: also the stand-alone K^-1 launch of a plan that does not stream it.  -> 0, or < 0 with msg set
int walk_steps(Sim& s, int numeric, int want_grad, bool with_kinv, char* msg, int msglen) {
    Walker w;
    const int rc = w.run(s, numeric, want_grad, msg, msglen);
    if (rc < 0) return rc;
    if (rc != (int)s.p.steps.size()) {
        snprintf(msg, msglen, "an exchange step (%d) in a plan that is executed alone", rc);
        return -5;
    }
    if (with_kinv && want_grad && !s.p.kinv_streamed && numeric) run_kinv(s);
    return 0;
}

}  // namespace

extern "C" {

// report[0] = |L L^T - A|_F / |A|_F, [1] = max |X L - I|, [2] = max |S - S^T| over the lower/upper pairs that hold X,
// [3] = max |K^-1 - X^T X| / max |X^T X| (want_grad only), [4] = number of races, [5] = steps, [6] = tasks, [7] = events
// numeric = 0: race check only (any size); 1: also execute.  slack = extra rows of capacity (stride = (ld + slack)^2).
// mutate (self-test of the checker): 1 = the first bulk launch forgets to wait for the chain; 2 = the main stream forgets
// the final join; 3 = the second macro's first leaf forgets its event wait
int plan_sim(int nblk, int numeric, int want_grad, int slack, int mutate, double* report, char* msg, int msglen) {
    Sim s;
    s.ld = (int64_t)nblk * NB;
    s.stride = (s.ld + slack) * (s.ld + slack);
    const char* bdiv = getenv("PLAN_SIM_BATCH_DIV");     // the plan of a BATCHED pass of about that many sets (tile sizes by set class)
    build_plan(s.p, nblk, s.ld, s.stride, plan_opts_from_env(), bdiv ? atoi(bdiv) : 1);
    if (mutate == 1) {
        for (Step& st : s.p.steps)
            if (st.strm == 1 && st.wait_ev > 0) { st.wait_ev = 0; break; }
    } else if (mutate == 2) {
        for (Step& st : s.p.steps)
            if (st.kind == 2) st.wait_ev = 0;
    } else if (mutate == 3) {
        int seen = 0;
        for (Step& st : s.p.steps)
            if (st.strm == 0 && st.wait_ev > 0 && st.kind != 2 && ++seen == 1) { st.wait_ev = 0; break; }
    }
    std::vector<double> A0;
    prepare(s, numeric != 0, A0, report);
    if (msg && msglen) msg[0] = 0;
    {
        const int rc = walk_steps(s, numeric, want_grad, true, msg, msglen);
        if (rc < 0) return rc;
    }
    const int64_t N = s.ld;
    report[4] = (double)s.races;
    if (s.races && msg) snprintf(msg, msglen, "%d races; first: %s", s.races, s.first_race.c_str());
    if (!numeric) return s.races ? 1 : 0;
    // ---- numeric checks ----
    const double* L = s.mem.data() + (int64_t)BUF_L * s.stride;
    const double* S = s.mem.data() + (int64_t)BUF_S * s.stride;
    const double* Ki = s.mem.data() + (int64_t)BUF_A * s.stride;
    double num = 0.0, den = 0.0;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double v = 0.0;
            for (int64_t k = 0; k <= j; ++k) v += L[i * s.ld + k] * L[j * s.ld + k];
            const double d = v - A0[(size_t)i * N + j];
            num += d * d;
            den += A0[(size_t)i * N + j] * A0[(size_t)i * N + j];
        }
    report[0] = sqrt(num / den);
    double e1 = 0.0, e2 = 0.0;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double v = 0.0;
            for (int64_t k = j; k <= i; ++k) v += S[i * s.ld + k] * L[k * s.ld + j];   // X = lower part of S
            e1 = std::max(e1, fabs(v - (i == j ? 1.0 : 0.0)));
            e2 = std::max(e2, fabs(S[i * s.ld + j] - S[j * s.ld + i]));
        }
    report[1] = std::isnan(e1) ? 1e300 : e1;
    report[2] = std::isnan(e2) ? 1e300 : e2;
    for (int64_t i = 0; i < N && !std::isnan(report[1]); ++i)
        for (int64_t j = 0; j <= i; ++j)
            if (std::isnan(S[i * s.ld + j]) || std::isnan(L[i * s.ld + j])) report[1] = 1e300;
    if (want_grad) {
        double e3 = 0.0, mx = 0.0;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = 0; j <= i; ++j) {
                double v = 0.0;
                for (int64_t k = i; k < N; ++k) v += S[k * s.ld + i] * S[k * s.ld + j];   // sum_k X[k][i] X[k][j]
                const double got = Ki[i * s.ld + j];
                e3 = std::isnan(got) ? 1e300 : std::max(e3, fabs(v - got));
                mx = std::max(mx, fabs(v));
                if (i / 64 == j / 64) {   // the gradient reduction reads the diagonal 64x64 tiles of K^-1 in full
                    const double up = Ki[j * s.ld + i];
                    e3 = std::isnan(up) ? 1e300 : std::max(e3, fabs(v - up));
                }
            }
        report[3] = e3 / mx;
    }
    return s.races ? 1 : 0;
}

// A SHARDED evaluation (mfgp_eval_sharded) of `size` ranks, every rank's plan executed on its own copy of the matrices:
//   each rank: its plan (the Cholesky in full, B / X^T work for its own block rows) -- race-checked like any plan;
//   exchange: every 128-row block of S from its owner to all (full rows), then lower part of S <- transpose of the upper part;
//   each rank: the stand-alone K^-1 launch restricted to its block rows.
// Checked against the UNSHARDED plan executed the same way: L, S (after the exchange) and every rank's own rows of K^-1 must be
// BITWISE what one rank computes alone (same tasks, same arithmetic), on top of X L = I.
// report[0] = races (all ranks), [1] = max |X L - I| (rank 0), [2] = number of words of L / S that differ from the single run
// (all ranks), [3] = words of K^-1 (own rows, lower 128-tiles) that differ, [4] = tasks of the largest rank plan / tasks of the
// unsharded plan
static int plan_sim_group(int nblk, int size, bool dist, double* report, char* msg, int msglen) {
    const PlanOpts opts = plan_opts_from_env();
    const int64_t ld = (int64_t)nblk * NB, stride = ld * ld;
    for (int i = 0; i < 8; ++i) report[i] = 0.0;
    if (msg && msglen) msg[0] = 0;
    Sim ref;
    ref.ld = ld; ref.stride = stride;
    {
        PlanOpts o = opts;
        o.kinv_stream = 0;                       // reference: K^-1 as the stand-alone launch (the streamed form is bitwise equal
        build_plan(ref.p, nblk, ld, stride, o);  // on the GPU, not in this simulator, which adds a chunk's sum at its end)
    }
    std::vector<double> A0;
    prepare(ref, true, A0, nullptr);
    int rc = walk_steps(ref, 1, 1, true, msg, msglen);
    if (rc < 0) return rc;
    std::vector<Sim> rk((size_t)size);
    std::vector<Walker> wk((size_t)size);
    size_t max_tasks = 0;
    double comm_words = 0;
    int n_exchanges = 0, n_exchanges_carry = 0;
    for (int r = 0; r < size; ++r) {
        Sim& s = rk[(size_t)r];
        s.ld = ld; s.stride = stride;
        Shard sh;
        sh.rank = r; sh.size = size;
        PlanOpts o = opts;
        o.dist_chol = dist ? 1 : 0;            // (the size rule would turn it on from 256 block columns only)
        build_plan(s.p, nblk, ld, stride, o, 1, sh);
        std::vector<double> dummy;
        prepare(s, true, dummy, nullptr);
    }
    // every rank runs up to its next exchange step; the exchange steps come in the same order on every rank (the planner emits them
    // for all); the data moves between the ranks' matrices; on to the next
    for (;;) {
        std::vector<int> at((size_t)size);
        for (int r = 0; r < size; ++r) {
            at[(size_t)r] = wk[(size_t)r].run(rk[(size_t)r], 1, 1, msg, msglen);
            if (at[(size_t)r] < 0) return at[(size_t)r];
        }
        const bool done0 = at[0] == (int)rk[0].p.steps.size();
        for (int r = 0; r < size; ++r) {
            const bool done = at[(size_t)r] == (int)rk[(size_t)r].p.steps.size();
            if (done != done0) {
                snprintf(msg, msglen, "rank %d and rank 0 disagree about the number of exchange steps", r);
                return -6;
            }
            if (!done) {
                const Step& a = rk[(size_t)r].p.steps[(size_t)at[(size_t)r]], &b = rk[0].p.steps[(size_t)at[0]];
                if (a.kind != b.kind || a.blk != b.blk || a.carry != b.carry) {
                    snprintf(msg, msglen, "rank %d meets exchange (%d, column %d) where rank 0 meets (%d, column %d)", r, a.kind, a.blk, b.kind, b.blk);
                    return -6;
                }
            }
        }
        if (done0) break;
        const Step& st = rk[0].p.steps[(size_t)at[0]];
        const int c = st.blk;
        auto copy_block = [&](int buf, int bi, int bj, int own) {
            const int64_t g0 = (int64_t)buf * stride + (int64_t)bi * NB * ld + (int64_t)bj * NB;
            for (int r = 0; r < size; ++r) {
                if (r == own) continue;
                for (int row = 0; row < NB; ++row)
                    memcpy(rk[(size_t)r].mem.data() + g0 + (int64_t)row * ld, rk[(size_t)own].mem.data() + g0 + (int64_t)row * ld, NB * sizeof(double));
            }
            comm_words += (double)NB * NB;
        };
        if (st.kind == STEP_COMM_DIAG) {
            copy_block(BUF_L, c, c, shard_owner(c, size));
            copy_block(BUF_S, c, c, shard_owner(c, size));
        } else {
            for (int i = c + 1; i < nblk; ++i) copy_block(BUF_L, i, c, shard_owner(i, size));
            if (st.carry) {
                copy_block(BUF_L, c + 1, c + 1, shard_owner(c + 1, size));
                copy_block(BUF_S, c + 1, c + 1, shard_owner(c + 1, size));
            }
            n_exchanges_carry += st.carry ? 1 : 0;
        }
        n_exchanges += 1;
        for (int r = 0; r < size; ++r) wk[(size_t)r].resume();
    }
    report[5] = comm_words * 8.0;     // bytes every rank receives or sends through the Cholesky's exchange steps (per copy of a block)
    report[6] = n_exchanges;          // exchange steps on the chain / of those, all-gathers that carried the next diagonal block along
    report[7] = n_exchanges_carry;
    for (int r = 0; r < size; ++r) {
        Sim& s = rk[(size_t)r];
        report[0] += s.races;
        if (s.races && msg && !msg[0]) snprintf(msg, msglen, "rank %d: %d races; first: %s", r, s.races, s.first_race.c_str());
        size_t nt = 0;
        for (const Step& st : s.p.steps) if (st.kind == 1) nt += (size_t)st.count;
        max_tasks = std::max(max_tasks, nt + (size_t)s.p.kinv_step.count);
    }
    {
        size_t nt = 0;
        for (const Step& st : ref.p.steps) if (st.kind == 1) nt += (size_t)st.count;
        report[4] = (double)max_tasks / (double)(nt + (size_t)ref.p.kinv_step.count);
    }
    // exchange: row block b of S from its owner to everybody, then the lower part from the upper one
    const int64_t so = (int64_t)BUF_S * stride;
    for (int b = 0; b < nblk; ++b) {
        const int own = shard_owner(b, size);
        const double* src = rk[(size_t)own].mem.data() + so + (int64_t)b * NB * ld;
        for (int r = 0; r < size; ++r)
            if (r != own) memcpy(rk[(size_t)r].mem.data() + so + (int64_t)b * NB * ld, src, (size_t)NB * ld * sizeof(double));
    }
    for (int r = 0; r < size; ++r) {
        double* S = rk[(size_t)r].mem.data() + so;
        for (int64_t i = 0; i < ld; ++i)
            for (int64_t k = i + 1; k < ld; ++k) S[k * ld + i] = S[i * ld + k];
        run_kinv(rk[(size_t)r]);
    }
    // compare with the single-rank run
    double diffLS = 0, diffK = 0;
    const double* Lr = ref.mem.data() + (int64_t)BUF_L * stride;
    const double* Sr = ref.mem.data() + so;
    const double* Kr = ref.mem.data() + (int64_t)BUF_A * stride;
    for (int r = 0; r < size; ++r) {
        const double* L = rk[(size_t)r].mem.data() + (int64_t)BUF_L * stride;
        const double* S = rk[(size_t)r].mem.data() + so;
        const double* K = rk[(size_t)r].mem.data() + (int64_t)BUF_A * stride;
        for (int64_t i = 0; i < ld; ++i)
            for (int64_t j = 0; j < ld; ++j) {
                if (j <= i && memcmp(&L[i * ld + j], &Lr[i * ld + j], 8) != 0) diffLS += 1;
                if (memcmp(&S[i * ld + j], &Sr[i * ld + j], 8) != 0) diffLS += 1;
                if (j / NB <= i / NB && shard_owner((int)(i / NB), size) == r && memcmp(&K[i * ld + j], &Kr[i * ld + j], 8) != 0 &&
                    (j <= i || i / 64 == j / 64))
                    diffK += 1;
            }
    }
    report[2] = diffLS;
    report[3] = diffK;
    {
        const double* L = rk[0].mem.data() + (int64_t)BUF_L * stride;
        const double* S = rk[0].mem.data() + so;
        double e1 = 0.0;
        for (int64_t i = 0; i < ld; ++i)
            for (int64_t j = 0; j <= i; ++j) {
                double v = 0.0;
                for (int64_t k = j; k <= i; ++k) v += S[i * ld + k] * L[k * ld + j];
                e1 = std::max(e1, fabs(v - (i == j ? 1.0 : 0.0)));
            }
        report[1] = std::isnan(e1) ? 1e300 : e1;
    }
    return report[0] > 0 ? 1 : 0;
}

// SCHEDULE ONLY (no arithmetic: any size) of a group's plans -- at the sizes the plans are made for, N = 8192 ... 32768 on 8 ranks: every
// rank's plan race-free, every wait behind its record, the exchange steps met in the same order by every rank.
// report[0] = races (all ranks), [4] = largest rank's tasks / tasks of the unsharded plan, [5] = bytes through the Cholesky's exchange
// steps, [6] = exchange steps per rank
int plan_sim_group_schedule(int nblk, int size, int dist, double* report, char* msg, int msglen) {
    const PlanOpts opts = plan_opts_from_env();
    const int64_t ld = (int64_t)nblk * NB, stride = ld * ld;
    for (int i = 0; i < 8; ++i) report[i] = 0.0;
    if (msg && msglen) msg[0] = 0;
    size_t ref_tasks = 0;
    {
        Plan ref;
        PlanOpts o = opts;
        o.kinv_stream = 0;
        build_plan(ref, nblk, ld, stride, o);
        for (const Step& st : ref.steps) if (st.kind == 1) ref_tasks += (size_t)st.count;
        ref_tasks += (size_t)ref.kinv_step.count;
    }
    std::vector<std::vector<std::pair<int, int>>> order((size_t)size);
    size_t max_tasks = 0;
    for (int r = 0; r < size; ++r) {
        Sim s;                                   // one rank at a time: the race cells of 256 block columns are 1 M per matrix
        s.ld = ld; s.stride = stride;
        Shard sh;
        sh.rank = r; sh.size = size;
        PlanOpts o = opts;
        o.dist_chol = dist ? 1 : 0;
        build_plan(s.p, nblk, ld, stride, o, 1, sh);
        std::vector<double> dummy;
        prepare(s, false, dummy, nullptr);
        Walker w;
        for (;;) {
            const int at = w.run(s, 0, 1, msg, msglen);
            if (at < 0) return at;
            if (at == (int)s.p.steps.size()) break;
            const Step& st = s.p.steps[(size_t)at];
            order[(size_t)r].push_back({st.kind * 2 + st.carry, st.blk});
            if (r == 0) report[5] += 8.0 * NB * NB * (st.kind == STEP_COMM_DIAG ? 2 : nblk - 1 - st.blk + 2 * st.carry);
            w.resume();
        }
        report[0] += s.races;
        if (s.races && msg && !msg[0]) snprintf(msg, msglen, "rank %d: %d races; first: %s", r, s.races, s.first_race.c_str());
        size_t nt = 0;
        for (const Step& st : s.p.steps) if (st.kind == 1) nt += (size_t)st.count;
        max_tasks = std::max(max_tasks, nt + (size_t)s.p.kinv_step.count);
        if (order[(size_t)r] != order[0]) {
            snprintf(msg, msglen, "rank %d meets its exchange steps in another order than rank 0 (%zu against %zu steps)", r,
                     order[(size_t)r].size(), order[0].size());
            return -6;
        }
    }
    report[4] = (double)max_tasks / (double)ref_tasks;
    report[6] = (double)order[0].size();
    return report[0] > 0 ? 1 : 0;
}

int plan_sim_sharded(int nblk, int size, double* report, char* msg, int msglen) { return plan_sim_group(nblk, size, false, report, msg, msglen); }
// the same with the DISTRIBUTED Cholesky (plan.h Shard::dist): every rank runs only its own rows of the panels and of the trailing
// updates, the diagonal blocks and the panel columns travel through the plan's exchange steps (executed here in lock step over the
// ranks' copies of the matrices).  report[5] = bytes moved by those steps.
int plan_sim_dist(int nblk, int size, double* report, char* msg, int msglen) { return plan_sim_group(nblk, size, true, report, msg, msglen); }

// out[0..7] = main-stream launches, bulk launches, waits on the main stream, records on the main stream, waits on bulk,
// records on bulk, tasks (without the stand-alone K^-1 launch), gradient-only tasks
void plan_stats(int nblk, double* out) {
    Plan p;
    const int64_t ld = (int64_t)nblk * NB;
    build_plan(p, nblk, ld, ld * ld);
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (const Step& st : p.steps) {
        const int b = st.strm == 1;
        if (st.kind != 2) out[b] += 1;
        if (st.wait_ev > 0) out[b ? 4 : 2] += 1;
        out[b ? 5 : 3] += (st.rec_ev > 0) + (st.rec_ev_final > 0);
        if (st.kind == 1) { out[6] += st.gcount > 0 ? st.gcount : st.count; out[7] += st.gcount > 0 ? st.gcount - st.count : 0; }
    }
}

// per-step table for tools/plan_flops.py: out[7*i + ..] = {stream, kind, tile, tasks (gradient variant), executed Gflop
// (2 T^2 klen per task), longest klen, shortest klen}; returns the number of steps (at most max_steps are written)
int plan_steps(int nblk, int want_grad, double* out, int max_steps) {
    Plan p;
    const int64_t ld = (int64_t)nblk * NB;
    build_plan(p, nblk, ld, ld * ld);
    int n = 0;
    for (const Step& st : p.steps) {
        if (n >= max_steps) break;
        double* o = out + 7 * n++;
        o[0] = st.strm; o[1] = st.kind; o[2] = st.tile; o[3] = o[4] = o[5] = o[6] = 0;
        if (st.kind != 1) continue;
        const bool g = want_grad && st.gcount > 0;
        const int cnt = g ? st.gcount : st.count, first = g ? st.gfirst : st.first;
        int kmax = 0, kmin = 1 << 30;
        double fl = 0;
        for (int k = 0; k < cnt; ++k) {
            const GemmTask& t = p.tasks[first + k];
            fl += 2.0 * st.tile * st.tile * t.klen;
            kmax = std::max(kmax, (int)t.klen); kmin = std::min(kmin, (int)t.klen);
        }
        o[3] = cnt; o[4] = fl * 1e-9; o[5] = kmax; o[6] = cnt ? kmin : 0;
    }
    return n;
}

}  // extern "C"
