// api_sharded.hip -- one evaluation shared by the ranks of the handle's communicator: mfgp_eval_sharded, the leader /
// follower form, the per-rank measurement hook, and the row-block K build of north_star (mfgp_kbuild_rows / _owned_rows /
// mfgp_dev_matrix / mfgp_eval_prebuilt).  The collectives themselves are in comm_rccl.hip.  Split out of mfgp_api.hip in round 6.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "mfgp_internal.h"
#include "api_shared.h"

using namespace mfgp;

extern "C" {

// ---- sharded evaluation ------------------------------------------------------------------------------------------
// One evaluation across the `size` ranks of the handle's communicator (one process per GPU).
// Every rank runs the Cholesky in full -- its serial chain does not shard -- but only ITS share of the other two thirds of
// the flops: the rows of X^T (the image of the identity) and, after ONE exchange of those rows, the rows of K^-1 and the
// gradient's tile sums (plan.h Shard: 128-row blocks, serpentine block-cyclic).  No result bit differs from mfgp_eval:
// every tile is computed by exactly the tasks the single evaluation runs, the tile sums of the gradient meet in ONE array
// (sum over ranks of arrays that are zero where a rank holds nothing) and are finished in the same fixed order.
static int ensure_shard_plan(mfgp_handle* h, int rank, int size) {
    // (the measured collective cost is an input of the plan -- it decides whether the Cholesky is distributed too -- and is the
    // communicator's: a plan made before the calibration, or for another group, is planned again)
    const double coll_us = (h->comm && h->comm_size == size) ? h->coll_us : 0.0;
    if (h->pls.nblk == h->nblk && h->pls.ld == h->Np && h->pls.shard.rank == rank && h->pls.shard.size == size &&
        h->pls.stride == (int64_t)h->cap * h->cap && h->pls.shard.coll_us == coll_us)
        return 0;
    Shard sh;
    sh.rank = rank; sh.size = size; sh.coll_us = coll_us;
    build_plan(h->pls, h->nblk, h->Np, (int64_t)h->cap * h->cap, h->pl.opts, 1, sh);
    while ((int)h->evpool.size() < h->pls.n_events) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        h->evpool.push_back(e);
    }
    const size_t need = h->pls.tasks.size();
    if (need > h->tasks_s_cap) {
        if (h->dtasks_s) HIPCHK(h, hipFree(h->dtasks_s));
        h->tasks_s_cap = need + need / 2 + 1024;
        HIPCHK(h, hipMalloc(&h->dtasks_s, h->tasks_s_cap * sizeof(GemmTask)));
    }
    HIPCHK(h, hipMemcpyAsync(h->dtasks_s, h->pls.tasks.data(), need * sizeof(GemmTask), hipMemcpyHostToDevice, h->stream));
    // the exchange's layout: block b (128 x (Np - 128 b) doubles of the upper part of S) at offset off[b] of its owner's chunk
    std::vector<long long> off((size_t)h->nblk), fill((size_t)size, 0);
    for (int b = 0; b < h->nblk; ++b) {
        const int own = shard_owner(b, size);
        off[(size_t)b] = fill[(size_t)own];
        fill[(size_t)own] += 128LL * (h->Np - 128LL * b);
    }
    h->shard_chunk = *std::max_element(fill.begin(), fill.end());
    if (h->pls.shard.dist) {   // one panel column, padded to the largest rank's share, + the diagonal message
        const size_t need = std::max(((size_t)(h->nblk / size + 2) * NB * NB + 2 * (size_t)NB * NB + 2) * size, 2 * (size_t)NB * NB + 2);
        if (need > h->dist_cap) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (h->ddist) HIPCHK(h, hipFree(h->ddist));
            h->ddist = nullptr;
            h->dist_cap = need;
            HIPCHK(h, hipMalloc(&h->ddist, need * sizeof(double)));
        }
    }
    // staging: the workspace matrix W wherever size x chunk fits it (always at sizes worth sharding: the chunks sum to ~Np^2 / 2);
    // a few blocks on many ranks pad beyond that -- then a buffer of its own
    if ((long long)size * h->shard_chunk > (long long)h->cap * h->cap && (size_t)size * (size_t)h->shard_chunk > h->stage_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->dstage) HIPCHK(h, hipFree(h->dstage));
        h->dstage = nullptr;
        h->stage_cap = (size_t)size * (size_t)h->shard_chunk;
        HIPCHK(h, hipMalloc(&h->dstage, h->stage_cap * sizeof(double)));
    }
    if (h->nblk > h->shard_off_cap) {
        if (h->dshard_off) HIPCHK(h, hipFree(h->dshard_off));
        h->shard_off_cap = h->nblk + 64;
        HIPCHK(h, hipMalloc(&h->dshard_off, (size_t)h->shard_off_cap * sizeof(long long)));
    }
    HIPCHK(h, hipMemcpyAsync(h->dshard_off, off.data(), (size_t)h->nblk * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// the pass of rank `rank` of `size`; exchange = false: without the collectives (what one rank's GPU does, timed by
// mfgp_dbg_eval_as_rank for the projections of DESIGN.md section 7 -- its results are NOT an evaluation's)
static int sharded_pass(mfgp_handle* h, const double* theta, double noise, double jitter, bool want_grad, int rank, int size,
                        bool exchange) {
    int rc = ensure_shard_plan(h, rank, size);
    if (rc) return rc;
    rc = set_params(h, theta, noise, jitter);
    if (rc) return rc;
    hipStream_t s = h->stream;
    const int Np = (int)h->Np;
    h->launches = 0;
    *h->hinfo = 0;
    ++h->epoch;
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[0], s));
    launch_kbuild_tri(s, h->spec, h->dX, (int)h->N, Np, h->buf[BUF_A], Np);
    h->launches++;
    if (h->stage_timing) HIPCHK(h, hipEventRecord(h->ev[1], s));
    for (const Step& st : h->pls.steps)
        if (run_step(h, st, false, 0, h->dtasks_s) != 0) return -1;
    if (exchange && size > 1) {
        // the rows of X^T to everybody: this rank's blocks packed into its chunk of the staging buffer (the workspace matrix W: the
        // image of the identity it held is dead once the sweep has joined), ONE in-place ncclAllGather, the others' blocks unpacked
        double* stage = (long long)size * h->shard_chunk <= (long long)h->cap * h->cap ? h->buf[BUF_W] : h->dstage;
        launch_shard_rows_copy(s, h->buf[BUF_S], Np, h->nblk, stage, h->dshard_off, h->shard_chunk, rank, size, false);
        rc = comm_allgather_chunks(h, stage, (size_t)h->shard_chunk, s);
        if (rc) return rc;
        launch_shard_rows_copy(s, h->buf[BUF_S], Np, h->nblk, stage, h->dshard_off, h->shard_chunk, rank, size, true);
        h->launches += 2;
    }
    launch_mirror_lower(s, h->buf[BUF_S], Np, Np);                              // X (lower part) from X^T (upper part)
    h->launches++;
    if (h->stage_timing) HIPCHK(h, hipEventRecord(h->ev[2], s));
    launch_rowdot(s, h->buf[BUF_S], Np, h->dY, h->dz, Np, Np, 0);
    launch_alpha_finish(s, h->buf[BUF_S], Np, h->dz, h->dalpha, Np, h->dlogdet, h->nblk, h->dres);
    h->launches += 2;
    if (h->stage_timing || (!want_grad && h->timing)) HIPCHK(h, hipEventRecord(h->ev[3], s));
    if (want_grad) {
        if (run_step(h, h->pls.kinv_step, true, 0, h->dtasks_s) != 0) return -1;   // this rank's rows of K^-1
        if (h->stage_timing) HIPCHK(h, hipEventRecord(h->ev[4], s));
        const size_t npart = (size_t)grad_num_partials(Np) * (MFGP_MAX_THETA + 1);
        HIPCHK(h, hipMemsetAsync(h->dpart, 0, npart * sizeof(double), s));
        launch_grad_tiles(s, h->spec, h->dX, h->buf[BUF_A], Np, h->dalpha, (int)h->N, Np, h->dpart, rank, size);
        if (exchange) {
            rc = comm_allreduce_sum(h, h->dpart, npart, s);
            if (rc) return rc;
        }
        launch_grad_finish(s, h->spec, h->dpart, Np, h->dres + 64);
        h->launches += 3;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[5], s));
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

static int sharded_finish(mfgp_handle* h, bool want_grad);
static int shard_broken(mfgp_handle* h, int rc);

int32_t mfgp_eval_sharded(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, double* nlml,
                          double* grad) {
    int rc = check_ready(h, "mfgp_eval_sharded");
    if (rc) return rc;
    if (!theta) return fail(h, -1, "mfgp_eval_sharded: theta is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    // everything that can be refused WITHOUT a collective in flight is checked first (as mfgp_sharded_lead does): a non-finite
    // parameter or a failed allocation is an ordinary, recoverable error of this call -- the communicator stays
    for (int i = 0; i < h->spec.np; ++i)
        if (!(theta[i] > 0.0) || !isfinite(theta[i])) return fail(h, -1, "parameters must be positive and finite");
    if (!(noise >= 0.0) || !(jitter >= 0.0)) return fail(h, -1, "noise and jitter must be >= 0");
    if (h->comm_aborted) return fail(h, -4, "mfgp_eval_sharded: the group's communicator was aborted after a failed pass");
    rc = ensure_shard_plan(h, h->comm_rank, h->comm_size);
    if (rc) return rc;
    const bool group = h->comm && h->comm_size > 1;      // (a failure inside a pass the peers run too: see shard_broken)
    rc = sharded_pass(h, theta, noise, jitter, want_grad != 0, h->comm_rank, h->comm_size, true);
    if (rc) return group ? shard_broken(h, rc) : rc;
    rc = sharded_finish(h, want_grad != 0);
    if (rc < 0 && group) return shard_broken(h, rc);
    if (rc) return rc;
    if (nlml) *nlml = 0.5 * ((double)h->N * 1.8378770664093453 + h->logdet + h->quad);
    if (want_grad && grad)
        for (int i = 0; i < h->spec.np + 1; ++i) grad[i] = h->grad[i];
    return 0;
}

// ---- leader / follower form -----------------------------------------------------------------------------------------
// A fit's sequential evaluations are driven by ONE optimiser (scipy L-BFGS-B on the leader, rank 0 of the group); the other
// ranks of the group have no optimiser of their own to keep in step: they SERVE -- mfgp_sharded_serve blocks, takes each
// evaluation's hyper-parameters from the leader (one broadcast of a 64-double control block on the communicator), runs its
// share, and returns when the leader releases the group (mfgp_sharded_release).
constexpr int CTL_WORDS = 64;     // [0] op (1 evaluate, 0 release)  [1] want_grad  [2] noise  [3] jitter  [4] P  [5 ..] theta
static_assert(5 + MFGP_MAX_THETA <= CTL_WORDS, "control block holds every parameter");

static int ctl_exchange(mfgp_handle* h, double* ctl, bool leader) {
    if (!h->dctl) {
        HIPCHK(h, hipMalloc(&h->dctl, CTL_WORDS * sizeof(double)));
        HIPCHK(h, hipHostMalloc(&h->hctl, CTL_WORDS * sizeof(double), hipHostMallocDefault));
    }
    hipStream_t s = h->stream;
    if (leader) {
        memcpy(h->hctl, ctl, CTL_WORDS * sizeof(double));
        HIPCHK(h, hipMemcpyAsync(h->dctl, h->hctl, CTL_WORDS * sizeof(double), hipMemcpyHostToDevice, s));
    }
    const int rc = comm_bcast_words(h, h->dctl, CTL_WORDS, 0, s);
    if (rc) return rc;
    if (!leader) {
        HIPCHK(h, hipMemcpyAsync(h->hctl, h->dctl, CTL_WORDS * sizeof(double), hipMemcpyDeviceToHost, s));
        if (int rs = comm_stream_wait(h, s, "mfgp_sharded_serve: waiting for the leader's control block")) return rs;
        memcpy(ctl, h->hctl, CTL_WORDS * sizeof(double));
    }
    return 0;
}

// a failure on this rank AFTER the control block told the group to start a pass: its collectives can no longer be matched
static int shard_broken(mfgp_handle* h, int rc) {
    const std::string why = h->err;
    comm_abort(h);                        // first: it also ends a collective of this rank that waits for a peer, so that the streams drain
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    (void)hipStreamSynchronize(h->stream);
    h->err = why + " [inside a pass the group had already started: the communicator was aborted, no further collective is issued]";
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return rc;
}

static int sharded_finish(mfgp_handle* h, bool want_grad) {
    if (int rs = comm_stream_wait(h, h->stream, "sharded evaluation: waiting for the pass (all-gather / all-reduce with the group)")) return rs;
    const bool streamed_flag = h->pl.kinv_streamed;       // finish_eval's flop accounting looks at the handle's own plan:
    h->pl.kinv_streamed = false;                          // a sharded pass never streams K^-1
    const double sweep0 = h->cum.cholinv_flops, kinv0 = h->cum.kinv_flops;
    const int rc = finish_eval(h, want_grad);
    h->pl.kinv_streamed = streamed_flag;
    h->kinv_valid = false;                                // (this rank holds only its own rows of K^-1)
    // finish_eval credited a whole evaluation; THIS RANK executed its share: the Cholesky in full or -- distributed -- a
    // G-th of it, a G-th of the inverse's and of K^-1's N^3 / 3 (serpentine ownership: the ranks' shares agree within 0.5 %).  The
    // counters of a multi-rank job are therefore per-rank EXECUTED work, and bench.py's roofline on such a line is labelled so.
    const int G = std::max(1, h->pls.shard.size);
    if (G > 1) {
        const double n3 = (double)h->Np * h->Np * h->Np / 3.0;
        h->cum.cholinv_flops = sweep0 + n3 * ((h->pls.shard.dist ? 1.0 / G : 1.0) + 1.0 / G);
        if (want_grad) h->cum.kinv_flops = kinv0 + n3 / G;
    }
    return rc;
}

int32_t mfgp_sharded_lead(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, double* nlml,
                          double* grad) {
    int rc = check_ready(h, "mfgp_sharded_lead");
    if (rc) return rc;
    if (!theta) return fail(h, -1, "mfgp_sharded_lead: theta is NULL");
    if (h->comm_rank != 0) return fail(h, -1, "mfgp_sharded_lead: only rank 0 of the handle's communicator leads");
    HIPCHK(h, hipSetDevice(h->device));
    for (int i = 0; i < h->spec.np; ++i)      // (checked BEFORE the followers are told: a refused call must not leave them mid-pass)
        if (!(theta[i] > 0.0) || !isfinite(theta[i])) return fail(h, -1, "parameters must be positive and finite");
    if (!(noise >= 0.0) || !(jitter >= 0.0)) return fail(h, -1, "noise and jitter must be >= 0");
    if (h->comm_aborted) return fail(h, -4, "mfgp_sharded_lead: the group's communicator was aborted after a failed pass");
    rc = ensure_shard_plan(h, 0, h->comm_size);          // (allocations BEFORE the followers are told: a failure here leaves them waiting, not mid-pass)
    if (rc) return rc;
    double ctl[CTL_WORDS] = {1.0, want_grad ? 1.0 : 0.0, noise, jitter, (double)h->spec.np};
    for (int i = 0; i < h->spec.np; ++i) ctl[5 + i] = theta[i];
    rc = ctl_exchange(h, ctl, true);
    if (rc) return rc < 0 ? shard_broken(h, rc) : rc;
    if (h->dbg_fail_sharded_in > 0 && --h->dbg_fail_sharded_in == 0)
        return shard_broken(h, fail(h, -2, "mfgp_sharded_lead: injected failure (mfgp_dbg_fail_sharded_after)"));
    rc = sharded_pass(h, theta, noise, jitter, want_grad != 0, 0, h->comm_size, true);
    if (rc) return shard_broken(h, rc);
    rc = sharded_finish(h, want_grad != 0);
    if (rc < 0) return shard_broken(h, rc);
    if (rc) return rc;
    if (nlml) *nlml = 0.5 * ((double)h->N * 1.8378770664093453 + h->logdet + h->quad);
    if (want_grad && grad)
        for (int i = 0; i < h->spec.np + 1; ++i) grad[i] = h->grad[i];
    return 0;
}

int32_t mfgp_sharded_release(mfgp_handle* h) {
    if (!h) return fail(h, -1, "mfgp_sharded_release: NULL");
    if (h->comm_rank != 0) return fail(h, -1, "mfgp_sharded_release: only rank 0 of the handle's communicator leads");
    if (h->comm_aborted) return fail(h, -4, "mfgp_sharded_release: the group's communicator was aborted after a failed pass; nothing to release");
    HIPCHK(h, hipSetDevice(h->device));
    double ctl[CTL_WORDS] = {0.0};
    const int rc = ctl_exchange(h, ctl, true);
    if (rc) return rc;
    return comm_stream_wait(h, h->stream, "mfgp_sharded_release: waiting for the followers to take the release");
}

int32_t mfgp_dbg_fail_sharded_after(mfgp_handle* h, int32_t n) {
    if (!h || n < 0) return fail(h, -1, "mfgp_dbg_fail_sharded_after: bad argument");
    h->dbg_fail_sharded_in = n;
    return 0;
}

int32_t mfgp_sharded_serve(mfgp_handle* h, int64_t* served) {
    int rc = check_ready(h, "mfgp_sharded_serve");
    if (rc) return rc;
    if (!h->comm || h->comm_rank == 0) return fail(h, -1, "mfgp_sharded_serve: for ranks > 0 of the handle's communicator");
    HIPCHK(h, hipSetDevice(h->device));
    int64_t n = 0;
    for (;;) {
        double ctl[CTL_WORDS];
        rc = ctl_exchange(h, ctl, false);
        if (rc) return rc;
        if (ctl[0] == 0.0) break;
        if ((int)ctl[4] != h->spec.np)       // (the leader is inside the pass already: its collectives must not wait for this rank)
            return shard_broken(h, fail(h, -1, "mfgp_sharded_serve: the leader's kernel has another parameter count"));
        const bool g = ctl[1] != 0.0;
        if (h->dbg_fail_sharded_in > 0 && --h->dbg_fail_sharded_in == 0)
            return shard_broken(h, fail(h, -2, "mfgp_sharded_serve: injected failure (mfgp_dbg_fail_sharded_after)"));
        rc = sharded_pass(h, ctl + 5, ctl[2], ctl[3], g, h->comm_rank, h->comm_size, true);
        if (rc) return shard_broken(h, rc);
        rc = sharded_finish(h, g);      // > 0: not positive definite -- the leader sees the same pivot and decides what comes next
        if (rc < 0) return shard_broken(h, rc);
        ++n;
    }
    if (served) *served = n;
    return 0;
}

// test / measurement hook: the device work of rank `rank` of `size` for one evaluation, WITHOUT the exchange steps; *ms = its
// duration (HIP events).  The handle is left without a valid factorisation.
int32_t mfgp_dbg_eval_as_rank(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, int32_t rank,
                              int32_t size, double* ms) {
    int rc = check_ready(h, "mfgp_dbg_eval_as_rank");
    if (rc) return rc;
    if (!theta || !ms || size < 1 || rank < 0 || rank >= size) return fail(h, -1, "mfgp_dbg_eval_as_rank: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    const bool t0 = h->timing;
    h->timing = true;
    rc = sharded_pass(h, theta, noise, jitter, want_grad != 0, rank, size, false);
    h->timing = t0;
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *ms = ev_ms(h->ev[0], h->ev[want_grad ? 5 : 3]);
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

int32_t mfgp_kbuild_rows(mfgp_handle* h, const double* theta, double noise, double jitter, int64_t row_begin,
                         int64_t row_end) {
    int rc = check_ready(h, "mfgp_kbuild_rows");
    if (rc) return rc;
    if (!theta) return fail(h, -1, "mfgp_kbuild_rows: theta is NULL");
    if (row_begin < 0 || row_end > h->Np || row_begin >= row_end || row_begin % 64 || row_end % 64)
        return fail(h, -1, "mfgp_kbuild_rows: rows must be a non-empty range of multiples of 64 within the padded size");
    HIPCHK(h, hipSetDevice(h->device));
    rc = set_params(h, theta, noise, jitter);
    if (rc) return rc;
    launch_kbuild_rows(h->stream, h->spec, h->dX, (int)h->N, (int)h->Np, h->buf[BUF_A], (int)h->Np,
                       (int)row_begin, (int)row_end);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

// the rows of every 128-row block that rank `rank` of `size` owns (mfgp_row_block_owner): what a rank builds before mfgp_allgather_rows
int32_t mfgp_kbuild_owned_rows(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t rank, int32_t size) {
    int rc = check_ready(h, "mfgp_kbuild_owned_rows");
    if (rc) return rc;
    if (!theta) return fail(h, -1, "mfgp_kbuild_owned_rows: theta is NULL");
    if (size < 1 || rank < 0 || rank >= size) return fail(h, -1, "mfgp_kbuild_owned_rows: need 0 <= rank < size");
    HIPCHK(h, hipSetDevice(h->device));
    rc = set_params(h, theta, noise, jitter);
    if (rc) return rc;
    for (int b = 0; b < h->nblk; ++b) {
        if (shard_owner(b, size) != rank) continue;
        int e = b + 1;                             // (consecutive owned blocks -- the turning points of the serpentine -- in one launch)
        while (e < h->nblk && shard_owner(e, size) == rank) ++e;
        launch_kbuild_rows(h->stream, h->spec, h->dX, (int)h->N, (int)h->Np, h->buf[BUF_A], (int)h->Np, b * NB, e * NB);
        b = e - 1;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

int32_t mfgp_dev_matrix(mfgp_handle* h, void** ptr, int64_t* padded_n) {
    if (!h || !ptr || !padded_n) return fail(h, -1, "mfgp_dev_matrix: NULL");
    if (!h->have_data) return fail(h, -1, "mfgp_dev_matrix: mfgp_set_data not called");
    *ptr = h->buf[BUF_A];
    *padded_n = h->Np;
    return 0;
}

int32_t mfgp_eval_prebuilt(mfgp_handle* h, int32_t want_grad, double* nlml, double* grad) {
    int rc = check_ready(h, "mfgp_eval_prebuilt");
    if (rc) return rc;
    if (!h->params_set) return fail(h, -1, "mfgp_eval_prebuilt: mfgp_kbuild_rows not called");
    HIPCHK(h, hipSetDevice(h->device));
    rc = enqueue_eval(h, nullptr, 0.0, 0.0, want_grad != 0, true);
    if (rc) return rc;
    rc = finish_eval(h, want_grad != 0);
    if (rc) return rc;
    if (nlml) *nlml = 0.5 * ((double)h->N * 1.8378770664093453 + h->logdet + h->quad);
    if (want_grad && grad)
        for (int i = 0; i < h->spec.np + 1; ++i) grad[i] = h->grad[i];
    return 0;
}

}  // extern "C"
