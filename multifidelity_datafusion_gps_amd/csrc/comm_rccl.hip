// comm_rccl.hip -- the exchange steps of the multi-GPU path: one process per GPU, one RCCL
// communicator per engine handle, collectives enqueued on the handle's own stream.
//
//   mfgp_allgather_rows  : the K(X,X) row-block layout of north_star / SURVEY 8(e3): every rank has built the rows of the
//                          128-row blocks it owns (mfgp_kbuild_owned_rows: serpentine block-cyclic deal); the LOWER part of every
//                          block is packed by owner and ONE ncclAllGather of equal chunks completes the lower triangle of Ky on
//                          every rank (4 Np (Np + 128) bytes in all: half of what full rows would move).
//   mfgp_comm_state      : 0 / size / -1 = no communicator / alive / ABORTED (comm_abort: a rank whose shared pass failed, or whose
//                          peers went silent, tears the communicator down without them and refuses every further collective).
//   mfgp_allgather_host  : the small gathers -- predictive (mean, variance) row blocks (SURVEY 8(e1), 16 B per test row)
//                          and restart results (8(e2)) -- host buffer -> device staging -> ncclAllGather -> host.
//   mfgp_rows_download / mfgp_rows_upload : the same row blocks through host memory, for transports other than RCCL
//                          (multi-process tests on a one-GPU box, where RCCL refuses two ranks on one device).
//
// librccl is opened lazily (dlopen) by mfgp_comm_unique_id / mfgp_comm_init: a single-GPU process never loads it, and
// libmfgp_hip.so itself has no link-time dependency on it.  The unique id travels between the ranks as 128 opaque bytes
// over whatever the host side uses for rendezvous (sharding.SocketComm: TCP on 127.0.0.1).
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <rccl/rccl.h>
#include "mfgp_internal.h"

using namespace mfgp;

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // (optional) tears a communicator down WITHOUT the peers' cooperation
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    // (optional: only mfgp_eval_sharded needs them)
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string load_error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            api.load_error = std::string("cannot dlopen librccl: ") + dlerror();
            return;
        }
        bool ok = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(api.lib, n);
            if (!p) { ok = false; api.load_error = std::string("librccl lacks ") + n; }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (ok) {
            api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.lib, "ncclCommAbort"));
            api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(api.lib, "ncclBroadcast"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
        }
        if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    });
    return api;
}

int rccl_fail(mfgp_handle* h, const char* what, ncclResult_t r) {
    return fail(h, -4, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}

// ncclAllGather, or -- test hook mfgp_dbg_fail_collective_after -- the error RCCL would have returned without the call being made
ncclResult_t all_gather(mfgp_handle* h, const void* send, void* recv, size_t count, hipStream_t s) {
    if (h->dbg_fail_collective_in > 0 && --h->dbg_fail_collective_in == 0) return ncclInternalError;
    return rccl().AllGather(send, recv, count, ncclDouble, static_cast<ncclComm_t>(h->comm), s);
}

// a gather whose ncclAllGather was refused: the peers are inside (or on their way into) a collective this rank will never join.
// Exactly what shard_broken does for a shared pass: tear the communicator down without them (which also poisons the handle's
// further collective calls), let what this call had already enqueued drain, report -4 -- the peers give up at their deadline
int gather_broken(mfgp_handle* h, hipStream_t s, const char* what, ncclResult_t r) {
    const int rc = rccl_fail(h, what, r);
    const std::string why = h->err;
    comm_abort(h);
    (void)hipStreamSynchronize(s);
    h->err = why + " [the communicator was aborted, no further collective is issued]";
    return rc;
}
}  // namespace

namespace mfgp {
void comm_release(mfgp_handle* h) {
    if (h) { h->coll_us = 0.0; for (double& v : h->calib) v = 0.0; }
    if (h && h->comm) {
        rccl().CommDestroy(static_cast<ncclComm_t>(h->comm));
        h->comm = nullptr;
        h->comm_rank = 0;
        h->comm_size = 1;
    }
    if (h) h->comm_aborted = false;
}

// A collective of this rank can no longer be matched by its peers (a failure after the control block told them to start a pass;
// a peer that went silent): the communicator is torn down without them -- ncclCommAbort also ends RCCL kernels of this rank that
// still wait for a peer -- and the handle remembers it: every further collective call on it is refused (-4) instead of being
// enqueued against peers that are somewhere else in the protocol (mismatched collectives never return).
void comm_abort(mfgp_handle* h) {
    if (!h || !h->comm) return;
    RcclApi& api = rccl();
    if (api.CommAbort) api.CommAbort(static_cast<ncclComm_t>(h->comm));
    // (without ncclCommAbort the communicator is leaked rather than destroyed: ncclCommDestroy waits for the peers)
    h->comm = nullptr;
    h->comm_aborted = true;
}

static int comm_usable(mfgp_handle* h, const char* who) {
    if (h->comm_aborted)
        return fail(h, -4, std::string(who) + ": the handle's communicator was aborted after a failed collective; the process must end");
    return 0;
}

// Waiting for a stream that carries a collective: hipStreamSynchronize never returns when a peer is gone (the RCCL kernel waits for
// its data for ever).  With a communicator of more than one rank the wait is a poll with a deadline (MFGP_SHARD_TIMEOUT_S, default
// 600 s -- a pass takes milliseconds, a leader's optimiser step between two passes less); past it the communicator is aborted
// (which ends the waiting kernel) and the call fails with -4: the rank exits with an error instead of hanging the job.
static double shard_timeout_s() {
    const char* v = getenv("MFGP_SHARD_TIMEOUT_S");
    const double x = v && *v ? atof(v) : 0.0;
    return x > 0.0 ? x : 600.0;
}
int comm_stream_wait(mfgp_handle* h, hipStream_t s, const char* what) {
    if (!h->comm || h->comm_size <= 1) {
        HIPCHK(h, hipStreamSynchronize(s));
        return 0;
    }
    const double limit = shard_timeout_s();
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) {
            h->err = std::string(what) + ": " + hipGetErrorString(e);
            comm_abort(h);
            return -2;
        }
        if ((spins & 63) == 63) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > limit) {
                comm_abort(h);
                (void)hipStreamSynchronize(s);      // (the aborted collective's kernel ends; what was enqueued behind it drains)
                return fail(h, -4, std::string(what) + ": no progress for " + std::to_string((int)limit) +
                                       " s -- a peer of the group is gone or somewhere else in the protocol; the communicator was aborted");
            }
            if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(el > 0.1 ? 500 : 20));
        }
    }
}

int comm_allgather_chunks(mfgp_handle* h, double* base, size_t chunk, hipStream_t s) {
    if (int rc = comm_usable(h, "ncclAllGather (rows of X^T)")) return rc;
    if (!h->comm || h->comm_size <= 1) return 0;
    ncclResult_t r = all_gather(h, base + (size_t)h->comm_rank * chunk, base, chunk, s);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather (rows of X^T)", r);
    return 0;
}

int comm_bcast_words(mfgp_handle* h, double* dev, size_t count, int root, hipStream_t s) {
    if (int rc = comm_usable(h, "ncclBroadcast (control block)")) return rc;
    if (!h->comm || h->comm_size <= 1) return 0;
    RcclApi& api = rccl();
    if (!api.Broadcast) return fail(h, -4, "librccl lacks ncclBroadcast");
    ncclResult_t r = api.Broadcast(dev, dev, count, ncclDouble, root, static_cast<ncclComm_t>(h->comm), s);
    if (r != ncclSuccess) return rccl_fail(h, "ncclBroadcast (control block)", r);
    return 0;
}

int comm_allreduce_sum(mfgp_handle* h, double* buf, size_t count, hipStream_t s) {
    if (int rc = comm_usable(h, "ncclAllReduce (gradient tile partials)")) return rc;
    if (!h->comm || h->comm_size <= 1) return 0;
    RcclApi& api = rccl();
    if (!api.AllReduce) return fail(h, -4, "librccl lacks ncclAllReduce");
    ncclResult_t r = api.AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(h->comm), s);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllReduce (gradient tile partials)", r);
    return 0;
}
}  // namespace mfgp

extern "C" {

int32_t mfgp_comm_unique_id(uint8_t* out128) {
    if (!out128) return fail(nullptr, -1, "mfgp_comm_unique_id: NULL");
    RcclApi& api = rccl();
    if (!api.lib) return fail(nullptr, -4, "mfgp_comm_unique_id: " + api.load_error);
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", r);
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof id);
    return 0;
}

int32_t mfgp_comm_init(mfgp_handle* h, const uint8_t* id128, int32_t rank, int32_t size) {
    if (!h || !id128) return fail(h, -1, "mfgp_comm_init: NULL argument");
    if (size < 1 || rank < 0 || rank >= size) return fail(h, -1, "mfgp_comm_init: need 0 <= rank < size");
    RcclApi& api = rccl();
    if (!api.lib) return fail(h, -4, "mfgp_comm_init: " + api.load_error);
    // every "nothing hangs" guarantee of the collective calls rests on ncclCommAbort (it ends a collective kernel of this rank that
    // waits for an absent peer): a communicator of several ranks is not created without it
    if (size > 1 && !api.CommAbort) return fail(h, -4, "mfgp_comm_init: librccl lacks ncclCommAbort");
    HIPCHK(h, hipSetDevice(h->device));
    comm_release(h);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t r = api.CommInitRank(&c, size, id, rank);
    if (r != ncclSuccess) return rccl_fail(h, "ncclCommInitRank", r);
    h->comm = c;
    h->comm_rank = rank;
    h->comm_size = size;
    return 0;
}

int32_t mfgp_comm_destroy(mfgp_handle* h) {
    if (!h) return fail(h, -1, "mfgp_comm_destroy: NULL");
    (void)hipSetDevice(h->device);
    comm_release(h);
    return 0;
}

int32_t mfgp_comm_state(mfgp_handle* h) {
    if (!h) return 0;
    return h->comm_aborted ? -1 : (h->comm ? h->comm_size : 0);
}

// What one small collective of THIS communicator costs on a stream, measured (the planner's choice between the
// replicated and the distributed Cholesky of a shared evaluation rested on a constant derived from a one-GPU projection).
// Collective: every rank of the communicator calls it with the same `reps`.  Two shapes, the two exchange steps of a distributed
// Cholesky's block column (plan.h): ncclBroadcast of a diagonal message (2 x 128^2 + 2 doubles, root 0) and an in-place
// ncclAllGather of `panel_blocks` 128 x 128 blocks per rank (0: what a column in the middle of the handle's current matrix
// carries per rank, at least one block) -- each timed `reps` times by the host clock from the enqueue to the stream running dry
// (that is what the serial chain waits for), after two untimed repetitions (connections are set up lazily); medians.  Rank 0's
// medians are then broadcast, so that every rank of the group holds the SAME figures and plans alike:
//   out[0] broadcast us, out[1] all-gather us, out[2] all-gather bytes per rank, out[3] all-gather GB/s (bytes received / time),
//   out[4] reps, out[5] the worst of this rank's OWN two medians (diagnostic).
// Stored on the handle: coll_us = the larger of out[0], out[1] + 8 us for the pack and unpack launches around an exchange.
int32_t mfgp_comm_calibrate(mfgp_handle* h, int32_t reps, int32_t panel_blocks, double* out6) {
    if (!h) return fail(h, -1, "mfgp_comm_calibrate: NULL");
    if (int rc = comm_usable(h, "mfgp_comm_calibrate")) return rc;
    if (reps < 1 || reps > 1000 || panel_blocks < 0) return fail(h, -1, "mfgp_comm_calibrate: need 1 <= reps <= 1000, panel_blocks >= 0");
    for (double& v : h->calib) v = 0.0;
    h->coll_us = 0.0;
    if (!h->comm || h->comm_size <= 1) {
        if (out6) for (int i = 0; i < 6; ++i) out6[i] = 0.0;
        return 0;
    }
    RcclApi& api = rccl();
    if (!api.Broadcast) return fail(h, -4, "librccl lacks ncclBroadcast");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int size = h->comm_size;
    const size_t diag = 2 * (size_t)NB * NB + 2;
    const int blocks = panel_blocks > 0 ? panel_blocks : std::max(1, (h->have_data ? h->nblk / 2 : 32) / size);
    const size_t chunk = (size_t)blocks * NB * NB;
    const size_t need = std::max(diag, chunk * size) + 8;
    if (need > h->stage_cap) {
        HIPCHK(h, hipStreamSynchronize(s));
        if (h->dstage) HIPCHK(h, hipFree(h->dstage));
        h->dstage = nullptr;
        h->stage_cap = need;
        HIPCHK(h, hipMalloc(&h->dstage, h->stage_cap * sizeof(double)));
    }
    HIPCHK(h, hipMemsetAsync(h->dstage, 0, need * sizeof(double), s));
    HIPCHK(h, hipStreamSynchronize(s));
    std::vector<double> tb, tg;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int it = 0; it < reps + 2; ++it) {
        const auto t0 = now();
        ncclResult_t r = api.Broadcast(h->dstage, h->dstage, diag, ncclDouble, 0, static_cast<ncclComm_t>(h->comm), s);
        if (r != ncclSuccess) return gather_broken(h, s, "mfgp_comm_calibrate: ncclBroadcast", r);
        if (int rc = comm_stream_wait(h, s, "mfgp_comm_calibrate: waiting for a broadcast")) return rc;
        if (it >= 2) tb.push_back(std::chrono::duration<double>(now() - t0).count() * 1e6);
    }
    for (int it = 0; it < reps + 2; ++it) {
        const auto t0 = now();
        ncclResult_t r = all_gather(h, h->dstage + (size_t)h->comm_rank * chunk, h->dstage, chunk, s);
        if (r != ncclSuccess) return gather_broken(h, s, "mfgp_comm_calibrate: ncclAllGather", r);
        if (int rc = comm_stream_wait(h, s, "mfgp_comm_calibrate: waiting for an all-gather")) return rc;
        if (it >= 2) tg.push_back(std::chrono::duration<double>(now() - t0).count() * 1e6);
    }
    std::sort(tb.begin(), tb.end());
    std::sort(tg.begin(), tg.end());
    double mine[2] = {tb[tb.size() / 2], tg[tg.size() / 2]}, agreed[2] = {0, 0};
    // rank 0's medians to everybody (through the staging buffer: the collectives work on device memory)
    HIPCHK(h, hipMemcpy(h->dstage, mine, sizeof mine, hipMemcpyHostToDevice));
    ncclResult_t r = api.Broadcast(h->dstage, h->dstage, 2, ncclDouble, 0, static_cast<ncclComm_t>(h->comm), s);
    if (r != ncclSuccess) return gather_broken(h, s, "mfgp_comm_calibrate: ncclBroadcast (agreed figures)", r);
    if (int rc = comm_stream_wait(h, s, "mfgp_comm_calibrate: waiting for the agreed figures")) return rc;
    HIPCHK(h, hipMemcpy(agreed, h->dstage, sizeof agreed, hipMemcpyDeviceToHost));
    h->calib[0] = agreed[0];
    h->calib[1] = agreed[1];
    h->calib[2] = (double)(chunk * sizeof(double));
    h->calib[3] = agreed[1] > 0 ? (double)(chunk * sizeof(double)) * (size - 1) / (agreed[1] * 1e-6) / 1e9 : 0.0;
    h->calib[4] = reps;
    h->calib[5] = std::max(mine[0], mine[1]);
    h->coll_us = std::max(agreed[0], agreed[1]) + 8.0;
    if (out6) for (int i = 0; i < 6; ++i) out6[i] = h->calib[i];
    return 0;
}

// What the planner makes of it for the handle's current matrix and communicator -- the decision mfgp_eval_sharded /
// mfgp_sharded_lead will plan under (MFGP_DIST_CHOL = 0 / 1 overrides it):
//   out[0] 1 = the Cholesky distributed over the group, 0 = replicated on every rank; out[1] projected saving ms; out[2] cost ms
//   (collectives x measured us); out[3] collectives on the chain; out[4] the measured us per collective (0: never calibrated);
//   out[5] 1 if an environment switch forced the choice
int32_t mfgp_shard_decision(mfgp_handle* h, double* out6) {
    if (!h || !out6) return fail(h, -1, "mfgp_shard_decision: NULL argument");
    if (!h->have_data) return fail(h, -1, "mfgp_shard_decision: mfgp_set_data not called");
    const int size = h->comm ? h->comm_size : 1;
    const DistDecision d = dist_cholesky_pays(h->nblk, size, h->coll_us);
    const int forced = h->pl.opts.dist_chol;
    out6[0] = size > 1 && (forced >= 0 ? forced != 0 : d.dist) ? 1.0 : 0.0;
    out6[1] = d.saving_ms; out6[2] = d.cost_ms; out6[3] = d.collectives; out6[4] = h->coll_us; out6[5] = forced >= 0 ? 1.0 : 0.0;
    return 0;
}

// The planner's rule itself, without a handle (pure host arithmetic: testable where there is no GPU)
int32_t mfgp_dist_cholesky_pays(int32_t nblk, int32_t size, double coll_us, double* saving_ms, double* cost_ms) {
    const DistDecision d = dist_cholesky_pays(nblk, size, coll_us);
    if (saving_ms) *saving_ms = d.saving_ms;
    if (cost_ms) *cost_ms = d.cost_ms;
    return d.dist ? 1 : 0;
}

int32_t mfgp_dbg_fail_collective_after(mfgp_handle* h, int32_t n) {
    if (!h || n < 0) return fail(h, -1, "mfgp_dbg_fail_collective_after: bad argument");
    h->dbg_fail_collective_in = n;
    return 0;
}

int32_t mfgp_row_block_owner(int32_t block, int32_t size) { return block < 0 ? -1 : shard_owner(block, size); }

// The row blocks of Ky: 128-row blocks dealt to the ranks in the serpentine block-cyclic order of a sharded
// evaluation (plan.h shard_owner) -- with CONTIGUOUS blocks per rank the last rank's rows are full-width while the factorisation
// reads only the lower triangle, so the largest message stays 8 Np^2 / size bytes whatever is cut; dealt this way every rank's
// blocks hold the same share of the triangle and ONE in-place all-gather of equal packed chunks moves 4 Np (Np + 128) bytes in
// all instead of 8 Np^2 (round 5: half the bytes of an exchange that costs 6 x the local build).
int32_t mfgp_allgather_rows(mfgp_handle* h) {
    if (!h) return fail(h, -1, "mfgp_allgather_rows: NULL");
    if (int rc = comm_usable(h, "mfgp_allgather_rows")) return rc;
    if (!h->comm) return fail(h, -1, "mfgp_allgather_rows: no communicator (mfgp_comm_init)");
    if (!h->have_data) return fail(h, -1, "mfgp_allgather_rows: mfgp_set_data not called");
    h->factorized = h->kinv_valid = h->grad_valid = false;
    const int size = h->comm_size, rank = h->comm_rank, nblk = h->nblk;
    if (size <= 1) return 0;                       // the group of one holds every block already
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (h->row_off_nblk != nblk || h->row_off_size != size) {
        // block b's lower part (128 x 128 (b + 1) doubles) at offset off[b] of its owner's chunk; chunks padded to the largest
        std::vector<long long> off((size_t)nblk), fill((size_t)size, 0);
        for (int b = 0; b < nblk; ++b) {
            const int own = shard_owner(b, size);
            off[(size_t)b] = fill[(size_t)own];
            fill[(size_t)own] += 128LL * 128LL * (b + 1);
        }
        h->row_chunk = *std::max_element(fill.begin(), fill.end());
        if (nblk > h->row_off_cap) {
            HIPCHK(h, hipStreamSynchronize(s));
            if (h->drow_off) HIPCHK(h, hipFree(h->drow_off));
            h->drow_off = nullptr;
            h->row_off_cap = nblk + 64;
            HIPCHK(h, hipMalloc(&h->drow_off, (size_t)h->row_off_cap * sizeof(long long)));
        }
        HIPCHK(h, hipMemcpyAsync(h->drow_off, off.data(), (size_t)nblk * sizeof(long long), hipMemcpyHostToDevice, s));
        HIPCHK(h, hipStreamSynchronize(s));        // (off is a local)
        h->row_off_nblk = nblk;
        h->row_off_size = size;
    }
    // staging: the workspace matrix W (idle until the factorisation starts) wherever size x chunk fits it, else a buffer of its own
    const size_t need = (size_t)size * (size_t)h->row_chunk;
    double* stage = h->buf[BUF_W];
    if (need > (size_t)h->cap * (size_t)h->cap) {
        if (need > h->stage_cap) {
            HIPCHK(h, hipStreamSynchronize(s));
            if (h->dstage) HIPCHK(h, hipFree(h->dstage));
            h->dstage = nullptr;
            h->stage_cap = need;
            HIPCHK(h, hipMalloc(&h->dstage, h->stage_cap * sizeof(double)));
        }
        stage = h->dstage;
    }
    double* A = h->buf[BUF_A];
    launch_shard_rows_copy(s, A, (int)h->Np, nblk, stage, h->drow_off, h->row_chunk, rank, size, false, true);
    ncclResult_t r = all_gather(h, stage + (size_t)rank * (size_t)h->row_chunk, stage, (size_t)h->row_chunk, s);
    if (r != ncclSuccess) return gather_broken(h, s, "ncclAllGather (row blocks of Ky, lower part)", r);
    launch_shard_rows_copy(s, A, (int)h->Np, nblk, stage, h->drow_off, h->row_chunk, rank, size, true, true);
    if (int rc = comm_stream_wait(h, s, "mfgp_allgather_rows: waiting for the all-gather of the row blocks")) return rc;
    HIPCHK(h, hipGetLastError());
    return 0;
}

int32_t mfgp_allgather_host(mfgp_handle* h, const double* send, int64_t count, double* recv) {
    if (!h || !send || !recv) return fail(h, -1, "mfgp_allgather_host: NULL argument");
    if (int rc = comm_usable(h, "mfgp_allgather_host")) return rc;
    if (!h->comm) return fail(h, -1, "mfgp_allgather_host: no communicator (mfgp_comm_init)");
    if (count < 1) return fail(h, -1, "mfgp_allgather_host: count < 1");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t total = (size_t)count * (size_t)h->comm_size;
    if (total > h->stage_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->dstage) HIPCHK(h, hipFree(h->dstage));
        h->dstage = nullptr;
        h->stage_cap = total + total / 2;
        HIPCHK(h, hipMalloc(&h->dstage, h->stage_cap * sizeof(double)));
    }
    double* mine = h->dstage + (size_t)h->comm_rank * count;
    HIPCHK(h, hipMemcpyAsync(mine, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    ncclResult_t r = all_gather(h, mine, h->dstage, (size_t)count, h->stream);
    if (r != ncclSuccess) return gather_broken(h, h->stream, "ncclAllGather (host vectors)", r);   // (the copy above drains there)
    // the deadline wait BEFORE the copy back: a copy into pageable memory blocks the host until the stream reaches it
    if (int rc = comm_stream_wait(h, h->stream, "mfgp_allgather_host: waiting for the all-gather")) return rc;
    HIPCHK(h, hipMemcpy(recv, h->dstage, total * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

static int rows_check(mfgp_handle* h, int64_t row_begin, int64_t row_end, const void* p, const char* who) {
    if (!h || !p) return fail(h, -1, std::string(who) + ": NULL argument");
    if (!h->have_data) return fail(h, -1, std::string(who) + ": mfgp_set_data not called");
    if (row_begin < 0 || row_end > h->Np || row_begin >= row_end)
        return fail(h, -1, std::string(who) + ": rows must be a non-empty range within the padded size");
    return 0;
}

int32_t mfgp_rows_download(mfgp_handle* h, int64_t row_begin, int64_t row_end, double* out) {
    int rc = rows_check(h, row_begin, row_end, out, "mfgp_rows_download");
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->buf[BUF_A] + (size_t)row_begin * h->Np, (size_t)(row_end - row_begin) * h->Np * sizeof(double),
                        hipMemcpyDeviceToHost));
    return 0;
}

int32_t mfgp_rows_upload(mfgp_handle* h, int64_t row_begin, int64_t row_end, const double* in) {
    int rc = rows_check(h, row_begin, row_end, in, "mfgp_rows_upload");
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->buf[BUF_A] + (size_t)row_begin * h->Np, in, (size_t)(row_end - row_begin) * h->Np * sizeof(double),
                        hipMemcpyHostToDevice));
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

}  // extern "C"
