// comm_rccl.hip -- the exchange steps of the multi-GPU path: one process per GPU, one RCCL
// communicator per engine handle, collectives enqueued on the handle's own stream.
//
//   mfgp_allgather_rows  : the K(X,X) row-block layout of north_star / SURVEY 8(e3): every rank has built its block of
//                          full rows of Ky in place (mfgp_kbuild_rows); ONE in-place ncclAllGather over xGMI completes the
//                          matrix on every rank (per-rank message 8 Np^2 / size bytes).
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
#include <mutex>
#include <rccl/rccl.h>
#include "mfgp_internal.h"

using namespace mfgp;

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
}  // namespace

namespace mfgp {
void comm_release(mfgp_handle* h) {
    if (h && h->comm) {
        rccl().CommDestroy(static_cast<ncclComm_t>(h->comm));
        h->comm = nullptr;
        h->comm_rank = 0;
        h->comm_size = 1;
    }
}

int comm_allgather_chunks(mfgp_handle* h, double* base, size_t chunk, hipStream_t s) {
    if (!h->comm || h->comm_size <= 1) return 0;
    ncclResult_t r = rccl().AllGather(base + (size_t)h->comm_rank * chunk, base, chunk, ncclDouble,
                                      static_cast<ncclComm_t>(h->comm), s);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather (rows of X^T)", r);
    return 0;
}

int comm_bcast_words(mfgp_handle* h, double* dev, size_t count, int root, hipStream_t s) {
    if (!h->comm || h->comm_size <= 1) return 0;
    RcclApi& api = rccl();
    if (!api.Broadcast) return fail(h, -4, "librccl lacks ncclBroadcast");
    ncclResult_t r = api.Broadcast(dev, dev, count, ncclDouble, root, static_cast<ncclComm_t>(h->comm), s);
    if (r != ncclSuccess) return rccl_fail(h, "ncclBroadcast (control block)", r);
    return 0;
}

int comm_allreduce_sum(mfgp_handle* h, double* buf, size_t count, hipStream_t s) {
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

int32_t mfgp_allgather_rows(mfgp_handle* h) {
    if (!h) return fail(h, -1, "mfgp_allgather_rows: NULL");
    if (!h->comm) return fail(h, -1, "mfgp_allgather_rows: no communicator (mfgp_comm_init)");
    if (!h->have_data) return fail(h, -1, "mfgp_allgather_rows: mfgp_set_data not called");
    const int64_t Np = h->Np;
    if (Np % (64 * (int64_t)h->comm_size) != 0)
        return fail(h, -1, "mfgp_allgather_rows: the padded size must split into equal 64-row multiples per rank");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t count = (size_t)(Np / h->comm_size) * (size_t)Np;   // doubles per rank
    double* A = h->buf[BUF_A];
    // in place: rank r's block already sits at its final position (sendbuff == recvbuff + r * count)
    ncclResult_t r = rccl().AllGather(A + (size_t)h->comm_rank * count, A, count, ncclDouble,
                                      static_cast<ncclComm_t>(h->comm), h->stream);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather (row blocks)", r);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

int32_t mfgp_allgather_host(mfgp_handle* h, const double* send, int64_t count, double* recv) {
    if (!h || !send || !recv) return fail(h, -1, "mfgp_allgather_host: NULL argument");
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
    ncclResult_t r = rccl().AllGather(mine, h->dstage, (size_t)count, ncclDouble, static_cast<ncclComm_t>(h->comm),
                                      h->stream);
    if (r != ncclSuccess) return rccl_fail(h, "ncclAllGather (host vectors)", r);
    HIPCHK(h, hipMemcpyAsync(recv, h->dstage, total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
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
