// api_batch.hip -- mfgp_eval_batch: B hyper-parameter points in ONE pass of the plan (the lock-stepped restarts of the reference's
// recipe, src/abstractMFGP.py:131-137), the batch slab's memory policy (mfgp_mem_info / mfgp_batch_mem).  Split out of mfgp_api.hip in
// round 6; the shared pieces of an evaluation are in api_shared.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "mfgp_internal.h"
#include "api_shared.h"

using namespace mfgp;

extern "C" {

// ---- batched evaluation ------------------------------------------------------------------------------------------
#define MFGP_MAX_BATCH_SETS 16

// device + pinned bytes `sets` matrix sets of a batch take on this handle (the slab dominates: 32 cap^2 bytes per set)
static size_t batch_bytes(const mfgp_handle* h, int sets) {
    const size_t cap = (size_t)h->cap;
    const size_t per = 4 * cap * cap + 2 * cap + cap / NB + (size_t)grad_num_partials((int)cap) * (MFGP_MAX_THETA + 1) + mfgp_handle::BRES;
    return (size_t)sets * per * sizeof(double);
}

// MFGP_BATCH_MEM_CAP (bytes; unset / 0: none): the most ONE handle's batch slab may take -- for a host application that shares the
// device, and for the tests of the fallback below
static size_t batch_mem_cap() {
    const char* v = getenv("MFGP_BATCH_MEM_CAP");
    if (!v || !*v) return 0;
    const double x = atof(v);
    return x > 0 ? (size_t)x : 0;
}

// -> 0, MFGP_ERR_OOM (everything released again; nothing else of the handle touched), or -2 (another HIP error)
static int alloc_batch(mfgp_handle* h, int want) {
    const size_t cap = (size_t)h->cap;
    hipError_t e = hipSuccess;
    auto dev = [&](double** p, size_t n) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(double)); };
    dev(&h->bslab, (size_t)want * 4 * cap * cap);
    dev(&h->bz, (size_t)want * cap);
    dev(&h->balpha, (size_t)want * cap);
    dev(&h->blogdet, (size_t)want * (cap / NB));
    dev(&h->bpart, (size_t)want * grad_num_partials((int)cap) * (MFGP_MAX_THETA + 1));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->bhres), (size_t)want * mfgp_handle::BRES * sizeof(double), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->bdres), h->bhres, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();               // (an allocation failure is not sticky, but it is the "last error" until read)
        free_batch(h);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation)
            return fail(h, MFGP_ERR_OOM, "mfgp_eval_batch: out of device memory for " + std::to_string(want) + " matrix sets (" +
                                             std::to_string(batch_bytes(h, want) >> 20) + " MiB)");
        return fail(h, -2, std::string("mfgp_eval_batch: allocating the batch slab: ") + hipGetErrorString(e));
    }
    memset(h->bhres, 0, (size_t)want * mfgp_handle::BRES * sizeof(double));
    h->bsets = want;
    h->bsets_cap = h->cap;
    return 0;
}

// Memory policy of a batch (round 5): a request the device (or MFGP_BATCH_MEM_CAP) cannot hold is its own status, MFGP_ERR_OOM --
// never a generic HIP error -- and leaves the handle usable: the sets it held before are still there (or re-allocated), every other
// call works, and the caller retries with fewer sets (engine.LockstepLane does) or with single evaluations, which need no slab.
static int ensure_batch(mfgp_handle* h, int B) {
    if (B <= h->bsets && h->bsets_cap == h->cap) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int held = h->bsets_cap == h->cap ? h->bsets : 0;
    const int want = std::max(B, held);
    const size_t limit = batch_mem_cap();
    if (limit && batch_bytes(h, want) > limit)
        return fail(h, MFGP_ERR_OOM, "mfgp_eval_batch: " + std::to_string(want) + " matrix sets (" + std::to_string(batch_bytes(h, want) >> 20) +
                                         " MiB) exceed MFGP_BATCH_MEM_CAP (" + std::to_string(limit >> 20) + " MiB)");
    free_batch(h);
    int rc = alloc_batch(h, want);
    if (rc == MFGP_ERR_OOM && held > 0) {
        const std::string why = h->err;
        if (alloc_batch(h, held) != 0) free_batch(h);      // (what was just released fits again unless somebody else took it meanwhile)
        h->err = why;
    }
    return rc;
}

// the batch's plan: the handle's plan with the 128-tile threshold divided by the number of sets a launch carries (classes
// 1 / 2 / 3-4 / 5-8 / 9-16, so that a fit's rounds of 4 and then 3 evaluations share one plan)
static int ensure_batch_plan(mfgp_handle* h, int B) {
    const int div = B >= 9 ? 9 : (B >= 5 ? 5 : (B >= 3 ? 3 : B));
    if (h->plb_div == div && h->plb.nblk == h->nblk && h->plb.ld == h->Np) return 0;
    build_plan(h->plb, h->nblk, h->Np, (int64_t)h->cap * h->cap, h->pl.opts, div);
    while ((int)h->evpool.size() < h->plb.n_events) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        h->evpool.push_back(e);
    }
    const size_t need = h->plb.tasks.size();
    if (need > h->tasks_b_cap) {
        if (h->dtasks_b) HIPCHK(h, hipFree(h->dtasks_b));
        h->tasks_b_cap = need + need / 2 + 1024;
        HIPCHK(h, hipMalloc(&h->dtasks_b, h->tasks_b_cap * sizeof(GemmTask)));
    }
    HIPCHK(h, hipMemcpyAsync(h->dtasks_b, h->plb.tasks.data(), need * sizeof(GemmTask), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->plb_div = div;
    return 0;
}

// B independent objective(+gradient) evaluations on the SAME data and kernel structure at B hyper-parameter points, as ONE
// pass of the plan: every launch of the sweep carries the B matrix sets side by side (leaf: B workgroups; tile GEMMs: the
// task list x B), so the serial Cholesky chain -- which leaves most of the GPU idle at N <= 4096 -- is paid once for all
// of them, and the bulk launches are B times fuller.  Each evaluation's arithmetic is the single evaluation's, tile for
// tile: results are bitwise those of mfgp_eval at the same point.
int32_t mfgp_eval_batch(mfgp_handle* h, int32_t B, const double* thetas, const double* noises, const double* jitters,
                        int32_t want_grad, double* nlml, double* grads, int32_t* status) {
    int rc = check_ready(h, "mfgp_eval_batch");
    if (rc) return rc;
    if (!thetas || !noises || !jitters || !nlml || !status || (want_grad && !grads))
        return fail(h, -1, "mfgp_eval_batch: NULL argument");
    if (B < 1 || B > MFGP_MAX_BATCH_SETS) return fail(h, -1, "mfgp_eval_batch: 1 <= B <= 16");
    HIPCHK(h, hipSetDevice(h->device));
    const int np = h->spec.np;
    std::vector<KernSpecDev> specs((size_t)B, h->spec);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < np; ++i) {
            const double v = thetas[(size_t)b * np + i];
            if (!(v > 0.0) || !isfinite(v)) return fail(h, -1, "mfgp_eval_batch: parameters must be positive and finite");
            specs[b].theta[i] = v;
        }
        if (!(noises[b] >= 0.0) || !(jitters[b] >= 0.0)) return fail(h, -1, "mfgp_eval_batch: noise and jitter must be >= 0");
        specs[b].theta[np] = noises[b];
        specs[b].theta[np + 1] = jitters[b];
        specs[b].D = h->D;
    }
    rc = ensure_batch(h, B);
    if (rc) return rc;
    rc = ensure_batch_plan(h, B);
    if (rc) return rc;
    hipStream_t s = h->stream;
    const size_t cap = (size_t)h->cap, set = 4 * cap * cap;
    const int Np = (int)h->Np;
    const bool grad = want_grad != 0;
    constexpr int BRES = mfgp_handle::BRES;
    h->launches = 0;
    for (int b = 0; b < B; ++b) {   // (the previous call synchronised: the pinned blocks are the host's to write)
        double* r = h->bhres + (size_t)b * BRES;
        *reinterpret_cast<int*>(r + 30) = 0;
        for (int i = 0; i < np; ++i) r[128 + i] = specs[b].theta[i];        // the gradient's finishing kernel divides by them
    }
    ++h->epoch;
    // from the first launch on, an error exit waits for what is already in flight on both streams: the next call rewrites the mapped
    // result blocks and the per-set parameter words from the host
    auto bail = [&](int code) {
        (void)hipStreamSynchronize(h->stream);
        if (h->stream2) (void)hipStreamSynchronize(h->stream2);
        return code;
    };
#define HIPCHK_BAIL(call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return bail(-2);                                                                    \
        }                                                                                       \
    } while (0)
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[0], s));
    launch_kbuild_tri_batch(s, specs.data(), B, h->dX, (int)h->N, Np, h->bslab, Np, (long long)set);
    h->launches += 1;
    if (h->stage_timing) HIPCHK_BAIL(hipEventRecord(h->ev[1], s));
    const bool stream_kinv = grad && h->plb.kinv_streamed;
    for (const Step& st : h->plb.steps)
        if (run_step(h, st, stream_kinv, B) != 0) return bail(-1);
    if (h->stage_timing) HIPCHK_BAIL(hipEventRecord(h->ev[2], s));
    double* const S0 = h->bslab + (size_t)BUF_S * cap * cap;
    launch_rowdot(s, S0, Np, h->dY, h->bz, Np, Np, 0, B, (long long)set, 0, (long long)cap);                   // z = X y
    launch_alpha_finish(s, S0, Np, h->bz, h->balpha, Np, h->blogdet, h->nblk, h->bdres, B, (long long)set, (long long)cap,
                        (int)(cap / NB), BRES);
    h->launches += 2;
    if (grad) {
        if (!stream_kinv && run_step(h, h->plb.kinv_step, true, B) != 0) return bail(-1);
        const size_t npart = (size_t)grad_num_partials((int)cap) * (MFGP_MAX_THETA + 1);
        launch_grad_batch(s, specs.data(), B, h->dX, h->bslab, (long long)set, Np, h->balpha, (long long)cap, (int)h->N, Np,
                          h->bpart, (long long)npart, h->bdres + 64, BRES, h->bdres + 128, BRES);
        h->launches += 2;
    }
    if (h->timing) HIPCHK_BAIL(hipEventRecord(h->ev[5], s));
    HIPCHK_BAIL(hipGetLastError());
    HIPCHK_BAIL(hipStreamSynchronize(s));
    HIPCHK_BAIL(hipGetLastError());
#undef HIPCHK_BAIL
    // accounting: B evaluations, timed as one pass
    const double npd = (double)h->Np;
    mfgp_timings& t = h->tm;
    memset(&t, 0, sizeof t);
    if (h->stage_timing) {
        t.kbuild_ms = ev_ms(h->ev[0], h->ev[1]);
        t.cholinv_ms = ev_ms(h->ev[1], h->ev[2]);
    }
    if (h->timing) t.total_ms = ev_ms(h->ev[0], h->ev[5]);
    t.timed = h->stage_timing ? 3 : (h->timing ? 1 : 0);
    t.kbuild_bytes = B * 4.0 * npd * (npd + 64.0);
    t.kinv_flops = (grad && !stream_kinv) ? B * npd * npd * npd / 3.0 : 0.0;   // (a streamed plan counts K^-1 inside the sweep)
    t.cholinv_flops = B * (stream_kinv ? 3.0 : 2.0) * npd * npd * npd / 3.0;
    t.n_launches = h->launches;
    if (h->timing) h->cum.timed_evals += B;
    h->cum.evals += B;
    h->cum.grad_evals += grad ? B : 0;
    h->cum.kbuild_ms += t.kbuild_ms;
    h->cum.cholinv_ms += t.cholinv_ms;
    h->cum.total_ms += t.total_ms;
    h->cum.kbuild_bytes += t.kbuild_bytes;
    h->cum.kinv_flops += t.kinv_flops;
    h->cum.cholinv_flops += t.cholinv_flops;
    for (int b = 0; b < B; ++b) {
        const double* r = h->bhres + (size_t)b * BRES;
        const int info = *reinterpret_cast<const int*>(r + 30);
        status[b] = info;
        nlml[b] = 0.5 * ((double)h->N * 1.8378770664093453 + r[1] + r[0]);
        if (grad)
            for (int i = 0; i < np + 1; ++i) grads[(size_t)b * (np + 1) + i] = r[64 + i];
    }
    return 0;
}

// what the host layer sizes a batch from (engine.LockstepLane / AbstractMFGP._ard_lockstep): free / total bytes of the handle's device,
// the bytes `sets` matrix sets of a batch would take on this handle at its current capacity, and how many it holds already
int32_t mfgp_mem_info(mfgp_handle* h, int64_t* free_bytes, int64_t* total_bytes) {
    if (!h || !free_bytes || !total_bytes) return fail(h, -1, "mfgp_mem_info: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    size_t f = 0, t = 0;
    HIPCHK(h, hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return 0;
}

int32_t mfgp_batch_mem(mfgp_handle* h, int32_t sets, int64_t* bytes, int64_t* cap_bytes, int32_t* sets_held) {
    int rc = check_ready(h, "mfgp_batch_mem");
    if (rc) return rc;
    if (sets < 0 || !bytes || !cap_bytes || !sets_held) return fail(h, -1, "mfgp_batch_mem: bad argument");
    *bytes = (int64_t)batch_bytes(h, sets);
    *cap_bytes = (int64_t)batch_mem_cap();
    *sets_held = h->bsets_cap == h->cap ? h->bsets : 0;
    return 0;
}

}  // extern "C"
