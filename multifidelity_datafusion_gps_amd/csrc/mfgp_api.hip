// mfgp_api.hip -- host side of libmfgp_hip.so: the C-ABI of include/mfgp.h, device-memory
// ownership, and the planner that turns "factorise / invert / predict" into lists of tile-GEMM
// tasks (gemm_f64.hip) and leaf launches (leaf_f64.hip).
//
// Algorithm of one evaluation (i.e. GPy ExactGaussianInference.inference +
// kern.update_gradients_full, restated MI355X-first):
//   A  <- lower(K(theta)) + (noise+jitter) I                     [1 launch, HBM bound]
//   (L, X=L^-1) <- cholinv(A)   recursive on 128-blocks:          [leaf + MFMA tile GEMMs]
//        cholinv(A11); L21 = A21 X11^T; A22 -= L21 L21^T; cholinv(A22); X21 = -X22 (L21 X11)
//      the inverse is kept "mirrored" in S (S = X + X^T - diag) so that every product in the
//      recursion, K^-1 = X^T X and the predictive V = Kx X^T are K-contiguous "NT" tile GEMMs.
//   z = X y ; alpha = X^T z ; logdet = 2 sum log diag(L) ; nlml = .5 (N log 2pi + logdet + z.z)
//   Kinv <- lower(X^T X)                                          [1 launch, MFMA bound, N^3/3 flops]
//   grad <- -0.5 sum (alpha alpha^T - Kinv) o dK/dtheta           [fused tile reduction]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "mfgp_internal.h"

using namespace mfgp;

thread_local std::string mfgp::g_err;

static int upload_tasks(mfgp_handle* h) {
    const size_t need = h->pl.tasks.size();
    if (need > h->tasks_cap) {
        if (h->dtasks) HIPCHK(h, hipFree(h->dtasks));
        h->tasks_cap = need + need / 2 + 1024;
        HIPCHK(h, hipMalloc(&h->dtasks, h->tasks_cap * sizeof(GemmTask)));
    }
    HIPCHK(h, hipMemcpyAsync(h->dtasks, h->pl.tasks.data(), need * sizeof(GemmTask), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// One step of a plan.  nbatch > 0: over the handle's batch sets (mfgp_eval_batch) instead of its own slab -- the same launch
// with one more grid dimension.  -> 0, or -1 when the planner asked for a kernel that does not exist (h->err says which).
static int run_step(mfgp_handle* h, const Step& s, bool want_grad = true, int nbatch = 0, const GemmTask* tasks = nullptr) {
    const GemmTask* const dtasks = tasks ? tasks : (nbatch > 0 ? h->dtasks_b : h->dtasks);
    const bool batched = nbatch > 0;
    hipStream_t st = (s.strm == 1 && h->stream2) ? h->stream2 : h->stream;
    const long long bstride = batched ? 4LL * h->cap * h->cap : 0;
    auto base = [&](int k) { return batched ? h->bslab + (size_t)k * h->cap * h->cap : h->buf[k]; };
    if (s.wait_ev > 0) (void)hipStreamWaitEvent(st, h->evpool[s.wait_ev - 1], 0);
    if (s.kind == 0) {
        if (batched)
            launch_leaf(st, base(BUF_A), base(BUF_L), base(BUF_S), (int)h->Np, s.blk, h->blogdet,
                        reinterpret_cast<int*>(h->bdres + 30), nbatch, bstride, (int)(h->cap / NB), 2 * mfgp_handle::BRES,
                        h->dflag + 1, h->epoch);
        else
            launch_leaf(st, h->buf[BUF_A], h->buf[BUF_L], h->buf[BUF_S], (int)h->Np, s.blk, h->dlogdet, h->dinfo, 1, 0, 0, 0,
                        h->dflag, h->epoch);
        h->launches++;
    } else if (s.kind == 1) {
        // a launch that carries a chunk of the K^-1 accumulation has a second task list for gradient evaluations
        const bool g = want_grad && s.gcount > 0;
        const int n = g ? s.gcount : s.count;
        if (n > 0) {
            if (launch_gemm(st, s.tile, dtasks + (g ? s.gfirst : s.first), n, base(s.a), base(s.b), base(s.c),
                            s.c2 >= 0 ? base(s.c2) : nullptr, (int)h->Np, s.role, batched ? nbatch : 1, bstride,
                            batched ? h->dflag + 1 : h->dflag, h->epoch) != 0)
                return fail(h, -1, "planner bug: no tile-GEMM kernel for tile " + std::to_string(s.tile) + ", role " +
                                       std::to_string(s.role));
            h->launches++;
        }
    } else if (s.kind == STEP_COMM_DIAG || s.kind == STEP_COMM_PANEL) {
        // exchange steps of a distributed Cholesky (this rank's plan of a sharded evaluation: never batched); on the main stream,
        // where the handle's collectives run.  Without a communicator (mfgp_dbg paths) the data simply stays where it is.
        const int size = h->pls.shard.size, rank = h->pls.shard.rank, c = s.blk, Np = (int)h->Np;
        if (s.kind == STEP_COMM_DIAG) {
            const int root = shard_owner(c, size);
            if (root == rank) launch_dist_diag_copy(st, h->buf[BUF_L], h->buf[BUF_S], Np, c, h->ddist, h->dlogdet, h->dinfo, false);   // (pack)
            if (int rc = comm_bcast_words(h, h->ddist, 2 * (size_t)NB * NB + 2, root, st)) return rc;
            if (root != rank)
                launch_dist_diag_copy(st, h->buf[BUF_L], h->buf[BUF_S], Np, c, h->ddist, h->dlogdet, h->dinfo, true, h->dflag, h->epoch);
        } else {
            std::vector<int> cnt((size_t)size, 0);
            for (int i = c + 1; i < h->nblk; ++i) cnt[(size_t)shard_owner(i, size)]++;
            const long long chunk = (long long)*std::max_element(cnt.begin(), cnt.end()) * NB * NB;
            launch_dist_panel_copy(st, h->buf[BUF_L], Np, h->nblk, c, h->ddist, chunk, rank, size, false);
            if (int rc = comm_allgather_chunks(h, h->ddist, (size_t)chunk, st)) return rc;
            launch_dist_panel_copy(st, h->buf[BUF_L], Np, h->nblk, c, h->ddist, chunk, rank, size, true);
        }
        h->launches += 2;
    }   // kind 2: join -- the wait above is all there is
    if (s.rec_ev > 0) (void)hipEventRecord(h->evpool[s.rec_ev - 1], st);
    if (s.rec_ev_final > 0) (void)hipEventRecord(h->evpool[s.rec_ev_final - 1], st);
    return 0;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* mfgp_last_error(mfgp_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

static int create_body(mfgp_handle* h, int device_id);

int32_t mfgp_create(int32_t device_id, mfgp_handle** out) {
    if (!out) return fail(nullptr, -1, "mfgp_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, -3, std::string("mfgp_create: no HIP device available (") +
                                     (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + ")");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, -1, "mfgp_create: bad device id");
    mfgp_handle* h = new mfgp_handle();
    h->device = device_id;
    const int rc = create_body(h, device_id);
    if (rc != 0) {   // report through the global slot (the caller never sees this handle) and release what was made
        g_err = "mfgp_create: " + h->err;
        mfgp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

static int create_body(mfgp_handle* h, int device_id) {
    HIPCHK(h, hipSetDevice(device_id));
    int prio_lo = 0, prio_hi = 0;  // lo = least urgent (numerically greatest), hi = most urgent
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // the main stream carries the serial chain (leaf -> panel -> narrow update): most urgent, so that its
    // workgroups take the first CU a bulk-update workgroup vacates
    HIPCHK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
    // bulk-update stream: least urgent.  (Measured and retired -- tools/gemm_lab/RETIRED.md: other priority pairs move nothing;
    // a CU mask that keeps CUs out of this stream's reach for the leaf gives the leaf its CU and costs the bulk stream as much.)
    HIPCHK(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo));
    // timing events: no system-scope fence at the record either (more precise stamps, and cheaper: see build_plans)
    for (auto& ev : h->ev) HIPCHK(h, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
    // the scalar results (quadratic form, log-det, gradient, pivot status) are written by the kernels straight into
    // pinned, device-mapped host memory: no copy kernel at the end of a call and no fill kernel for the status at its
    // start (each costs ~5 us plus a gap; an evaluation at N <= 128 is ~75 us of GPU time in all)
    HIPCHK(h, hipHostMalloc(&h->hres, 128 * sizeof(double), hipHostMallocMapped));   // [0,1] scalars, [30] status, [48..51] append, [64..] gradient
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->dres), h->hres, 0));
    memset(h->hres, 0, 128 * sizeof(double));
    HIPCHK(h, hipHostMalloc(&h->hio, (size_t)(mfgp_handle::IO_IN + 2 * mfgp_handle::IO_OUT) * sizeof(double), hipHostMallocMapped));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->dio), h->hio, 0));
    h->dinfo = reinterpret_cast<int*>(h->dres + 30);   // the pivot status lives beside the results
    h->hinfo = reinterpret_cast<int*>(h->hres + 30);
    // ... and, for the kernels, a failure mark per matrix set in DEVICE memory (the status word is a PCIe read away): a mark holds
    // the number of the evaluation that failed, so nothing has to clear it between evaluations
    HIPCHK(h, hipMalloc(&h->dflag, (size_t)(1 + MFGP_BATCH_MAX) * sizeof(int)));
    HIPCHK(h, hipMemset(h->dflag, 0, (size_t)(1 + MFGP_BATCH_MAX) * sizeof(int)));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device_id));
    char tmp[256];
    snprintf(tmp, sizeof tmp, "mfgp_hip %s %s CUs=%d", prop.gcnArchName, prop.name, prop.multiProcessorCount);
    h->info_str = tmp;
    return 0;
}

const char* mfgp_device_info(mfgp_handle* h) { return h ? h->info_str.c_str() : ""; }

#ifndef MFGP_SRC_HASH
#define MFGP_SRC_HASH "unknown"
#endif
const char* mfgp_build_id(void) { return MFGP_SRC_HASH; }

static void free_batch(mfgp_handle* h) {
    for (double** p : {&h->bslab, &h->bz, &h->balpha, &h->blogdet, &h->bpart}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    if (h->bhres) hipHostFree(h->bhres);
    h->bhres = h->bdres = nullptr;
    h->bsets = 0;
    h->bsets_cap = 0;
    h->plb_div = 0;      // (the batch plan's offsets follow the slab's capacity too)
}

static void free_mats(mfgp_handle* h) {
    free_batch(h);       // (its layout follows the slab's capacity)
    if (h->slab) hipFree(h->slab);
    h->slab = nullptr;
    for (auto& b : h->buf) b = nullptr;
    for (double** p : {&h->dX, &h->dY, &h->dz, &h->dalpha, &h->dlogdet, &h->dpart, &h->dvec, &h->dvec2}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    h->cap = 0;
}

int32_t mfgp_destroy(mfgp_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    free_mats(h);
    if (h->dXs) hipFree(h->dXs);
    if (h->dXc) hipFree(h->dXc);
    if (h->dm) hipFree(h->dm);
    if (h->doffs) hipFree(h->doffs);
    if (h->dAug) hipFree(h->dAug);
    if (h->dtasks) hipFree(h->dtasks);
    if (h->dtasks_b) hipFree(h->dtasks_b);
    if (h->dtasks_s) hipFree(h->dtasks_s);
    if (h->dshard_off) hipFree(h->dshard_off);
    if (h->drow_off) hipFree(h->drow_off);
    if (h->ddist) hipFree(h->ddist);
    if (h->dctl) hipFree(h->dctl);
    if (h->dflag) hipFree(h->dflag);
    if (h->hctl) hipHostFree(h->hctl);
    comm_release(h);
    if (h->dstage) hipFree(h->dstage);
    if (h->hres) hipHostFree(h->hres);
    if (h->hio) hipHostFree(h->hio);
    for (auto& ev : h->ev) if (ev) hipEventDestroy(ev);
    for (auto& ev : h->evpool) hipEventDestroy(ev);
    if (h->stream2) hipStreamDestroy(h->stream2);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

static int build_plans(mfgp_handle* h) {
    build_plan(h->pl, h->nblk, h->Np, (int64_t)h->cap * h->cap, plan_opts_from_env());
    h->plb_div = 0;      // the batch plan follows: rebuilt, under the same switches, when the next batch arrives
    h->pls.nblk = 0;     // ... and so does the plan of a sharded evaluation
    while ((int)h->evpool.size() < h->pl.n_events) {
        hipEvent_t e;
        // the plan's events order kernels of ONE device across the handle's two streams: no system-scope fence (cache
        // write-back / invalidate for the host and for other devices) when they are recorded -- ~2 % of an evaluation at
        // N = 3072 .. 6144; results reach the host behind hipStreamSynchronize
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        h->evpool.push_back(e);
    }
    return upload_tasks(h);
}

int32_t mfgp_set_data(mfgp_handle* h, const double* X, int64_t N, int32_t D, const double* Y) {
    if (!h || !X || !Y) return fail(h, -1, "mfgp_set_data: NULL argument");
    if (N < 1 || D < 1 || D > 32) return fail(h, -1, "mfgp_set_data: need N >= 1 and 1 <= D <= 32 (LDS staging of the covariance kernels)");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t Np = (N + NB - 1) / NB * NB;
    bool realloc_ = false;
    if (Np > h->cap || D != h->D) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // growth is geometric (x1.25, rounded to the block) once a handle has to grow: an adaptation run that adds one
        // row per step (src/abstractMFGP.py:320,354) would otherwise free and re-allocate the four Np^2 buffers at
        // every 128-row boundary (N_hf 512 -> 8192: 60 times; now 13)
        int64_t cap = std::max(Np, h->cap);
        if (h->cap > 0 && Np > h->cap) cap = std::max(Np, (h->cap + h->cap / 4 + NB - 1) / NB * NB);
        free_mats(h);
        // ONE slab for the four Np^2 matrices A | L | S | W: the sweep plan addresses all of them from the slab base, so a
        // single launch can mix tasks whose operands live in different matrices
        HIPCHK(h, hipMalloc(&h->slab, (size_t)4 * cap * cap * sizeof(double)));
        for (int k = 0; k < 4; ++k) h->buf[k] = h->slab + (size_t)k * cap * cap;
        realloc_ = true;
        HIPCHK(h, hipMalloc(&h->dX, (size_t)cap * D * sizeof(double)));
        for (double** p : {&h->dY, &h->dz, &h->dalpha, &h->dvec, &h->dvec2})
            HIPCHK(h, hipMalloc(p, (size_t)cap * sizeof(double)));
        HIPCHK(h, hipMalloc(&h->dlogdet, (size_t)(cap / NB) * sizeof(double)));
        HIPCHK(h, hipMalloc(&h->dpart, (size_t)grad_num_partials((int)cap) * (MFGP_MAX_THETA + 1) * sizeof(double)));
        h->cap = cap;
    }
    const bool replan = (Np != h->Np) || realloc_;   // (the plan's offsets depend on the slab stride = cap^2)
    h->N = N; h->Np = Np; h->D = D; h->nblk = (int)(Np / NB);
    h->stage_timing = Np >= 4096;   // four more event records per evaluation: ~20 us, 5 % of an evaluation at N = 1024
    if (const char* e = getenv("MFGP_STAGE_TIMING")) h->stage_timing = atoi(e) != 0;
    // no timing events at all below that size unless asked for (MFGP_TIMING=1: start / end stamps only): an optimiser never
    // reads them, and the two records of an evaluation (three of a predict) are ~7 us of the ~56 (~80) us a small one takes
    h->timing = h->stage_timing;
    if (const char* e = getenv("MFGP_TIMING")) h->timing = h->timing || atoi(e) != 0;
    HIPCHK(h, hipMemsetAsync(h->dX, 0, (size_t)Np * D * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->dY, 0, (size_t)Np * sizeof(double), h->stream));
    HIPCHK(h, hipMemcpyAsync(h->dX, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->dY, Y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_data = true;
    h->factorized = h->kinv_valid = h->grad_valid = false;
    h->spec.D = D;
    if (replan) return build_plans(h);
    return 0;
}

int32_t mfgp_num_params(const mfgp_kern_part* parts, int32_t n_parts) {
    if (!parts || n_parts < 1 || n_parts > MFGP_MAX_PARTS) return -1;
    int np = 0;
    for (int f = 0; f < n_parts; ++f) {
        const int base = parts[f].type & ~MFGP_KERN_ARD;
        if (base < 0 || base > MFGP_KERN_MATERN52 || parts[f].col_end <= parts[f].col_begin) return -1;
        np += 1 + ((parts[f].type & MFGP_KERN_ARD) ? parts[f].col_end - parts[f].col_begin : 1);
    }
    return np;
}

int32_t mfgp_set_kernel(mfgp_handle* h, const mfgp_kern_part* parts, int32_t n_parts) {
    if (!h || !parts) return fail(h, -1, "mfgp_set_kernel: NULL argument");
    if (n_parts < 1 || n_parts > MFGP_MAX_PARTS) return fail(h, -1, "mfgp_set_kernel: 1..6 parts supported");
    KernSpecDev sp{};
    sp.nf = n_parts;
    sp.D = h->D;
    sp.ng = 0;
    sp.np = 0;
    for (int f = 0; f < n_parts; ++f) {
        const mfgp_kern_part& p = parts[f];
        const int base = p.type & ~MFGP_KERN_ARD;
        if (p.type < 0 || base > MFGP_KERN_MATERN52) return fail(h, -1, "mfgp_set_kernel: unknown kernel type");
        if (p.col_begin < 0 || p.col_end <= p.col_begin || p.col_end > 32)
            return fail(h, -1, "mfgp_set_kernel: bad column range");
        if (f > 0 && p.term < parts[f - 1].term) return fail(h, -1, "mfgp_set_kernel: term ids must be ascending");
        sp.type[f] = base; sp.c0[f] = p.col_begin; sp.c1[f] = p.col_end; sp.term[f] = p.term;
        sp.toff[f] = sp.np;
        sp.nl[f] = (p.type & MFGP_KERN_ARD) ? p.col_end - p.col_begin : 1;
        sp.np += 1 + sp.nl[f];
        if (sp.np > MFGP_MAX_THETA) return fail(h, -1, "mfgp_set_kernel: more than MFGP_MAX_THETA kernel parameters");
        int g = -1;
        for (int k = 0; k < sp.ng; ++k)
            if (sp.gc0[k] == p.col_begin && sp.gc1[k] == p.col_end) g = k;
        if (g < 0) {
            if (sp.ng == MFGP_MAX_GROUPS) return fail(h, -1, "mfgp_set_kernel: at most 3 distinct column ranges");
            g = sp.ng++;
            sp.gc0[g] = p.col_begin; sp.gc1[g] = p.col_end;
        }
        sp.gidx[f] = g;
    }
    h->spec = sp;
    h->have_kernel = true;
    h->factorized = h->kinv_valid = h->grad_valid = false;
    return 0;
}

static int check_ready(mfgp_handle* h, const char* who) {
    if (!h) return fail(nullptr, -1, std::string(who) + ": NULL handle");
    if (!h->have_data) return fail(h, -1, std::string(who) + ": mfgp_set_data not called");
    if (!h->have_kernel) return fail(h, -1, std::string(who) + ": mfgp_set_kernel not called");
    for (int f = 0; f < h->spec.nf; ++f)
        if (h->spec.c1[f] > h->D) return fail(h, -1, std::string(who) + ": kernel column range exceeds D");
    h->spec.D = h->D;
    return 0;
}

// enqueue K-build + cholinv + solve (+ K^-1 + gradient); no host sync
static int set_params(mfgp_handle* h, const double* theta, double noise, double jitter) {
    // the hyper-parameters ride in the kernel arguments (KernSpecDev::theta): nothing to upload
    const int np = h->spec.np;
    for (int i = 0; i < np; ++i) {
        if (!(theta[i] > 0.0) || !isfinite(theta[i])) return fail(h, -1, "parameters must be positive and finite");
        h->theta[i] = theta[i];
        h->spec.theta[i] = theta[i];
    }
    if (!(noise >= 0.0) || !(jitter >= 0.0)) return fail(h, -1, "noise and jitter must be >= 0");
    h->noise = noise; h->jitter = jitter;
    h->params_set = true;
    h->spec.theta[np] = noise;
    h->spec.theta[np + 1] = jitter;
    return 0;
}

static int enqueue_eval(mfgp_handle* h, const double* theta, double noise, double jitter, bool want_grad,
                        bool prebuilt = false) {
    hipStream_t s = h->stream;
    h->launches = 0;
    if (!prebuilt) {
        const int rc_ = set_params(h, theta, noise, jitter);
        if (rc_) return rc_;
    }
    *h->hinfo = 0;   // (the previous call synchronised the stream)
    ++h->epoch;
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[0], s));
    if (!prebuilt) {
        launch_kbuild_tri(s, h->spec, h->dX, (int)h->N, (int)h->Np, h->buf[BUF_A], (int)h->Np);
        h->launches++;
    }
    const bool stages = h->stage_timing;   // an event record costs 6-8 us of stream time: per-stage stamps only where that is noise
    if (stages) HIPCHK(h, hipEventRecord(h->ev[1], s));
    const bool stream_kinv = want_grad && h->pl.kinv_streamed;
    for (const Step& st : h->pl.steps)
        if (run_step(h, st, stream_kinv) != 0) return -1;
    if (stages) HIPCHK(h, hipEventRecord(h->ev[2], s));
    launch_rowdot(s, h->buf[BUF_S], (int)h->Np, h->dY, h->dz, (int)h->Np, (int)h->Np, 0);       // z = X y
    launch_alpha_finish(s, h->buf[BUF_S], (int)h->Np, h->dz, h->dalpha, (int)h->Np, h->dlogdet, h->nblk, h->dres);   // alpha = X^T z; z^T z, log-det
    h->launches += 2;
    if (stages || (!want_grad && h->timing)) HIPCHK(h, hipEventRecord(h->ev[3], s));
    if (want_grad) {
        if (!stream_kinv && run_step(h, h->pl.kinv_step) != 0) return -1;   // (streamed plans have accumulated K^-1 behind the chain already)
        if (stages) HIPCHK(h, hipEventRecord(h->ev[4], s));
        launch_grad(s, h->spec, h->dX, h->buf[BUF_A], (int)h->Np, h->dalpha, (int)h->N, (int)h->Np,
                    h->dpart, h->dres + 64);
        h->launches += 2;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[5], s));
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

static int finish_eval(mfgp_handle* h, bool want_grad) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    mfgp_timings& t = h->tm;
    memset(&t, 0, sizeof t);
    if (h->stage_timing) {
        t.kbuild_ms = ev_ms(h->ev[0], h->ev[1]);
        t.cholinv_ms = ev_ms(h->ev[1], h->ev[2]);
        t.solve_ms = ev_ms(h->ev[2], h->ev[3]);
        if (want_grad) {
            t.kinv_ms = ev_ms(h->ev[3], h->ev[4]);
            t.grad_ms = ev_ms(h->ev[4], h->ev[5]);
        }
    }
    if (h->timing) t.total_ms = ev_ms(h->ev[0], h->ev[want_grad ? 5 : 3]);
    t.timed = h->stage_timing ? 3 : (h->timing ? 1 : 0);
    if (h->timing) h->cum.timed_evals += 1;
    const double np = (double)h->Np;
    t.kbuild_bytes = 4.0 * np * (np + 64.0);
    // a streamed plan accumulates K^-1 inside the sweep (between the cholinv stamps): its N^3/3 flops are counted there
    const bool streamed = want_grad && h->pl.kinv_streamed;
    t.kinv_flops = streamed ? 0.0 : np * np * np / 3.0;
    t.cholinv_flops = (streamed ? 3.0 : 2.0) * np * np * np / 3.0;
    t.n_launches = h->launches;
    h->cum.evals += 1;
    h->cum.grad_evals += want_grad ? 1 : 0;
    h->cum.kbuild_ms += t.kbuild_ms;
    h->cum.cholinv_ms += t.cholinv_ms;
    h->cum.solve_ms += t.solve_ms;
    h->cum.kinv_ms += t.kinv_ms;
    h->cum.grad_ms += t.grad_ms;
    h->cum.total_ms += t.total_ms;
    h->cum.kbuild_bytes += t.kbuild_bytes;
    h->cum.kinv_flops += want_grad ? t.kinv_flops : 0.0;
    h->cum.cholinv_flops += t.cholinv_flops;
    h->quad = h->hres[0];
    h->logdet = h->hres[1];
    h->kinv_valid = want_grad;
    h->grad_valid = want_grad;
    if (want_grad)
        for (int i = 0; i < h->spec.np + 1; ++i) h->grad[i] = h->hres[64 + i];
    const int info = *h->hinfo;
    if (info != 0) {
        h->factorized = false;
        h->kinv_valid = h->grad_valid = false;
        h->err = "Cholesky failed: non-positive pivot at index " + std::to_string(info);
        return info;
    }
    h->factorized = true;
    return 0;
}

int32_t mfgp_eval(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad,
                  double* nlml, double* grad) {
    int rc = check_ready(h, "mfgp_eval");
    if (rc) return rc;
    if (!theta) return fail(h, -1, "mfgp_eval: theta is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    rc = enqueue_eval(h, theta, noise, jitter, want_grad != 0);
    if (rc) return rc;
    rc = finish_eval(h, want_grad != 0);
    if (rc) return rc;
    if (nlml) *nlml = 0.5 * ((double)h->N * 1.8378770664093453 + h->logdet + h->quad);
    if (want_grad && grad)
        for (int i = 0; i < h->spec.np + 1; ++i) grad[i] = h->grad[i];
    return 0;
}

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
        const size_t need = std::max((size_t)(h->nblk / size + 2) * size * NB * NB, 2 * (size_t)NB * NB + 2);
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
    const int rc = finish_eval(h, want_grad);
    h->pl.kinv_streamed = streamed_flag;
    h->kinv_valid = false;                                // (this rank holds only its own rows of K^-1)
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

static double prior_variance(const mfgp_handle* h);
static int ensure_xs(mfgp_handle* h, int rows_p);

// rank-1 append at fixed hyper-parameters (the adaptation loop of src/abstractMFGP.py:320,354 grows the
// training set by one row per step).  O(N^2): one covariance row, two triangular mat-vecs with the stored inverse
// factor (l = X k, w = X^T l: 8 Np^2 bytes in all), one finishing kernel that also brings alpha up to date in O(N).
// Returns 0 = appended; 1 = no padding slot left (N is a multiple of 128: the caller re-uploads and refactorises);
// >1 = not positive definite with the new row.
int32_t mfgp_append_row(mfgp_handle* h, const double* x_new, double y_new) {
    int rc = check_ready(h, "mfgp_append_row");
    if (rc) return rc;
    if (!x_new) return fail(h, -1, "mfgp_append_row: x_new is NULL");
    if (!h->factorized) return fail(h, -1, "mfgp_append_row: no valid factorisation");
    if (h->N >= h->Np) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int n = (int)h->N, D = h->D;
    const int64_t Np = h->Np;
    // stage the new row as a 64-row zero-padded panel operand; X[n] / Y[n] are written by the finishing kernel, and only
    // if the extension is positive definite (a rejected append leaves the handle's data untouched)
    rc = ensure_xs(h, 128);
    if (rc) return rc;
    memset(h->hio, 0, (size_t)64 * D * sizeof(double));      // (pinned staging: one asynchronous copy, see mfgp_predict)
    memcpy(h->hio, x_new, (size_t)D * sizeof(double));
    HIPCHK(h, hipMemcpyAsync(h->dXs, h->hio, (size_t)64 * D * sizeof(double), hipMemcpyHostToDevice, s));
    // k = K(x_new, X[0:n]) -> row 0 of W (0 in the padded columns) ; l = X k ; w = X^T l.  The first pass runs to the end of
    // row n's 128-block: rows n .. of S are still identity rows, so l[n ..] = k[n ..] = 0 -- the second pass reads l in whole
    // 128-column chunks (masked by its column range, but the operand has to be finite)
    launch_kbuild_panel(s, h->spec, h->dXs, 64, h->dX, n, (int)Np, h->buf[BUF_W], (int)Np);
    launch_rowdot(s, h->buf[BUF_S], (int)Np, h->buf[BUF_W], h->dvec, ((n >> 7) + 1) << 7, (int)Np, 0);
    launch_rowdot(s, h->buf[BUF_S], (int)Np, h->dvec, h->dvec2, n, n, 1);
    const double kdiag = prior_variance(h) + h->noise + h->jitter;
    launch_append_finish(s, h->buf[BUF_L], h->buf[BUF_S], (int)Np, n, h->dvec, h->dvec2, h->dz, h->dalpha, kdiag, y_new,
                         h->dres + 48, h->dX, h->dXs, D, h->dY);
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (h->hres[51] != 0.0) {
        h->err = "mfgp_append_row: the extended matrix is not positive definite";
        return n + 2;
    }
    h->N = n + 1;
    h->logdet += 2.0 * log(h->hres[48]);
    h->quad += h->hres[49] * h->hres[49];
    h->kinv_valid = h->grad_valid = false;
    return 0;
}

int32_t mfgp_factorize(mfgp_handle* h, const double* theta, double noise, double jitter) {
    return mfgp_eval(h, theta, noise, jitter, 0, nullptr, nullptr);
}

int32_t mfgp_nlml(mfgp_handle* h, double* value) {
    if (!h || !value) return fail(h, -1, "mfgp_nlml: NULL argument");
    if (!h->factorized) return fail(h, -1, "mfgp_nlml: no valid factorisation");
    *value = 0.5 * ((double)h->N * 1.8378770664093453 + h->logdet + h->quad);
    return 0;
}

int32_t mfgp_nlml_grad(mfgp_handle* h, double* grad) {
    if (!h || !grad) return fail(h, -1, "mfgp_nlml_grad: NULL argument");
    if (!h->factorized) return fail(h, -1, "mfgp_nlml_grad: no valid factorisation");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->grad_valid) {
        hipStream_t s = h->stream;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[3], s));
        if (run_step(h, h->pl.kinv_step) != 0) return -1;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[4], s));
        launch_grad(s, h->spec, h->dX, h->buf[BUF_A], (int)h->Np, h->dalpha, (int)h->N, (int)h->Np,
                    h->dpart, h->dres + 64);
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[5], s));
        HIPCHK(h, hipStreamSynchronize(s));
        HIPCHK(h, hipGetLastError());
        h->tm.kinv_ms = h->timing ? ev_ms(h->ev[3], h->ev[4]) : 0.f;
        h->tm.grad_ms = h->timing ? ev_ms(h->ev[4], h->ev[5]) : 0.f;
        for (int i = 0; i < h->spec.np + 1; ++i) h->grad[i] = h->hres[64 + i];
        h->kinv_valid = h->grad_valid = true;
        h->cum.grad_evals += 1;
        h->cum.kinv_ms += h->tm.kinv_ms;
        h->cum.grad_ms += h->tm.grad_ms;
        h->cum.total_ms += h->tm.kinv_ms + h->tm.grad_ms;
        h->cum.kinv_flops += (double)h->Np * h->Np * h->Np / 3.0;
    }
    for (int i = 0; i < h->spec.np + 1; ++i) grad[i] = h->grad[i];
    return 0;
}

// k(x, x) of the handle's stationary covariance at its current parameters: sum over the terms of the product of their variances
// (GPy Kdiag)
static double prior_variance(const mfgp_handle* h) {
    double kss = 0.0, prod = 1.0;
    int cur = h->spec.term[0];
    for (int f = 0; f < h->spec.nf; ++f) {
        if (h->spec.term[f] != cur) { kss += prod; prod = 1.0; cur = h->spec.term[f]; }
        prod *= h->theta[h->spec.toff[f]];
    }
    return kss + prod;
}

// make room for a predictive panel of rows_p rows in h->dXs
static int ensure_xs(mfgp_handle* h, int rows_p) {
    const int D = h->D;
    if (rows_p > h->xs_cap_rows || D != h->xs_cap_D) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->dXs) HIPCHK(h, hipFree(h->dXs));
        h->dXs = nullptr;
        h->xs_cap_rows = std::max(rows_p, h->xs_cap_rows);
        h->xs_cap_D = D;
        HIPCHK(h, hipMalloc(&h->dXs, (size_t)h->xs_cap_rows * D * sizeof(double)));
    }
    return 0;
}

// mean (and variance) of the `rows` test rows already resident (zero padded to rows_p) in h->dXs, in stream order
// `pinned`: the results are written by the kernels straight into the handle's device-mapped pinned memory and copied to
// mean / var by the host after the synchronisation (no device-to-host copy commands)
static int predict_chunk(mfgp_handle* h, int64_t rows, int rows_p, double* mean, double* var, int want_var,
                         int include_noise, double* pan_ms, double* var_ms, bool pinned = false) {
    hipStream_t s = h->stream;
    double* const mean_dev = pinned ? h->dio + mfgp_handle::IO_IN : h->dvec;
    double* const var_dev = pinned ? h->dio + mfgp_handle::IO_IN + mfgp_handle::IO_OUT : h->dvec2;
    const int64_t Np = h->Np;
    int rc;
    // <= 64 test rows (the DIRECT callback / acquisition case): bandwidth-bound products instead of a padded tile GEMM --
    // up to 16 rows on the VALU behind one coalesced read of the triangle (trimv_f64.hip: panel, product, ONE finishing launch for
    // mean and variance), 17 .. 64 rows the MFMA multi-vector form
    static const bool skinny_on = !(getenv("MFGP_SKINNY") && atoi(getenv("MFGP_SKINNY")) == 0);
    const bool few = skinny_on && rows <= 16;
    const bool skinny = want_var && skinny_on && rows <= 64 && !few;
    const int rows16 = rows <= 16 ? 1 : (rows <= 32 ? 2 : 4);
    if (want_var && !skinny && !few && h->pl.predv_rows != rows_p) {
        // (re)plan the variance product for this panel height; keep the cholinv/kinv tasks
        plan_predv(h->pl, rows_p);
        rc = upload_tasks(h);
        if (rc) return rc;
    }
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[6], s));
    if (few) {
        const int R = rows <= 1 ? 1 : (rows <= 2 ? 2 : (rows <= 4 ? 4 : (rows <= 8 ? 8 : 16)));
        launch_kbuild_panel(s, h->spec, h->dXs, 64, h->dX, (int)h->N, (int)Np, h->buf[BUF_W], (int)Np);
        h->launches += 1;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[7], s));
        if (want_var) {
            h->kinv_valid = false;  // V overwrites the K^-1 storage
            launch_predv_rows(s, R, h->buf[BUF_W], h->buf[BUF_S], h->buf[BUF_A], (int)Np, (int)Np, h->dalpha, mean_dev, (int)rows);
            launch_predv_finish(s, (int)rows, h->buf[BUF_A], (int)Np, (int)Np, prior_variance(h), include_noise ? h->noise : 0.0,
                                var_dev);
            h->launches += 2;
        } else {
            launch_rowdot(s, h->buf[BUF_W], (int)Np, h->dalpha, mean_dev, (int)rows, (int)Np, 2);
            h->launches += 1;
        }
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[8], s));
        if (!pinned) {
            HIPCHK(h, hipMemcpyAsync(mean, h->dvec, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
            if (want_var) HIPCHK(h, hipMemcpyAsync(var, h->dvec2, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    } else {
        launch_kbuild_panel(s, h->spec, h->dXs, rows_p, h->dX, (int)h->N, (int)Np, h->buf[BUF_W], (int)Np);
        launch_rowdot(s, h->buf[BUF_W], (int)Np, h->dalpha, mean_dev, rows_p, (int)Np, 2);
        h->launches += 2;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[7], s));
        if (!pinned) HIPCHK(h, hipMemcpyAsync(mean, h->dvec, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        if (want_var) {
            h->kinv_valid = false;  // V overwrites the K^-1 storage
            const int vrows = skinny ? 16 * rows16 : rows_p;
            if (skinny) launch_predv_skinny(s, rows16, h->buf[BUF_W], h->buf[BUF_S], h->buf[BUF_A], (int)Np, (int)Np);
            else if (run_step(h, h->pl.predv_step) != 0) return -1;
            launch_rowsumsq(s, h->buf[BUF_A], (int)Np, h->dvec2, vrows, (int)Np);
            launch_finish_var(s, h->spec, h->dvec2, var_dev, vrows, include_noise ? h->noise : 0.0);
            h->launches += 2;
            if (h->timing) HIPCHK(h, hipEventRecord(h->ev[8], s));
            if (!pinned) HIPCHK(h, hipMemcpyAsync(var, h->dvec2, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    }
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (pinned) {
        memcpy(mean, h->hio + mfgp_handle::IO_IN, (size_t)rows * sizeof(double));
        if (want_var) memcpy(var, h->hio + mfgp_handle::IO_IN + mfgp_handle::IO_OUT, (size_t)rows * sizeof(double));
    }
    if (h->timing) {
        *pan_ms += ev_ms(h->ev[6], h->ev[7]);
        if (want_var) *var_ms += ev_ms(h->ev[7], h->ev[8]);
    }
    return 0;
}

static void predict_account(mfgp_handle* h, int64_t Nstar, double pan_ms, double var_ms, bool want_var) {
    h->tm.predict_panel_ms = pan_ms;
    h->tm.predict_var_ms = var_ms;
    h->cum.predicts += 1;
    h->cum.predict_rows += (double)Nstar;
    h->cum.predict_ms += pan_ms + var_ms;
    h->cum.predict_panel_ms += pan_ms;
    h->cum.predict_var_ms += var_ms;
    if (want_var) {
        h->cum.predict_var_flops += (double)h->Np * (double)h->Np * (double)Nstar;      // the work, timed or not
        if (h->timing) h->cum.timed_predict_var_flops += (double)h->Np * (double)h->Np * (double)Nstar;
    }
    h->tm.timed = h->timing ? (h->tm.timed | 1) : h->tm.timed;
    h->tm.n_launches = h->launches;
}

int32_t mfgp_predict(mfgp_handle* h, const double* Xstar, int64_t Nstar, double* mean, double* var,
                     int32_t want_var, int32_t include_noise) {
    int rc = check_ready(h, "mfgp_predict");
    if (rc) return rc;
    if (!Xstar || !mean || (want_var && !var)) return fail(h, -1, "mfgp_predict: NULL argument");
    if (Nstar < 1) return fail(h, -1, "mfgp_predict: Nstar < 1");
    if (!h->factorized) return fail(h, -1, "mfgp_predict: no valid factorisation (call mfgp_factorize / mfgp_eval)");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int D = h->D;
    const int64_t Np = h->Np;
    double pan_ms = 0, var_ms = 0;
    h->launches = 0;
    for (int64_t r0 = 0; r0 < Nstar; r0 += Np) {
        const int64_t rows = std::min(Np, Nstar - r0);
        const int rows_p = (int)((rows + NB - 1) / NB * NB);
        rc = ensure_xs(h, rows_p);
        if (rc) return rc;
        // vrows of the skinny variance path can exceed rows (16 / 32 / 64): the output slots hold rows_p
        const bool pinned = (int64_t)rows_p * D <= mfgp_handle::IO_IN && rows_p <= mfgp_handle::IO_OUT;
        if (pinned) {   // zero-padded rows assembled in pinned memory by the host, ONE asynchronous copy command
            memcpy(h->hio, Xstar + r0 * D, (size_t)rows * D * sizeof(double));
            memset(h->hio + rows * D, 0, (size_t)(rows_p - rows) * D * sizeof(double));
            HIPCHK(h, hipMemcpyAsync(h->dXs, h->hio, (size_t)rows_p * D * sizeof(double), hipMemcpyHostToDevice, s));
        } else {
            HIPCHK(h, hipMemsetAsync(h->dXs, 0, (size_t)rows_p * D * sizeof(double), s));
            HIPCHK(h, hipMemcpyAsync(h->dXs, Xstar + r0 * D, (size_t)rows * D * sizeof(double), hipMemcpyHostToDevice, s));
        }
        rc = predict_chunk(h, rows, rows_p, mean + r0, want_var ? var + r0 : nullptr, want_var, include_noise, &pan_ms,
                           &var_ms, pinned);
        if (rc) return rc;
    }
    predict_account(h, Nstar, pan_ms, var_ms, want_var != 0);
    return 0;
}

// ---- device-resident level chaining ------------------------------------------------------
static int ensure_chain(mfgp_handle* lf, int64_t rows, int c) {
    const int d = lf->D;
    if (rows > lf->ch_rows || c > lf->ch_c || d != lf->ch_D) {   // (a handle reused at another input width re-allocates)
        HIPCHK(lf, hipStreamSynchronize(lf->stream));
        if (lf->dXc) HIPCHK(lf, hipFree(lf->dXc));
        if (lf->dm) HIPCHK(lf, hipFree(lf->dm));
        if (lf->doffs) HIPCHK(lf, hipFree(lf->doffs));
        if (lf->dAug) HIPCHK(lf, hipFree(lf->dAug));
        lf->dXc = lf->dm = lf->doffs = lf->dAug = nullptr;
        lf->ch_rows = std::max(rows, lf->ch_rows);
        lf->ch_c = std::max(c, lf->ch_c);
        lf->ch_D = d;
        HIPCHK(lf, hipMalloc(&lf->dXc, (size_t)lf->ch_rows * d * sizeof(double)));
        HIPCHK(lf, hipMalloc(&lf->dm, (size_t)lf->ch_rows * lf->ch_c * sizeof(double)));
        HIPCHK(lf, hipMalloc(&lf->doffs, (size_t)lf->ch_c * d * sizeof(double)));
        HIPCHK(lf, hipMalloc(&lf->dAug, (size_t)lf->ch_rows * (d + lf->ch_c) * sizeof(double)));
    }
    return 0;
}

// On lf->stream: upload `rows` base points, push the (rows*c, d) stencil stack through the low-fidelity posterior
// mean.  Leaves the base points in lf->dXc and the means, (rows, c) row-major, in lf->dm.  No host synchronisation.
// `s`: the stream everything is enqueued on -- lf's own, or the consuming level's (mfgp_predict_chained: one stream for both
// levels, no cross-stream hop; nothing else runs on lf meanwhile, every API call ends synchronised).
static int chain_lf_means(mfgp_handle* lf, const double* Xhost, int64_t rows, const double* offs_host, int c, hipStream_t s) {
    const int d = lf->D;
    const int64_t Np = lf->Np;
    int rc = ensure_chain(lf, rows, c);
    if (rc) return rc;
    const int64_t T = rows * c;
    rc = ensure_xs(lf, (int)std::min<int64_t>(Np, (T + NB - 1) / NB * NB));
    if (rc) return rc;
    if ((rows + c) * d <= mfgp_handle::IO_IN) {   // small batch: through pinned memory (see mfgp_predict), copies stay asynchronous
        memcpy(lf->hio, Xhost, (size_t)rows * d * sizeof(double));
        memcpy(lf->hio + rows * d, offs_host, (size_t)c * d * sizeof(double));
        HIPCHK(lf, hipMemcpyAsync(lf->dXc, lf->hio, (size_t)rows * d * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(lf, hipMemcpyAsync(lf->doffs, lf->hio + rows * d, (size_t)c * d * sizeof(double), hipMemcpyHostToDevice, s));
    } else {
        HIPCHK(lf, hipMemcpyAsync(lf->doffs, offs_host, (size_t)c * d * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(lf, hipMemcpyAsync(lf->dXc, Xhost, (size_t)rows * d * sizeof(double), hipMemcpyHostToDevice, s));
    }
    for (int64_t t0 = 0; t0 < T; t0 += Np) {
        const int n = (int)std::min(Np, T - t0);
        const int n_p = (n + NB - 1) / NB * NB;
        launch_stencil_rows(s, lf->dXc, lf->doffs, d, c, t0, n, n_p, lf->dXs);
        launch_kbuild_panel(s, lf->spec, lf->dXs, n_p, lf->dX, (int)lf->N, (int)Np, lf->buf[BUF_W], (int)Np);
        // the padded rows n..n_p of the mean land in dvec's tail, never in dm: write through dvec, then copy
        launch_rowdot(s, lf->buf[BUF_W], (int)Np, lf->dalpha, lf->dvec, n_p, (int)Np, 2);
        HIPCHK(lf, hipMemcpyAsync(lf->dm + t0, lf->dvec, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        lf->launches += 3;
    }
    return 0;
}

static int chain_check(mfgp_handle* lf, const double* X, int64_t N, const double* offs, int c, const char* who) {
    int rc = check_ready(lf, who);
    if (rc) return rc;
    if (!X || !offs) return fail(lf, -1, std::string(who) + ": NULL argument");
    if (N < 1 || c < 1) return fail(lf, -1, std::string(who) + ": need N >= 1 and c >= 1");
    if (!lf->factorized) return fail(lf, -1, std::string(who) + ": the low-fidelity level has no valid factorisation");
    return 0;
}

int32_t mfgp_augment(mfgp_handle* lf, const double* X, int64_t N, const double* offsets, int32_t c, double* out) {
    int rc = chain_check(lf, X, N, offsets, c, "mfgp_augment");
    if (rc) return rc;
    if (!out) return fail(lf, -1, "mfgp_augment: NULL argument");
    HIPCHK(lf, hipSetDevice(lf->device));
    const int d = lf->D, w = d + c;
    const int64_t chunk = lf->Np;
    lf->launches = 0;
    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t rows = std::min(chunk, N - r0);
        rc = chain_lf_means(lf, X + r0 * d, rows, offsets, c, lf->stream);
        if (rc) return rc;
        launch_assemble_aug(lf->stream, lf->dXc, lf->dm, (int)rows, (int)rows, d, c, lf->dAug, w);
        HIPCHK(lf, hipMemcpyAsync(out + r0 * w, lf->dAug, (size_t)rows * w * sizeof(double), hipMemcpyDeviceToHost,
                                  lf->stream));
        HIPCHK(lf, hipStreamSynchronize(lf->stream));
        HIPCHK(lf, hipGetLastError());
    }
    return 0;
}

int32_t mfgp_predict_chained(mfgp_handle* h, mfgp_handle* lf, const double* Xstar, int64_t Nstar, const double* offsets,
                             int32_t c, double* mean, double* var, int32_t want_var, int32_t include_noise,
                             double* aug_out) {
    int rc = check_ready(h, "mfgp_predict_chained");
    if (rc) return rc;
    if (!lf) return fail(h, -1, "mfgp_predict_chained: NULL low-fidelity handle");
    if (lf == h) return fail(h, -1, "mfgp_predict_chained: the two levels must be distinct handles");
    rc = chain_check(lf, Xstar, Nstar, offsets, c, "mfgp_predict_chained");
    if (rc) return fail(h, rc, std::string("mfgp_predict_chained: low-fidelity level: ") + lf->err);
    if (!mean || (want_var && !var)) return fail(h, -1, "mfgp_predict_chained: NULL argument");
    if (!h->factorized) return fail(h, -1, "mfgp_predict_chained: no valid factorisation (call mfgp_factorize / mfgp_eval)");
    if (h->device != lf->device) return fail(h, -1, "mfgp_predict_chained: the two levels live on different devices");
    if (h->D != lf->D + c) return fail(h, -1, "mfgp_predict_chained: this level has D != d_lf + c columns");
    HIPCHK(h, hipSetDevice(h->device));
    const int d = lf->D, D = h->D;
    const int64_t Np = h->Np;
    double pan_ms = 0, var_ms = 0;
    h->launches = 0;
    lf->launches = 0;
    for (int64_t r0 = 0; r0 < Nstar; r0 += Np) {
        const int64_t rows = std::min(Np, Nstar - r0);
        const int rows_p = (int)((rows + NB - 1) / NB * NB);
        rc = ensure_xs(h, rows_p);
        if (rc) return rc;
        // both levels on THIS level's stream: the low-fidelity means, the augmented rows (straight into this level's panel
        // input) and this level's predict follow each other in stream order -- no event, no cross-stream hop (~12 us)
        rc = chain_lf_means(lf, Xstar + r0 * d, rows, offsets, c, h->stream);
        if (rc) return fail(h, rc, std::string("mfgp_predict_chained: low-fidelity level: ") + lf->err);
        launch_assemble_aug(h->stream, lf->dXc, lf->dm, (int)rows, rows_p, d, c, h->dXs, D);
        if (aug_out)
            HIPCHK(h, hipMemcpyAsync(aug_out + r0 * D, h->dXs, (size_t)rows * D * sizeof(double), hipMemcpyDeviceToHost,
                                     h->stream));
        h->launches += lf->launches + 1;
        lf->launches = 0;
        rc = predict_chunk(h, rows, rows_p, mean + r0, want_var ? var + r0 : nullptr, want_var, include_noise, &pan_ms,
                           &var_ms, rows_p <= mfgp_handle::IO_OUT);
        if (rc) return rc;
    }
    predict_account(h, Nstar, pan_ms, var_ms, want_var != 0);
    return 0;
}

// ---- read-back ---------------------------------------------------------------------------------------
static int copy_block(mfgp_handle* h, const double* dsrc, double* out, int mode) {
    // mode 0: lower triangle only (zeros above); 1: symmetric from lower; 2: as stored
    const int64_t N = h->N, Np = h->Np;
    std::vector<double> tmp((size_t)Np * Np);
    HIPCHK(h, hipMemcpy(tmp.data(), dsrc, (size_t)Np * Np * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j < N; ++j) {
            double v;
            if (mode == 0) v = (j <= i) ? tmp[i * Np + j] : 0.0;
            else if (mode == 1) v = (j <= i) ? tmp[i * Np + j] : tmp[j * Np + i];
            else v = tmp[i * Np + j];
            out[i * N + j] = v;
        }
    return 0;
}

int32_t mfgp_get_K(mfgp_handle* h, double* out) {
    int rc = check_ready(h, "mfgp_get_K");
    if (rc) return rc;
    if (!out) return fail(h, -1, "mfgp_get_K: NULL");
    if (!h->params_set) return fail(h, -1, "mfgp_get_K: no parameters yet (call mfgp_eval / mfgp_factorize first)");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t N = h->N;
    double* d = nullptr;
    HIPCHK(h, hipMalloc(&d, (size_t)N * N * sizeof(double)));
    launch_kbuild_full(h->stream, h->spec, h->dX, (int)N, (int)h->Np, d, (int)N);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, d, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipFree(d));
    return 0;
}
int32_t mfgp_get_L(mfgp_handle* h, double* out) {
    if (!h || !out) return fail(h, -1, "mfgp_get_L: NULL");
    if (!h->factorized) return fail(h, -1, "mfgp_get_L: no valid factorisation");
    HIPCHK(h, hipSetDevice(h->device));
    return copy_block(h, h->buf[BUF_L], out, 0);
}
int32_t mfgp_get_Linv(mfgp_handle* h, double* out) {
    if (!h || !out) return fail(h, -1, "mfgp_get_Linv: NULL");
    if (!h->factorized) return fail(h, -1, "mfgp_get_Linv: no valid factorisation");
    HIPCHK(h, hipSetDevice(h->device));
    return copy_block(h, h->buf[BUF_S], out, 0);
}
int32_t mfgp_get_Kinv(mfgp_handle* h, double* out) {
    if (!h || !out) return fail(h, -1, "mfgp_get_Kinv: NULL");
    if (!h->factorized || !h->kinv_valid) return fail(h, -1, "mfgp_get_Kinv: K^-1 not available (evaluate with want_grad)");
    HIPCHK(h, hipSetDevice(h->device));
    return copy_block(h, h->buf[BUF_A], out, 1);
}
int32_t mfgp_get_alpha(mfgp_handle* h, double* out) {
    if (!h || !out) return fail(h, -1, "mfgp_get_alpha: NULL");
    if (!h->factorized) return fail(h, -1, "mfgp_get_alpha: no valid factorisation");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(out, h->dalpha, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
int32_t mfgp_get_counters(mfgp_handle* h, mfgp_counters* out, int32_t reset) {
    if (!h || !out) return fail(h, -1, "mfgp_get_counters: NULL");
    *out = h->cum;
    if (reset) memset(&h->cum, 0, sizeof h->cum);
    return 0;
}
int32_t mfgp_device_synchronize(mfgp_handle* h) {
    if (!h) return fail(h, -1, "mfgp_device_synchronize: NULL");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    return 0;
}
int32_t mfgp_get_timings(mfgp_handle* h, mfgp_timings* out) {
    if (!h || !out) return fail(h, -1, "mfgp_get_timings: NULL");
    *out = h->tm;
    return 0;
}

// ---- test hooks ---------------------------------------------------------------------------------------
int32_t mfgp_dbg_gemm_nt(mfgp_handle* h, const double* A, const double* B, double* C, int32_t M, int32_t N,
                         int32_t K, double alpha, double beta, int32_t tile) {
    if (!h || !A || !B || !C) return fail(h, -1, "mfgp_dbg_gemm_nt: NULL");
    const bool chain = (tile == -64);   // -64: the serial-chain variant of the 64-tile kernel (mfgp_gemm_nt_f64_chain)
    if (chain) tile = 64;
    if ((tile != 128 && tile != 64 && tile != 32) || M % tile || N % tile || K % BK || K < BK)
        return fail(h, -1, "mfgp_dbg_gemm_nt: M, N must be multiples of the tile (128, 64, 32) and K of 32");
    HIPCHK(h, hipSetDevice(h->device));
    // one common leading dimension
    const int ld = std::max(K, N);
    double *dA, *dB, *dC;
    GemmTask* dt;
    HIPCHK(h, hipMalloc(&dA, (size_t)M * ld * 8));
    HIPCHK(h, hipMalloc(&dB, (size_t)N * ld * 8));
    HIPCHK(h, hipMalloc(&dC, (size_t)M * ld * 8));
    HIPCHK(h, hipMemcpy2D(dA, (size_t)ld * 8, A, (size_t)K * 8, (size_t)K * 8, M, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy2D(dB, (size_t)ld * 8, B, (size_t)K * 8, (size_t)K * 8, N, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy2D(dC, (size_t)ld * 8, C, (size_t)N * 8, (size_t)N * 8, M, hipMemcpyHostToDevice));
    std::vector<GemmTask> ts;
    for (int i = 0; i < M / tile; ++i)
        for (int j = 0; j < N / tile; ++j) {
            GemmTask t{};
            t.a_off = (int64_t)i * tile * ld;
            t.b_off = (int64_t)j * tile * ld;
            t.c_off = (int64_t)i * tile * ld + j * tile;
            t.c2_off = -1;
            t.klen = K;
            t.alpha = alpha; t.beta = beta;
            ts.push_back(t);
        }
    HIPCHK(h, hipMalloc(&dt, ts.size() * sizeof(GemmTask)));
    HIPCHK(h, hipMemcpy(dt, ts.data(), ts.size() * sizeof(GemmTask), hipMemcpyHostToDevice));
    if (launch_gemm(h->stream, tile, dt, (int)ts.size(), dA, dB, dC, nullptr, ld, chain ? 3 : 0) != 0)
        return fail(h, -1, "mfgp_dbg_gemm_nt: no kernel for this tile");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy2D(C, (size_t)N * 8, dC, (size_t)ld * 8, (size_t)N * 8, M, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dt);
    return 0;
}

int32_t mfgp_dbg_leaf(mfgp_handle* h, const double* A, double* Lout, double* Xout, double* logdet_half) {
    if (!h || !A || !Lout || !Xout || !logdet_half) return fail(h, -1, "mfgp_dbg_leaf: NULL");
    HIPCHK(h, hipSetDevice(h->device));
    double *dA, *dL, *dS, *dl;
    int* di;
    const size_t bytes = (size_t)NB * NB * 8;
    HIPCHK(h, hipMalloc(&dA, bytes)); HIPCHK(h, hipMalloc(&dL, bytes)); HIPCHK(h, hipMalloc(&dS, bytes));
    HIPCHK(h, hipMalloc(&dl, 8)); HIPCHK(h, hipMalloc(&di, 4));
    HIPCHK(h, hipMemcpy(dA, A, bytes, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(di, 0, 4));
    launch_leaf(h->stream, dA, dL, dS, NB, 0, dl, di);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    int info = 0;
    HIPCHK(h, hipMemcpy(Lout, dL, bytes, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(Xout, dS, bytes, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(logdet_half, dl, 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&info, di, 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dL); hipFree(dS); hipFree(dl); hipFree(di);
    return info;
}

}  // extern "C"
