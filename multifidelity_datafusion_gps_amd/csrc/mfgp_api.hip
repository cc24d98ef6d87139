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
#include "api_shared.h"

using namespace mfgp;

thread_local std::string mfgp::g_err;

int upload_tasks(mfgp_handle* h) {
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
int run_step(mfgp_handle* h, const Step& s, bool want_grad, int nbatch, const GemmTask* tasks) {
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
            // per rank: its blocks of the panel column, padded to the largest share, then -- one exchange per column (Step::carry) --
            // room for the diagonal message of block c + 1 (used by its owner's chunk only)
            const long long panel_part = (long long)*std::max_element(cnt.begin(), cnt.end()) * NB * NB;
            const long long chunk = panel_part + (s.carry ? 2LL * NB * NB + 2 : 0);
            const int next_own = shard_owner(c + 1, size);
            launch_dist_panel_copy(st, h->buf[BUF_L], Np, h->nblk, c, h->ddist, chunk, rank, size, false);
            if (s.carry && next_own == rank)
                launch_dist_diag_copy(st, h->buf[BUF_L], h->buf[BUF_S], Np, c + 1, h->ddist + (long long)rank * chunk + panel_part, h->dlogdet,
                                      h->dinfo, false);
            if (int rc = comm_allgather_chunks(h, h->ddist, (size_t)chunk, st)) return rc;
            launch_dist_panel_copy(st, h->buf[BUF_L], Np, h->nblk, c, h->ddist, chunk, rank, size, true);
            if (s.carry && next_own != rank)
                launch_dist_diag_copy(st, h->buf[BUF_L], h->buf[BUF_S], Np, c + 1, h->ddist + (long long)next_own * chunk + panel_part,
                                      h->dlogdet, h->dinfo, true, h->dflag, h->epoch);
            if (s.carry) h->launches += 1;
        }
        h->launches += 2;
    }   // kind 2: join -- the wait above is all there is
    if (s.rec_ev > 0) (void)hipEventRecord(h->evpool[s.rec_ev - 1], st);
    if (s.rec_ev_final > 0) (void)hipEventRecord(h->evpool[s.rec_ev_final - 1], st);
    return 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
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

void free_batch(mfgp_handle* h) {
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
    // (dXc lies inside the allocation that starts at doffs: api_predict.hip ensure_chain)
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

int build_plans(mfgp_handle* h) {
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
    h->timing_small = false;        // ... and a predict of <= 64 rows (~80 us at N = 8192, three records ~6.5 us of it) is stamped only on request
    if (const char* e = getenv("MFGP_TIMING")) {
        h->timing = h->timing || atoi(e) != 0;
        h->timing_small = atoi(e) != 0;
    }
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

int check_ready(mfgp_handle* h, const char* who) {
    if (!h) return fail(nullptr, -1, std::string(who) + ": NULL handle");
    if (!h->have_data) return fail(h, -1, std::string(who) + ": mfgp_set_data not called");
    if (!h->have_kernel) return fail(h, -1, std::string(who) + ": mfgp_set_kernel not called");
    for (int f = 0; f < h->spec.nf; ++f)
        if (h->spec.c1[f] > h->D) return fail(h, -1, std::string(who) + ": kernel column range exceeds D");
    h->spec.D = h->D;
    return 0;
}

// enqueue K-build + cholinv + solve (+ K^-1 + gradient); no host sync
int set_params(mfgp_handle* h, const double* theta, double noise, double jitter) {
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

int enqueue_eval(mfgp_handle* h, const double* theta, double noise, double jitter, bool want_grad, bool prebuilt) {
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

int finish_eval(mfgp_handle* h, bool want_grad) {
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

}  // extern "C"
