// api_predict.hip -- what runs on a finished factorisation: mfgp_predict, the rank-1 append of the adaptation loop
// (8(f1)) and the device-resident level chaining (8(f3): mfgp_augment / mfgp_predict_chained).  Split out of mfgp_api.hip in round 6.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "mfgp_internal.h"
#include "api_shared.h"

using namespace mfgp;

extern "C" {

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
    if (!launch_kbuild_panel_few(s, h->spec, few_rows_packed(h->dXs, D, 1), 1, h->dX, n, (int)Np, h->buf[BUF_W], (int)Np))
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

// k(x, x) of the handle's stationary covariance at its current parameters: sum over the terms of the product of their variances
// (GPy Kdiag)
double prior_variance(const mfgp_handle* h) {
    double kss = 0.0, prod = 1.0;
    int cur = h->spec.term[0];
    for (int f = 0; f < h->spec.nf; ++f) {
        if (h->spec.term[f] != cur) { kss += prod; prod = 1.0; cur = h->spec.term[f]; }
        prod *= h->theta[h->spec.toff[f]];
    }
    return kss + prod;
}

// make room for a predictive panel of rows_p rows in h->dXs
int ensure_xs(mfgp_handle* h, int rows_p) {
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
static bool skinny_enabled() {
    static const bool on = !(getenv("MFGP_SKINNY") && atoi(getenv("MFGP_SKINNY")) == 0);
    return on;
}
// `src`: where the (at most 4) test rows are, if not in h->dXs -- a level-chained predict hands its augmented rows over unassembled
// (only when few_source_ok(h, rows) said the panel can be built from them)
static bool few_source_ok(const mfgp_handle* h, int64_t rows) { return skinny_enabled() && rows <= 4 && kbuild_panel_few_ok(h->spec); }
static int predict_chunk(mfgp_handle* h, int64_t rows, int rows_p, double* mean, double* var, int want_var,
                         int include_noise, double* pan_ms, double* var_ms, int64_t* timed_rows, bool pinned = false,
                         const FewRows* src = nullptr) {
    hipStream_t s = h->stream;
    double* const mean_dev = pinned ? h->dio + mfgp_handle::IO_IN : h->dvec;
    double* const var_dev = pinned ? h->dio + mfgp_handle::IO_IN + mfgp_handle::IO_OUT : h->dvec2;
    const int64_t Np = h->Np;
    int rc;
    // <= 64 test rows (the DIRECT callback / acquisition case): bandwidth-bound products instead of a padded tile GEMM, all in
    // trimv_f64.hip -- a few rows on the VALU behind one coalesced read of the triangle, up to 64 rows on the matrix pipe with S
    // staged in the same coalesced shape (through registers from Np = 3072, by LDS-DMA below); either way: panel, product (+ the
    // means), ONE finishing launch.  MFGP_PREDV2=0 keeps the matrix-pipe products on the LDS-DMA form at every size.
    const bool few = skinny_enabled() && rows <= 64;
    static const bool predv2_on = !(getenv("MFGP_PREDV2") && atoi(getenv("MFGP_PREDV2")) == 0);
    if (want_var && !few && h->pl.predv_rows != rows_p) {
        // (re)plan the variance product for this panel height; keep the cholinv/kinv tasks
        plan_predv(h->pl, rows_p);
        rc = upload_tasks(h);
        if (rc) return rc;
    }
    const bool stamp = h->timing && (!few || h->timing_small);
    if (stamp) HIPCHK(h, hipEventRecord(h->ev[6], s));
    if (few) {
        // (up to 4 rows -- the products below then read 1, 2 or 4 rows of the panel, the means `rows` -- one thread per training point
        // instead of 64-row tiles, where the kernel description allows)
        const int Rfew = rows <= 1 ? 1 : (rows <= 2 ? 2 : 4);
        const FewRows packed = few_rows_packed(h->dXs, h->D, Rfew);
        if (!(rows <= 4 && launch_kbuild_panel_few(s, h->spec, src ? *src : packed, Rfew, h->dX, (int)h->N, (int)Np, h->buf[BUF_W],
                                                   (int)Np))) {
            if (src) return fail(h, -1, "predict_chunk: unassembled test rows without the few-row panel");
            launch_kbuild_panel(s, h->spec, h->dXs, 64, h->dX, (int)h->N, (int)Np, h->buf[BUF_W], (int)Np);
        }
        h->launches += 1;
        if (stamp) HIPCHK(h, hipEventRecord(h->ev[7], s));
        if (want_var) {
            h->kinv_valid = false;  // V overwrites the K^-1 storage
            // up to 4 rows on the VALU; 5 .. 8 too below Np = 2048, from there the 16-row matrix-pipe forms are faster than the 8-row
            // VALU form (variance stage, ms, 8 rows VALU / 16 rows matrix pipe: Np = 2048 0.018 / 0.016, 3072 0.024 / 0.016,
            // 4096 0.028 / 0.023, 8192 0.060 / 0.054; 4 rows VALU: 0.014, 0.017, 0.020, 0.055)
            const double kss = prior_variance(h), add = include_noise ? h->noise : 0.0;
            if (rows <= 4 || (rows <= 8 && Np < 2048)) {
                const int R = rows <= 1 ? 1 : (rows <= 2 ? 2 : (rows <= 4 ? 4 : 8));
                launch_predv_rows(s, R, h->buf[BUF_W], h->buf[BUF_S], h->buf[BUF_A], (int)Np, (int)Np, h->dalpha, mean_dev, (int)rows);
                launch_predv_finish(s, (int)rows, h->buf[BUF_A], (int)Np, (int)Np, kss, add, var_dev);
                h->launches += 2;
            } else if (predv2_on && predv_mfma2_pays((int)rows, (int)Np)) {
                // the register-staged form: partial planes per share of the triangle, summed by its own finishing launch
                launch_predv_mfma2(s, (int)((rows + 15) / 16), h->buf[BUF_W], h->buf[BUF_S], h->buf[BUF_A], (int)Np, (int)Np,
                                   h->dalpha, mean_dev, (int)rows);
                launch_predv_finish_planes(s, (int)rows, h->buf[BUF_A], (int)Np, (int)Np, kss, add, var_dev);
                h->launches += 2;
            } else {
                // (the fragment-ordered copy of the panel: rows 64 .. 127 of the workspace matrix, which holds the 64-row panel)
                launch_predv_mfma(s, (int)((rows + 15) / 16), h->buf[BUF_W], h->buf[BUF_W] + 64 * Np, h->buf[BUF_S], h->buf[BUF_A],
                                  (int)Np, (int)Np, h->dalpha, mean_dev, (int)rows);
                launch_predv_finish(s, (int)rows, h->buf[BUF_A], (int)Np, (int)Np, kss, add, var_dev);
                h->launches += 3;
            }
        } else {
            launch_rowdot(s, h->buf[BUF_W], (int)Np, h->dalpha, mean_dev, (int)rows, (int)Np, 2);
            h->launches += 1;
        }
        if (stamp) HIPCHK(h, hipEventRecord(h->ev[8], s));
        if (!pinned) {
            HIPCHK(h, hipMemcpyAsync(mean, h->dvec, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
            if (want_var) HIPCHK(h, hipMemcpyAsync(var, h->dvec2, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    } else {
        launch_kbuild_panel(s, h->spec, h->dXs, rows_p, h->dX, (int)h->N, (int)Np, h->buf[BUF_W], (int)Np);
        launch_rowdot(s, h->buf[BUF_W], (int)Np, h->dalpha, mean_dev, rows_p, (int)Np, 2);
        h->launches += 2;
        if (stamp) HIPCHK(h, hipEventRecord(h->ev[7], s));
        if (!pinned) HIPCHK(h, hipMemcpyAsync(mean, h->dvec, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        if (want_var) {
            h->kinv_valid = false;  // V overwrites the K^-1 storage
            if (run_step(h, h->pl.predv_step) != 0) return -1;
            launch_rowsumsq(s, h->buf[BUF_A], (int)Np, h->dvec2, rows_p, (int)Np);
            launch_finish_var(s, h->spec, h->dvec2, var_dev, rows_p, include_noise ? h->noise : 0.0);
            h->launches += 2;
            if (stamp) HIPCHK(h, hipEventRecord(h->ev[8], s));
            if (!pinned) HIPCHK(h, hipMemcpyAsync(var, h->dvec2, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
        }
    }
    HIPCHK(h, hipStreamSynchronize(s));
    HIPCHK(h, hipGetLastError());
    if (pinned) {
        memcpy(mean, h->hio + mfgp_handle::IO_IN, (size_t)rows * sizeof(double));
        if (want_var) memcpy(var, h->hio + mfgp_handle::IO_IN + mfgp_handle::IO_OUT, (size_t)rows * sizeof(double));
    }
    if (stamp) {
        *pan_ms += ev_ms(h->ev[6], h->ev[7]);
        if (want_var) *var_ms += ev_ms(h->ev[7], h->ev[8]);
        *timed_rows += rows;
    }
    return 0;
}

static void predict_account(mfgp_handle* h, int64_t Nstar, double pan_ms, double var_ms, bool want_var, int64_t timed_rows) {
    h->tm.predict_panel_ms = pan_ms;
    h->tm.predict_var_ms = var_ms;
    h->cum.predicts += 1;
    h->cum.predict_rows += (double)Nstar;
    h->cum.predict_ms += pan_ms + var_ms;
    h->cum.predict_panel_ms += pan_ms;
    h->cum.predict_var_ms += var_ms;
    if (want_var) {
        h->cum.predict_var_flops += (double)h->Np * (double)h->Np * (double)Nstar;      // the work, timed or not
        h->cum.timed_predict_var_flops += (double)h->Np * (double)h->Np * (double)timed_rows;    // (the rows whose launches were stamped)
    }
    h->tm.timed = timed_rows > 0 ? (h->tm.timed | 1) : h->tm.timed;
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
    int64_t timed_rows = 0;
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
                           &var_ms, &timed_rows, pinned);
        if (rc) return rc;
    }
    predict_account(h, Nstar, pan_ms, var_ms, want_var != 0, timed_rows);
    return 0;
}

// ---- device-resident level chaining ------------------------------------------------------
static int ensure_chain(mfgp_handle* lf, int64_t rows, int c) {
    const int d = lf->D;
    if (rows > lf->ch_rows || c > lf->ch_c || d != lf->ch_D) {   // (a handle reused at another input width re-allocates)
        HIPCHK(lf, hipStreamSynchronize(lf->stream));
        // (dXc points INTO the allocation that starts at doffs: one host-to-device copy carries the stencil offsets and the base points)
        if (lf->dm) HIPCHK(lf, hipFree(lf->dm));
        if (lf->doffs) HIPCHK(lf, hipFree(lf->doffs));
        if (lf->dAug) HIPCHK(lf, hipFree(lf->dAug));
        lf->dXc = lf->dm = lf->doffs = lf->dAug = nullptr;
        lf->ch_rows = std::max(rows, lf->ch_rows);
        lf->ch_c = std::max(c, lf->ch_c);
        lf->ch_D = d;

        HIPCHK(lf, hipMalloc(&lf->dm, (size_t)lf->ch_rows * lf->ch_c * sizeof(double)));
        HIPCHK(lf, hipMalloc(&lf->doffs, ((size_t)lf->ch_c + (size_t)lf->ch_rows) * d * sizeof(double)));
        lf->dXc = lf->doffs + (size_t)lf->ch_c * d;
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
    if ((rows + lf->ch_c) * d <= mfgp_handle::IO_IN) {   // small batch: through pinned memory (see mfgp_predict), ONE asynchronous copy:
        // [stencil offsets, padded to the scratch's ch_c rows | base points] lands on [doffs | dXc], which are one allocation
        memcpy(lf->hio, offs_host, (size_t)c * d * sizeof(double));
        memcpy(lf->hio + (size_t)lf->ch_c * d, Xhost, (size_t)rows * d * sizeof(double));
        HIPCHK(lf, hipMemcpyAsync(lf->doffs, lf->hio, ((size_t)lf->ch_c + (size_t)rows) * d * sizeof(double), hipMemcpyHostToDevice, s));
    } else {
        HIPCHK(lf, hipMemcpyAsync(lf->doffs, offs_host, (size_t)c * d * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(lf, hipMemcpyAsync(lf->dXc, Xhost, (size_t)rows * d * sizeof(double), hipMemcpyHostToDevice, s));
    }
    for (int64_t t0 = 0; t0 < T; t0 += Np) {
        const int n = (int)std::min(Np, T - t0);
        const int n_p = (n + 63) / 64 * 64;      // the panel kernel works in 64-row tiles (round 6: was 128 -- a one-point callback built and
                                                 // read a 128 x N_lf panel, 16 MB at N_lf = 16384, for one row of it)
        // up to 4 rows: the stencil rows are formed inside the few-row panel kernel (covariance.hip) -- one launch less
        const FewRows st{lf->dXc, lf->doffs, nullptr, d, c, 0, n, (long long)t0};
        if (n <= 4 && launch_kbuild_panel_few(s, lf->spec, st, n <= 1 ? 1 : (n <= 2 ? 2 : 4), lf->dX, (int)lf->N, (int)Np,
                                              lf->buf[BUF_W], (int)Np)) {
            lf->launches += 2;
        } else {
            launch_stencil_rows(s, lf->dXc, lf->doffs, d, c, t0, n, n_p, lf->dXs);
            launch_kbuild_panel(s, lf->spec, lf->dXs, n_p, lf->dX, (int)lf->N, (int)Np, lf->buf[BUF_W], (int)Np);
            lf->launches += 3;
        }
        // the means of the n real rows only, straight to their place in dm
        launch_rowdot(s, lf->buf[BUF_W], (int)Np, lf->dalpha, lf->dm + t0, n, (int)Np, 2);
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
    int64_t timed_rows = 0;
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
        // up to 4 rows: this level's few-row panel kernel reads the augmented rows [Xc[i] | m[i]] where they are -- no assembling launch
        const bool unassembled = !aug_out && few_source_ok(h, rows);
        const FewRows augsrc{lf->dXc, nullptr, lf->dm, d, 1, c, (int)rows, 0};
        if (!unassembled) {
            launch_assemble_aug(h->stream, lf->dXc, lf->dm, (int)rows, rows_p, d, c, h->dXs, D);
            h->launches += 1;
        }
        if (aug_out)
            HIPCHK(h, hipMemcpyAsync(aug_out + r0 * D, h->dXs, (size_t)rows * D * sizeof(double), hipMemcpyDeviceToHost,
                                     h->stream));
        h->launches += lf->launches;
        lf->launches = 0;
        rc = predict_chunk(h, rows, rows_p, mean + r0, want_var ? var + r0 : nullptr, want_var, include_noise, &pan_ms,
                           &var_ms, &timed_rows, rows_p <= mfgp_handle::IO_OUT, unassembled ? &augsrc : nullptr);
        if (rc) return rc;
    }
    predict_account(h, Nstar, pan_ms, var_ms, want_var != 0, timed_rows);
    return 0;
}

}  // extern "C"
