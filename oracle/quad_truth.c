/* Quad-precision (IEEE binary128, gcc __float128 + libquadmath) evaluation of the exact-GP quantities of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- like everything under oracle/: only tests/ call it, as a checker.  It is NOT a restatement of GPy's
 * algorithm (that is oracle/gp_oracle.py, which follows the reference's call sites src/abstractMFGP.py:59-80,100-104,131-137 and
 * src/MFDataFusion.py:93-98,154-156 in fp64, the way GPy computes them).  This file evaluates the MATHEMATICAL quantities those call
 * sites ask for -- covariance matrix, log marginal likelihood, its gradient, predictive mean and latent variance -- on the same fp64
 * inputs with 113-bit arithmetic, so that its results are exact to far below one fp64 ulp even at cond(Ky) ~ 1e12.  The parity tests
 * use it to turn "two fp64 evaluations agree to tolerance t" into "each of them is within t of the true value", and to say which of
 * two disagreeing fp64 forms (GPy's explicit-inverse predictive variance against the triangular one) is the one that is off.
 *
 * Kernel description: the POD layout of include/mfgp.h / oracle/gp_oracle.py -- parts[4 f] = (type | ARD flag 0x100, col_begin,
 * col_end, term), theta = [variance_f, lengthscale(s)_f ...] per factor, K = sum over terms of the product of its factors.
 * Build: oracle/Makefile (gcc -O2 -fopenmp ... -lquadmath) -> oracle/libquad_truth.so
 */
#include <omp.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __float128 q_t;

/* threads of the parallel loops: set by the caller to the cores it may really use (a container's CPU quota is invisible to OpenMP, and
 * a team as wide as the machine on a 16-core share turns every one of the n fork-joins of the Cholesky into a scheduling storm) */
static int g_threads = 8;
void quad_set_threads(int t) { g_threads = t < 1 ? 1 : (t > 64 ? 64 : t); }

enum { K_RBF = 0, K_M32 = 1, K_M52 = 2, K_ARD = 0x100, MAX_PARTS = 16, MAX_THETA = 256 };

typedef struct {
    int nparts, nterms, P;
    int type[MAX_PARTS], c0[MAX_PARTS], c1[MAX_PARTS], term[MAX_PARTS], iv[MAX_PARTS], il[MAX_PARTS], nl[MAX_PARTS];
    q_t theta[MAX_THETA];
} spec_t;

static int make_spec(spec_t* s, int nparts, const int32_t* parts, const double* theta) {
    if (nparts < 1 || nparts > MAX_PARTS) return -1;
    s->nparts = nparts; s->nterms = 0;
    int i = 0;
    for (int f = 0; f < nparts; ++f) {
        s->type[f] = parts[4 * f]; s->c0[f] = parts[4 * f + 1]; s->c1[f] = parts[4 * f + 2]; s->term[f] = parts[4 * f + 3];
        if (s->term[f] + 1 > s->nterms) s->nterms = s->term[f] + 1;
        s->nl[f] = (s->type[f] & K_ARD) ? (s->c1[f] - s->c0[f]) : 1;
        s->iv[f] = i; s->il[f] = i + 1;
        i += 1 + s->nl[f];
        if (i > MAX_THETA) return -1;
    }
    s->P = i;
    for (int k = 0; k < i; ++k) s->theta[k] = (q_t)theta[k];
    return 0;
}

/* one factor at a pair of rows: value k_f and, if dl != NULL, its derivatives with respect to its lengthscale(s) */
static q_t factor(const spec_t* s, int f, const double* xa, const double* xb, q_t* dl) {
    const int base = s->type[f] & 0xff, ard = (s->type[f] & K_ARD) != 0;
    const q_t var = s->theta[s->iv[f]];
    q_t r2 = 0, w[64];
    const int nc = s->c1[f] - s->c0[f];
    for (int c = 0; c < nc; ++c) {
        const q_t ell = s->theta[s->il[f] + (ard ? c : 0)];
        const q_t dx = (q_t)xa[s->c0[f] + c] - (q_t)xb[s->c0[f] + c];
        const q_t t = dx * dx / (ell * ell);
        r2 += t;
        if (dl && c < 64) w[c] = t / ell;          /* Delta_c^2 / ell_c^3 */
    }
    const q_t r = sqrtq(r2);
    q_t k, g;                                      /* g: d k / d ell_c = g * Delta_c^2 / ell_c^3 */
    if (base == K_RBF) {
        k = var * expq(-0.5Q * r2); g = k;
    } else if (base == K_M32) {
        const q_t s3 = sqrtq(3.0Q), e = var * expq(-s3 * r);
        k = (1 + s3 * r) * e; g = 3 * e;
    } else {
        const q_t s5 = sqrtq(5.0Q), e = var * expq(-s5 * r);
        k = (1 + s5 * r + 5 * r2 / 3) * e; g = 5 * (1 + s5 * r) * e / 3;
    }
    if (dl) {
        if (ard) for (int c = 0; c < nc && c < 64; ++c) dl[c] = g * w[c];
        else { q_t sum = 0; for (int c = 0; c < nc && c < 64; ++c) sum += w[c]; dl[0] = g * sum; }
    }
    return k;
}

static q_t kval(const spec_t* s, const double* xa, const double* xb) {
    q_t tot = 0;
    for (int t = 0; t < s->nterms; ++t) {
        q_t p = 1; int any = 0;
        for (int f = 0; f < s->nparts; ++f) if (s->term[f] == t) { p *= factor(s, f, xa, xb, NULL); any = 1; }
        if (any) tot += p;
    }
    return tot;
}

/* in-place lower Cholesky of the n x n matrix A (row-major, lower part used); returns the failing pivot + 1, 0 on success */
static int cholesky(q_t* A, int n) {
    for (int j = 0; j < n; ++j) {
        q_t d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0)) return j + 1;
        d = sqrtq(d);
        A[(size_t)j * n + j] = d;
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int i = j + 1; i < n; ++i) {
            q_t v = A[(size_t)i * n + j];
            const q_t* ri = A + (size_t)i * n; const q_t* rj = A + (size_t)j * n;
            for (int k = 0; k < j; ++k) v -= ri[k] * rj[k];
            A[(size_t)i * n + j] = v / d;
        }
    }
    return 0;
}

static void solve_lower(const q_t* L, int n, q_t* b) {      /* b <- L^-1 b */
    for (int i = 0; i < n; ++i) {
        q_t v = b[i];
        for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * b[k];
        b[i] = v / L[(size_t)i * n + i];
    }
}
static void solve_upper_t(const q_t* L, int n, q_t* b) {    /* b <- L^-T b */
    for (int i = n - 1; i >= 0; --i) {
        q_t v = b[i];
        for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * b[k];
        b[i] = v / L[(size_t)i * n + i];
    }
}

/* Everything at once.  Outputs (each may be NULL): K (n x n, the covariance WITHOUT the diagonal shift), nlml, logdet (of Ky), alpha (n),
 * grad (P + 1: d nlml / d theta_k, then d nlml / d noise), mean (ns), var (ns: latent k** - ks Ky^-1 ks^T), Kinv (n x n).
 * Returns 0, -1 for a bad description, -2 out of memory, pivot + 1 if Ky is not positive definite in quad precision. */
int quad_gp(int n, int d, const double* X, const double* y, int nparts, const int32_t* parts, const double* theta, double noise, double jitter,
            int ns, const double* Xs, double* K_out, double* nlml, double* logdet, double* alpha_out, double* grad, double* mean, double* var,
            double* Kinv_out) {
    spec_t s;
    if (make_spec(&s, nparts, parts, theta) || n < 1 || d < 1) return -1;
    for (int f = 0; f < s.nparts; ++f) if (s.c0[f] < 0 || s.c1[f] > d || s.c1[f] - s.c0[f] > 64 || s.c1[f] <= s.c0[f]) return -1;
    const size_t nn = (size_t)n * n;
    q_t* L = (q_t*)malloc(nn * sizeof(q_t));
    q_t* a = (q_t*)malloc((size_t)n * sizeof(q_t));
    if (!L || !a) { free(L); free(a); return -2; }
    const q_t shift = (q_t)noise + (q_t)jitter;
#pragma omp parallel for schedule(dynamic, 8) num_threads(g_threads)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const q_t k = kval(&s, X + (size_t)i * d, X + (size_t)j * d);
            L[(size_t)i * n + j] = k + (i == j ? shift : 0);
            if (K_out) { K_out[(size_t)i * n + j] = (double)k; K_out[(size_t)j * n + i] = (double)k; }
        }
    const int piv = cholesky(L, n);
    if (piv) { free(L); free(a); return piv; }
    q_t ld = 0;
    for (int i = 0; i < n; ++i) ld += 2 * logq(L[(size_t)i * n + i]);
    for (int i = 0; i < n; ++i) a[i] = (q_t)y[i];
    solve_lower(L, n, a);
    solve_upper_t(L, n, a);
    q_t ya = 0;
    for (int i = 0; i < n; ++i) ya += (q_t)y[i] * a[i];
    if (nlml) *nlml = (double)(0.5Q * ya + 0.5Q * ld + 0.5Q * n * logq(2 * M_PIq));
    if (logdet) *logdet = (double)ld;
    if (alpha_out) for (int i = 0; i < n; ++i) alpha_out[i] = (double)a[i];

    if (grad || Kinv_out) {
        q_t* W = (q_t*)malloc(nn * sizeof(q_t));        /* Ky^-1, column by column */
        if (!W) { free(L); free(a); return -2; }
#pragma omp parallel num_threads(g_threads)
        {
            q_t* col = (q_t*)malloc((size_t)n * sizeof(q_t));
#pragma omp for schedule(dynamic, 4)
            for (int j = 0; j < n; ++j) {
                memset(col, 0, (size_t)n * sizeof(q_t));
                col[j] = 1;
                solve_lower(L, n, col);
                solve_upper_t(L, n, col);
                for (int i = 0; i < n; ++i) W[(size_t)i * n + j] = col[i];
            }
            free(col);
        }
        if (Kinv_out) for (size_t k = 0; k < nn; ++k) Kinv_out[k] = (double)W[k];
        if (grad) {
            /* d nlml / d p = 1/2 tr((Ky^-1 - alpha alpha^T) dK/dp) */
            const int P = s.P;
            q_t* acc = (q_t*)calloc((size_t)(P + 1) * n, sizeof(q_t));     /* per row, summed afterwards in a fixed order */
            if (!acc) { free(W); free(L); free(a); return -2; }
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
            for (int i = 0; i < n; ++i) {
                q_t kf[MAX_PARTS], dl[MAX_PARTS][64];
                q_t* ai = acc + (size_t)i * (P + 1);
                for (int j = 0; j < n; ++j) {
                    const q_t w = W[(size_t)i * n + j] - a[i] * a[j];
                    for (int f = 0; f < s.nparts; ++f) kf[f] = factor(&s, f, X + (size_t)i * d, X + (size_t)j * d, dl[f]);
                    for (int f = 0; f < s.nparts; ++f) {
                        q_t others = 1;
                        for (int g = 0; g < s.nparts; ++g) if (g != f && s.term[g] == s.term[f]) others *= kf[g];
                        ai[s.iv[f]] += w * others * kf[f] / s.theta[s.iv[f]];
                        for (int c = 0; c < s.nl[f]; ++c) ai[s.il[f] + c] += w * others * dl[f][c];
                    }
                    if (i == j) ai[P] += w;
                }
            }
            for (int k = 0; k <= P; ++k) {
                q_t t = 0;
                for (int i = 0; i < n; ++i) t += acc[(size_t)i * (P + 1) + k];
                grad[k] = (double)(0.5Q * t);
            }
            free(acc);
        }
        free(W);
    }

    if (ns > 0 && Xs && (mean || var)) {
        q_t kss = 0;
        for (int t = 0; t < s.nterms; ++t) {
            q_t p = 1; int any = 0;
            for (int f = 0; f < s.nparts; ++f) if (s.term[f] == t) { p *= s.theta[s.iv[f]]; any = 1; }
            if (any) kss += p;
        }
#pragma omp parallel num_threads(g_threads)
        {
            q_t* v = (q_t*)malloc((size_t)n * sizeof(q_t));
#pragma omp for schedule(dynamic, 4)
            for (int r = 0; r < ns; ++r) {
                q_t m = 0;
                for (int j = 0; j < n; ++j) { v[j] = kval(&s, Xs + (size_t)r * d, X + (size_t)j * d); m += v[j] * a[j]; }
                if (mean) mean[r] = (double)m;
                if (var) {
                    solve_lower(L, n, v);
                    q_t ss = 0;
                    for (int j = 0; j < n; ++j) ss += v[j] * v[j];
                    var[r] = (double)(kss - ss);
                }
            }
            free(v);
        }
    }
    free(L); free(a);
    return 0;
}
