// covariance.hip -- covariance-matrix construction and the fused NLML-gradient reduction.
//
// Replaces GPy's RBF.K / Matern32.K / Matern52.K, Prod.K, Add.K (call sites
// src/abstractMFGP.py:60,77-80) and kern.update_gradients_full + Gaussian.exact_inference_gradients
// (a8).  GPy forms r^2 as |x|^2+|x'|^2-2x.x' through a DGEMM and several N^2 temporaries; here each
// 64x64 output tile stages its two 64xD row blocks of X in LDS (transposed, so a thread reads its
// 4 rows / 4 columns with wide conflict-free LDS reads) and accumulates the squared differences
// directly -- one pass, 8 B written per element, nothing else touches HBM.
//
// Covariance structure (mfgp_kern_part): K = sum_terms prod_{f in term} var_f * shape_f(r_f / l_f),
// r_f^2 over the factor's column range; factors with the same range share r^2 ("leader").
#include <stdlib.h>
#include "mfgp_internal.h"

namespace mfgp {

constexpr int KT = 64;        // tile edge
constexpr int XP = KT + 2;    // LDS pitch of the transposed X blocks

// exp(x) for x <= 0 (every covariance shape here decays): Cody-Waite reduction by ln 2, degree-13 Taylor in
// Horner/FMA form, v_ldexp_f64 -- about 22 VALU instructions instead of ~90 for the general ocml exp, max
// relative error 2.2e-16 over [-745, 0] (measured against numpy on 4e6 points).  The K-build and the gradient
// reduction are co-limited by exp throughput, not by HBM, so this is their main lever.
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -746.0);      // n = -1076 below: p 2^-1076 < 2^-1075 rounds to an exact 0 (a clamp at -745 would return the smallest
                              // denormal for EVERY x <= -745 -- harmless in K, but a gradient weight r^2 / l^3 can be 1e300)
    const double n = rint(x * 1.4426950408889634);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// 1 / l^2 of a lengthscale, finite: below l ~ 1e-154 the reciprocal overflows, and 0 * inf on the diagonal (r = 0) would poison K with
// NaN where GPy's 0 / l = 0 gives the variance (the optimiser's softplus domain reaches l ~ 1e-308: a line search gone astray, handled
// like any other point).  SCALED_R2_MAX bounds a scaled squared distance wherever it multiplies a covariance that has already
// underflowed to 0 (gradient weights, Matern polynomials): 0 * 1e300 = 0, 0 * inf = NaN.  Neither touches a value in the normal range.
constexpr double SCALED_R2_MAX = 1e300;
__host__ __device__ __forceinline__ double inv_sq(double l) { return fmin(1.0 / (l * l), 1.7976931348623157e308); }

// stage rows [row0, row0+64) of X (row-major, D columns) transposed into LDS: sx[d*XP + r]
__device__ __forceinline__ void stage_rows(const double* __restrict__ X, int64_t row0, int D, double* sx) {
    const double* src = X + row0 * D;
    for (int e = threadIdx.x; e < KT * D; e += 256) {
        const int r = e / D, d = e - r * D;
        sx[d * XP + r] = src[e];
    }
}

// column index of a thread's c-th output column: two 16-byte pairs, 256 B contiguous per 16 lanes
__device__ __forceinline__ int col_of(int tx, int c) { return (c < 2) ? (2 * tx + c) : (32 + 2 * tx + (c - 2)); }

// squared distances of a thread's 4x4 pairs over the column range [c0, c1)
__device__ __forceinline__ void pair_r2(const double* sxi, const double* sxj, int ty, int tx, int c0, int c1,
                                        double (&r2)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) r2[e] = 0.0;
    for (int d = c0; d < c1; ++d) {
        const d2_t xi0 = *reinterpret_cast<const d2_t*>(sxi + d * XP + 4 * ty);
        const d2_t xi1 = *reinterpret_cast<const d2_t*>(sxi + d * XP + 4 * ty + 2);
        const d2_t xa = *reinterpret_cast<const d2_t*>(sxj + d * XP + 2 * tx);
        const d2_t xb = *reinterpret_cast<const d2_t*>(sxj + d * XP + 32 + 2 * tx);
        const double xi[4] = {xi0.x, xi0.y, xi1.x, xi1.y};
        const double xj[4] = {xa.x, xa.y, xb.x, xb.y};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double df = xi[r] - xj[c];
                r2[r * 4 + c] = __builtin_fma(df, df, r2[r * 4 + c]);
            }
    }
}

// the same with one inverse squared lengthscale per column (MFGP_KERN_ARD): r2[e] = sum_d (x_d - x'_d)^2 / l_d^2, and the
// share of ONE column d of it (the per-column gradient needs it)
// (the lengthscales are read as sp.theta[lbase + d - c0]: a pointer into the by-value kernel argument would send it to scratch)
__device__ __forceinline__ void pair_r2_ard(const KernSpecDev& sp, const double* sxi, const double* sxj, int ty, int tx, int c0,
                                            int c1, int lbase, double (&r2)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) r2[e] = 0.0;
    for (int d = c0; d < c1; ++d) {
        const double w = inv_sq(sp.theta[lbase + (d - c0)]);
        const d2_t xi0 = *reinterpret_cast<const d2_t*>(sxi + d * XP + 4 * ty);
        const d2_t xi1 = *reinterpret_cast<const d2_t*>(sxi + d * XP + 4 * ty + 2);
        const d2_t xa = *reinterpret_cast<const d2_t*>(sxj + d * XP + 2 * tx);
        const d2_t xb = *reinterpret_cast<const d2_t*>(sxj + d * XP + 32 + 2 * tx);
        const double xi[4] = {xi0.x, xi0.y, xi1.x, xi1.y};
        const double xj[4] = {xa.x, xa.y, xb.x, xb.y};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double df = xi[r] - xj[c];
                r2[r * 4 + c] = __builtin_fma(df * df, w, r2[r * 4 + c]);
            }
    }
}

// scaled squared distances of factor f (isotropic: r^2 / l^2 is applied by the caller through inv_l2; ARD: already scaled)
__device__ __forceinline__ double factor_r2(const KernSpecDev& sp, int f, const double* sxi, const double* sxj, int ty, int tx,
                                            double (&r2)[16]) {
    if (sp.nl[f] == 1) {
        pair_r2(sxi, sxj, ty, tx, sp.c0[f], sp.c1[f], r2);
        return inv_sq(sp.theta[sp.toff[f] + 1]);
    }
    pair_r2_ard(sp, sxi, sxj, ty, tx, sp.c0[f], sp.c1[f], sp.toff[f] + 1, r2);
    return 1.0;
}

// One factor of a product term for the 16 pairs of a thread (wave-uniform shape per call).  RBF factors only add
// to the term's exponent (expo) and variance product: a term of m RBF factors costs ONE exp per pair, not m
// (k1*k2 = s1 s2 exp(-r1^2/2l1^2 - r2^2/2l2^2)); Matern factors multiply prod directly.
__device__ __forceinline__ void apply_factor(int type, double var, double inv_l2, const double (&r2)[16],
                                             double (&prod)[16], double (&expo)[16], double& varprod) {
    if (type == MFGP_KERN_RBF) {
        varprod *= var;
#pragma unroll
        for (int e = 0; e < 16; ++e) expo[e] = __builtin_fma(-0.5 * inv_l2, r2[e], expo[e]);
    } else if (type == MFGP_KERN_MATERN32) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double s3r = 1.7320508075688772 * sqrt(fmin(r2[e] * inv_l2, SCALED_R2_MAX));
            prod[e] *= var * (1.0 + s3r) * exp_nonpos(-s3r);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double rs2 = fmin(r2[e] * inv_l2, SCALED_R2_MAX);
            const double s5r = 2.23606797749979 * sqrt(rs2);
            prod[e] *= var * (1.0 + s5r + (5.0 / 3.0) * rs2) * exp_nonpos(-s5r);
        }
    }
}

// close a product term: fold the accumulated RBF exponent and variance product into prod
__device__ __forceinline__ void finish_term(double (&prod)[16], double (&expo)[16], double& varprod, bool any_rbf) {
    if (any_rbf) {
#pragma unroll
        for (int e = 0; e < 16; ++e) prod[e] *= varprod * exp_nonpos(expo[e]);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) expo[e] = 0.0;
    varprod = 1.0;
}

// K of the 16 pairs: factor-major (runtime loop over factors, each re-accumulating its own r^2 from LDS:
// a few FMAs, and it keeps everything in ~100 VGPRs with no private-memory arrays)
__device__ __forceinline__ void cov_values(const KernSpecDev& sp, const double* sxi,
                                           const double* sxj, int ty, int tx, double (&K)[16]) {
    double prod[16], r2[16], expo[16], varprod = 1.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        K[e] = 0.0;
        prod[e] = 1.0;
        expo[e] = 0.0;
    }
    int cur = sp.term[0];
    bool any_rbf = false;
    for (int f = 0; f < sp.nf; ++f) {
        if (sp.term[f] != cur) {
            finish_term(prod, expo, varprod, any_rbf);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                K[e] += prod[e];
                prod[e] = 1.0;
            }
            cur = sp.term[f];
            any_rbf = false;
        }
        const double var = sp.theta[sp.toff[f]];
        const double inv_l2 = factor_r2(sp, f, sxi, sxj, ty, tx, r2);
        any_rbf = any_rbf || (sp.type[f] == MFGP_KERN_RBF);
        apply_factor(sp.type[f], var, inv_l2, r2, prod, expo, varprod);
    }
    finish_term(prod, expo, varprod, any_rbf);
#pragma unroll
    for (int e = 0; e < 16; ++e) K[e] += prod[e];
}

enum { MODE_TRI = 0, MODE_PANEL = 1, MODE_FULL = 2, MODE_ROWS = 3 };

// MODE_TRI   : lower-triangle tiles of Ky = K + (noise+jitter) I on the padded Np grid (identity padding)
// MODE_PANEL : Kx[r][c] = k(Xs[r], X[c]); rows from Xs (Nrows padded rows), 0 for c >= N
// MODE_FULL  : full K (no noise) into an N x N buffer, bounds-checked
// MODE_ROWS  : a block of FULL rows of Ky (all Np columns, noise + identity padding as MODE_TRI): the unit a rank
//              builds when K(X,X) is sharded by row blocks and all-gathered; bi is offset by `row_tile0`
template <int MODE>
__global__ __launch_bounds__(256) void mfgp_kbuild_f64(KernSpecDev sp, const double* __restrict__ Xr,
                                                       const double* __restrict__ Xc, int N, int Np,
                                                       double* __restrict__ out, int ld, int row_tile0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sxi = smem;
    double* sxj = smem + sp.D * XP;
    int bi, bj;
    if (MODE == MODE_TRI) {
        const int b = blockIdx.x;
        int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= b) ++i;
        while (i * (i + 1) / 2 > b) --i;
        bi = i;
        bj = b - i * (i + 1) / 2;
    } else {
        bi = blockIdx.y + row_tile0;
        bj = blockIdx.x;
    }
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    stage_rows(Xr, (int64_t)bi * KT, sp.D, sxi);
    stage_rows(Xc, (int64_t)bj * KT, sp.D, sxj);
    __syncthreads();

    double Kv[16];
    cov_values(sp, sxi, sxj, ty, tx, Kv);
    const double diag_add = (MODE == MODE_TRI || MODE == MODE_ROWS) ? (sp.theta[sp.np] + sp.theta[sp.np + 1]) : 0.0;

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = bi * KT + 4 * ty + r;
        double v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gj = bj * KT + col_of(tx, c);
            double k = Kv[r * 4 + c];
            if (MODE == MODE_TRI || MODE == MODE_ROWS) {
                if (gi >= N || gj >= N) k = (gi == gj) ? 1.0 : 0.0;
                else if (gi == gj) k += diag_add;
            } else if (MODE == MODE_PANEL) {
                if (gj >= N) k = 0.0;
            }
            v[c] = k;
        }
        if (MODE == MODE_FULL) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gj = bj * KT + col_of(tx, c);
                if (gi < N && gj < N) out[(int64_t)gi * ld + gj] = v[c];
            }
        } else {
            double* p = out + (int64_t)gi * ld + bj * KT;
            *reinterpret_cast<d2_t*>(p + 2 * tx) = (d2_t){v[0], v[1]};
            *reinterpret_cast<d2_t*>(p + 32 + 2 * tx) = (d2_t){v[2], v[3]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path for the two covariance structures the reference actually builds:
//     one RBF over a column range:                 K = v1 exp(-cb r2_B)
//     the NARGP composite k1(aug) k2(std) + k3(std): K = v1 exp(-(ca r2_A + cb r2_B)) + v2 exp(-cc r2_B)
// with the two squared distances (augmentation columns A, input columns B) accumulated ONCE per pair -- k2 and k3 share
// r2_B -- and every per-factor constant (v1 = s1 s2, ca = 1/2l1^2, ...) folded on the host.  The generic kernel above
// re-accumulates r^2 per factor and carries the kernel description in registers: 216 VGPRs, 2 waves per SIMD.
// ------------------------------------------------------------------------------------------------
struct Rbf2Spec {
    double v1, ca, cb, v2, cc, diag_add;
    int32_t a0, a1, b0, b1, has2, D;
};

struct Rbf2Batch { Rbf2Spec s[MFGP_BATCH_MAX]; };   // the parameter sets of a batched evaluation, by value (1.1 KB of kernel arguments)

template <int MODE>
__device__ __forceinline__ void kbuild_rbf2_body(const Rbf2Spec& sp, const double* __restrict__ Xr,
                                                 const double* __restrict__ Xc, int N, int Np,
                                                 double* __restrict__ out, int ld, int row_tile0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sxi = smem;
    double* sxj = smem + sp.D * XP;
    int bi, bj;
    if (MODE == MODE_TRI) {
        const int b = blockIdx.x;
        int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= b) ++i;
        while (i * (i + 1) / 2 > b) --i;
        bi = i;
        bj = b - i * (i + 1) / 2;
    } else {
        bi = blockIdx.y + row_tile0;
        bj = blockIdx.x;
    }
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    stage_rows(Xr, (int64_t)bi * KT, sp.D, sxi);
    stage_rows(Xc, (int64_t)bj * KT, sp.D, sxj);
    __syncthreads();

    double r2a[16], r2b[16];
    pair_r2(sxi, sxj, ty, tx, sp.b0, sp.b1, r2b);
    if (sp.a1 > sp.a0) {
        pair_r2(sxi, sxj, ty, tx, sp.a0, sp.a1, r2a);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) r2a[e] = 0.0;
    }
    double Kv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Kv[e] = sp.v1 * exp_nonpos(-__builtin_fma(sp.ca, r2a[e], sp.cb * r2b[e]));
    if (sp.has2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Kv[e] = __builtin_fma(sp.v2, exp_nonpos(-sp.cc * r2b[e]), Kv[e]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = bi * KT + 4 * ty + r;
        double v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gj = bj * KT + col_of(tx, c);
            double k = Kv[r * 4 + c];
            if (MODE == MODE_TRI || MODE == MODE_ROWS) {
                if (gi >= N || gj >= N) k = (gi == gj) ? 1.0 : 0.0;
                else if (gi == gj) k += sp.diag_add;
            } else if (MODE == MODE_PANEL) {
                if (gj >= N) k = 0.0;
            }
            v[c] = k;
        }
        if (MODE == MODE_FULL) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gj = bj * KT + col_of(tx, c);
                if (gi < N && gj < N) out[(int64_t)gi * ld + gj] = v[c];
            }
        } else {
            double* p = out + (int64_t)gi * ld + bj * KT;
            *reinterpret_cast<d2_t*>(p + 2 * tx) = (d2_t){v[0], v[1]};
            *reinterpret_cast<d2_t*>(p + 32 + 2 * tx) = (d2_t){v[2], v[3]};
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void mfgp_kbuild_rbf2_f64(Rbf2Spec sp, const double* __restrict__ Xr,
                                                            const double* __restrict__ Xc, int N, int Np,
                                                            double* __restrict__ out, int ld, int row_tile0) {
    kbuild_rbf2_body<MODE>(sp, Xr, Xc, N, Np, out, ld, row_tile0);
}
// the lower-triangle build for the B parameter sets of a batched evaluation: blockIdx.y = set, its matrix bstride elements on
__global__ __launch_bounds__(256) void mfgp_kbuild_rbf2_batch_f64(Rbf2Batch specs, const double* __restrict__ X, int N, int Np,
                                                                  double* __restrict__ out, int ld, long long bstride) {
    const Rbf2Spec sp = specs.s[blockIdx.y];
    kbuild_rbf2_body<MODE_TRI>(sp, X, X, N, Np, out + blockIdx.y * bstride, ld, 0);
}

// does the description match the fast path?  (host side; the generic kernel remains for everything else: Matern factors,
// other products / sums)
static bool rbf2_match(const KernSpecDev& s, Rbf2Spec& o) {
    static const bool on = !(getenv("MFGP_KBUILD_FAST") && atoi(getenv("MFGP_KBUILD_FAST")) == 0);
    if (!on) return false;
    for (int f = 0; f < s.nf; ++f)
        if (s.type[f] != MFGP_KERN_RBF || s.nl[f] != 1) return false;   // (ARD factors take the generic kernels)
    o = Rbf2Spec{};
    o.D = s.D;
    o.diag_add = s.theta[s.np] + s.theta[s.np + 1];
    auto half_inv_l2 = [&](int f) { return 0.5 * inv_sq(s.theta[2 * f + 1]); };
    if (s.nf == 1) {
        o.b0 = s.c0[0]; o.b1 = s.c1[0];
        o.v1 = s.theta[0]; o.cb = half_inv_l2(0);
        return true;
    }
    if (s.nf == 3 && s.term[0] == s.term[1] && s.term[2] != s.term[1] && s.c0[1] == s.c0[2] && s.c1[1] == s.c1[2]) {
        o.a0 = s.c0[0]; o.a1 = s.c1[0];
        o.b0 = s.c0[1]; o.b1 = s.c1[1];
        o.v1 = s.theta[0] * s.theta[2]; o.ca = half_inv_l2(0); o.cb = half_inv_l2(1);
        o.v2 = s.theta[4]; o.cc = half_inv_l2(2); o.has2 = 1;
        return true;
    }
    return false;
}

static size_t kb_lds(int D) { return (size_t)2 * D * XP * sizeof(double); }

void launch_kbuild_tri(hipStream_t s, const KernSpecDev& spec, const double* X,
                       int N, int Np, double* A, int ld) {
    const int nt = Np / KT;
    Rbf2Spec f;
    if (rbf2_match(spec, f)) {
        hipLaunchKernelGGL((mfgp_kbuild_rbf2_f64<MODE_TRI>), dim3(nt * (nt + 1) / 2), dim3(256), kb_lds(spec.D), s, f, X, X,
                           N, Np, A, ld, 0);
        return;
    }
    hipLaunchKernelGGL((mfgp_kbuild_f64<MODE_TRI>), dim3(nt * (nt + 1) / 2), dim3(256), kb_lds(spec.D), s,
                       spec, X, X, N, Np, A, ld, 0);
}
void launch_kbuild_tri_batch(hipStream_t s, const KernSpecDev* specs, int nbatch, const double* X, int N, int Np, double* A,
                             int ld, long long bstride) {
    const int nt = Np / KT;
    Rbf2Batch fb;
    bool fast = nbatch <= MFGP_BATCH_MAX;
    for (int b = 0; fast && b < nbatch; ++b) fast = rbf2_match(specs[b], fb.s[b]);
    if (fast) {
        hipLaunchKernelGGL(mfgp_kbuild_rbf2_batch_f64, dim3(nt * (nt + 1) / 2, nbatch), dim3(256), kb_lds(specs[0].D), s, fb, X, N,
                           Np, A, ld, bstride);
        return;
    }
    for (int b = 0; b < nbatch; ++b) launch_kbuild_tri(s, specs[b], X, N, Np, A + b * bstride, ld);
}
void launch_kbuild_panel(hipStream_t s, const KernSpecDev& spec, const double* Xs, int Nsp,
                         const double* X, int N, int Np, double* Kx, int ld) {
    Rbf2Spec f;
    if (rbf2_match(spec, f)) {
        hipLaunchKernelGGL((mfgp_kbuild_rbf2_f64<MODE_PANEL>), dim3(Np / KT, Nsp / KT), dim3(256), kb_lds(spec.D), s, f, Xs, X,
                           N, Np, Kx, ld, 0);
        return;
    }
    hipLaunchKernelGGL((mfgp_kbuild_f64<MODE_PANEL>), dim3(Np / KT, Nsp / KT), dim3(256), kb_lds(spec.D), s,
                       spec, Xs, X, N, Np, Kx, ld, 0);
}
// A panel of 1, 2 or 4 test rows (the N* = 1 callback of the reference's maximiser, the low-fidelity level's row of a level-chained
// predict, the new row of a rank-1 append): one thread per training point instead of 64 x 64 tiles of which one row is wanted.  The
// arithmetic of a pair is kbuild_rbf2_body's, operation for operation (squared distances by fma in column order, test minus training
// point; the same exp_nonpos) -- the panel rows are the same bits as the tile kernel's, and so are the means formed from them.
// The test rows are read where they are (FewRows, mfgp_internal.h): a packed block; the stencil rows Xc[t / c] + offs[t % c] of a
// level-chained predict (what mfgp_stencil_rows_f64 would write first); or the augmented rows [Xc[t] | m[t]] of its next level (what
// mfgp_assemble_aug_f64 would) -- two launches less on the chained N* = 1 call.
template <int R>
__global__ __launch_bounds__(256) void mfgp_kpanel_few_rbf2_f64(Rbf2Spec sp, FewRows q, const double* __restrict__ X, int N, int Np,
                                                                double* __restrict__ out, int ld) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    const int D = sp.D;
    const double* xj = X + (int64_t)j * D;
    // the R test rows (wave-uniform: scalar loads): row r = [pa[r][0 .. da) + po[r][0 .. da) | pm[r][0 .. dm)], a zero row from n on
    const double *pa[R], *po[R], *pm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = (int)q.t0 + r;
        pa[r] = q.a + (int64_t)(t / q.c) * q.da;
        po[r] = q.offs ? q.offs + (int64_t)(t % q.c) * q.da : nullptr;
        pm[r] = q.m ? q.m + (int64_t)t * q.dm : nullptr;
    }
    double r2a[R], r2b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) r2a[r] = r2b[r] = 0.0;
    const bool two = sp.a1 > sp.a0;
    for (int d0 = 0; d0 < D; d0 += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = xj[min(d0 + u, D - 1)];       // four loads in flight (the rows of X are 8 D bytes apart)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = d0 + u;
            const bool inb = d < D && d >= sp.b0 && d < sp.b1, ina = d < D && two && d >= sp.a0 && d < sp.a1;
            if (!(inb || ina)) continue;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double xs = 0.0;
                if (r < q.n) xs = d < q.da ? (po[r] ? pa[r][d] + po[r][d] : pa[r][d]) : pm[r][d - q.da];
                const double df = xs - x[u];
                if (inb) r2b[r] = __builtin_fma(df, df, r2b[r]);      // (ascending d within either range: kbuild_rbf2_body's order)
                if (ina) r2a[r] = __builtin_fma(df, df, r2a[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double k = sp.v1 * exp_nonpos(-__builtin_fma(sp.ca, r2a[r], sp.cb * r2b[r]));
        if (sp.has2) k = __builtin_fma(sp.v2, exp_nonpos(-sp.cc * r2b[r]), k);
        if (j >= N) k = 0.0;
        out[(int64_t)r * ld + j] = k;
    }
}
bool kbuild_panel_few_ok(const KernSpecDev& spec) {
    Rbf2Spec f;
    return rbf2_match(spec, f);
}
// -> false: the kernel description is not one of the fast path's (the caller builds the 64-row tile panel instead)
bool launch_kbuild_panel_few(hipStream_t s, const KernSpecDev& spec, const FewRows& q, int R, const double* X, int N, int Np,
                             double* Kx, int ld) {
    Rbf2Spec f;
    if (R > 4 || !rbf2_match(spec, f)) return false;
    const dim3 grid((Np + 255) / 256), blk(256);
    if (R <= 1) hipLaunchKernelGGL((mfgp_kpanel_few_rbf2_f64<1>), grid, blk, 0, s, f, q, X, N, Np, Kx, ld);
    else if (R == 2) hipLaunchKernelGGL((mfgp_kpanel_few_rbf2_f64<2>), grid, blk, 0, s, f, q, X, N, Np, Kx, ld);
    else hipLaunchKernelGGL((mfgp_kpanel_few_rbf2_f64<4>), grid, blk, 0, s, f, q, X, N, Np, Kx, ld);
    return true;
}
void launch_kbuild_full(hipStream_t s, const KernSpecDev& spec, const double* X,
                        int N, int Np, double* out, int ld) {
    hipLaunchKernelGGL((mfgp_kbuild_f64<MODE_FULL>), dim3(Np / KT, Np / KT), dim3(256), kb_lds(spec.D), s,
                       spec, X, X, N, Np, out, ld, 0);
}
void launch_kbuild_rows(hipStream_t s, const KernSpecDev& spec, const double* X, int N, int Np,
                        double* A, int ld, int row_begin, int row_end) {
    Rbf2Spec f;
    if (rbf2_match(spec, f)) {
        hipLaunchKernelGGL((mfgp_kbuild_rbf2_f64<MODE_ROWS>), dim3(Np / KT, (row_end - row_begin) / KT), dim3(256),
                           kb_lds(spec.D), s, f, X, X, N, Np, A, ld, row_begin / KT);
        return;
    }
    hipLaunchKernelGGL((mfgp_kbuild_f64<MODE_ROWS>), dim3(Np / KT, (row_end - row_begin) / KT), dim3(256), kb_lds(spec.D), s,
                       spec, X, X, N, Np, A, ld, row_begin / KT);
}

// ------------------------------------------------------------------------------------------------
// gradient reduction: with G = alpha alpha^T - Ky^-1 (lower-triangle tiles, off-diagonal tiles x2)
//   Sv_f = sum G_ij * termprod_f(i,j)          -> dNLML/dvar_f = -0.5 * Sv_f / var_f
//   Sl_f = sum G_ij * termprod_f(i,j) * g_f    -> dNLML/dl_f   = -0.5 * Sl_f / l_f
//   Sn   = sum_i G_ii                          -> dNLML/dnoise = -0.5 * Sn
// Per-tile partials go to `partials`; a second single-block kernel adds them in a fixed order
// (bitwise reproducible) and applies the factors.
// ------------------------------------------------------------------------------------------------
constexpr int NSUM = MFGP_MAX_THETA + 1;   // stride of a tile's partial sums: slot i < P = parameter i (include/mfgp.h layout), slot P = noise

// g / s of one factor for the 16 pairs, with g = (dk/dl) l / k of the ISOTROPIC factor and s the scaled squared distance:
// the column shares of an ARD factor are (dk/dl_d) l_d / k = (g / s) (x_d - x'_d)^2 / l_d^2  (finite at s = 0)
__device__ __forceinline__ void factor_logderiv_over_s(int type, const double (&s)[16], double (&rho)[16]) {
    if (type == MFGP_KERN_RBF) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rho[e] = 1.0;
    } else if (type == MFGP_KERN_MATERN32) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rho[e] = 3.0 / (1.0 + 1.7320508075688772 * sqrt(s[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double s5r = 2.23606797749979 * sqrt(s[e]);
            rho[e] = (5.0 / 3.0) * (1.0 + s5r) / (1.0 + s5r + (5.0 / 3.0) * s[e]);
        }
    }
}

// sum of v over the 256 threads of the block in a fixed order (bitwise reproducible) -> *dst, by thread 0
__device__ __forceinline__ void block_sum_to(double v, double* sred, double* dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) *dst = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    __syncthreads();
}

// A SHARDED evaluation (mfgp_eval_sharded) reduces only the tiles whose rows of K^-1 this rank holds: `shard` = rank | size << 16
// (0: every tile); the other tiles' slots of `partials` stay as the caller zeroed them, the ranks' arrays are then summed
// (x + 0 = x: the sum is bitwise the single rank's array) and finished as usual.  Same ownership rule as plan.h shard_owner.
__device__ __forceinline__ bool tile_row_is_mine(int bi, int shard) {
    const int size = shard >> 16, rank = shard & 0xffff;
    if (size <= 1) return true;
    const int x = ((bi * KT) / 128) % (2 * size);
    return (x < size ? x : 2 * size - 1 - x) == rank;
}

__global__ __launch_bounds__(256) void mfgp_grad_tiles_f64(KernSpecDev sp, const double* __restrict__ X,
                                                           const double* __restrict__ Kinv, int ld,
                                                           const double* __restrict__ alpha, int N,
                                                           double* __restrict__ partials, int shard) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sxi = smem;
    double* sxj = smem + sp.D * XP;
    double* sred = smem + 2 * sp.D * XP;  // [4] wave sums
    const int b = blockIdx.x;
    int bi = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
    while (bi * (bi + 1) / 2 > b) --bi;
    if (!tile_row_is_mine(bi, shard)) return;
    const int bj = b - bi * (bi + 1) / 2;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    double* out = partials + (int64_t)b * NSUM;
    stage_rows(X, (int64_t)bi * KT, sp.D, sxi);
    stage_rows(X, (int64_t)bj * KT, sp.D, sxj);
    __syncthreads();

    // G = w * (alpha_i alpha_j - Kinv_ij), zero outside the real N x N block
    const double w = (bi == bj) ? 1.0 : 2.0;
    double G[16], sn = 0.0;
    {
        double aj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) aj[c] = alpha[bj * KT + col_of(tx, c)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = bi * KT + 4 * ty + r;
            const double ai = alpha[gi];
            const double* kp = Kinv + (int64_t)gi * ld + bj * KT;
            const d2_t k01 = *reinterpret_cast<const d2_t*>(kp + 2 * tx);
            const d2_t k23 = *reinterpret_cast<const d2_t*>(kp + 32 + 2 * tx);
            const double kin[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gj = bj * KT + col_of(tx, c);
                const bool valid = (gi < N) && (gj < N);
                const double g = valid ? w * (ai * aj[c] - kin[c]) : 0.0;
                G[r * 4 + c] = g;
                if (gi == gj) sn += g;
            }
        }
    }
    block_sum_to(sn, sred, out + sp.np);

    // term by term: pass A forms the term product, pass B the per-factor (per-column, for ARD factors) lengthscale weights
    int f0 = 0;
    while (f0 < sp.nf) {
        int f1 = f0 + 1;
        while (f1 < sp.nf && sp.term[f1] == sp.term[f0]) ++f1;
        double prod[16], r2[16], expo[16], varprod = 1.0;
        bool any_rbf = false;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            prod[e] = 1.0;
            expo[e] = 0.0;
        }
        for (int f = f0; f < f1; ++f) {
            const double inv_l2 = factor_r2(sp, f, sxi, sxj, ty, tx, r2);
            any_rbf = any_rbf || (sp.type[f] == MFGP_KERN_RBF);
            apply_factor(sp.type[f], sp.theta[sp.toff[f]], inv_l2, r2, prod, expo, varprod);
        }
        finish_term(prod, expo, varprod, any_rbf);
        double sv = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            prod[e] *= G[e];
            sv += prod[e];
        }
        for (int f = f0; f < f1; ++f) {
            block_sum_to(sv, sred, out + sp.toff[f]);          // every factor of the term shares sum G K_term
            double rho[16];
            const double inv_l2 = factor_r2(sp, f, sxi, sxj, ty, tx, r2);
#pragma unroll
            for (int e = 0; e < 16; ++e) r2[e] = fmin(r2[e] * inv_l2, SCALED_R2_MAX);      // scaled squared distance s
            factor_logderiv_over_s(sp.type[f], r2, rho);
            if (sp.nl[f] == 1) {
                double sl = 0.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) sl = __builtin_fma(prod[e] * rho[e], r2[e], sl);
                block_sum_to(sl, sred, out + sp.toff[f] + 1);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) rho[e] *= prod[e];
                for (int d = sp.c0[f]; d < sp.c1[f]; ++d) {    // column d's share of s: (x_d - x'_d)^2 / l_d^2
                    double col[16];
                    pair_r2_ard(sp, sxi, sxj, ty, tx, d, d + 1, sp.toff[f] + 1 + (d - sp.c0[f]), col);
                    double sl = 0.0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) sl = __builtin_fma(rho[e], fmin(col[e], SCALED_R2_MAX), sl);
                    block_sum_to(sl, sred, out + sp.toff[f] + 1 + (d - sp.c0[f]));
                }
            }
        }
        f0 = f1;
    }
}

__global__ __launch_bounds__(256) void mfgp_grad_finish_f64(KernSpecDev sp,
                                                            const double* __restrict__ partials, int nblocks,
                                                            double* __restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int i = blockIdx.x;  // one block per sum, fixed summation order -> bitwise reproducible
    double s = 0.0;
    for (int b = tid; b < nblocks; b += 256) s += partials[(int64_t)b * NSUM + i];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double S = red[0];
        if (i == sp.np) out[sp.np] = -0.5 * S;                     // noise
        else if (i < sp.np) out[i] = -0.5 * S / sp.theta[i];      // slot i = parameter i: sum G K_term / var, sum G K_term g / l
    }
}

// Fast path of the gradient reduction for the same two structures as mfgp_kbuild_rbf2_f64: with K1 = v1 exp(-(ca r2_A + cb
// r2_B)) and K2 = v2 exp(-cc r2_B) every sum the seven derivatives need is one of
//     s1 = sum G K1,  s1a = sum G K1 r2_A,  s1b = sum G K1 r2_B,  s2 = sum G K2,  s2b = sum G K2 r2_B,  sn = sum_i G_ii
// (dk/dl l/k = r^2/l^2 for an RBF factor; both factors of the product term share s1) -- ONE pass over the pairs instead of
// the generic kernel's two passes per term with r^2 re-accumulated per factor.  Same partials layout as the generic kernel.
__device__ __forceinline__ void grad_rbf2_body(const Rbf2Spec& sp, const double* __restrict__ X,
                                               const double* __restrict__ Kinv, int ld,
                                               const double* __restrict__ alpha, int N,
                                               double* __restrict__ partials, int shard = 0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sxi = smem;
    double* sxj = smem + sp.D * XP;
    double* red = smem + 2 * sp.D * XP;  // [6][256]
    const int b = blockIdx.x;
    int bi = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
    while (bi * (bi + 1) / 2 > b) --bi;
    if (!tile_row_is_mine(bi, shard)) return;
    const int bj = b - bi * (bi + 1) / 2;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    stage_rows(X, (int64_t)bi * KT, sp.D, sxi);
    stage_rows(X, (int64_t)bj * KT, sp.D, sxj);
    __syncthreads();

    const double w = (bi == bj) ? 1.0 : 2.0;   // off-diagonal tiles stand for their mirror image too
    double G[16], sn = 0.0;
    {
        double aj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) aj[c] = alpha[bj * KT + col_of(tx, c)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = bi * KT + 4 * ty + r;
            const double ai = alpha[gi];
            const double* kp = Kinv + (int64_t)gi * ld + bj * KT;
            const d2_t k01 = *reinterpret_cast<const d2_t*>(kp + 2 * tx);
            const d2_t k23 = *reinterpret_cast<const d2_t*>(kp + 32 + 2 * tx);
            const double kin[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int gj = bj * KT + col_of(tx, c);
                const double g = ((gi < N) && (gj < N)) ? w * (ai * aj[c] - kin[c]) : 0.0;
                G[r * 4 + c] = g;
                if (gi == gj) sn += g;
            }
        }
    }
    double r2a[16], r2b[16];
    pair_r2(sxi, sxj, ty, tx, sp.b0, sp.b1, r2b);
    if (sp.a1 > sp.a0) {
        pair_r2(sxi, sxj, ty, tx, sp.a0, sp.a1, r2a);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) r2a[e] = 0.0;
    }
    double s1 = 0.0, s1a = 0.0, s1b = 0.0, s2 = 0.0, s2b = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const double gk = G[e] * (sp.v1 * exp_nonpos(-__builtin_fma(sp.ca, r2a[e], sp.cb * r2b[e])));
        s1 += gk;
        s1a = __builtin_fma(gk, r2a[e], s1a);
        s1b = __builtin_fma(gk, r2b[e], s1b);
    }
    if (sp.has2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double gk = G[e] * (sp.v2 * exp_nonpos(-sp.cc * r2b[e]));
            s2 += gk;
            s2b = __builtin_fma(gk, r2b[e], s2b);
        }
    }
    red[0 * 256 + tid] = s1; red[1 * 256 + tid] = s1a; red[2 * 256 + tid] = s1b;
    red[3 * 256 + tid] = s2; red[4 * 256 + tid] = s2b; red[5 * 256 + tid] = sn;
    __syncthreads();
    // fixed-order block reduction (bitwise reproducible), then scatter into the generic layout [2f] = Sv_f, [2f+1] = Sl_f
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < 6; i += 4) {
        double v = (red[i * 256 + lane] + red[i * 256 + 64 + lane]) + (red[i * 256 + 128 + lane] + red[i * 256 + 192 + lane]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) {
            double* out = partials + (int64_t)b * NSUM;
            if (sp.has2) {   // factors: 0 = k1 (columns A), 1 = k2 (columns B), 2 = k3 (columns B)
                if (i == 0) { out[0] = v; out[2] = v; }
                if (i == 1) out[1] = v * (2.0 * sp.ca);
                if (i == 2) out[3] = v * (2.0 * sp.cb);
                if (i == 3) out[4] = v;
                if (i == 4) out[5] = v * (2.0 * sp.cc);
            } else {
                if (i == 0) out[0] = v;
                if (i == 2) out[1] = v * (2.0 * sp.cb);
            }
            if (i == 5) out[sp.has2 ? 6 : 2] = v;       // noise slot = P (6 for the composite, 2 for one RBF)
        }
    }
}

__global__ __launch_bounds__(256) void mfgp_grad_rbf2_f64(Rbf2Spec sp, const double* __restrict__ X,
                                                          const double* __restrict__ Kinv, int ld,
                                                          const double* __restrict__ alpha, int N,
                                                          double* __restrict__ partials, int shard) {
    grad_rbf2_body(sp, X, Kinv, ld, alpha, N, partials, shard);
}
// the B sets of a batched evaluation: blockIdx.y = set
__global__ __launch_bounds__(256) void mfgp_grad_rbf2_batch_f64(Rbf2Batch specs, const double* __restrict__ X,
                                                                const double* __restrict__ Kinv, long long kstride, int ld,
                                                                const double* __restrict__ alpha, long long astride, int N,
                                                                double* __restrict__ partials, long long pstride) {
    const Rbf2Spec sp = specs.s[blockIdx.y];
    grad_rbf2_body(sp, X, Kinv + blockIdx.y * kstride, ld, alpha + blockIdx.y * astride, N, partials + blockIdx.y * pstride);
}
// mfgp_grad_finish_f64 for the B sets: blockIdx.y = set; the parameters it divides by come from device-readable memory
__global__ __launch_bounds__(256) void mfgp_grad_finish_batch_f64(int np, const double* __restrict__ thetas, int tstride,
                                                                  const double* __restrict__ partials, long long pstride,
                                                                  int nblocks, double* __restrict__ out, int ostride) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int i = blockIdx.x;
    partials += blockIdx.y * pstride;
    double s = 0.0;
    for (int b = tid; b < nblocks; b += 256) s += partials[(int64_t)b * NSUM + i];
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double S = red[0];
        double* o = out + (int64_t)blockIdx.y * ostride;
        if (i == np) o[np] = -0.5 * S;
        else if (i < np) o[i] = -0.5 * S / thetas[(int64_t)blockIdx.y * tstride + i];
    }
}

int grad_num_partials(int Np) {
    const int nt = Np / KT;
    return nt * (nt + 1) / 2;
}

void launch_grad_tiles(hipStream_t s, const KernSpecDev& spec, const double* X, const double* Kinv, int ld,
                       const double* alpha, int N, int Np, double* partials, int shard_rank, int shard_size) {
    const int nb = grad_num_partials(Np);
    const size_t lds = kb_lds(spec.D) + (size_t)8 * sizeof(double);
    const int shard = shard_size > 1 ? (shard_rank | (shard_size << 16)) : 0;
    Rbf2Spec f;
    if (rbf2_match(spec, f)) {
        hipLaunchKernelGGL(mfgp_grad_rbf2_f64, dim3(nb), dim3(256), kb_lds(spec.D) + (size_t)256 * 6 * sizeof(double), s, f, X,
                           Kinv, ld, alpha, N, partials, shard);
    } else {
        hipLaunchKernelGGL(mfgp_grad_tiles_f64, dim3(nb), dim3(256), lds, s, spec, X, Kinv, ld, alpha, N,
                           partials, shard);
    }
}
void launch_grad_finish(hipStream_t s, const KernSpecDev& spec, const double* partials, int Np, double* out) {
    hipLaunchKernelGGL(mfgp_grad_finish_f64, dim3(spec.np + 1), dim3(256), 0, s, spec, partials, grad_num_partials(Np), out);
}
void launch_grad(hipStream_t s, const KernSpecDev& spec, const double* X,
                 const double* Kinv, int ld, const double* alpha, int N, int Np, double* partials,
                 double* out) {
    launch_grad_tiles(s, spec, X, Kinv, ld, alpha, N, Np, partials, 0, 1);
    launch_grad_finish(s, spec, partials, Np, out);
}

void launch_grad_batch(hipStream_t s, const KernSpecDev* specs, int nbatch, const double* X, const double* Kinv, long long kstride,
                       int ld, const double* alpha, long long astride, int N, int Np, double* partials, long long pstride,
                       double* out, int ostride, const double* thetas, int tstride) {
    const int nb = grad_num_partials(Np);
    Rbf2Batch fb;
    bool fast = nbatch <= MFGP_BATCH_MAX;
    for (int b = 0; fast && b < nbatch; ++b) fast = rbf2_match(specs[b], fb.s[b]);
    if (!fast) {
        for (int b = 0; b < nbatch; ++b)
            launch_grad(s, specs[b], X, Kinv + b * kstride, ld, alpha + b * astride, N, Np, partials + b * pstride,
                        out + (int64_t)b * ostride);
        return;
    }
    hipLaunchKernelGGL(mfgp_grad_rbf2_batch_f64, dim3(nb, nbatch), dim3(256), kb_lds(specs[0].D) + (size_t)256 * 6 * sizeof(double),
                       s, fb, X, Kinv, kstride, ld, alpha, astride, N, partials, pstride);
    hipLaunchKernelGGL(mfgp_grad_finish_batch_f64, dim3(specs[0].np + 1, nbatch), dim3(256), 0, s, specs[0].np, thetas, tstride,
                       partials, pstride, nb, out, ostride);
}

// var[i] = max(kss - ss[i], 1e-15) + add,  kss = sum_terms prod var_f  (GPy Kdiag of a stationary kernel)
__global__ void mfgp_finish_var_f64(KernSpecDev sp,
                                    const double* __restrict__ ss, double* __restrict__ var, int n, double add) {
    double kss = 0.0, prod = 1.0;
    int cur = sp.term[0];
    for (int f = 0; f < sp.nf; ++f) {
        if (sp.term[f] != cur) {
            kss += prod;
            prod = 1.0;
            cur = sp.term[f];
        }
        prod *= sp.theta[sp.toff[f]];
    }
    kss += prod;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v = kss - ss[i];
        if (!(v > 1e-15)) v = 1e-15;
        var[i] = v + add;
    }
}

void launch_finish_var(hipStream_t s, const KernSpecDev& spec, const double* ss,
                       double* var, int n, double add) {
    hipLaunchKernelGGL(mfgp_finish_var_f64, dim3((n + 255) / 256), dim3(256), 0, s, spec, ss, var, n, add);
}

}  // namespace mfgp
