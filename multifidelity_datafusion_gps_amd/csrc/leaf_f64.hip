// leaf_f64.hip -- Cholesky factor AND its inverse of one 128x128 diagonal block, inside one workgroup.
//
// This is the only serial piece of the factorisation (the chain of N pivots); everything above the
// leaves is tile GEMMs (gemm_f64.hip).  Replaces LAPACK dpotrf + dtrtri on the diagonal blocks
// behind GPy's jitchol / pdinv.
//
// The block lives in LDS (pitch 130 doubles: fragment reads of v_mfma_f64_16x16x4 hit 32 distinct
// 8-byte bank pairs).  Phase 1 factorises in place by 16-column panels: a 16x16 micro-Cholesky,
// a thread-per-row substitution for the rows below, and an MFMA rank-16 update of the trailing part.
// Phase 2 inverts in place (right-to-left block columns, LAPACK dtrti2 order): the eight 16x16
// diagonal inverses are solved concurrently, then X[ib][jb] = -sum_kb X[ib][kb] (L[kb][jb] X[jb][jb])
// on MFMA.
#include "mfgp_internal.h"

namespace mfgp {

constexpr int LP = 130;  // LDS pitch (doubles)

__device__ __forceinline__ d4_t mfma(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void mfgp_leaf_cholinv_f64(const double* __restrict__ A, double* Lout,
                                                                double* S, int ld, int blk,
                                                                double* logdet_part, int* info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sL = smem;             // 128 x LP
    double* sc = smem + 128 * LP;  // scratch: [0] pivot, [1..16] column, [32..47] 1/l_kk, [64..67] reduce

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 15;
    const int q = lane >> 4;
    const int64_t g0 = (int64_t)blk * NB * ld + (int64_t)blk * NB;  // offset of the diagonal block

    // ---- load (16 B per lane, whole rows coalesced) ------------------------------------------------
    for (int e = tid; e < NB * NB / 2; e += 256) {
        const int row = e >> 6, c2 = e & 63;
        const d2_t v = *reinterpret_cast<const d2_t*>(A + g0 + (int64_t)row * ld + 2 * c2);
        *reinterpret_cast<d2_t*>(sL + row * LP + 2 * c2) = v;
    }
    __syncthreads();

    // ---- phase 1: blocked Cholesky, 16-column panels --------------------------------------------------
    for (int jb = 0; jb < 8; ++jb) {
        const int base = jb * 16;
        // (a) micro-Cholesky of the 16x16 diagonal block: thread (i, k) owns one element
        {
            const int i = tid >> 4, k = tid & 15;
            double a = sL[(base + i) * LP + base + k];
            for (int j = 0; j < 16; ++j) {
                if (i == j && k == j) sc[0] = a;
                __syncthreads();
                double d = sc[0];
                if (!(d > 0.0)) {  // not positive definite (or NaN): record the first failing pivot
                    if (tid == 0 && *info == 0) *info = blk * NB + base + j + 1;
                    d = 1.0;
                }
                const double rinv = rsqrt(d);
                if (k == j && i >= j) {
                    a = (i == j) ? d * rinv : a * rinv;
                    sc[1 + i] = a;
                    if (i == j) sc[32 + j] = rinv;
                }
                __syncthreads();
                if (k > j && i >= k) a -= sc[1 + i] * sc[1 + k];
            }
            sL[(base + i) * LP + base + k] = (k <= i) ? a : 0.0;
        }
        __syncthreads();
        // (b) rows below the diagonal block: x L_jj^T = a by forward substitution, one thread per row
        {
            const int nrows = NB - base - 16;
            if (tid < nrows) {
                double* rowp = sL + (base + 16 + tid) * LP + base;
                double x[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) x[k] = rowp[k];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    double s = x[k];
#pragma unroll
                    for (int m = 0; m < k; ++m) s -= x[m] * sL[(base + k) * LP + base + m];
                    x[k] = s * sc[32 + k];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) rowp[k] = x[k];
            }
        }
        __syncthreads();
        // (c) trailing update on MFMA: C[ib][kb] -= L[ib][jb] L[kb][jb]^T for jb < kb <= ib < 8
        {
            const int m = 7 - jb;
            const int nblk = m * (m + 1) / 2;
            for (int idx = wave; idx < nblk; idx += 4) {
                // unrank idx -> (ii >= kk) in 0..m-1
                int ii = 0, rem = idx;
                while (rem > ii) { rem -= ii + 1; ++ii; }
                const int ib = jb + 1 + ii, kb = jb + 1 + rem;
                d4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = sL[(ib * 16 + q + 4 * r) * LP + kb * 16 + fr];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double av = -sL[(ib * 16 + fr) * LP + base + 4 * s + q];
                    const double bv = sL[(kb * 16 + fr) * LP + base + 4 * s + q];
                    acc = mfma(av, bv, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sL[(ib * 16 + q + 4 * r) * LP + kb * 16 + fr] = acc[r];
            }
        }
        __syncthreads();
    }

    // ---- write L (zeros above the diagonal) and the half log-determinant ------------------------------
    for (int e = tid; e < NB * NB / 2; e += 256) {
        const int row = e >> 6, c2 = e & 63;
        d2_t v = *reinterpret_cast<const d2_t*>(sL + row * LP + 2 * c2);
        if (2 * c2 > row) v.x = 0.0;
        if (2 * c2 + 1 > row) v.y = 0.0;
        *reinterpret_cast<d2_t*>(Lout + g0 + (int64_t)row * ld + 2 * c2) = v;
    }
    {
        double v = (tid < NB) ? log(sL[tid * LP + tid]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sc[64 + wave] = v;
        __syncthreads();
        if (tid == 0) logdet_part[blk] = (sc[64] + sc[65]) + (sc[66] + sc[67]);
    }
    __syncthreads();

    // ---- phase 2: in-place inverse ---------------------------------------------------------------------
    // (a) the eight 16x16 diagonal inverses, one thread per column, all at once
    {
        double x[16];
        const int b = tid >> 4, k = tid & 15;
        if (tid < 128) {
            const double* Lb = sL + (b * 16) * LP + b * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double s = (i == k) ? 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < i; ++m) s -= Lb[i * LP + m] * x[m];
                x[i] = s / Lb[i * LP + i];
            }
        }
        __syncthreads();
        if (tid < 128) {
            double* Lb = sL + (b * 16) * LP + b * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) Lb[i * LP + k] = x[i];  // zeros above the diagonal come out of the solve
        }
    }
    __syncthreads();
    // (b) block columns right to left
    for (int jb = 6; jb >= 0; --jb) {
        const int base = jb * 16;
        // T[kb] = L[kb][jb] * X[jb][jb]   (in place)
        for (int kb = jb + 1 + wave; kb < 8; kb += 4) {
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
            double av[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = sL[(kb * 16 + fr) * LP + base + 4 * s + q];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double bv = sL[(base + 4 * s + q) * LP + base + fr];  // X_jj[m = 4s+q][col fr]
                acc = mfma(av[s], bv, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sL[(kb * 16 + q + 4 * r) * LP + base + fr] = acc[r];
        }
        __syncthreads();
        // X[ib][jb] = - sum_{kb = jb+1..ib} X[ib][kb] * T[kb]   (accumulate in registers, then overwrite T)
        d4_t out[2];
        int nout = 0;
        for (int ib = jb + 1 + wave; ib < 8; ib += 4) {
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
            for (int kb = jb + 1; kb <= ib; ++kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = -sL[(ib * 16 + fr) * LP + kb * 16 + 4 * s + q];
                    const double b = sL[(kb * 16 + 4 * s + q) * LP + base + fr];
                    acc = mfma(a, b, acc);
                }
            }
            if (nout == 0) out[0] = acc; else out[1] = acc;
            ++nout;
        }
        __syncthreads();
        {
            int n = 0;
            for (int ib = jb + 1 + wave; ib < 8; ib += 4) {
                const d4_t acc = (n == 0) ? out[0] : out[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) sL[(ib * 16 + q + 4 * r) * LP + base + fr] = acc[r];
                ++n;
            }
        }
        __syncthreads();
    }

    // ---- write X mirrored: S[r][c] = X[max(r,c)][min(r,c)] -------------------------------------------
    for (int e = tid; e < NB * NB; e += 256) {
        const int row = e >> 7, col = e & 127;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        S[g0 + (int64_t)row * ld + col] = sL[hi * LP + lo];
    }
}

void launch_leaf(hipStream_t s, const double* A, double* Lout, double* S, int ld, int blk,
                 double* logdet_part, int* info) {
    constexpr size_t lds = (size_t)(128 * LP + 96) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfgp_leaf_cholinv_f64),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(mfgp_leaf_cholinv_f64, dim3(1), dim3(256), lds, s, A, Lout, S, ld, blk,
                       logdet_part, info);
}

}  // namespace mfgp
