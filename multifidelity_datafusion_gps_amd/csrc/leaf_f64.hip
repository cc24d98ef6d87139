// leaf_f64.hip -- Cholesky factor AND its inverse of one 128x128 diagonal block, inside one workgroup.
//
// This is the only serial piece of the factorisation (the chain of N pivots); everything above the
// leaves is tile GEMMs (gemm_f64.hip).  Replaces LAPACK dpotrf + dtrtri on the diagonal blocks
// behind GPy's jitchol / pdinv.
//
// The block lives in LDS (pitch 130 doubles: fragment reads of v_mfma_f64_16x16x4 hit 32 distinct
// 8-byte bank pairs).  8 waves.
// Phase 1 factorises in place by 16-column panels with look-ahead:
//   * the 16x16 diagonal micro-Cholesky runs in the REGISTERS of wave 0 (lane = row, 16 columns per
//     lane, pivots and column entries broadcast with v_readlane: no LDS round trip, no barrier inside),
//   * rows below the diagonal block: one thread per row, forward substitution against L_jj (LDS broadcast),
//   * the rank-16 trailing update runs on MFMA; the next panel's block column is updated first, then
//     wave 0 factorises the next diagonal block WHILE waves 1-7 finish the rest of the update.
// Phase 2 inverts in place (right-to-left block columns, LAPACK dtrti2 order): the eight 16x16
// diagonal inverses are solved concurrently, then X[ib][jb] = -sum_kb X[ib][kb] (L[kb][jb] X[jb][jb])
// on MFMA.
#include "mfgp_internal.h"

namespace mfgp {

constexpr int LP = 130;       // LDS pitch (doubles)
constexpr int LEAF_THREADS = 512;
constexpr int SC_RINV = 0;    // scratch: 1/l_kk for the 128 pivots
constexpr int SC_RED = 128;   // 8 partial sums

__device__ __forceinline__ d4_t mfma(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to full double precision: hardware seed + two Newton steps (short dependent chain:
// this sits on the pivot-to-pivot critical path 128 times per leaf)
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    return y;
}

// 16x16 Cholesky in the registers of one wave.  blk -> element (0,0) of the diagonal block in LDS.
__device__ __forceinline__ void micro_chol16(double* blk, double* rinv_out, int lane, int* info, int pivot0) {
    const int i = lane & 15;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const d2_t v = *reinterpret_cast<const d2_t*>(blk + i * LP + k);
        a[k] = v.x;
        a[k + 1] = v.y;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = readlane_f64(a[j], j);
        if (!(d > 0.0)) {  // wave-uniform: not positive definite (or NaN) -> record the first failing pivot
            if (lane == 0 && *info == 0) *info = pivot0 + j + 1;
            d = 1.0;
        }
        const double y = fast_rsqrt(d);
        a[j] = (i == j) ? d * y : a[j] * y;  // column j: l_ij for rows i >= j (rows above hold unused values)
        if (lane == 0) rinv_out[j] = y;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
            const double lkj = readlane_f64(a[j], k);
            a[k] = __builtin_fma(-a[j], lkj, a[k]);  // meaningful for rows i >= k
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            d2_t v;
            v.x = (k <= i) ? a[k] : 0.0;
            v.y = (k + 1 <= i) ? a[k + 1] : 0.0;
            *reinterpret_cast<d2_t*>(blk + i * LP + k) = v;
        }
    }
}

// C[ib][kb] -= L[ib][jb] L[kb][jb]^T on 16x16 blocks of the LDS matrix
__device__ __forceinline__ void update_block(double* sL, int ib, int kb, int jb, int fr, int q) {
    d4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sL[(ib * 16 + q + 4 * r) * LP + kb * 16 + fr];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double av = -sL[(ib * 16 + fr) * LP + jb * 16 + 4 * s + q];
        const double bv = sL[(kb * 16 + fr) * LP + jb * 16 + 4 * s + q];
        acc = mfma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sL[(ib * 16 + q + 4 * r) * LP + kb * 16 + fr] = acc[r];
}

__global__ __launch_bounds__(LEAF_THREADS, 1) void mfgp_leaf_cholinv_f64(const double* __restrict__ A,
                                                                         double* Lout, double* S, int ld, int blk,
                                                                         double* logdet_part, int* info,
                                                                         unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sL = smem;             // 128 x LP
    double* sc = smem + 128 * LP;  // scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 15;
    const int q = lane >> 4;
    const int64_t g0 = (int64_t)blk * NB * ld + (int64_t)blk * NB;  // offset of the diagonal block
#define STAMP(i) do { if (stamps && tid == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);

    // ---- load (16 B per lane, whole rows coalesced) ------------------------------------------------
    for (int e = tid; e < NB * NB / 2; e += LEAF_THREADS) {
        const int row = e >> 6, c2 = e & 63;
        const d2_t v = *reinterpret_cast<const d2_t*>(A + g0 + (int64_t)row * ld + 2 * c2);
        *reinterpret_cast<d2_t*>(sL + row * LP + 2 * c2) = v;
    }
    __syncthreads();

    STAMP(1);
    // ---- phase 1: blocked Cholesky with look-ahead ---------------------------------------------------
    if (wave == 0) micro_chol16(sL, sc + SC_RINV, lane, info, blk * NB);
    __syncthreads();
    STAMP(2);
    for (int jb = 0; jb < 8; ++jb) {
        const int base = jb * 16;
        // rows below the diagonal block: x L_jj^T = a by forward substitution, one thread per row
        {
            const int nrows = NB - base - 16;
            if (tid < nrows) {
                double* rowp = sL + (base + 16 + tid) * LP + base;
                const double* Lj = sL + base * LP + base;
                const double* rinv = sc + SC_RINV + base;
                double x[16];
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const d2_t v = *reinterpret_cast<const d2_t*>(rowp + k);
                    x[k] = v.x;
                    x[k + 1] = v.y;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    double s = x[k];
#pragma unroll
                    for (int m = 0; m < k; ++m) s = __builtin_fma(-x[m], Lj[k * LP + m], s);
                    x[k] = s * rinv[k];
                }
#pragma unroll
                for (int k = 0; k < 16; k += 2)
                    *reinterpret_cast<d2_t*>(rowp + k) = (d2_t){x[k], x[k + 1]};
            }
        }
        __syncthreads();
        if (jb == 0) STAMP(3);
        if (jb == 7) break;
        // priority: block column jb+1 gets panel jb's update first (one block per wave)
        {
            const int ib = jb + 1 + wave;
            if (ib < 8) update_block(sL, ib, jb + 1, jb, fr, q);
        }
        __syncthreads();
        if (jb == 0) STAMP(4);
        // wave 0 factorises the next diagonal block while waves 1-7 finish the trailing update
        if (wave == 0) {
            micro_chol16(sL + (base + 16) * LP + base + 16, sc + SC_RINV + base + 16, lane, info,
                         blk * NB + base + 16);
        } else {
            const int m = 6 - jb;  // block columns jb+2 .. 7
            const int nblk = m * (m + 1) / 2;
            for (int idx = wave - 1; idx < nblk; idx += 7) {
                int ii = 0, rem = idx;
                while (rem > ii) { rem -= ii + 1; ++ii; }
                update_block(sL, jb + 2 + ii, jb + 2 + rem, jb, fr, q);
            }
        }
        __syncthreads();
        if (jb == 0) STAMP(5);
    }
    STAMP(6);

    // ---- write L (zeros above the diagonal) and the half log-determinant ------------------------------
    for (int e = tid; e < NB * NB / 2; e += LEAF_THREADS) {
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
        if (lane == 0) sc[SC_RED + wave] = v;
        __syncthreads();
        if (tid == 0) logdet_part[blk] = sc[SC_RED] + sc[SC_RED + 1];  // rows live in waves 0 and 1
    }

    STAMP(7);
    // ---- phase 2: in-place inverse ---------------------------------------------------------------------
    // (a) the eight 16x16 diagonal inverses, one thread per column, all at once
    {
        double x[16];
        const int b = tid >> 4, k = tid & 15;
        if (tid < 128) {
            const double* Lb = sL + (b * 16) * LP + b * 16;
            const double* rinv = sc + SC_RINV + b * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double s = (i == k) ? 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < i; ++m) s = __builtin_fma(-Lb[i * LP + m], x[m], s);
                x[i] = s * rinv[i];
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
    STAMP(8);
    // (b) block columns right to left
    for (int jb = 6; jb >= 0; --jb) {
        const int base = jb * 16;
        // T[kb] = L[kb][jb] * X[jb][jb]   (in place, one block per wave)
        {
            const int kb = jb + 1 + wave;
            if (kb < 8) {
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
        }
        __syncthreads();
        // X[ib][jb] = - sum_{kb = jb+1..ib} X[ib][kb] * T[kb]   (accumulate in registers, then overwrite T)
        const int ib = jb + 1 + wave;
        d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
        if (ib < 8) {
            for (int kb = jb + 1; kb <= ib; ++kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = -sL[(ib * 16 + fr) * LP + kb * 16 + 4 * s + q];
                    const double b = sL[(kb * 16 + 4 * s + q) * LP + base + fr];
                    acc = mfma(a, b, acc);
                }
            }
        }
        __syncthreads();
        if (ib < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sL[(ib * 16 + q + 4 * r) * LP + base + fr] = acc[r];
        }
        __syncthreads();
    }

    STAMP(9);
    // ---- write X mirrored: S[r][c] = X[max(r,c)][min(r,c)] -------------------------------------------
    for (int e = tid; e < NB * NB; e += LEAF_THREADS) {
        const int row = e >> 7, col = e & 127;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        S[g0 + (int64_t)row * ld + col] = sL[hi * LP + lo];
    }
    __syncthreads();
    STAMP(10);
#undef STAMP
}

void launch_leaf(hipStream_t s, const double* A, double* Lout, double* S, int ld, int blk,
                 double* logdet_part, int* info, unsigned long long* stamps) {
    constexpr size_t lds = (size_t)(128 * LP + 160) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfgp_leaf_cholinv_f64),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(mfgp_leaf_cholinv_f64, dim3(1), dim3(LEAF_THREADS), lds, s, A, Lout, S, ld, blk,
                       logdet_part, info, stamps);
}

}  // namespace mfgp
