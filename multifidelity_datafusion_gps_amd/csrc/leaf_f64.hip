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
constexpr int SC_LT = 144;    // 16x16 transposed copy of the current diagonal factor: LT[m*16 + k] = L_jj[k][m]
constexpr int SC_SIZE = SC_LT + 256;

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
// Branch-free inside the pivot loop: a failed pivot (d <= 0 or NaN) is replaced by 1 and its index kept.
__device__ __forceinline__ void micro_chol16(double* blk, double* rinv_out, double* lt_out, int lane, int* info,
                                             int pivot0) {
    const int i = lane & 15;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const d2_t v = *reinterpret_cast<const d2_t*>(blk + i * LP + k);
        a[k] = v.x;
        a[k + 1] = v.y;
    }
    double my_rinv = 0.0;
    int fail = 0;  // 1-based index of the first non-positive pivot inside this block (wave-uniform)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = readlane_f64(a[j], j);
        const bool ok = d > 0.0;
        fail = (!ok && fail == 0) ? j + 1 : fail;
        d = ok ? d : 1.0;
        const double y = fast_rsqrt(d);
        a[j] *= y;                      // l_ij for rows i > j; row j itself: a_jj * y = sqrt(d) when ok
        a[j] = (i == j && !ok) ? 1.0 : a[j];
        my_rinv = (i == j) ? y : my_rinv;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
            const double lkj = readlane_f64(a[j], k);
            a[k] = __builtin_fma(-a[j], lkj, a[k]);  // meaningful for rows i >= k
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 4 broadcasts in flight (SGPR pressure)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < 16) {
        rinv_out[i] = my_rinv;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            d2_t v;
            v.x = (k <= i) ? a[k] : 0.0;
            v.y = (k + 1 <= i) ? a[k + 1] : 0.0;
            *reinterpret_cast<d2_t*>(blk + i * LP + k) = v;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) lt_out[k * 16 + i] = (k <= i) ? a[k] : 0.0;  // column k of L_jj, contiguous
    }
    if (fail != 0 && lane == 0 && *info == 0) *info = pivot0 + fail;
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
    {
        constexpr int NLD = NB * NB / 2 / LEAF_THREADS;  // 16 loads in flight per thread before the first LDS store
        d2_t v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = tid + u * LEAF_THREADS;
            const int row = e >> 6, c2 = e & 63;
            v[u] = *reinterpret_cast<const d2_t*>(A + g0 + (int64_t)row * ld + 2 * c2);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = tid + u * LEAF_THREADS;
            const int row = e >> 6, c2 = e & 63;
            *reinterpret_cast<d2_t*>(sL + row * LP + 2 * c2) = v[u];
        }
    }
    __syncthreads();

    STAMP(1);
    // ---- phase 1: blocked Cholesky with look-ahead ---------------------------------------------------
    if (wave == 0) micro_chol16(sL, sc + SC_RINV, sc + SC_LT, lane, info, blk * NB);
    __syncthreads();
    STAMP(2);
    for (int jb = 0; jb < 8; ++jb) {
        const int base = jb * 16;
        // rows below the diagonal block: x L_jj^T = a by forward substitution, one thread per row
        {
            const int nrows = NB - base - 16;
            if (tid < nrows) {
                double* rowp = sL + (base + 16 + tid) * LP + base;
                // an opaque per-lane zero keeps the (wave-uniform) L_jj reads in VGPRs: hipcc otherwise moves every
                // broadcast value to an SGPR with v_readfirstlane and spills ~130 SGPRs in this loop nest
                int vz;
                asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
                const double* LT = sc + SC_LT + vz;
                const double* rinv = sc + SC_RINV + base + vz;
                double x[16];
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const d2_t v = *reinterpret_cast<const d2_t*>(rowp + k);
                    x[k] = v.x;
                    x[k + 1] = v.y;
                }
                // right-looking: after x[m] is final, eliminate it from every later unknown (independent FMAs)
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    x[m] *= rinv[m];
#pragma unroll
                    for (int k = m + 1; k < 16; ++k) x[k] = __builtin_fma(-x[m], LT[m * 16 + k], x[k]);
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
            micro_chol16(sL + (base + 16) * LP + base + 16, sc + SC_RINV + base + 16, sc + SC_LT, lane, info,
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
    // (b) recursive assembly X21 = -X22 (L21 X11) over node half-sizes hs = 1, 2, 4 blocks; all nodes of a
    //     level are independent; one or two 16x16 outputs per wave, accumulated in registers (the products
    //     read blocks that other waves overwrite, so each half-step is compute | barrier | write | barrier)
    for (int hs = 1; hs <= 4; hs <<= 1) {
        const int nout = 4 * hs;  // (8 / (2 hs)) nodes x hs^2 outputs
        for (int half = 0; half < 2; ++half) {
            d4_t acc[2];
            int oi[2], oj[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc[u] = (d4_t){0.0, 0.0, 0.0, 0.0};
                const int o = wave + 8 * u;
                oi[u] = -1;
                oj[u] = 0;
                if (o < nout) {
                    const int node = o / (hs * hs), w = o % (hs * hs);
                    const int b0 = 2 * hs * node;
                    const int i = b0 + hs + w / hs, j = b0 + w % hs;
                    oi[u] = i;
                    oj[u] = j;
                    // half 0: T[i][j]   = sum_{k=j}^{b0+hs-1} L[i][k] X[k][j]
                    // half 1: X21[i][j] = - sum_{k=b0+hs}^{i} X[i][k] T[k][j]
                    const int k0 = half == 0 ? j : b0 + hs;
                    const int k1 = half == 0 ? b0 + hs - 1 : i;
                    for (int k = k0; k <= k1; ++k) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            double a = sL[(i * 16 + fr) * LP + k * 16 + 4 * s4 + q];
                            const double b = sL[(k * 16 + 4 * s4 + q) * LP + j * 16 + fr];
                            if (half == 1) a = -a;
                            acc[u] = mfma(a, b, acc[u]);
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (oi[u] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sL[(oi[u] * 16 + q + 4 * r) * LP + oj[u] * 16 + fr] = acc[u][r];
                }
            }
            __syncthreads();
        }
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
    constexpr size_t lds = (size_t)(128 * LP + SC_SIZE) * sizeof(double);
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
