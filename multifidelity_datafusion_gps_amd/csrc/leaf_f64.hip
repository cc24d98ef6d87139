// leaf_f64.hip -- Cholesky factor AND its inverse of one 128x128 diagonal block, inside one workgroup.
//
// This is the only serial piece of the factorisation (the chain of N pivots); everything above the
// leaves is tile GEMMs (gemm_f64.hip).  Replaces LAPACK dpotrf + dtrtri on the diagonal blocks
// behind GPy's jitchol / pdinv.
//
// The block lives in LDS (pitch 130 doubles: fragment reads of v_mfma_f64_16x16x4 hit 32 distinct
// 8-byte bank pairs; 138 KB in all).  8 waves, 16-column panels with look-ahead:
//   * the 16x16 diagonal micro-Cholesky runs in the REGISTERS of wave 0 (lane = row, 16 columns per lane; the entry the
//     next pivot waits for broadcast with v_readlane, the others through an LDS broadcast issued ahead of their use),
//   * the inverse of the 16x16 diagonal factor comes out of the same pivot loop (lanes 16-31, the same instructions), and
//     the other rows of the panel are solved as MFMA block products with it,
//   * the inverse of the WHOLE block rides on the panel loop as the augmented system [A; I] (the image of the identity in
//     the upper triangle of the LDS block), exactly like the planner's sweep one level up,
//   * per panel, wave 0 runs the critical path back to back -- the solve of block (jb+1, jb), panel jb's update of the next
//     diagonal block, its micro-Cholesky -- while waves 1-7 solve the rest of the panel and, after ONE barrier, do every other
//     rank-16 update and write the finished panel out in the micro-Cholesky's shadow (two workgroup barriers per panel);
//     every solved 16x16 block leaves for global memory straight from the solving wave's accumulators (solve_block), the
//     shadow pass keeps the diagonal block and the lower strip of the mirrored inverse.
// Measured history and the variants that lost: LABBOOK.md, tools/gemm_lab/RETIRED.md, profiles/r05_leaf_*.txt.
#include <stdlib.h>
#include <mutex>
#include "mfgp_internal.h"

namespace mfgp {

constexpr int LP = 130;       // LDS pitch (doubles)
constexpr int LEAF_THREADS = 512;
constexpr int SC_RED = 0;      // scratch: 8 partial sums
constexpr int SC_SIZE = 144;    // 8 partial sums (+ 8 spare), then the micro-Cholesky's 2 x 64 column broadcast slots

__device__ __forceinline__ d4_t mfma(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to full double precision: hardware seed y0 (>= 20 good bits) and ONE third-order correction
//   e = 1 - d y0^2 ,   1/sqrt(d) = y0 (1 - e)^-1/2 = y0 (1 + e/2 + 3e^2/8 + O(e^3)) ,   |e|^3 < 2^-60
// -- four dependent operations behind the seed instead of the six of two Newton steps: this chain sits on the pivot-to-pivot
// critical path 128 times per leaf.
__device__ __forceinline__ double fast_rsqrt(double d) {
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = __builtin_fma(-d * y0, y0, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y0 * e, p, y0);
}

// 16x16 Cholesky AND the inverse of the factor in the registers of one wave.  blk -> element (0,0) of the diagonal
// block in LDS.  Lanes 0-15: lane i holds row i of the block, v[k] = A[i][k] -> L[i][k].  Lanes 16-31: lane 16+c holds
// column c of the inverse Y = L^-1, v[k] = delta_kc - sum_{m<k} l_km Y[m][c] until pivot k, then Y[k][c].
// The inverse rides on the factorisation's own broadcasts: row j of Y is Y[j][:] = (e_j - sum_{m<j} l_jm Y[m][:]) / l_jj,
// and l_kj -- broadcast at pivot j to eliminate column j from row k -- is exactly the coefficient with which the
// finished row j of Y enters row k's sum.  Both lane groups execute the SAME instructions (v[j] *= y_j, v[k] -= v[j] * l_kj):
// the inverse costs nothing per pivot and no extra communication.
template <int YP = LP>
__device__ __forceinline__ void micro_chol16(double* blk, double* y_out, int lane, int* info, int pivot0, double* col_buf,
                                             int* flag, int epoch) {
    const int i = lane & 15;
    const bool inv_lane = (lane & 16) != 0;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {   // inverse lanes start from the identity: column c of I
        const d2_t t = *reinterpret_cast<const d2_t*>(blk + i * LP + k);
        v[k] = inv_lane ? (k == i ? 1.0 : 0.0) : t.x;
        v[k + 1] = inv_lane ? (k + 1 == i ? 1.0 : 0.0) : t.y;
    }
    // Nothing in the pivot loop looks at the sign of a pivot (the loop is bound by instruction issue: every instruction per
    // pivot counts 128 times per leaf): a pivot d <= 0 or NaN makes its own diagonal entry NaN (rsq(d < 0) = NaN, 0 * inf = NaN)
    // and everything after it, so the first non-positive diagonal entry of the result IS the first failed pivot.
    //
    // Column j reaches the other lanes two ways.  The ONE entry the next pivot waits for (l_{j+1,j}) is broadcast with
    // v_readlane (two instructions + the FMA, no memory latency).  The other 14 - j entries go through LDS: the factor
    // lanes store their l_ij (one ds_write), every lane reads them back two at a time from the same address (broadcast
    // reads), so an update costs half a read + one FMA instead of two v_readlane + one FMA -- a third of the loop's
    // instructions.  The reads are issued before the next pivot's reciprocal-square-root chain and consumed after it.
    // col_buf: 2 slots (j & 1) of 64 doubles.  EVERY lane stores (entry = its lane index; only entries 0-15, the factor lanes,
    // are read): no exec masking around the store.  The slot base goes through an opaque per-lane zero, or the compiler
    // materialises every read's absolute LDS address in a VGPR of its own (s_add + v_mov per read) instead of base + offset.
    int zero = 0;
    __asm__ volatile("" : "+v"(zero));
    double* const col = col_buf + zero;
    double r[16];                         // pivot j's entries l_kj for the columns k >= j + 2, in flight
    double y = fast_rsqrt(readlane_f64(v[0], 0));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= 1) {   // pivot j-1's remaining updates (columns j+1 .. 15; column j was the readlane path)
#pragma unroll
            for (int k = j + 1; k < 16; ++k) v[k] = __builtin_fma(-v[j - 1], r[k], v[k]);
        }
        v[j] *= y;                      // factor lanes: l_ij (row j itself: sqrt(d)); inverse lanes: Y[j][c]
        if (j < 15) {
            const double l = readlane_f64(v[j], j + 1);
            v[j + 1] = __builtin_fma(-v[j], l, v[j + 1]);
            const double dn = readlane_f64(v[j + 1], j + 1);
            const double y0 = __builtin_amdgcn_rsq(dn);
            if (j < 14) {
                col[(j & 1) * 64 + lane] = v[j];
                // lanes exchange data through memory here: without a (wavefront-scope, instruction-free) fence the compiler
                // treats the reads below as private to each lane and forwards older values to the lanes that did not store
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double* const cj = col + (j & 1) * 64;
                const int k0 = (j + 2) | 1;               // first odd column index >= j + 2 ...
                if (((j + 2) & 1) == 0) {                 // ... reached through one aligned pair, or directly
                    const d2_t t = *reinterpret_cast<const d2_t*>(cj + j + 2);
                    r[j + 2] = t.x; r[j + 3] = t.y;
                } else {
                    r[j + 2] = cj[j + 2];
                }
#pragma unroll
                for (int k = k0 + 1; k < 16; k += 2) {    // aligned pairs (k even, k + 1 <= 15)
                    const d2_t t = *reinterpret_cast<const d2_t*>(cj + k);
                    r[k] = t.x; r[k + 1] = t.y;
                }
            }
            // (the reads must be ISSUED here, a whole reciprocal-square-root chain ahead of their first use: left to itself the
            // compiler sinks them next to that use and puts the LDS latency on the pivot-to-pivot path)
            __builtin_amdgcn_sched_barrier(0);
            const double en = __builtin_fma(-dn * y0, y0, 1.0);
            const double p = __builtin_fma(0.375, en, 0.5);
            y = __builtin_fma(y0 * en, p, y0);
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            d2_t t;
            t.x = (k <= i) ? v[k] : 0.0;
            t.y = (k + 1 <= i) ? v[k + 1] : 0.0;
            *reinterpret_cast<d2_t*>(blk + i * LP + k) = t;
        }
    } else if (lane < 32) {
#pragma unroll
        for (int k = 0; k < 16; ++k) y_out[k * YP + i] = v[k];   // column i of Y_jj (zero above the diagonal)
    }
    // first failed pivot, if any: the first diagonal entry that is not a positive number (read back: same wave, LDS is in order)
    const double dg = blk[i * LP + i];
    const unsigned long long bad = __ballot(lane < 16 && !(dg > 0.0));
    if (bad != 0 && lane == 0) {
        if (*info == 0) *info = pivot0 + __ffsll((long long)bad);
        if (flag) *flag = epoch;    // device-resident mark of a failed evaluation: the sweep's later launches return at once
    }
}

// 16x16x16 block products: four v_mfma_f64_16x16x4 on one accumulator.  (The v_mfma_f64_4x4x4_4b form the tile GEMM moved to in
// round 3 makes the leaf SLOWER, 27 -> 36 us alone: the block products are not what bounds the leaf -- wave 0's micro-Cholesky is,
// it shares SIMD 0 with a worker wave, and four times as many MFMA issues on that SIMD stretch its VALU chain.  Retired:
// tools/gemm_lab/RETIRED.md.)

// rows of block ib below the diagonal block jb:  X = A Y_jj^T  (x L_jj^T = a), one 16x16 block per wave on MFMA
// gL / gS != nullptr (the panel loop's solves; round 5): the block is FINAL -- it also leaves for global memory straight from the
// accumulators, 16 lanes a contiguous 128-byte row segment: a block below the diagonal into the factor, a block above it (X^T[ib,jb])
// into the upper part of the mirrored inverse, with zeros in its place in the factor -- instead of a pass of the seven worker waves
// over the LDS block in the micro-Cholesky's shadow, where THEY are the longer side (-2 % per evaluation at N <= 2048, level above)
template <int YP = LP>
__device__ __forceinline__ void solve_block(double* sL, const double* Y, int ib, int jb, int fr, int q, double* gL = nullptr,
                                            double* gS = nullptr, int ld = 0) {
    d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double av = sL[(ib * 16 + fr) * LP + jb * 16 + 4 * s + q];
        const double bv = Y[fr * YP + 4 * s + q];   // B[k][n] = Y[n][k]
        acc = mfma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sL[(ib * 16 + q + 4 * r) * LP + jb * 16 + fr] = acc[r];
    if (gL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long o = (long long)(ib * 16 + q + 4 * r) * ld + jb * 16 + fr;
            if (ib > jb) {
                gL[o] = acc[r];
            } else {
                gL[o] = 0.0;
                gS[o] = acc[r];
            }
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


// ---- leaf v3: the inverse rides on the factorisation --------------------------------------------------------------------
// Same block-in-LDS, 16-column-panel factorisation as v2, but the inverse is no longer a second phase: it is produced by
// the augmented system [A; I] INSIDE the panel loop, as idle-wave work in the shadow of wave 0's micro-Cholesky (which is
// the leaf's critical path: 8 x ~5.7k cycles).  With B the running image of the identity, kept in the UPPER triangle of
// the LDS block (the lower triangle holds A -> L; the upper part of the input is never used):
//     X^T[I,jb] = B[I,jb] Y_jj^T            (I < jb:  the same product as the panel solve L[I,jb] = A[I,jb] Y_jj^T)
//     B[I,J]   -= X^T[I,jb] L[J,jb]^T       (I <= jb < J; X^T[jb,jb] = Y_jj^T, first touch of row jb)
// Per panel that is 7 solves (one per wave, as many as v2's worst case) and up to 16 extra rank-16 updates spread over the
// seven waves that wait for the micro-Cholesky anyway; v2's phase 2 (17k of its 90k cycles) is gone.
constexpr int YP16 = 18;                     // pitch of a 16x16 inverse diagonal factor Y_jj
constexpr int SY_SIZE = 2 * 16 * YP16;       // TWO of them (panel jb's, and the next one being produced beside it): Y_jb is dead once
                                             // panel jb's solves, first touches and output are through -- 138 KB of LDS in all, so
                                             // that a 16 KB chain workgroup of another evaluation still fits on the CU beside the leaf

// first touch of row jb of B:  B[jb, J] = -Y_jj^T L[J, jb]^T
__device__ __forceinline__ void bfirst_block(double* sL, const double* Y, int jb, int J, int fr, int q) {
    d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double av = -Y[(4 * s + q) * YP16 + fr];                       // A[row fr][k] = Y^T[fr][k] = Y[k][fr]
        const double bv = sL[(J * 16 + fr) * LP + jb * 16 + 4 * s + q];      // B[k][n = fr] = L[J*16 + fr][jb*16 + k]
        acc = mfma(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sL[(jb * 16 + q + 4 * r) * LP + J * 16 + fr] = acc[r];
}


__device__ __forceinline__ void leaf_body_v3(double* smem, const double* __restrict__ A, double* Lout, double* S, int ld, int blk,
                                             double* logdet_part, int* info, int* flag, int epoch) {
    double* sL = smem;                       // 128 x LP: lower = A -> L, strictly upper 16-blocks = B -> X^T
    double* sY = smem + 128 * LP;            // 2 x (16 x YP16): Y_jj = L_jj^-1, slot jb & 1
    double* sc = sY + SY_SIZE;               // scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 15;
    const int q = lane >> 4;
    const int64_t g0 = (int64_t)blk * NB * ld + (int64_t)blk * NB;  // offset of the diagonal block
    // load: only the 16-blocks on and below the diagonal (the upper part of the input is never used: that triangle of the
    // LDS block holds B).  Wave 0 takes the first diagonal tile alone and starts its micro-Cholesky at once, the other seven
    // waves bring in the remaining 35 tiles meanwhile.
    if (wave == 0) {
        d2_t v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;            // 16 rows x 8 pairs
            v[u] = *reinterpret_cast<const d2_t*>(A + g0 + (int64_t)(e >> 3) * ld + 2 * (e & 7));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;
            *reinterpret_cast<d2_t*>(sL + (e >> 3) * LP + 2 * (e & 7)) = v[u];
        }
        micro_chol16<YP16>(sL, sY, lane, info, blk * NB, sc + 16, flag, epoch);    // (same wave wrote the tile: LDS program order suffices)
    } else {
        // the 35 other tiles of the lower block triangle (tile tl = I(I+1)/2 + J, J <= I), 128 pairs of doubles each
        constexpr int NLD = 10;   // ceil(35 * 128 / 448 lanes)
        const int t7 = tid - 64;
        d2_t v[NLD];
        int off[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int g = t7 + u * 448;
            off[u] = -1;
            if (g < 35 * 128) {
                const int tl = 1 + g / 128, pr = g % 128;             // tile 1..35 in row-major lower-triangle order
                int I = 0, rem = tl;
                while (rem > I) { rem -= I + 1; ++I; }                // tl = I(I+1)/2 + J
                const int J = rem;
                const int row = 16 * I + (pr >> 3), col = 16 * J + 2 * (pr & 7);
                off[u] = row * LP + col;
                v[u] = *reinterpret_cast<const d2_t*>(A + g0 + (int64_t)row * ld + col);
            }
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if (off[u] >= 0) *reinterpret_cast<d2_t*>(sL + off[u]) = v[u];
    }
    __syncthreads();
    // output of panel jb by `nthr` threads (t = 0 .. nthr-1): what the solves did not write -- the diagonal 16-block of L, and the
    // mirrored inverse S[r][c] = X[max(r,c)][min(r,c)] for max(r,c) in block jb: X[hi][lo] = X^T[lo][hi] sits in the UPPER part
    // of sL, the diagonal 16-block in sY
    double* const gL = Lout + g0;        // the solving waves write their finished blocks themselves (solve_block)
    double* const gS = S + g0;
    auto write_panel = [&](int jb, int t, int nthr) {
        for (int e = t; e < 16 * 8; e += nthr) {          // the diagonal 16-block of L (zeros above its diagonal); the rest left with the solves
            const int row = 16 * jb + (e >> 3), c2 = 8 * jb + (e & 7);
            d2_t v = *reinterpret_cast<const d2_t*>(sL + row * LP + 2 * c2);
            if (2 * c2 > row) v.x = 0.0;
            if (2 * c2 + 1 > row) v.y = 0.0;
            *reinterpret_cast<d2_t*>(Lout + g0 + (int64_t)row * ld + 2 * c2) = v;
        }
        const int w = 16 * jb + 16;                       // the strip: rows of block jb, columns 0 .. w (row-major writes)
        const double* Yb = sY + (jb & 1) * 16 * YP16;
        for (int e = t; e < 16 * w; e += nthr) {
            const int r = 16 * jb + e / w, c = e % w;
            double v;
            if (c >= 16 * jb) {                           // inside the diagonal block
                const int a = r & 15, b = c & 15;
                v = a >= b ? Yb[a * YP16 + b] : Yb[b * YP16 + a];
            } else {
                v = sL[c * LP + r];                       // X[r][c] = X^T[c][r]
            }
            S[g0 + (int64_t)r * ld + c] = v;
        }
        // (its mirror image -- rows above the block -- left with the solves too)
    };
    // Panel loop (round 5: two workgroup barriers per panel instead of three, and the pivot wave's work contiguous).  The leaf's
    // critical path runs through ONE block per panel: the sub-diagonal block (jb+1, jb) is solved, the next diagonal block (jb+1, jb+1)
    // takes panel jb's update, and its micro-Cholesky produces Y_{jb+1}, which the next panel's solves wait for.  Wave 0 does all three
    // back to back (the two hand-overs go through LDS within the wave: program order, no barrier); the other waves solve the rest of
    // the panel meanwhile, and after ONE barrier do everything else of panel jb -- the rest of block column jb+1 first -- in the shadow
    // of the micro-Cholesky.  Every block still receives the same updates in the same order: the results are bitwise those of the
    // three-phase loop (phase A solve | barrier | phase B block column jb+1 | barrier | phase C micro-Cholesky beside the rest | barrier).
    auto wave_handover = [&]() {           // LDS written by this wave is read by other lanes of the SAME wave next
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int jb = 0; jb < 8; ++jb) {
        const double* Yj = sY + (jb & 1) * 16 * YP16;
        // phase A -- panel "solves", one 16x16 block per wave: rows below the diagonal give L[ib,jb], rows above give X^T[ib,jb].
        // Wave 0 takes the critical block row jb+1 (the wave that owns that row takes wave 0's row 0), then updates the next diagonal block.
        if (jb == 7) {
            if (wave != 7) solve_block<YP16>(sL, Yj, wave, 7, fr, q, gL, gS, ld);
            __syncthreads();
            break;
        }
        if (wave == 0) {
            solve_block<YP16>(sL, Yj, jb + 1, jb, fr, q, gL, gS, ld);
            wave_handover();
            update_block(sL, jb + 1, jb + 1, jb, fr, q);
        } else {
            const int ib = (wave == jb + 1) ? 0 : wave;
            if (ib != jb) solve_block<YP16>(sL, Yj, ib, jb, fr, q, gL, gS, ld);
        }
        __syncthreads();
        if (wave == 0) {
            // the next diagonal block is factorised (and inverted) while waves 1-7 do the rest of panel jb
            micro_chol16<YP16>(sL + (jb * 16 + 16) * LP + jb * 16 + 16, sY + ((jb + 1) & 1) * 16 * YP16, lane, info,
                               blk * NB + jb * 16 + 16, sc + 16, flag, epoch);
        } else {
            const int nP = 6 - jb;                   // block column jb+1 of A below its diagonal block: the next panel's solves read it
            const int m = 6 - jb;                    // A: block columns jb+2 .. 7, lower blocks
            const int nA = m * (m + 1) / 2;
            const int w = 7 - jb;                    // B: rows 0 .. jb, block columns jb+1 .. 7
            const int nB = (jb + 1) * w;
            for (int idx = wave - 1; idx < nP + nA + nB; idx += 7) {
                if (idx < nP) {
                    update_block(sL, jb + 2 + idx, jb + 1, jb, fr, q);
                } else if (idx < nP + nA) {
                    int ii = 0, rem = idx - nP;
                    while (rem > ii) { rem -= ii + 1; ++ii; }
                    update_block(sL, jb + 2 + ii, jb + 2 + rem, jb, fr, q);
                } else {
                    const int t = idx - nP - nA;
                    const int I = t / w, J = jb + 1 + t % w;
                    if (I == jb) bfirst_block(sL, Yj, jb, J, fr, q);
                    else update_block(sL, I, J, jb, fr, q);     // (I < jb: "L[I][jb]" read there is X^T[I,jb])
                }
            }
            // Block column jb of L and row / column block jb of X are final since this panel's solves: write them out now,
            // in the shadow of the micro-Cholesky, panel by panel (16 + <= 32 KB each) instead of 256 KB after the last one.
            write_panel(jb, tid - 64, 448);
        }
        __syncthreads();
    }
    // ---- the last panel's share of the output and the half log-determinant ----
    write_panel(7, tid, LEAF_THREADS);
    {
        double v = (tid < NB) ? log(sL[tid * LP + tid]) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sc[SC_RED + wave] = v;
        __syncthreads();
        if (tid == 0) logdet_part[blk] = sc[SC_RED] + sc[SC_RED + 1];  // rows live in waves 0 and 1
    }
    __syncthreads();
}

// One workgroup per matrix set: blockIdx.x = b selects set b of a batched evaluation (mfgp_eval_batch: the sets lie `bstride`
// elements apart, their log-det partials `ldstride` apart, their pivot status words `istride` ints apart); a single evaluation
// is the batch of one.
__global__ __launch_bounds__(LEAF_THREADS, 1) void mfgp_leaf_cholinv_f64(const double* __restrict__ A,
                                                                         double* Lout, double* S, int ld, int blk,
                                                                         double* logdet_part, int* info, long long bstride,
                                                                         int ldstride, int istride, int* flag, int epoch) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // flag[set] == epoch: an earlier diagonal block of THIS evaluation was not positive definite -- nothing downstream is used
    // (the host reports the pivot), so the rest of the sweep returns at once instead of costing a whole factorisation
    if (flag) {
        flag += blockIdx.x;
        if (*flag == epoch) return;
    }
    const long long off = (long long)blockIdx.x * bstride;
    leaf_body_v3(smem, A + off, Lout + off, S + off, ld, blk, logdet_part + (int)blockIdx.x * ldstride,
                 info + (int)blockIdx.x * istride, flag, epoch);
}


void launch_leaf(hipStream_t s, const double* A, double* Lout, double* S, int ld, int blk,
                 double* logdet_part, int* info, int nbatch, long long bstride, int ldstride, int istride, int* flag, int epoch) {
    constexpr size_t lds = (size_t)(128 * LP + SY_SIZE + SC_SIZE) * sizeof(double);
    static std::once_flag attr_once[MFGP_MAX_DEVICES];   // per device, thread-safe (see launch_gemm)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(attr_once[dev & (MFGP_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfgp_leaf_cholinv_f64),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL(mfgp_leaf_cholinv_f64, dim3(nbatch > 0 ? nbatch : 1), dim3(LEAF_THREADS), lds, s, A, Lout, S, ld, blk,
                       logdet_part, info, bstride, ldstride, istride, flag, epoch);
}

}  // namespace mfgp
