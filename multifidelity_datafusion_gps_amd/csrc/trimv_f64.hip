// trimv_f64.hip -- the bandwidth-bound triangular (multi-)vector products with the stored inverse factor.
//
//     V[i][j] = sum_{k in range(j)} W[i][k] * M[j][k]        i < R right-hand sides, j < nrows
//     range(j):  mode 0: k <= j (lower part)   mode 1: j <= k < ncols (upper part)   mode 2: k < ncols
//
// What is in this file (round 6; everything the adaptation loop of src/abstractMFGP.py:317-359 calls between two refits):
//   trimv_wave / mfgp_trimv_f64          the body: R <= 4 right-hand sides on the VALU (launch_rowdot: R = 1)
//   mfgp_alpha_finish_f64                alpha = X^T z + the solve's scalars in one launch
//   mfgp_predv_rows_f64 / _lds_f64       predict with <= 8 test rows: the variance product + the means in one launch
//   mfgp_panel_fragments_f64, mfgp_predv_mfma_f64   9 .. 64 test rows below Np = 3072: the same product on the matrix pipe, S by LDS-DMA
//   mfgp_predv_mfma2_f64, mfgp_predv_finish_planes_f64   9 .. 64 test rows from Np = 3072: S and a shared W tile staged through registers,
//                                        equal shares of the triangle per workgroup, partial planes summed by the finish
//   mfgp_predv_finish_f64                var = max(k** - |V|^2, 1e-15) + noise, one workgroup per test row
//   mean_segment / mean_row / mfgp_rowmean_f64   the predictive means: a full-length row's sum as the ordered sum of its 2048-column
//                                        segments' sums, so that a few long rows split along k inside a workgroup (mode 2 of launch_rowdot)
//
// One kernel body serves every O(N^2) pass of the path: z = X y and alpha = X^T z behind GPy's dpotrs (a5), the
// predictive mean K(X*,X) alpha and -- few test rows, the N* = 1 callback of the reference's DIRECT maximiser
// (src/adaptation_maximizers/scipydirect_wrapper.py:22-24) -- the variance product V = K(X*,X) X^T (a11), and the two passes
// l = X k, w = X^T l of a rank-1 append (8(f1)).  Each is bound by ONE read of the 4 Np^2-byte triangle of the mirrored inverse
// S = X + X^T - diag, so the only thing that matters is the shape of the reads:
//   * a wave owns JR consecutive rows j and walks along k: one wave-instruction (global_load_dwordx4, 16 B per lane) covers
//     1 KiB contiguous of ONE row -- the full-rate shape of the guide (>= 256 contiguous bytes per row);
//   * U 128-column chunks per batch and the next batch already in flight while this one is consumed (register double
//     buffer): 2 * U * JR KiB of S per wave outstanding;
//   * every wave gets the same number of bytes: row group g is paired with group G-1-g (short + long row of the triangle);
//   * the R right-hand sides are re-read from L1 / L2 (R / JR of the S bytes), the products run on the VALU (2 R FMAs per
//     16 bytes of S -- idle beside the read), per-lane partial sums are folded in a FIXED order (DPP butterflies inside a
//     16-lane row, then rows 0..3): results are deterministic and independent of the launch.
// Every chunk is masked by the row's own k range with v_cndmask (selects, not multiplications: whatever lies beyond the
// range never reaches a sum), so interior, diagonal and tail chunks are one code path; chunk indices past a group's last
// are clamped to it for the ADDRESS (an L1 hit) and masked out by their intended k.
//
// Requirements: ld, ldw >= the column count rounded up to 128 (every matrix / vector of the engine is padded to Np).
#include <type_traits>
#include "mfgp_internal.h"

namespace mfgp {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int srclane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srclane),
                            __builtin_amdgcn_readlane(__double2loint(v), srclane));
}
// sum over the 64 lanes in a fixed order, the same value in every lane
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror: every lane of a 16-lane row holds the row's sum
    return ((lane_f64(v, 0) + lane_f64(v, 16)) + lane_f64(v, 32)) + lane_f64(v, 48);
}

struct TrimvGroup {   // wave-uniform description of one row group
    int j0;           // first row
    int c_lo, nb;     // first chunk, number of U-batches
    int c_last;       // last chunk that intersects the group's range (address clamp)
};

template <int JR, int U>
__device__ __forceinline__ TrimvGroup trimv_group(int g, int nrows, int ncols, int mode) {
    TrimvGroup t;
    t.j0 = g * JR;
    const int jl = min(t.j0 + JR, nrows) - 1;                 // last real row of the group
    const int nc = (ncols + 127) >> 7;
    t.c_lo = mode == 1 ? (t.j0 >> 7) : 0;
    t.c_last = (mode == 0 ? (jl >> 7) : nc - 1);
    const int n = t.c_last - t.c_lo + 1;
    t.nb = n > 0 ? (n + U - 1) / U : 0;
    return t;
}

// the body: wave `wv` of `nw` (nw = ceil(G / 2), G = ceil(nrows / JR) row groups)
template <int R, int JR, int U>
__device__ __forceinline__ void trimv_wave(const double* __restrict__ M, int ld, const double* __restrict__ W, int ldw,
                                           double* __restrict__ V, int ldv, int nrows, int ncols, int mode, int wv, int lane) {
    const int G = (nrows + JR - 1) / JR;
    const int ga = wv, gb = G - 1 - wv;
    if (ga > gb) return;
    const TrimvGroup ta = trimv_group<JR, U>(ga, nrows, ncols, mode);
    TrimvGroup tb = trimv_group<JR, U>(gb, nrows, ncols, mode);
    if (gb == ga) tb.nb = 0;
    const int nbt = ta.nb + tb.nb;

    d2_t sb[2][U][JR], wb[2][U][R];
    double acc[JR][R];
#pragma unroll
    for (int r = 0; r < JR; ++r)
#pragma unroll
        for (int i = 0; i < R; ++i) acc[r][i] = 0.0;

    // batch b of the wave -> (group, first chunk)
    // (b past the wave's last batch: the last one again -- the issue stays unconditional, so the compiler's vmcnt counting keeps
    // the double buffer: a branch around the loads makes it wait for everything in flight)
    auto issue = [&](int b, d2_t (&s)[U][JR], d2_t (&w)[U][R]) {
        b = min(b, nbt - 1);
        const bool second = b >= ta.nb;
        const TrimvGroup& t = second ? tb : ta;
        const int c0 = t.c_lo + (second ? b - ta.nb : b) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(c0 + u, t.c_last);
            const int k = (c << 7) + 2 * lane;
#pragma unroll
            for (int r = 0; r < JR; ++r) {
                const int j = min(t.j0 + r, nrows - 1);
                s[u][r] = *reinterpret_cast<const d2_t*>(M + (int64_t)j * ld + k);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) w[u][i] = *reinterpret_cast<const d2_t*>(W + (int64_t)i * ldw + k);
        }
    };
    auto consume = [&](int b, const d2_t (&s)[U][JR], const d2_t (&w)[U][R]) {
        const bool second = b >= ta.nb;
        const TrimvGroup& t = second ? tb : ta;
        const int c0 = t.c_lo + (second ? b - ta.nb : b) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = ((c0 + u) << 7) + 2 * lane;     // the INTENDED columns (a clamped chunk masks itself out)
#pragma unroll
            for (int r = 0; r < JR; ++r) {
                const int j = t.j0 + r;
                const int lo = mode == 1 ? j : 0;
                int hi = mode == 0 ? j + 1 : ncols;
                if (j >= nrows || c0 + u > t.c_last) hi = 0;
                const double m0 = (k >= lo && k < hi) ? s[u][r].x : 0.0;
                const double m1 = (k + 1 >= lo && k + 1 < hi) ? s[u][r].y : 0.0;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    acc[r][i] = __builtin_fma(m0, w[u][i].x, acc[r][i]);
                    acc[r][i] = __builtin_fma(m1, w[u][i].y, acc[r][i]);
                }
            }
        }
    };
    auto flush = [&](const TrimvGroup& t) {
#pragma unroll
        for (int r = 0; r < JR; ++r)
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double s = wave_sum_f64(acc[r][i]);
                if (lane == 0 && t.j0 + r < nrows) V[(int64_t)i * ldv + t.j0 + r] = s;
                acc[r][i] = 0.0;
            }
    };

    if (nbt == 0) {   // (rows without any column: only possible with ncols == 0)
        flush(ta);
        if (gb != ga) flush(tb);
        return;
    }
    if (ta.nb == 0) flush(ta);
    issue(0, sb[0], wb[0]);
    int b = 0;
    for (; b + 2 <= nbt; b += 2) {
        issue(b + 1, sb[1], wb[1]);
        consume(b, sb[0], wb[0]);
        if (b == ta.nb - 1) flush(ta);
        issue(b + 2, sb[0], wb[0]);
        consume(b + 1, sb[1], wb[1]);
        if (b + 1 == ta.nb - 1) flush(ta);
    }
    if (b < nbt) {
        consume(b, sb[0], wb[0]);
        if (b == ta.nb - 1) flush(ta);
    }
    if (gb != ga) flush(tb);
}

// blockIdx.y = set b of a batched evaluation: M, W, V move by b * (mstride, wstride, vstride) elements (0 for a single one)
template <int R, int JR, int U>
__global__ __launch_bounds__(256) void mfgp_trimv_f64(const double* __restrict__ M, int ld, const double* __restrict__ W, int ldw,
                                                      double* __restrict__ V, int ldv, int nrows, int ncols, int mode,
                                                      long long mstride, long long wstride, long long vstride) {
    M += blockIdx.y * mstride; W += blockIdx.y * wstride; V += blockIdx.y * vstride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    trimv_wave<R, JR, U>(M, ld, W, ldw, V, ldv, nrows, ncols, mode, wv, lane);
}

// alpha = X^T z (mode 1 over the mirrored S: the upper part holds X^T) AND, in the launch's extra last workgroup, the scalars of
// the solve -- z^T z and the log-det from the leaf's partials; they need z only, which the previous launch completed: one launch
// floor less per evaluation.  blockIdx.y = set b of a batched evaluation: S moves by b * sstride, z and alpha by b * vstride, the
// log-det partials by b * ldstride, the scalars by b * scstride elements (all 0 for a single one)
template <int JR, int U>
__global__ __launch_bounds__(256) void mfgp_alpha_finish_f64(const double* __restrict__ S, int ld, const double* __restrict__ z,
                                                             double* __restrict__ alpha, int Np, const double* __restrict__ logdet_part,
                                                             int nblk, double* __restrict__ scalars, long long sstride,
                                                             long long vstride, int ldstride, int scstride) {
    S += blockIdx.y * sstride; z += blockIdx.y * vstride; alpha += blockIdx.y * vstride;
    logdet_part += blockIdx.y * ldstride; scalars += blockIdx.y * scstride;
    if (blockIdx.x == gridDim.x - 1) {   // the extra workgroup: scalars
        __shared__ double red[256];
        const int tid = threadIdx.x;
        double s = 0.0;
        for (int i = tid; i < Np; i += 256) s += z[i] * z[i];
        red[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            scalars[0] = red[0];
            double ldet = 0.0;
            for (int b = 0; b < nblk; ++b) ldet += logdet_part[b];
            scalars[1] = 2.0 * ldet;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    trimv_wave<1, JR, U>(S, ld, z, 0, alpha, 0, Np, Np, 1, wv, lane);
}

// The means of a small predict ride in the variance product's launch: W[i] . alpha for the `rows` real test rows (the segmented
// row sums below, the same as launch_rowdot's mode 2 forms them -- the mean of a test row does not depend on the size of the
// batch it travels in).  They take the FIRST workgroups of the grid, two rows per wave: as the
// last workgroup of the launch -- rounds 6's first form -- one CU read all the rows alone behind everybody else (64 rows x 64 KiB
// at N = 8192: a 0.03 - 0.08 ms tail, the whole difference between the 16- and the 8-row product).
static inline int mean_blocks(int rows) { return ((rows + 1) / 2 + 3) / 4; }

// ---- full-length row sums (the predictive means  W[i] . alpha) ------------------------------------------------------------
// A row's sum is DEFINED as the sum, in order, of its SEGMENT sums: segment s = the MEAN_SEG chunks (of 128 columns) from
// s MEAN_SEG on; inside a segment lane l runs one FMA chain over its two columns of every chunk in order, and the 64 chains are
// folded in the fixed order of wave_sum_f64.  Every mean of the engine is formed this way (mfgp_rowmean_f64 = launch_rowdot's
// mode 2, the mean blocks of the small predicts' product launches), whoever walks the row: one wave alone, segment after segment,
// or -- a few rows of many columns, the low-fidelity mean of a level-chained N* = 1 predict is ONE row of N_lf columns -- the eight
// waves of a workgroup, a segment each (a row of 16384 columns: one round of 2 x 16 KiB requests instead of 64 dependent ones at
// the start of the round, 8 after the first fix).  The same bits either way, so the mean of a test row does not depend on the
// batch it travels in.  (Rows of up to MEAN_SEG chunks are one segment: the sums of rounds 1-5.)
constexpr int MEAN_SEG = 16;
__device__ __forceinline__ double mean_segment(const double* __restrict__ row, const double* __restrict__ x, int sg, int nchunk,
                                               int ncols, int lane) {
    d2_t mv[MEAN_SEG], xv[MEAN_SEG];
#pragma unroll
    for (int u = 0; u < MEAN_SEG; ++u) {                      // the whole segment requested at once
        const int c = min(sg * MEAN_SEG + u, nchunk - 1);
        mv[u] = *reinterpret_cast<const d2_t*>(row + (c << 7) + 2 * lane);
        xv[u] = *reinterpret_cast<const d2_t*>(x + (c << 7) + 2 * lane);
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < MEAN_SEG; ++u) {
        const int c = sg * MEAN_SEG + u, k = (c << 7) + 2 * lane;
        const double m0 = (c < nchunk && k < ncols) ? mv[u].x : 0.0;      // (selects, as everywhere: what lies beyond never reaches a sum)
        const double m1 = (c < nchunk && k + 1 < ncols) ? mv[u].y : 0.0;
        acc = __builtin_fma(m0, xv[u].x, acc);
        acc = __builtin_fma(m1, xv[u].y, acc);
    }
    return wave_sum_f64(acc);
}
// one wave, one row: every lane returns the row's sum
__device__ __forceinline__ double mean_row(const double* __restrict__ row, const double* __restrict__ x, int ncols, int lane) {
    const int nchunk = (ncols + 127) >> 7, nseg = (nchunk + MEAN_SEG - 1) / MEAN_SEG;
    double tot = 0.0;
    for (int sg = 0; sg < nseg; ++sg) tot += mean_segment(row, x, sg, nchunk, ncols, lane);
    return tot;
}
__device__ __forceinline__ void predv_mean_block(const double* __restrict__ W, int ld, const double* __restrict__ alpha,
                                                 double* __restrict__ mean, int rows, int Np, int block, int wave, int lane) {
    // (two rows per wave, a short and a long end of the list, as the launch's grid was sized: mean_blocks)
    const int wv = block * 4 + wave;
    const int ja = wv, jb = rows - 1 - wv;
    if (ja > jb) return;
    const double sa = mean_row(W + (int64_t)ja * ld, alpha, Np, lane);
    if (lane == 0) mean[ja] = sa;
    if (jb != ja) {
        const double sb = mean_row(W + (int64_t)jb * ld, alpha, Np, lane);
        if (lane == 0) mean[jb] = sb;
    }
}
// y[j] = M[j] . x over ncols columns: the eight waves of a workgroup share 8 / wps rows, wps = min(8, segments of a row rounded down
// to a power of two) waves per row -- wave w of a row takes its segments w, w + wps, ... -- and one lane per row adds the segment
// sums in order.  (wps = 1: a wave walks its row alone, rows of up to 2048 columns.)
static inline int rowmean_wps(int ncols) {
    const int nseg = (((ncols + 127) >> 7) + MEAN_SEG - 1) / MEAN_SEG;
    return nseg >= 8 ? 8 : (nseg >= 4 ? 4 : (nseg >= 2 ? 2 : 1));
}
__global__ __launch_bounds__(512) void mfgp_rowmean_f64(const double* __restrict__ M, int ld, const double* __restrict__ x,
                                                        double* __restrict__ y, int nrows, int ncols, int wps) {
    __shared__ double part[512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 8 / wps, slot = wave / wps, w = wave % wps;
    const int j = blockIdx.x * rpw + slot;
    const int nchunk = (ncols + 127) >> 7, nseg = (nchunk + MEAN_SEG - 1) / MEAN_SEG;     // (nseg <= 512: launch_rowdot)
    if (wps == 1) {                                      // (whole workgroup: no barrier ahead)
        if (j < nrows) {
            const double sv = mean_row(M + (int64_t)j * ld, x, ncols, lane);
            if (lane == 0) y[j] = sv;
        }
        return;
    }
    double* const mine = part + (wps == 8 ? 0 : slot * 8);                                  // (wps < 8: nseg <= 7)
    if (j < nrows) {
        const double* row = M + (int64_t)j * ld;
        for (int sg = w; sg < nseg; sg += wps) {
            const double sv = mean_segment(row, x, sg, nchunk, ncols, lane);
            if (lane == 0) mine[sg] = sv;
        }
    }
    __syncthreads();
    if (w == 0 && lane == 0 && j < nrows) {
        double tot = 0.0;
        for (int sg = 0; sg < nseg; ++sg) tot += mine[sg];
        y[j] = tot;
    }
}

static inline int trimv_blocks(int nrows, int JR) {
    const int G = (nrows + JR - 1) / JR;
    return ((G + 1) / 2 + 3) / 4;
}

// rows per wave / chunks per batch of the single-vector form: 4 rows x 2 chunks (measured at N = 8192, variance stage incl. its
// finishing launch / append: "4 2" 0.0498 / 0.132 ms, "1 8" 0.0508 / 0.139, "2 4" 0.0555 / 0.143: profiles/r06_trimv_shapes.txt)
constexpr int TRIMV1_JR = 4, TRIMV1_U = 2;

void launch_rowdot(hipStream_t s, const double* M, int ld, const double* x, double* y, int nrows, int ncols, int mode, int nbatch,
                   long long mstride, long long xstride, long long ystride) {
    if (nrows <= 0) return;
    if (mode == 2 && nbatch <= 1) {
        // the means: the segmented row sums above, a row's segments spread over up to eight waves (measured, mean-only predict call at
        // N = 16384, 4-rows-per-wave walk / one row per wave, 8 chunks per batch / this: 1 row 0.060 / 0.034 / 0.026 ms, 64 rows
        // 0.110 / 0.049 / 0.030, 1024 rows 0.138 / 0.084 / 0.083, 16384 rows 0.868 / 0.873 / 0.856)
        const int nseg = (((ncols + 127) >> 7) + MEAN_SEG - 1) / MEAN_SEG;
        int wps = nseg <= 512 ? rowmean_wps(ncols) : 1;      // (the segment sums of a row wait in 512 LDS slots)
        if (nseg <= 2 && nrows >= 2048) wps = 1;             // (two segments and many rows: 0.059 / 0.080 / 0.159 ms against 0.063 / 0.085 /
                                                             // 0.168 at 2048 / 4096 / 8192 rows, N = 4096; the same bits)
        const int rpw = 8 / wps;
        hipLaunchKernelGGL(mfgp_rowmean_f64, dim3((nrows + rpw - 1) / rpw), dim3(512), 0, s, M, ld, x, y, nrows, ncols, wps);
        return;
    }
    hipLaunchKernelGGL((mfgp_trimv_f64<1, TRIMV1_JR, TRIMV1_U>), dim3(trimv_blocks(nrows, TRIMV1_JR), nbatch > 0 ? nbatch : 1), dim3(256), 0, s, M, ld,
                       x, 0, y, 0, nrows, ncols, mode, mstride, xstride, ystride);
}

void launch_alpha_finish(hipStream_t s, const double* S, int ld, const double* z, double* alpha, int Np, const double* logdet_part,
                         int nblk, double* scalars, int nbatch, long long sstride, long long vstride, int ldstride, int scstride) {
    hipLaunchKernelGGL((mfgp_alpha_finish_f64<4, 2>), dim3(trimv_blocks(Np, 4) + 1, nbatch > 0 ? nbatch : 1), dim3(256), 0, s, S, ld,
                       z, alpha, Np, logdet_part, nblk, scalars, sstride, vstride, ldstride, scstride);
}

// Predict with <= 8 test rows: V[i][j] = sum_{k <= j} W[i][k] X[j][k], i < R (R in {1, 2, 4, 8}: the caller rounds its row
// count up; the panel W holds at least that many rows), j < Np, X = L^-1 read from the lower part of the mirrored S -- and, in
// the launch's first workgroups, the means W[i] . alpha of the `rows` real test rows (predv_mean_block).
template <int R, int JR, int U>
__global__ __launch_bounds__(256) void mfgp_predv_rows_f64(const double* __restrict__ S, int ld, const double* __restrict__ W,
                                                           double* __restrict__ V, int Np, const double* __restrict__ alpha,
                                                           double* __restrict__ mean, int rows) {
    const int lane = threadIdx.x & 63;
    const int nmean = ((rows + 1) / 2 + 3) / 4;
    if ((int)blockIdx.x < nmean) {
        predv_mean_block(W, ld, alpha, mean, rows, Np, blockIdx.x, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
        return;
    }
    const int wv = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - nmean) * 4 + (threadIdx.x >> 6));
    trimv_wave<R, JR, U>(S, ld, W, ld, V, ld, Np, Np, 0, wv, lane);
}

// The same product for 8 test rows (written for 8 and 16; from 9 rows up the matrix-pipe form below is faster -- 16 rows: 0.080 ms
// here, issue-bound on the VALU, 0.068 there): the R right-hand sides would cost R / JR times the bytes of S in L2 -> L1 reads and
// fill the wave's memory queue (measured: 3.8 / 2.9 TB/s at R = 8 / 16 in the form above), so the NW waves of a workgroup walk
// the SAME chunks of NW JR consecutive rows in lock step and share the R x 128 tile of W through LDS: W crosses L1 once per
// NW JR rows of S, the fragments come out of LDS (ds_read_b128, conflict-free).  The tile is staged by LDS-DMA
// (global_load_lds_dwordx4: one wave-instruction lands 1 KiB = one row's chunk lane-linearly, no staging registers) D steps
// ahead into D + 1 buffers, and S is fetched D chunks ahead in registers (D + 1 register sets taking turns): vmcnt counts in
// order, so the wait for a tile retires everything issued before it -- the distance of the TILE is what bounds the bytes of S a
// wave keeps in flight (measured at R = 16: 3.3 TB/s with the tile fetched inside the step or two steps ahead, whatever the
// distance of S).  One barrier per chunk.  Row block b is paired with block NB-1-b as above; sums and their order per (i, j)
// are exactly those of trimv_wave -- a row's result does not depend on which of the two forms produced it.
// The waits on the DMA are counted by hand against the ISSUE order, which the sched_barriers pin (tile, then S, then the
// step's arithmetic); `hipcc -S` of this file shows, per step: WQ global_load_lds, JR global_load_dwordx4, ..., s_waitcnt
// vmcnt(JR + (D - 1)(WQ + JR)), s_barrier.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int R, int JR, int NW, int D>
__global__ __launch_bounds__(64 * NW) void mfgp_predv_rows_lds_f64(const double* __restrict__ S, int ld, const double* __restrict__ W,
                                                                  double* __restrict__ V, int Np, const double* __restrict__ alpha,
                                                                  double* __restrict__ mean, int rows) {
    static_assert(R % NW == 0, "the waves stage R / NW rows of the W tile each");
    constexpr int BR = NW * JR;                // rows of S per workgroup and block
    constexpr int NS = D + 1;
    constexpr int WQ = R / NW;
    __shared__ __attribute__((aligned(1024))) d2_t wl[NS][R][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmean = ((rows + 1) / 2 + 3) / 4;
    if ((int)blockIdx.x < nmean) {
        if (wave < 4) predv_mean_block(W, ld, alpha, mean, rows, Np, blockIdx.x, wave, lane);
        return;
    }
    const int NB = Np / BR;
    const int bA = (int)blockIdx.x - nmean, bB = NB - 1 - bA;                       // (NB is even: Np is a multiple of 128, 2 BR divides 128)
    const int nA = ((bA * BR + BR - 1) >> 7) + 1, nB = ((bB * BR + BR - 1) >> 7) + 1;
    const int T = nA + nB;
    auto row0 = [&](int t) { return (t < nA ? bA : bB) * BR + wave * JR; };
    auto chunk = [&](int t) { return t < nA ? t : t - nA; };
    d2_t ss[NS][JR];
    double acc[JR][R];
#pragma unroll
    for (int r = 0; r < JR; ++r)
#pragma unroll
        for (int i = 0; i < R; ++i) acc[r][i] = 0.0;
    auto load_s = [&](int t, d2_t (&s)[JR]) {
        t = min(t, T - 1);
        const double* p = S + (int64_t)row0(t) * ld + (chunk(t) << 7) + 2 * lane;
#pragma unroll
        for (int r = 0; r < JR; ++r) s[r] = *reinterpret_cast<const d2_t*>(p + (int64_t)r * ld);
    };
    const double* const w_lane = W + (int64_t)(wave * WQ) * ld + 2 * lane;
    auto dma_w = [&](int t, int buf) {          // this wave's WQ rows of tile t -> wl[buf]
        t = min(t, T - 1);
        const int c = chunk(t) << 7;
#pragma unroll
        for (int i = 0; i < WQ; ++i)
            __builtin_amdgcn_global_load_lds(w_lane + ((int64_t)i * ld + c), (lds_ptr_t)&wl[buf][wave * WQ + i][0], 16, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < D; ++u) dma_w(u, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tiles 0 .. D-1 have landed (this wave's part); once per workgroup
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int u = 0; u < D; ++u) {
        __builtin_amdgcn_sched_barrier(0);      // (program order = issue order: the waits below are counted against it)
        load_s(u, ss[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // one step: consume chunk t out of register set / tile buffer t mod NS, fetch chunk and tile t + D into the set and the
    // buffer step t - 1 consumed (a rotation by copies would have to wait for the loads it copies)
    auto step = [&](int t, int buf, const d2_t (&cur)[JR], d2_t (&nxt)[JR]) {
        dma_w(t + D, buf == 0 ? NS - 1 : buf - 1);   // the tile first, and kept first
        __builtin_amdgcn_sched_barrier(0);
        load_s(t + D, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int j0 = row0(t), k = (chunk(t) << 7) + 2 * lane;
#pragma unroll
        for (int r = 0; r < JR; ++r) {
            const double m0 = (k <= j0 + r) ? cur[r].x : 0.0;
            const double m1 = (k + 1 <= j0 + r) ? cur[r].y : 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const d2_t w = wl[buf][i][lane];
                acc[r][i] = __builtin_fma(m0, w.x, acc[r][i]);
                acc[r][i] = __builtin_fma(m1, w.y, acc[r][i]);
            }
        }
        if (t == nA - 1 || t == T - 1) {
#pragma unroll
            for (int r = 0; r < JR; ++r)
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const double v = wave_sum_f64(acc[r][i]);
                    if (lane == 0) V[(int64_t)i * ld + j0 + r] = v;
                    acc[r][i] = 0.0;
                }
        }
        // tile t + 1 (issued at the top of step t + 1 - D) has landed: younger than it are that step's S loads and the D - 1
        // steps since
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JR + (D - 1) * (WQ + JR)) : "memory");
        __builtin_amdgcn_s_barrier();
    };
    int buf = 0;
    for (int t = 0; t < T; t += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (t + u >= T) break;
            step(t + u, buf, ss[u], ss[(u + D) % NS]);
            buf = buf == NS - 1 ? 0 : buf + 1;
        }
    }
}

// 9 .. 64 test rows: the products move to the matrix pipe (v_mfma_f64_16x16x4: 16 test rows x 16 rows of S x 4 k per instruction),
// which wants a lane to hold ONE k of 16 DIFFERENT rows of S -- the opposite of the coalesced read.  Rounds 2-5 fetched the
// fragments in that shape straight from memory (32 B per lane: a wave-instruction = 16 rows x 128 B, 0.30 of the HBM peak: VERDICT
// r5 weak #2); here S goes through LDS by LDS-DMA in the coalesced shape (global_load_lds_dwordx4: one instruction = 2 rows x 512
// contiguous bytes, no staging registers) and the fragments come out of LDS.  The 16-byte chunks of a row sit XOR-swizzled by
// the row (on the per-lane SOURCE address of the DMA and on the fragment read alike), so the 16 lanes that read one k-slot of 16
// rows hit 16 different bank groups.  One workgroup = one block of 16 rows of S; its four waves take the block's 64-column stages
// in turn (wave w: stages w, w + 4, ...), each with its own 2 x 8 KiB of LDS and no barrier inside the loop, and their partial
// tiles meet through LDS at the end in a fixed order (deterministic).  Per stage and wave: the W fragments of the stage
// (RT x 16 doubles per lane, L2 hits) are requested, ONE wait retires them together with the stage's own DMA, the DMA of the wave's
// NEXT stage is issued and stays in flight behind the stage's 16 RT MFMAs.  Consecutive workgroups alternate between the short
// and the long end of the triangle (block 0, NB-1, 1, NB-2, ...), so that the workgroups resident together carry like sums of
// work; masks (k <= j) touch the last stage of a block only.
// The A operand of those MFMAs -- lane (r, q) supplies W[16 i + r][k + q] -- read from the row-major panel is 16 rows x 128 B per
// wave-instruction with the rows a whole ld apart (64 KiB at N = 8192: one L2 channel), the shape that held rounds 2-5's kernel at
// 0.3 of the HBM peak.  The panel is therefore re-laid once per predict (<= 4 MB, out of L2) into fragment order:
//     Wt[((i * Np / 4 + k / 4) * 16 + r) * 4 + k % 4] = W[16 i + r][k]
// so that the 64 lanes of a fragment load read 2 KiB contiguous.
__global__ __launch_bounds__(256) void mfgp_panel_fragments_f64(const double* __restrict__ W, int ld, double* __restrict__ Wt, int Np,
                                                                int RT) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one 32-byte piece: (tile i, k / 4, row r), r fastest
    const int64_t per_tile = (int64_t)(Np >> 2) * 16;
    if (idx >= per_tile * RT) return;
    const int i = (int)(idx / per_tile);
    const int64_t rem = idx - (int64_t)i * per_tile;
    const int kq = (int)(rem >> 4), r = (int)(rem & 15);
    const double* src = W + (int64_t)(16 * i + r) * ld + 4 * kq;
    const d2_t v0 = *reinterpret_cast<const d2_t*>(src), v1 = *reinterpret_cast<const d2_t*>(src + 2);
    double* dst = Wt + idx * 4;
    *reinterpret_cast<d2_t*>(dst) = v0;
    *reinterpret_cast<d2_t*>(dst + 2) = v1;
}

template <int RT, int KS>
__global__ __launch_bounds__(256, 2) void mfgp_predv_mfma_f64(const double* __restrict__ W, const double* __restrict__ Wt, const double* __restrict__ S,
                                                              double* __restrict__ V, int ld, int Np, const double* __restrict__ alpha,
                                                              double* __restrict__ mean, int rows) {
    constexpr int STAGE_B = 16 * KS * 8;         // bytes per stage: 16 rows x KS columns
    constexpr int NC = KS / 16;                  // 16-column chunks per stage
    constexpr int RPI = 1024 / (KS * 8);         // rows per DMA instruction (1 KiB): 2 x 512 B (KS = 64) or 4 x 256 B (KS = 32)
    constexpr int SPR = KS / 2;                  // 16-byte slots per row
    __shared__ __attribute__((aligned(1024))) char lds[4 * 2 * STAGE_B < 24576 ? 24576 : 4 * 2 * STAGE_B];   // (>= the reduction's 3 RT x 2 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmean = ((rows + 1) / 2 + 3) / 4;
    if ((int)blockIdx.x < nmean) {               // the means of the real test rows: the first workgroups of the launch
        predv_mean_block(W, ld, alpha, mean, rows, Np, blockIdx.x, wave, lane);
        return;
    }
    const int g = (Np >> 4) - 1 - ((int)blockIdx.x - nmean);   // longest block first (measured against two interleaved orders: r06 lab notes)
    const int j0 = g << 4;
    const int n = (j0 + 15) / KS + 1;            // stages of the block: columns 0 .. j0 + 15
    char* const my = lds + wave * (2 * STAGE_B);
    const int r = lane & 15, q = lane >> 4;
    // DMA: instruction u of a stage covers rows RPI u .. of the block; lane l lands at byte u * 1024 + 16 l = (row RPI u + l / SPR,
    // slot l % SPR) and therefore fetches chunk slot ^ (row & (SPR - 1) & 15) of that row
    const int d_row = lane / SPR, d_slot = lane % SPR;
    auto dma = [&](int t, int st) {
#pragma unroll
        for (int u = 0; u < 16 / RPI; ++u) {
            const int row = RPI * u + d_row;
            const double* src = S + (int64_t)(j0 + row) * ld + t * KS + 2 * (d_slot ^ (row & 15 & (SPR - 1)));
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(my + st * STAGE_B + u * 1024), 16, 0, 0);
        }
    };
    // W fragments of a stage (lane (r, q): 4 consecutive k per 16-column chunk, the same k order as the S fragment), in fragment order
    auto load_a = [&](int t, d2_t (&a)[RT][NC][2]) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double* wp = Wt + (((int64_t)i * (Np >> 2) + ((t * KS + 16 * c) >> 2) + q) * 16 + r) * 4;   // 2 KiB per wave-instruction pair
                a[i][c][0] = *reinterpret_cast<const d2_t*>(wp);
                a[i][c][1] = *reinterpret_cast<const d2_t*>(wp + 2);
            }
    };
    // acc[tile][4]: the four result registers of one 16 x 16 tile, D[row = q + 4 e][col = r].  (The v_mfma_f64_4x4x4_4b form the
    // tile GEMM uses -- four rotations of the A fragment, here by DPP row rotations of the loaded registers -- was measured against
    // this one in round 6 and lost, 0.092 / 0.160 ms against 0.085 / 0.130 at 32 / 64 test rows: the rotations' v_mov_dpp compete
    // with the MFMAs for the SIMD's issue slot.  The 16x16x4 kernel runs the matrix pipe 0.54 / 0.71 busy: profiles/r06_adapt_sq.txt.)
    double acc[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[t][x] = 0.0;
    auto compute = [&](int t, int st, const d2_t (&a)[RT][NC][2]) {
        const int kb = t * KS;
        const bool diag = kb + KS > j0;          // the stage reaches past the block's first diagonal entry: keep k <= j
        const char* const base = my + st * STAGE_B + r * (KS * 8);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int s0 = 8 * c + 2 * q, sw = r & (SPR - 1);
            d2_t b0 = *reinterpret_cast<const d2_t*>(base + ((s0 ^ sw) << 4));
            d2_t b1 = *reinterpret_cast<const d2_t*>(base + (((s0 + 1) ^ sw) << 4));
            if (diag) {
                const int kk = kb + 16 * c + 4 * q, j = j0 + r;
                if (kk + 0 > j) b0[0] = 0.0;
                if (kk + 1 > j) b0[1] = 0.0;
                if (kk + 2 > j) b1[0] = 0.0;
                if (kk + 3 > j) b1[1] = 0.0;
            }
            const double bv[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                d4_t d = (d4_t){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c][0][0], bv[0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c][0][1], bv[1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c][1][0], bv[2], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c][1][1], bv[3], d, 0, 0, 0);
                acc[i][0] = d[0]; acc[i][1] = d[1]; acc[i][2] = d[2]; acc[i][3] = d[3];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this stage's fragment reads are done before a DMA may overwrite it
    };
    // One step: ONE wait retires everything in flight -- the stage's DMA and its W fragments, both requested a whole step ago --
    // then the NEXT stage of this wave is requested (fragments into the other register set, DMA into the other buffer) and stays in
    // flight behind this stage's RT KS / 4 MFMAs.  The wait is the builtin, not inline assembly: the compiler's own wait-count
    // bookkeeping then knows the fragments have landed -- with an LDS-DMA pending it otherwise waits for vmcnt(0) at their first use,
    // i.e. for the prefetch.
    d2_t a0[RT][NC][2], a1[RT][NC][2];
    if (wave < n) { load_a(wave, a0); dma(wave, 0); }
    for (int t = wave; t < n; t += 8) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) expcnt(7) lgkmcnt(15)
        __builtin_amdgcn_sched_barrier(0);
        if (t + 4 < n) { load_a(t + 4, a1); dma(t + 4, 1); }
        __builtin_amdgcn_sched_barrier(0);
        compute(t, 0, a0);
        if (t + 4 >= n) break;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 8 < n) { load_a(t + 8, a0); dma(t + 8, 0); }
        __builtin_amdgcn_sched_barrier(0);
        compute(t + 4, 1, a1);
    }
    // the four partial tiles: waves 1..3 through LDS (every wave's DMAs have been retired by its last wait), summed by wave 0 in a
    // fixed order
    __syncthreads();
    double* const red = reinterpret_cast<double*>(lds);
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(((wave - 1) * RT + i) * 4 + e) * 64 + lane] = acc[i][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double v = ((acc[i][e] + red[((0 * RT + i) * 4 + e) * 64 + lane]) + red[((1 * RT + i) * 4 + e) * 64 + lane]) +
                                 red[((2 * RT + i) * 4 + e) * 64 + lane];
                V[(int64_t)(i * 16 + q + 4 * e) * ld + j0 + r] = v;
            }
    }
}

void launch_predv_mfma(hipStream_t s, int RT, const double* W, double* Wt, const double* S, double* V, int ld, int Np,
                       const double* alpha, double* mean, int rows) {
    RT = RT <= 1 ? 1 : (RT == 2 ? 2 : (RT == 3 ? 3 : 4));
    const int64_t pieces = (int64_t)(Np >> 2) * 16 * RT;
    hipLaunchKernelGGL(mfgp_panel_fragments_f64, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, W, ld, Wt, Np, RT);
    const dim3 grid(Np / 16 + mean_blocks(rows)), blk(256);
    // (stage width 64 columns where the W fragments of two stages fit the registers, 32 from three row tiles up)
#define PREDV_MFMA(rt, k) hipLaunchKernelGGL((mfgp_predv_mfma_f64<rt, k>), grid, blk, 0, s, W, Wt, S, V, ld, Np, alpha, mean, rows)
    if (RT == 1) PREDV_MFMA(1, 64);
    else if (RT == 2) PREDV_MFMA(2, 64);
    else if (RT == 3) PREDV_MFMA(3, 32);
    else PREDV_MFMA(4, 32);
#undef PREDV_MFMA
}

// ---- the register-staged form of the same product (9 .. 64 test rows from Np = 3072: predv_mfma2_pays) --------------------------
// In the kernel above every wave fetches its own copy of the W fragments from L2 (RT x the bytes of S), a workgroup is one 16-row
// block (its waves split the block's stages and meet through LDS), and the panel is re-laid in a launch of its own.  Here:
//   * a workgroup = four waves = four 16-row blocks of S (64 consecutive rows) that walk the SAME 32-column stages in step, so
//     the stage's W tile (RT x 16 rows x 32 columns) is fetched ONCE per workgroup -- each wave a quarter, straight from the
//     row-major panel: an instruction = 4 rows x 256 contiguous bytes, the full-rate shape, no re-laid copy -- and shared in LDS;
//   * S and the W quarter are register loads in that coalesced shape issued P stages ahead; the compiler's own counted
//     s_waitcnt vmcnt retires exactly one stage per step, nothing else shares the counter (no LDS-DMA);
//   * LDS is a transposition buffer only: 4 KiB per wave for S (own wave: no barrier), 2 x RT x 4 KiB for the W tile (written
//     for stage s + 1 while stage s is consumed: one barrier per stage), 16-byte chunks XOR-swizzled by the row as above;
//   * the triangle is cut into equal shares: with the (group, stage) pairs in one line (group G has 2 G + 2 stages), workgroup b
//     takes L consecutive stages [b L, (b + 1) L) -- every workgroup the same number of bytes and of MFMAs, two resident per
//     CU, no reduction between waves -- and writes one partial 64-column tile per group it touches into plane (b - first
//     workgroup of the group); the finishing launch adds a column's planes in that fixed order (deterministic) before it squares.
// What bounds it (profiles/r06_adapt_sq.txt): the issue rate of v_mfma_f64_16x16x4 (64 busy cycles, one per ~100: gemm_f64.hip's
// header) -- 64 test rows are 4.3 GFLOP = 0.086 ms at that rate against 0.085 measured for the launch, 32 rows 0.043 against
// 0.054; the read of S alone is 0.046.  Four stages ahead instead of two measured the same or slower, and so did the
// v_mfma_f64_4x4x4_4b form in this kernel (S fragment rotated by four LDS reads: 0.084 / 0.123 ms at 32 / 64 rows; by DPP
// row_ror: 0.080 / 0.117; the 16x16x4 form then: 0.070 / 0.102) -- the rotations cost more issue slots than the shape saves.
template <int RT, int P>
__global__ __launch_bounds__(256, 2) void mfgp_predv_mfma2_f64(const double* __restrict__ W, const double* __restrict__ S,
                                                               double* __restrict__ Vp, int ld, int Np, const double* __restrict__ alpha,
                                                               double* __restrict__ mean, int rows, int L) {
    static_assert(P == 2 || P == 4, "the W tile's buffer index is the parity of the step");
    constexpr int WT_B = RT * 4096;              // bytes of a stage's W tile
    __shared__ __attribute__((aligned(1024))) char lds[2 * WT_B + 4 * 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmean = ((rows + 1) / 2 + 3) / 4;
    if ((int)blockIdx.x < nmean) {
        predv_mean_block(W, ld, alpha, mean, rows, Np, blockIdx.x, wave, lane);
        return;
    }
    const int b = (int)blockIdx.x - nmean;
    const int NG = Np >> 6;
    const int T = NG * (NG + 1);                 // stages of the whole triangle: group G holds stages G (G + 1) .. (G + 1)(G + 2) - 1
    int lin = b * L;                             // first stage of this workgroup, in the line
    const int lin_end = min(lin + L, T);
    int G = (int)((sqrt(4.0 * (double)lin + 1.0) - 1.0) * 0.5);
    while (G * (G + 1) > lin) --G;
    while ((G + 1) * (G + 2) <= lin) ++G;
    char* const Tb = lds + 2 * WT_B + wave * 4096;
    const int r = lane & 15, q = lane >> 4;
    const int plane = 64 * ld;                   // doubles between two planes

    while (lin < lin_end) {
        const int g_lo = G * (G + 1);
        const int s_lo = lin - g_lo, s_hi = min(lin_end - g_lo, 2 * G + 2);   // stages [s_lo, s_hi) of group G
        const int n = s_hi - s_lo;
        const int j0 = 64 * G + 16 * wave;
        const double* const Srow = S + (int64_t)(j0 + q) * ld + 2 * r;       // lane (row q of an instruction's four, chunk r)
        // this wave's quarter of a tile, instruction 0: rows 4 (wave RT + v) + q of the row-major panel, chunk r
        const double* const Wsh = W + (int64_t)(4 * wave * RT + q) * ld + 2 * r;
        d2_t sS[P][4], sW[P][RT];
        auto issue = [&](int st, d2_t (&xs)[4], d2_t (&xw)[RT]) {             // S of stage st, W quarter of stage st + 1
            const int a = min(st, s_hi - 1), aw = min(st + 1, s_hi - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) xs[u] = *reinterpret_cast<const d2_t*>(Srow + (int64_t)(4 * u) * ld + 32 * a);
#pragma unroll
            for (int v = 0; v < RT; ++v) xw[v] = *reinterpret_cast<const d2_t*>(Wsh + (int64_t)(4 * v) * ld + 32 * aw);
        };
        auto put_w = [&](int buf, const d2_t (&xw)[RT]) {
#pragma unroll
            for (int v = 0; v < RT; ++v) {
                const int row = 4 * (wave * RT + v) + q;
                *reinterpret_cast<d2_t*>(lds + buf * WT_B + row * 256 + ((r ^ (row & 15)) << 4)) = xw[v];
            }
        };
        double acc[RT][4];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.0;
        {   // the first stage's W tile, then P stages of requests
            d2_t w0[RT];
#pragma unroll
            for (int v = 0; v < RT; ++v) w0[v] = *reinterpret_cast<const d2_t*>(Wsh + (int64_t)(4 * v) * ld + 32 * s_lo);
            __builtin_amdgcn_sched_barrier(0);   // (first in the queue: its wait leaves the P stages in flight)
#pragma unroll
            for (int k = 0; k < P; ++k) {        // in THIS order: set 0 oldest, as it is on the loop's back edge
                issue(s_lo + k, sS[k], sW[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
            put_w(0, w0);
        }
        __syncthreads();
        // one step; KK = k mod P names the register set.  (The k loop below has ONE exit: with an exit inside the unrolled body
        // the compiler routes it through the loop's latch, and its wait-count bookkeeping then merges "one step since set 0 was
        // requested" into the loop header -- an s_waitcnt vmcnt(0) every P steps.)
        auto step = [&](auto KK, int k) {
            constexpr int kk = decltype(KK)::value;
            const int st = s_lo + k;
            // into LDS: this wave's stage of S, its quarter of the NEXT stage's W tile
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = 4 * u + q;
                *reinterpret_cast<d2_t*>(Tb + row * 256 + ((r ^ row) << 4)) = sS[kk][u];
            }
            put_w((kk + 1) & 1, sW[kk]);
            issue(st + P, sS[kk], sW[kk]);
            // the stage: lane (r, q) multiplies k = 16 c + 4 q + m of row j0 + r (v_mfma_f64_16x16x4, the fragment order of the
            // kernel above)
            const int kb = 32 * st;
            const bool diag = kb + 32 > j0;
            const char* const wt = lds + (kk & 1) * WT_B;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int s0 = 8 * c + 2 * q;
                d2_t b0 = *reinterpret_cast<const d2_t*>(Tb + r * 256 + ((s0 ^ r) << 4));
                d2_t b1 = *reinterpret_cast<const d2_t*>(Tb + r * 256 + (((s0 + 1) ^ r) << 4));
                if (diag) {
                    const int kx = kb + 16 * c + 4 * q, j = j0 + r;
                    if (kx + 0 > j) b0[0] = 0.0;
                    if (kx + 1 > j) b0[1] = 0.0;
                    if (kx + 2 > j) b1[0] = 0.0;
                    if (kx + 3 > j) b1[1] = 0.0;
                }
                const double bv[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const d2_t a0 = *reinterpret_cast<const d2_t*>(wt + (16 * i + r) * 256 + ((s0 ^ r) << 4));
                    const d2_t a1 = *reinterpret_cast<const d2_t*>(wt + (16 * i + r) * 256 + (((s0 + 1) ^ r) << 4));
                    d4_t d = (d4_t){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], bv[0], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], bv[1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], bv[2], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], bv[3], d, 0, 0, 0);
                    acc[i][0] = d[0]; acc[i][1] = d[1]; acc[i][2] = d[2]; acc[i][3] = d[3];
                }
            }
            __syncthreads();     // the next stage's W tile is complete; this stage's is free
        };
        const int nfull = n - n % P;
        for (int k0 = 0; k0 < nfull; k0 += P) {
            step(std::integral_constant<int, 0>{}, k0);
            step(std::integral_constant<int, 1>{}, k0 + 1);
            if constexpr (P > 2) {
                step(std::integral_constant<int, 2>{}, k0 + 2);
                step(std::integral_constant<int, 3>{}, k0 + 3);
            }
        }
        if (nfull < n) step(std::integral_constant<int, 0>{}, nfull);
        if (nfull + 1 < n) step(std::integral_constant<int, 1>{}, nfull + 1);
        if constexpr (P > 2)
            if (nfull + 2 < n) step(std::integral_constant<int, 2>{}, nfull + 2);
        // the partial tile of this (group, workgroup): plane = workgroups since the group's first
        double* const out = Vp + (int64_t)(b - g_lo / L) * plane + j0 + r;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(int64_t)(i * 16 + q + 4 * e) * ld] = acc[i][e];
        lin = g_lo + s_hi;
        ++G;
    }
}

// its finish, one workgroup per test row:  var[i] = max(kss - sum_j (sum_p Vp[p][i][j])^2, 1e-15) + add, the planes of a column's
// group in their fixed order
__global__ __launch_bounds__(1024) void mfgp_predv_finish_planes_f64(const double* __restrict__ Vp, int ld, int Np, int L, double kss,
                                                                     double add, double* __restrict__ var) {
    __shared__ double red[16];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t plane = (int64_t)64 * ld;
    const double* v = Vp + (int64_t)i * ld;
    double sv = 0.0;
    for (int j = 2 * tid; j < Np; j += 2048) {
        const int G = j >> 6;
        const int cnt = ((G + 1) * (G + 2) - 1) / L - (G * (G + 1)) / L + 1;
        d2_t t = (d2_t){0.0, 0.0};
        for (int p0 = 0; p0 < cnt; p0 += 8) {
            d2_t c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = *reinterpret_cast<const d2_t*>(v + (int64_t)min(p0 + u, cnt - 1) * plane + j);
                if (p0 + u >= cnt) c[u] = (d2_t){0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += c[u];
        }
        sv = __builtin_fma(t.x, t.x, sv);
        sv = __builtin_fma(t.y, t.y, sv);
    }
    sv = wave_sum_f64(sv);
    if (lane == 0) red[wave] = sv;
    __syncthreads();
    if (tid == 0) {
        double x = 0.0;
        for (int w = 0; w < 16; ++w) x += red[w];
        x = kss - x;
        if (!(x > 1e-15)) x = 1e-15;
        var[i] = x + add;
    }
}

// stages per workgroup: two workgroups per CU, all resident at once, every one the same share of the triangle (measured at
// Np = 8192, 32 / 64 test rows, variance stage: 512 shares 0.066 / 0.098 ms, 768 with three workgroups per CU 0.067 / 0.115,
// 1024 0.069 / 0.102 -- more shares are more planes)
int predv_mfma2_stages(int Np) {
    const int NG = Np >> 6, T = NG * (NG + 1);
    const int Lq = (T + 511) / 512;
    return Lq < 16 ? 16 : Lq;
}
// does the register-staged form pay?  Variance stage first form / this one, ms, at 16 / 32 / 48 / 64 test rows (r06 lab notes):
//   Np = 2048: 0.016 / 0.015, 0.022 / 0.021, 0.026 / 0.026, 0.029 / 0.031      Np = 3072: 0.019 / 0.017, 0.031 / 0.022, 0.036 / 0.028, 0.041 / 0.033
//   Np = 4096: 0.024 / 0.024, 0.036 / 0.030, 0.043 / 0.038, 0.053 / 0.048      Np = 6144: 0.038 / 0.032, 0.053 / 0.038, 0.064 / 0.047, 0.080 / 0.057
//   Np = 8192: 0.065 / 0.053, 0.088 / 0.061, 0.103 / 0.075, 0.127 / 0.092
bool predv_mfma2_pays(int rows, int Np) { return rows > 4 && Np >= 3072; }

void launch_predv_mfma2(hipStream_t s, int RT, const double* W, const double* S, double* Vp, int ld, int Np,
                        const double* alpha, double* mean, int rows) {
    RT = RT <= 1 ? 1 : (RT == 2 ? 2 : (RT == 3 ? 3 : 4));
    const int NG = Np >> 6, T = NG * (NG + 1), L = predv_mfma2_stages(Np);
    const dim3 grid((T + L - 1) / L + mean_blocks(rows)), blk(256);
    // (two stages ahead: four measured the same or slower -- 0.070 / 0.102 ms -- the product is bound by the issue rate of
    // v_mfma_f64_16x16x4, not by what is in flight)
#define PREDV_MFMA2(rt) hipLaunchKernelGGL((mfgp_predv_mfma2_f64<rt, 2>), grid, blk, 0, s, W, S, Vp, ld, Np, alpha, mean, rows, L)
    if (RT == 1) PREDV_MFMA2(1);
    else if (RT == 2) PREDV_MFMA2(2);
    else if (RT == 3) PREDV_MFMA2(3);
    else PREDV_MFMA2(4);
#undef PREDV_MFMA2
}
void launch_predv_finish_planes(hipStream_t s, int rows, const double* Vp, int ld, int Np, double kss, double add, double* var) {
    hipLaunchKernelGGL(mfgp_predv_finish_planes_f64, dim3(rows), dim3(1024), 0, s, Vp, ld, Np, predv_mfma2_stages(Np), kss, add, var);
}

void launch_predv_rows(hipStream_t s, int R, const double* W, const double* S, double* V, int ld, int Np, const double* alpha,
                       double* mean, int rows) {
    const dim3 blk(256);
#define TRIMVR(r, jr, u)                                                                                                  \
    hipLaunchKernelGGL((mfgp_predv_rows_f64<r, jr, u>), dim3(trimv_blocks(Np, jr) + mean_blocks(rows)), blk, 0, s, S, ld, W, V, Np, alpha, mean, rows)
    if (R <= 1) TRIMVR(1, TRIMV1_JR, TRIMV1_U);
    else if (R <= 2) TRIMVR(2, 4, 2);
    else if (R <= 4) TRIMVR(4, 4, 2);
    else   // 8 rows: the W tile through LDS (measured against the register form <8, 4, 2>: 0.056 / 0.072 ms)
        hipLaunchKernelGGL((mfgp_predv_rows_lds_f64<8, 4, 4, 2>), dim3(Np / 16 / 2 + mean_blocks(rows)), dim3(256), 0, s, S, ld, W, V, Np, alpha,
                           mean, rows);
#undef TRIMVR
}

// its finish, one workgroup per test row i:  var[i] = max(kss - sum_j V[i][j]^2, 1e-15) + add      (fixed-order sum)
__global__ __launch_bounds__(256) void mfgp_predv_finish_f64(const double* __restrict__ V, int ld, int Np, double kss, double add,
                                                             double* __restrict__ var) {
    __shared__ double red[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* v = V + (int64_t)i * ld;
    double sv = 0.0;
    for (int k0 = 2 * tid; k0 < Np; k0 += 512 * 8) {       // (Np is a multiple of 128: a chunk of 512 columns is whole or absent)
        d2_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                        // eight loads in flight, not a chain of Np / 512 round trips
            const int k = k0 + 512 * u;
            c[u] = *reinterpret_cast<const d2_t*>(v + min(k, Np - 2));
            if (k >= Np) c[u] = (d2_t){0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sv = __builtin_fma(c[u].x, c[u].x, sv);
            sv = __builtin_fma(c[u].y, c[u].y, sv);
        }
    }
    sv = wave_sum_f64(sv);
    if (lane == 0) red[wave] = sv;
    __syncthreads();
    if (tid == 0) {
        double x = kss - (((red[0] + red[1]) + red[2]) + red[3]);
        if (!(x > 1e-15)) x = 1e-15;
        var[i] = x + add;
    }
}
void launch_predv_finish(hipStream_t s, int rows, const double* V, int ld, int Np, double kss, double add, double* var) {
    hipLaunchKernelGGL(mfgp_predv_finish_f64, dim3(rows), dim3(256), 0, s, V, ld, Np, kss, add, var);
}

}  // namespace mfgp
