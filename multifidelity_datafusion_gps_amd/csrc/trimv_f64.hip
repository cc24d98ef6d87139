// trimv_f64.hip -- the bandwidth-bound triangular (multi-)vector products with the stored inverse factor.
//
//     V[i][j] = sum_{k in range(j)} W[i][k] * M[j][k]        i < R right-hand sides, j < nrows
//     range(j):  mode 0: k <= j (lower part)   mode 1: j <= k < ncols (upper part)   mode 2: k < ncols
//
// One kernel body serves every O(N^2) pass of the path: z = X y and alpha = X^T z behind GPy's dpotrs (a5), the
// predictive mean K(X*,X) alpha and -- R <= 16 test rows, the N* = 1 callback of the reference's DIRECT maximiser
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

static inline int trimv_blocks(int nrows, int JR) {
    const int G = (nrows + JR - 1) / JR;
    return ((G + 1) / 2 + 3) / 4;
}

// rows per wave / chunks per batch of the single-vector form: 4 rows x 2 chunks (measured at N = 8192, variance stage incl. its
// finishing launch / append: "4 2" 0.0498 / 0.132 ms, "1 8" 0.0508 / 0.139, "2 4" 0.0555 / 0.143: profiles/r06_trimv_shapes.txt).
// MFGP_TRIMV="JR U" selects "2 4" or "1 8" instead (lab switch, read once)
static void trimv1_shape(int& JR, int& U) {
    static int jr = 0, u = 0;
    if (!jr) {
        jr = 4; u = 2;
        if (const char* e = getenv("MFGP_TRIMV")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d %d", &a, &b) == 2) { jr = a; u = b; }
        }
    }
    JR = jr; U = u;
}

void launch_rowdot(hipStream_t s, const double* M, int ld, const double* x, double* y, int nrows, int ncols, int mode, int nbatch,
                   long long mstride, long long xstride, long long ystride) {
    if (nrows <= 0) return;
    int JR, U;
    trimv1_shape(JR, U);
    const dim3 blk(256);
    const int nb = nbatch > 0 ? nbatch : 1;
#define TRIMV1(jr, u)                                                                                                     \
    hipLaunchKernelGGL((mfgp_trimv_f64<1, jr, u>), dim3(trimv_blocks(nrows, jr), nb), blk, 0, s, M, ld, x, 0, y, 0, nrows, ncols, \
                       mode, mstride, xstride, ystride)
    if (JR == 1 && U == 8) TRIMV1(1, 8);
    else if (JR == 2 && U == 4) TRIMV1(2, 4);
    else TRIMV1(4, 2);
#undef TRIMV1
}

void launch_alpha_finish(hipStream_t s, const double* S, int ld, const double* z, double* alpha, int Np, const double* logdet_part,
                         int nblk, double* scalars, int nbatch, long long sstride, long long vstride, int ldstride, int scstride) {
    hipLaunchKernelGGL((mfgp_alpha_finish_f64<4, 2>), dim3(trimv_blocks(Np, 4) + 1, nbatch > 0 ? nbatch : 1), dim3(256), 0, s, S, ld,
                       z, alpha, Np, logdet_part, nblk, scalars, sstride, vstride, ldstride, scstride);
}

// Predict with <= 16 test rows: V[i][j] = sum_{k <= j} W[i][k] X[j][k], i < R (R in {1, 2, 4, 8, 16}: the caller rounds its row
// count up; the panel W holds at least that many rows), j < Np, X = L^-1 read from the lower part of the mirrored S -- and, in
// the launch's extra last workgroup, the means W[i] . alpha of the `rows` real test rows (the single-vector body over the panel:
// a row's sum is formed in the same order as in launch_rowdot's launches, whatever their shape -- the mean of a test row does
// not depend on the size of the batch it travels in).
template <int R, int JR, int U>
__global__ __launch_bounds__(256) void mfgp_predv_rows_f64(const double* __restrict__ S, int ld, const double* __restrict__ W,
                                                           double* __restrict__ V, int Np, const double* __restrict__ alpha,
                                                           double* __restrict__ mean, int rows) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == gridDim.x - 1) {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        trimv_wave<1, 2, 4>(W, ld, alpha, 0, mean, 0, rows, Np, 2, wv, lane);   // rows <= 16: at most 8 groups = 4 waves
        return;
    }
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    trimv_wave<R, JR, U>(S, ld, W, ld, V, ld, Np, Np, 0, wv, lane);
}

// The same product for 8 and 16 test rows: the R right-hand sides would cost R / JR times the bytes of S in L2 -> L1 reads and
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
    if (blockIdx.x == gridDim.x - 1) {
        if (wave < 4) trimv_wave<1, 2, 4>(W, ld, alpha, 0, mean, 0, rows, Np, 2, wave, lane);
        return;
    }
    const int NB = Np / BR;
    const int bA = blockIdx.x, bB = NB - 1 - bA;                       // (NB is even: Np is a multiple of 128, 2 BR divides 128)
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

void launch_predv_rows(hipStream_t s, int R, const double* W, const double* S, double* V, int ld, int Np, const double* alpha,
                       double* mean, int rows) {
    const dim3 blk(256);
    int JR1, U1;
    trimv1_shape(JR1, U1);
#define TRIMVR(r, jr, u)                                                                                                  \
    hipLaunchKernelGGL((mfgp_predv_rows_f64<r, jr, u>), dim3(trimv_blocks(Np, jr) + 1), blk, 0, s, S, ld, W, V, Np, alpha, mean, rows)
    if (R <= 1) {
        if (JR1 == 1 && U1 == 8) TRIMVR(1, 1, 8);
        else if (JR1 == 2 && U1 == 4) TRIMVR(1, 2, 4);
        else TRIMVR(1, 4, 2);
    } else if (R <= 2) TRIMVR(2, 4, 2);
    else if (R <= 4) TRIMVR(4, 4, 2);
    else {
        static const int lds = getenv("MFGP_TRIMV_LDS") ? atoi(getenv("MFGP_TRIMV_LDS")) : 1;
#define PREDV_LDS(r, jr, nw, d)                                                                                           \
    hipLaunchKernelGGL((mfgp_predv_rows_lds_f64<r, jr, nw, d>), dim3(Np / (nw * jr) / 2 + 1), dim3(64 * nw), 0, s, S, ld, W, V, Np, alpha, \
                       mean, rows)
        if (lds == 0) { if (R <= 8) TRIMVR(8, 4, 2); else TRIMVR(16, 4, 1); }
        else if (R <= 8) PREDV_LDS(8, 4, 4, 2);
        else PREDV_LDS(16, 2, 4, 3);
#undef PREDV_LDS
    }
#undef TRIMVR
}

// its finish, one workgroup per test row i:  var[i] = max(kss - sum_j V[i][j]^2, 1e-15) + add      (fixed-order sum)
__global__ __launch_bounds__(256) void mfgp_predv_finish_f64(const double* __restrict__ V, int ld, int Np, double kss, double add,
                                                             double* __restrict__ var) {
    __shared__ double red[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* v = V + (int64_t)i * ld;
    double sv = 0.0;
    for (int k = 2 * tid; k < Np; k += 512) {
        const d2_t c = *reinterpret_cast<const d2_t*>(v + k);
        sv = __builtin_fma(c.x, c.x, sv);
        sv = __builtin_fma(c.y, c.y, sv);
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
