// vecops.hip -- the bandwidth-bound vector pieces: triangular mat-vecs with the explicit inverse
// factor (alpha = L^-T (L^-1 y), replacing LAPACK dpotrs behind GPy's exact inference, SURVEY 8(a)
// a5), row sums of squares for the predictive variance (a11), the log-det / quadratic-form finish
// (a6) and the peak probes used by bench.py.
#include <algorithm>
#include <vector>
#include "mfgp_internal.h"

namespace mfgp {

__global__ __launch_bounds__(256) void mfgp_rowsumsq_f64(const double* __restrict__ M, int ld,
                                                         double* __restrict__ out, int nrows, int ncols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double* mp = M + (int64_t)row * ld;
    double s = 0.0;
    for (int k = 2 * lane; k < ncols; k += 128) {
        const d2_t m = *reinterpret_cast<const d2_t*>(mp + k);
        s += m.x * m.x + m.y * m.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void mfgp_finish_solve_f64(const double* __restrict__ z, int Np,
                                                             const double* __restrict__ logdet_part, int nblk,
                                                             double* __restrict__ scalars) {
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
        double ld = 0.0;
        for (int b = 0; b < nblk; ++b) ld += logdet_part[b];
        scalars[1] = 2.0 * ld;
    }
}

// rank-1 append: given l = X k (first n entries) and w = X^T l, finish the new row r = n of L and of
// the mirrored inverse S, the new z entry, alpha and the scalars.
//   d = sqrt(kdiag - l.l) ; L[r][0:n] = l, L[r][r] = d ; X[r][0:n] = -w/d, X[r][r] = 1/d (both triangles of S)
//   z[r] = (y_new - l.z) / d ; out = {d, z_r, l.l, status(0 ok / 1 not PD)}
//   alpha = X^T z with the new row [-w^T/d, 1/d] of X:  alpha[0:n] += X[r][0:n] z_r ; alpha[r] = z_r / d   (O(n): no third pass
//   over the triangle)
// Every workgroup forms the two dot products itself (n <= Np doubles out of L2, the same fixed order everywhere, so every
// workgroup holds the same d and z_r) and then writes its own slice of the new row / column -- the column S[i][r] is n scattered
// 8-byte stores, which one workgroup alone would issue for ~15 us.  The new training row itself (X[r] <- xs_new, Y[r] <- y_new)
// is committed HERE, by workgroup 0, and only on success: a rejected append leaves the handle's data exactly as it was (padding
// row r of X and Y stays zero).
__global__ __launch_bounds__(256) void mfgp_append_finish_f64(double* __restrict__ L, double* __restrict__ S, int ld,
                                                              int n, const double* __restrict__ l,
                                                              const double* __restrict__ w, double* __restrict__ z,
                                                              double* __restrict__ alpha, double kdiag, double y_new,
                                                              double* __restrict__ out, double* __restrict__ X,
                                                              const double* __restrict__ xs_new, int D, double* __restrict__ Y) {
    __shared__ double red[512];
    const int tid = threadIdx.x;
    // eight elements per thread in flight (clamped index + select: no branch around a load): one element per iteration is a chain of
    // 32 dependent L2 round trips at n = 8192 -- 16 us of this launch's 16.4 (profiles/r06_pmc.json, before)
    double ss = 0.0, lz = 0.0;
    for (int i0 = tid; i0 < n; i0 += 256 * 8) {
        double lv[8], zv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u, ic = min(i, n - 1);
            lv[u] = l[ic]; zv[u] = z[ic];
            if (i >= n) { lv[u] = 0.0; zv[u] = 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ss = __builtin_fma(lv[u], lv[u], ss);
            lz = __builtin_fma(lv[u], zv[u], lz);
        }
    }
    red[tid] = ss;
    red[256 + tid] = lz;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            red[tid] += red[tid + off];
            red[256 + tid] += red[256 + tid + off];
        }
        __syncthreads();
    }
    const double d2 = kdiag - red[0];
    if (!(d2 > 0.0)) {
        if (blockIdx.x == 0 && tid == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = red[0]; out[3] = 1.0; }
        return;
    }
    const double d = sqrt(d2), rd = 1.0 / d;
    const double zr = (y_new - red[256]) * rd;
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const double xv = -w[i] * rd;
        L[(int64_t)n * ld + i] = l[i];
        S[(int64_t)n * ld + i] = xv;
        S[(int64_t)i * ld + n] = xv;
        alpha[i] = __builtin_fma(xv, zr, alpha[i]);
    }
    if (blockIdx.x == 0) {
        if (tid < D) X[(int64_t)n * D + tid] = xs_new[tid];
        if (tid == 0) {
            Y[n] = y_new;
            L[(int64_t)n * ld + n] = d;
            S[(int64_t)n * ld + n] = rd;
            z[n] = zr;
            alpha[n] = zr * rd;
            out[0] = d; out[1] = zr; out[2] = red[0]; out[3] = 0.0;
        }
    }
}

void launch_append_finish(hipStream_t s, double* L, double* S, int ld, int n, const double* l, const double* w, double* z,
                          double* alpha, double kdiag, double y_new, double* out, double* X, const double* xs_new, int D, double* Y) {
    const int nwg = n >= 4096 ? 128 : (n >= 2048 ? 32 : (n >= 512 ? 8 : 1));
    hipLaunchKernelGGL(mfgp_append_finish_f64, dim3(nwg), dim3(256), 0, s, L, S, ld, n, l, w, z, alpha, kdiag, y_new, out, X,
                       xs_new, D, Y);
}

void launch_rowsumsq(hipStream_t s, const double* M, int ld, double* out, int nrows, int ncols) {
    hipLaunchKernelGGL(mfgp_rowsumsq_f64, dim3((nrows + 3) / 4), dim3(256), 0, s, M, ld, out, nrows, ncols);
}
void launch_finish_solve(hipStream_t s, const double* z, int Np, const double* logdet_part, int nblk,
                         double* scalars) {
    hipLaunchKernelGGL(mfgp_finish_solve_f64, dim3(1), dim3(256), 0, s, z, Np, logdet_part, nblk, scalars);
}

// lower part of the mirrored inverse S from its upper part: S[c][r] = S[r][c] for r < c, by 64x64 tiles through LDS (a sharded
// evaluation receives the ROWS of X^T -- the upper part -- from their owners and rebuilds X, the lower part, here).  Grid:
// the lower-triangle tiles (bi >= bj) as in the K build; tile (bi, bj) is written from tile (bj, bi).
__global__ __launch_bounds__(256) void mfgp_mirror_lower_f64(double* __restrict__ S, int ld) {
    __shared__ double t[64][65];
    const int b = blockIdx.x;
    int bi = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
    while (bi * (bi + 1) / 2 > b) --bi;
    const int bj = b - bi * (bi + 1) / 2;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) t[r][tx] = S[(int64_t)(bj * 64 + r) * ld + bi * 64 + tx];   // upper tile (bj, bi), row-wise
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        if (bi == bj && tx >= r) continue;                                                          // diagonal tile: strictly lower only
        S[(int64_t)(bi * 64 + r) * ld + bj * 64 + tx] = t[tx][r];
    }
}
// The exchange of a sharded evaluation moves only what the other ranks lack: for every 128-row block b of X^T its columns from
// the diagonal on (the upper part of S), packed densely -- block b is 128 x (Np - 128 b) doubles at offset off[b] of its owner's
// chunk -- so that ONE in-place ncclAllGather of equal chunks carries every rank's rows (the serpentine ownership makes the
// chunks equal to within a block; they are padded to the largest).  pack: this rank's blocks S -> stage; unpack: the others'
// blocks stage -> S.  One workgroup per (row of a block, block).  lower != 0: the columns UP TO the diagonal block instead
// (block b is 128 x 128 (b + 1) doubles) -- the row blocks of Ky that mfgp_allgather_rows exchanges: the factorisation reads
// only the lower triangle.
__global__ __launch_bounds__(256) void mfgp_shard_rows_copy_f64(double* __restrict__ S, int ld, double* __restrict__ stage,
                                                                const long long* __restrict__ off, long long chunk, int rank,
                                                                int size, int unpack, int lower) {
    const int b = blockIdx.y, r = blockIdx.x;
    const int x = b % (2 * size), own = x < size ? x : 2 * size - 1 - x;        // plan.h shard_owner
    if ((own == rank) == (unpack != 0)) return;
    const int w = lower ? (b + 1) * 128 : ld - b * 128;                            // (ld = Np)
    double* rowS = S + (long long)(b * 128 + r) * ld + (lower ? 0 : b * 128);
    double* rowP = stage + (long long)own * chunk + off[b] + (long long)r * w;
    const d2_t* src = reinterpret_cast<const d2_t*>(unpack ? rowP : rowS);
    d2_t* dst = reinterpret_cast<d2_t*>(unpack ? rowS : rowP);
    for (int k = threadIdx.x; k < w / 2; k += 256) dst[k] = src[k];
}
void launch_shard_rows_copy(hipStream_t s, double* S, int ld, int nblk, double* stage, const long long* off, long long chunk,
                            int rank, int size, bool unpack, bool lower) {
    hipLaunchKernelGGL(mfgp_shard_rows_copy_f64, dim3(128, nblk), dim3(256), 0, s, S, ld, stage, off, chunk, rank, size,
                       unpack ? 1 : 0, lower ? 1 : 0);
}

// ---- exchange steps of a distributed Cholesky (plan.h Shard::dist) ------------------------------------------------------
// COMM_DIAG(c): the diagonal blocks L_cc, X_cc (mirrored) and the leaf's two words (half log-det of the block, pivot status) in one
// contiguous message of 2 * 128^2 + 2 doubles: packed by the owner, unpacked by everybody else.  One workgroup per block row.
__global__ __launch_bounds__(128) void mfgp_dist_diag_copy_f64(double* __restrict__ L, double* __restrict__ S, int ld, int c,
                                                               double* __restrict__ stage, double* __restrict__ logdet,
                                                               int* __restrict__ info, int unpack, int* __restrict__ flag, int epoch) {
    const int r = blockIdx.x, k = threadIdx.x;
    double* pl = L + (long long)(c * 128 + r) * ld + c * 128 + k;
    double* ps = S + (long long)(c * 128 + r) * ld + c * 128 + k;
    if (unpack) {
        *pl = stage[r * 128 + k];
        *ps = stage[128 * 128 + r * 128 + k];
        if (r == 0 && k == 0) {
            logdet[c] = stage[2 * 128 * 128];
            const int st = (int)stage[2 * 128 * 128 + 1];
            if (st != 0 && *info == 0) *info = st;
            if (st != 0 && flag) *flag = epoch;          // the owner's leaf failed: this rank's later launches return at once too
        }
    } else {
        stage[r * 128 + k] = *pl;
        stage[128 * 128 + r * 128 + k] = *ps;
        if (r == 0 && k == 0) {
            stage[2 * 128 * 128] = logdet[c];
            stage[2 * 128 * 128 + 1] = (double)*info;
        }
    }
}
void launch_dist_diag_copy(hipStream_t s, double* L, double* S, int ld, int c, double* stage, double* logdet, int* info, bool unpack,
                           int* flag, int epoch) {
    hipLaunchKernelGGL(mfgp_dist_diag_copy_f64, dim3(128), dim3(128), 0, s, L, S, ld, c, stage, logdet, info, unpack ? 1 : 0, flag,
                       epoch);
}
// COMM_PANEL(c): block column c of L below the diagonal.  Block (i, c), i > c, belongs to shard_owner(i); it is the k-th block of its
// owner in this column (k = the owner's blocks in (c, i)) and travels at [owner * chunk + k * 128^2) of the all-gather's buffer.
// pack: this rank's blocks L -> stage; unpack: the others' blocks stage -> L.  One workgroup per (row of a block, block row).
__global__ __launch_bounds__(128) void mfgp_dist_panel_copy_f64(double* __restrict__ L, int ld, int c, double* __restrict__ stage,
                                                                long long chunk, int rank, int size, int unpack) {
    const int i = c + 1 + blockIdx.y, r = blockIdx.x, t = threadIdx.x;
    const int x = i % (2 * size), own = x < size ? x : 2 * size - 1 - x;        // plan.h shard_owner
    if ((own == rank) == (unpack != 0)) return;
    int k = 0;
    for (int j = c + 1; j < i; ++j) {
        const int xj = j % (2 * size);
        k += ((xj < size ? xj : 2 * size - 1 - xj) == own);
    }
    double* pl = L + (long long)(i * 128 + r) * ld + c * 128 + t;
    double* pp = stage + (long long)own * chunk + (long long)k * (128 * 128) + r * 128 + t;
    if (unpack) *pl = *pp; else *pp = *pl;
}
void launch_dist_panel_copy(hipStream_t s, double* L, int ld, int nblk, int c, double* stage, long long chunk, int rank, int size,
                            bool unpack) {
    if (nblk - 1 - c <= 0) return;
    hipLaunchKernelGGL(mfgp_dist_panel_copy_f64, dim3(128, nblk - 1 - c), dim3(128), 0, s, L, ld, c, stage, chunk, rank, size,
                       unpack ? 1 : 0);
}

void launch_mirror_lower(hipStream_t s, double* S, int ld, int Np) {
    const int nt = Np / 64;
    hipLaunchKernelGGL(mfgp_mirror_lower_f64, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, S, ld);
}

// ---- device-resident level chaining -------------------------------------------------
// stencil rows t0 .. t0+n of the (rows*c, d) stack  T[i*c + j] = Xc[i] + offs[j]  (src/MFDataFusion.py:190-197:
// the low-fidelity level is evaluated at x + i*tau for every stencil offset i); rows n .. n_p are zero padding.
__global__ __launch_bounds__(256) void mfgp_stencil_rows_f64(const double* __restrict__ Xc, const double* __restrict__ offs,
                                                             int d, int c, int64_t t0, int n, int n_p,
                                                             double* __restrict__ T) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n_p * d) return;
    const int r = (int)(e / d), k = (int)(e % d);
    double v = 0.0;
    if (r < n) {
        const int64_t t = t0 + r;
        v = Xc[(t / c) * d + k] + offs[(t % c) * d + k];
    }
    T[e] = v;
}
// out[i] = [ Xc[i, 0:d] | m[i*c .. i*c + c) ] for i < rows, zeros for rows <= i < rows_p   (ld >= d + c)
__global__ __launch_bounds__(256) void mfgp_assemble_aug_f64(const double* __restrict__ Xc, const double* __restrict__ m,
                                                             int rows, int rows_p, int d, int c,
                                                             double* __restrict__ out, int ld) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int w = d + c;
    if (e >= (int64_t)rows_p * w) return;
    const int i = (int)(e / w), k = (int)(e % w);
    double v = 0.0;
    if (i < rows) v = (k < d) ? Xc[(int64_t)i * d + k] : m[(int64_t)i * c + (k - d)];
    out[(int64_t)i * ld + k] = v;
}
void launch_stencil_rows(hipStream_t s, const double* Xc, const double* offs, int d, int c, int64_t t0, int n, int n_p,
                         double* T) {
    const int64_t tot = (int64_t)n_p * d;
    hipLaunchKernelGGL(mfgp_stencil_rows_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xc, offs, d, c, t0, n,
                       n_p, T);
}
void launch_assemble_aug(hipStream_t s, const double* Xc, const double* m, int rows, int rows_p, int d, int c,
                         double* out, int ld) {
    const int64_t tot = (int64_t)rows_p * (d + c);
    hipLaunchKernelGGL(mfgp_assemble_aug_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xc, m, rows, rows_p, d,
                       c, out, ld);
}

}  // namespace mfgp
