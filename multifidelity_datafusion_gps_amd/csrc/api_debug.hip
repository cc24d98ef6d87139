// api_debug.hip -- test hooks of the C-ABI: one tile-GEMM launch and one leaf launch on caller-supplied operands (tests/test_gpu_kernels.py).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "mfgp_internal.h"
#include "api_shared.h"

using namespace mfgp;

extern "C" {

// ---- test hooks ---------------------------------------------------------------------------------------
int32_t mfgp_dbg_gemm_nt(mfgp_handle* h, const double* A, const double* B, double* C, int32_t M, int32_t N,
                         int32_t K, double alpha, double beta, int32_t tile) {
    if (!h || !A || !B || !C) return fail(h, -1, "mfgp_dbg_gemm_nt: NULL");
    const bool chain = (tile == -64);   // -64: the serial-chain variant of the 64-tile kernel (mfgp_gemm_nt_f64_chain)
    if (chain) tile = 64;
    if ((tile != 128 && tile != 64 && tile != 32) || M % tile || N % tile || K % BK || K < BK)
        return fail(h, -1, "mfgp_dbg_gemm_nt: M, N must be multiples of the tile (128, 64, 32) and K of 32");
    HIPCHK(h, hipSetDevice(h->device));
    // one common leading dimension
    const int ld = std::max(K, N);
    double *dA, *dB, *dC;
    GemmTask* dt;
    HIPCHK(h, hipMalloc(&dA, (size_t)M * ld * 8));
    HIPCHK(h, hipMalloc(&dB, (size_t)N * ld * 8));
    HIPCHK(h, hipMalloc(&dC, (size_t)M * ld * 8));
    HIPCHK(h, hipMemcpy2D(dA, (size_t)ld * 8, A, (size_t)K * 8, (size_t)K * 8, M, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy2D(dB, (size_t)ld * 8, B, (size_t)K * 8, (size_t)K * 8, N, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy2D(dC, (size_t)ld * 8, C, (size_t)N * 8, (size_t)N * 8, M, hipMemcpyHostToDevice));
    std::vector<GemmTask> ts;
    for (int i = 0; i < M / tile; ++i)
        for (int j = 0; j < N / tile; ++j) {
            GemmTask t{};
            t.a_off = (int64_t)i * tile * ld;
            t.b_off = (int64_t)j * tile * ld;
            t.c_off = (int64_t)i * tile * ld + j * tile;
            t.c2_off = -1;
            t.klen = K;
            t.alpha = alpha; t.beta = beta;
            ts.push_back(t);
        }
    HIPCHK(h, hipMalloc(&dt, ts.size() * sizeof(GemmTask)));
    HIPCHK(h, hipMemcpy(dt, ts.data(), ts.size() * sizeof(GemmTask), hipMemcpyHostToDevice));
    if (launch_gemm(h->stream, tile, dt, (int)ts.size(), dA, dB, dC, nullptr, ld, chain ? 3 : 0) != 0)
        return fail(h, -1, "mfgp_dbg_gemm_nt: no kernel for this tile");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpy2D(C, (size_t)N * 8, dC, (size_t)ld * 8, (size_t)N * 8, M, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dt);
    return 0;
}

int32_t mfgp_dbg_leaf(mfgp_handle* h, const double* A, double* Lout, double* Xout, double* logdet_half) {
    if (!h || !A || !Lout || !Xout || !logdet_half) return fail(h, -1, "mfgp_dbg_leaf: NULL");
    HIPCHK(h, hipSetDevice(h->device));
    double *dA, *dL, *dS, *dl;
    int* di;
    const size_t bytes = (size_t)NB * NB * 8;
    HIPCHK(h, hipMalloc(&dA, bytes)); HIPCHK(h, hipMalloc(&dL, bytes)); HIPCHK(h, hipMalloc(&dS, bytes));
    HIPCHK(h, hipMalloc(&dl, 8)); HIPCHK(h, hipMalloc(&di, 4));
    HIPCHK(h, hipMemcpy(dA, A, bytes, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(di, 0, 4));
    launch_leaf(h->stream, dA, dL, dS, NB, 0, dl, di);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    int info = 0;
    HIPCHK(h, hipMemcpy(Lout, dL, bytes, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(Xout, dS, bytes, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(logdet_half, dl, 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&info, di, 4, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dL); hipFree(dS); hipFree(dl); hipFree(di);
    return info;
}

}  // extern "C"
