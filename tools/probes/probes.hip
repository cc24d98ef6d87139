// probes.hip -- hardware probes of the MI355X: a bare fp64 MFMA loop and the HBM copy / write-only / read-only streams (round 6: cut
// down to what tests/test_gpu_kernels.py::test_probe_peaks reads; the per-shape, per-wave-count, VALU, int8 / bf16, placement and
// CU-mask probes behind LABBOOK's "Hardware / runtime facts" are in the history at commit a031ded).  TEST / TOOL CODE ONLY: built into tools/probes/libmfgp_probes.so, never linked into
// libmfgp_hip.so and not declared in include/mfgp.h.  The numbers in DESIGN.md section 3 ("The fp64 matrix pipe on
// gfx950") come from here (tools/probes/probes.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace mfgp {
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- probes ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfgp_probe_mfma_f64(double* out, int iters) {
    d4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;  // keep the loop alive
}

__global__ __launch_bounds__(256) void mfgp_probe_copy(const d2_t* __restrict__ src, d2_t* __restrict__ dst,
                                                       int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// write-only and read-only HBM streams (the K build is a pure write stream, the skinny variance product a pure read)
__global__ __launch_bounds__(256) void mfgp_probe_write(d2_t* __restrict__ dst, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (d2_t){v, v + (double)i};
}
__global__ __launch_bounds__(256) void mfgp_probe_read(const d2_t* __restrict__ src, int64_t n, double* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i < n; i += stride) { const d2_t v = src[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
void run_probe_bw(hipStream_t s, double* out2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int64_t bytes = (int64_t)1 << 30;
    d2_t* buf = nullptr;
    double* dummy = nullptr;
    hipMalloc(&buf, bytes);
    hipMalloc(&dummy, 64);
    float ms = 0.f;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 6; ++rep) {
            if (rep == 1) hipEventRecord(e0, s);
            if (which == 0) hipLaunchKernelGGL(mfgp_probe_write, dim3(4096), dim3(256), 0, s, buf, bytes / 16, 1.0);
            else hipLaunchKernelGGL(mfgp_probe_read, dim3(4096), dim3(256), 0, s, buf, bytes / 16, dummy);
        }
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        out2[which] = 5.0 * (double)bytes / (ms * 1e-3) / 1e9;
    }
    hipFree(buf);
    hipFree(dummy);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

void run_probe(hipStream_t s, double* mfma_tflops, double* copy_gbs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double* dummy = nullptr;
    hipMalloc(&dummy, 64);
    // MFMA: 256 CUs x 4 blocks x 4 waves, 8 independent accumulators each
    const int iters = 20000, blocks = 1024;
    hipLaunchKernelGGL(mfgp_probe_mfma_f64, dim3(blocks), dim3(256), 0, s, dummy, 100);  // warm
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfgp_probe_mfma_f64, dim3(blocks), dim3(256), 0, s, dummy, iters);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
    *mfma_tflops = flops / (ms * 1e-3) / 1e12;
    // copy 1 GiB
    const int64_t bytes = (int64_t)1 << 30;
    d2_t *src = nullptr, *dst = nullptr;
    hipMalloc(&src, bytes);
    hipMalloc(&dst, bytes);
    hipMemsetAsync(src, 1, bytes, s);
    hipLaunchKernelGGL(mfgp_probe_copy, dim3(4096), dim3(256), 0, s, src, dst, bytes / 16);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL(mfgp_probe_copy, dim3(4096), dim3(256), 0, s, src, dst, bytes / 16);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    *copy_gbs = 5.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
    hipFree(src);
    hipFree(dst);
    hipFree(dummy);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

}  // namespace mfgp

extern "C" {
// out2 = {bare fp64 MFMA TFLOP/s, 1 GiB device copy GB/s}
int32_t mfgp_probe_basic(int32_t device, double* out2) {
    if (!out2 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe(s, out2, out2 + 1);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// out2 = {write-only GB/s, read-only GB/s} over 1 GiB (16 B per lane, grid-stride): the streams the K build and the triangular
// products of trimv_f64.hip are held against
int32_t mfgp_probe_bw(int32_t device, double* out2) {
    if (!out2 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe_bw(s, out2);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
