// probes.hip -- hardware probes of the MI355X (bare fp64 / int8 / bf16 MFMA loops, v_fma_f64 loops, HBM copy / write-only /
// read-only streams).  TEST / TOOL CODE ONLY: built into tools/probes/libmfgp_probes.so, never linked into
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

// detailed MFMA probe: per-wave shader cycles (s_memtime) and 100 MHz real time around the loop
template <int NACC>
__global__ __launch_bounds__(256) void mfgp_probe_mfma_detail(unsigned long long* out, int iters) {
    d4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0 + (s == 12345.678 ? 1 : 0);
        out[2 * w + 1] = r1 - r0;
    }
}

// VALU probe: NCH independent v_fma_f64 chains per lane; MIX: also issue MFMAs from the same wave
template <int NCH, bool MIX>
__global__ __launch_bounds__(256) void mfgp_probe_valu_f64(double* out, int iters) {
    double x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    const double a = 1.0 - 1e-9, b = 1e-9;
    d4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) x[i] = __builtin_fma(x[i], a, b);
        if (MIX) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

// GEMM-like operand pattern: every FMA reads THREE distinct 64-bit VGPR operands (acc += a[i] * b[j]), 8 x 4 tile
__global__ __launch_bounds__(256) void mfgp_probe_valu3_f64(double* out, int iters) {
    double acc[8][4], a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = 1.0 - 1e-9 * (threadIdx.x + j);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(a[i], b[j], acc[i][j]);
        // keep a and b in VGPRs and changing, so nothing folds to constants / SGPRs
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                          "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s == 12345.678) out[0] = s;
}

// out4: {VALU-only TFLOP/s (2 waves/SIMD), VALU-only (4 waves/SIMD), mixed total TFLOP/s (2 waves/SIMD: 32 fma + 4 mfma per iter), mixed 4 waves/SIMD}
void run_probe_valu(hipStream_t s, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double* dummy = nullptr;
    hipMalloc(&dummy, 64);
    const int iters = 20000;
    for (int c = 0; c < 4; ++c) {
        const int blocks = (c & 1) ? 1024 : 512;
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            if (c < 2) hipLaunchKernelGGL((mfgp_probe_valu_f64<32, false>), dim3(blocks), dim3(256), 0, s, dummy, iters);
            else hipLaunchKernelGGL((mfgp_probe_valu_f64<32, true>), dim3(blocks), dim3(256), 0, s, dummy, iters);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        const double waves = blocks * 4.0;
        double flops = waves * iters * 32.0 * 64.0 * 2.0;
        if (c >= 2) flops += waves * iters * 4.0 * 2048.0;
        out[c] = flops / (ms * 1e-3) / 1e12;
    }
    for (int c = 0; c < 2; ++c) {   // out[4], out[5]: the three-VGPR-operand pattern at 2 / 4 waves per SIMD
        const int blocks = c ? 1024 : 512;
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(mfgp_probe_valu3_f64, dim3(blocks), dim3(256), 0, s, dummy, iters);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        out[4 + c] = blocks * 4.0 * iters * 32.0 * 64.0 * 2.0 / (ms * 1e-3) / 1e12;
    }
    hipFree(dummy);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

// out[3*c + 0..2] = {TFLOP/s, shader cycles per MFMA per wave (median), shader clock GHz} for the configs
//   c=0: 1 wave/SIMD x 8 acc, c=1: 2 waves/SIMD x 8 acc, c=2: 4 waves/SIMD x 8 acc, c=3: 1 wave/SIMD x 1 acc (dependent)
// low-precision matrix pipes, for sizing an fp64 emulation (Ozaki splitting) against the fp64 MFMA ceiling:
// v_mfma_i32_16x16x64_i8 (32768 int8 ops each) and v_mfma_f32_16x16x32_bf16 (16384 flops each), 8 accumulators per wave
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfgp_probe_mfma_i8(int* out, int iters) {
    v4i_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4i_t){0, 0, 0, 0};
    const v4i_t a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 123456789) out[0] = sum;
}
__global__ __launch_bounds__(256) void mfgp_probe_mfma_bf16(float* out, int iters) {
    v4f_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    v8bf_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.001f * (threadIdx.x & 7)); b[i] = (__bf16)(1.0f - 0.001f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.678f) out[0] = sum;
}
void run_probe_lowp(hipStream_t s, double* out2) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int* dummy = nullptr;
    hipMalloc(&dummy, 64);
    const int iters = 20000, blocks = 2048;   // 2 workgroups of 4 waves per SIMD pair: 8 waves per CU x 8
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            if (which == 0) hipLaunchKernelGGL(mfgp_probe_mfma_i8, dim3(blocks), dim3(256), 0, s, dummy, iters);
            else hipLaunchKernelGGL(mfgp_probe_mfma_bf16, dim3(blocks), dim3(256), 0, s, reinterpret_cast<float*>(dummy), iters);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 4.0 * iters * 8.0 * (which == 0 ? 32768.0 : 16384.0);
        out2[which] = ops / (ms * 1e-3) / 1e12;   // Tera-ops (int8) / TFLOP (bf16) per second
    }
    hipFree(dummy);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

void run_probe_detail(hipStream_t s, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* dbuf = nullptr;
    const int maxw = 1024 * 4;
    hipMalloc(&dbuf, sizeof(unsigned long long) * 2 * maxw);
    std::vector<unsigned long long> hb(2 * maxw);
    const int iters = 4000;
    for (int c = 0; c < 4; ++c) {
        const int blocks = (c == 1) ? 512 : (c == 2) ? 1024 : 256;
        const int nacc = (c == 3) ? 1 : 8;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            if (nacc == 8) hipLaunchKernelGGL((mfgp_probe_mfma_detail<8>), dim3(blocks), dim3(256), 0, s, dbuf, iters);
            else hipLaunchKernelGGL((mfgp_probe_mfma_detail<1>), dim3(blocks), dim3(256), 0, s, dbuf, iters * 8);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * 4;
        hipMemcpy(hb.data(), dbuf, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
        std::vector<double> cyc(nw), clk(nw);
        const double nm = (double)iters * 8.0;
        for (int w = 0; w < nw; ++w) {
            cyc[w] = (double)hb[2 * w] / nm;
            clk[w] = (double)hb[2 * w] / ((double)hb[2 * w + 1] * 10.0) ;  // cycles per ns = GHz (realtime ticks are 10 ns)
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        out[3 * c + 0] = (double)nw * nm * 2048.0 / (ms * 1e-3) / 1e12;
        out[3 * c + 1] = cyc[nw / 2];
        out[3 * c + 2] = clk[nw / 2];
    }
    hipFree(dbuf);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}


// ---- fp64 MFMA shapes, data dependence and clock (round 3) ------------------------------------------------------------
// SHAPE 0: v_mfma_f64_16x16x4_f64 (2048 flops), SHAPE 1: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks, 512 flops).
// ZERO: all-zero operands (the clock the chip holds depends on operand toggling: MI355X_MICROARCH.md "DVFS give-back").
// Eight independent accumulators per wave, four rotating operand pairs; stamps: shader cycles (s_memtime) and 100 MHz
// real time (s_memrealtime) around the loop -> cycles per MFMA per wave, in-kernel clock.  Launches are sized to run
// >= 20 ms so that GRBM_GUI_ACTIVE / 8 / wall time of a rocprofv3 --pmc pass is a valid effective clock as well.
__device__ __forceinline__ double probe_operand(unsigned seed) {
    unsigned long long h = (seed + 1u) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    // sign 0, exponent of 1.0 (or 0.5), 52 random mantissa bits: magnitudes in [0.5, 2), every mantissa bit toggles
    const unsigned long long bits = ((h & 1ull) ? 0x3FF0000000000000ull : 0x3FE0000000000000ull) | (h >> 12);
    return __longlong_as_double((long long)bits) * ((h & 2ull) ? 1.0 : -1.0);
}
template <int SHAPE, bool ZERO>
__global__ __launch_bounds__(256) void mfgp_probe_fp64_shape(unsigned long long* out, int iters) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = ZERO ? 0.0 : probe_operand(threadIdx.x * 8 + i + blockIdx.x * 2048);
        b[i] = ZERO ? 0.0 : probe_operand(threadIdx.x * 8 + 4 + i + blockIdx.x * 2048);
    }
    d4_t acc[8];
    double acc1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0}; acc1[i] = 0.0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
            else acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[(i >> 1) & 3], acc1[i], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (SHAPE == 0) ? acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] : acc1[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0 + (s == 12345.678 ? 1 : 0);
        out[2 * w + 1] = r1 - r0;
    }
}

// out[4*c + 0..3] = {TFLOP/s, shader cycles per MFMA per wave (median), in-kernel clock GHz (median), launch ms}
//   c = 0: 16x16x4 random 1 wave/SIMD   1: 16x16x4 random 2 w/SIMD   2: 16x16x4 random 4 w/SIMD   3: 16x16x4 ZERO 4 w/SIMD
//   c = 4: 4x4x4_4b random 1 w/SIMD     5: 4x4x4_4b random 2 w/SIMD  6: 4x4x4_4b random 4 w/SIMD  7: 4x4x4_4b ZERO 4 w/SIMD
void run_probe_fp64_shapes(hipStream_t s, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* dbuf = nullptr;
    const int maxw = 1024 * 4;
    hipMalloc(&dbuf, sizeof(unsigned long long) * 2 * maxw);
    std::vector<unsigned long long> hb(2 * maxw);
    for (int c = 0; c < 8; ++c) {
        const int shape = c / 4, v = c % 4;
        const int blocks = (v == 0) ? 256 : (v == 1) ? 512 : 1024;
        const bool zero = (v == 3);
        // ~25-45 ms per launch: 16x16x4 ~100-140 cycles per MFMA, 4x4x4 unknown (16-64): iterations per wave chosen per shape
        const int iters = (shape == 0 ? 64000 : 256000) / (blocks / 256);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            if (shape == 0 && !zero) hipLaunchKernelGGL((mfgp_probe_fp64_shape<0, false>), dim3(blocks), dim3(256), 0, s, dbuf, iters);
            else if (shape == 0) hipLaunchKernelGGL((mfgp_probe_fp64_shape<0, true>), dim3(blocks), dim3(256), 0, s, dbuf, iters);
            else if (!zero) hipLaunchKernelGGL((mfgp_probe_fp64_shape<1, false>), dim3(blocks), dim3(256), 0, s, dbuf, iters);
            else hipLaunchKernelGGL((mfgp_probe_fp64_shape<1, true>), dim3(blocks), dim3(256), 0, s, dbuf, iters);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * 4;
        hipMemcpy(hb.data(), dbuf, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
        std::vector<double> cyc(nw), clk(nw);
        const double nm = (double)iters * 8.0;
        for (int w = 0; w < nw; ++w) {
            cyc[w] = (double)hb[2 * w] / nm;
            clk[w] = (double)hb[2 * w] / ((double)hb[2 * w + 1] * 10.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        out[4 * c + 0] = (double)nw * nm * (shape == 0 ? 2048.0 : 512.0) / (ms * 1e-3) / 1e12;
        out[4 * c + 1] = cyc[nw / 2];
        out[4 * c + 2] = clk[nw / 2];
        out[4 * c + 3] = ms;
    }
    hipFree(dbuf);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

// ---- operand / result layout of v_mfma_f64_4x4x4_4b_f64, and whether its A-broadcast controls (CBSZ / ABID) act on f64 ----
template <int CBSZ, int ABID>
__global__ __launch_bounds__(64) void mfgp_probe_mfma444_layout(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], CBSZ, ABID, 0);
}
void run_probe_444_layout(hipStream_t s, const double* a, const double* b, const double* c, double* out7x64) {
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 7 * 512);
    hipMemcpyAsync(da, a, 512, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(db, b, 512, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(dc, c, 512, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<0, 0>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 0 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<1, 0>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 1 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<1, 1>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 2 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<2, 0>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 3 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<2, 1>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 4 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<2, 2>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 5 * 64);
    hipLaunchKernelGGL((mfgp_probe_mfma444_layout<2, 3>), dim3(1), dim3(64), 0, s, da, db, dc, dd + 6 * 64);
    hipMemcpyAsync(out7x64, dd, 7 * 512, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
}

// ---- where does the dispatcher put the workgroups of a launch that does not fill the chip?  (round 3) ---------------------------
// G workgroups of 256 threads with `lds` bytes of LDS each record (XCC id, HW_ID) and stay resident for ~hold_us, so that the
// placement of the whole grid is seen at once.  Question: does a grid of 2 x 250 64-KB workgroups leave six CUs EMPTY (a kernel
// that needs a whole CU could start at once) or twelve CUs half full?
__global__ __launch_bounds__(256) void mfgp_probe_placement(unsigned* out, long long hold_ticks) {
    extern __shared__ double lds_dummy[];
    if (threadIdx.x == 0) {
        lds_dummy[0] = 1.0;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID (id 4), 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID (id 20), bits 3:0
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(16);
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


// ---- sustained matrix load (round 3): does the 71 TFLOP/s of the ~30 ms bare loops hold over SECONDS? --------------------
// The bench keeps the fp64 matrix pipe busy for 1.5 s at a time; a part that is power- or thermally managed lowers its clock
// under such a load, and then the ceiling a sustained job can reach is lower than the burst figure.  `launches` back-to-back
// launches of the 4x4x4_4b loop (1 wave per SIMD, random operands, ~25 ms each): per launch TFLOP/s (events) and the
// in-kernel clock (s_memtime / s_memrealtime, median over waves).  shape 0 = 16x16x4 for comparison.
void run_probe_sustained(hipStream_t s, int shape, int launches, double* out) {
    std::vector<hipEvent_t> ev(launches + 1);
    for (auto& e : ev) hipEventCreate(&e);
    unsigned long long* dbuf = nullptr;
    const int blocks = 256, nw = blocks * 4;
    hipMalloc(&dbuf, sizeof(unsigned long long) * 2 * nw * (size_t)launches);
    const int iters = shape == 0 ? 64000 : 256000;
    hipEventRecord(ev[0], s);
    for (int l = 0; l < launches; ++l) {
        unsigned long long* o = dbuf + (size_t)l * 2 * nw;
        if (shape == 0) hipLaunchKernelGGL((mfgp_probe_fp64_shape<0, false>), dim3(blocks), dim3(256), 0, s, o, iters);
        else hipLaunchKernelGGL((mfgp_probe_fp64_shape<1, false>), dim3(blocks), dim3(256), 0, s, o, iters);
        hipEventRecord(ev[l + 1], s);
    }
    hipEventSynchronize(ev[launches]);
    std::vector<unsigned long long> hb(2 * (size_t)nw * launches);
    hipMemcpy(hb.data(), dbuf, sizeof(unsigned long long) * hb.size(), hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8.0;
    for (int l = 0; l < launches; ++l) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev[l], ev[l + 1]);
        std::vector<double> clk(nw);
        for (int w = 0; w < nw; ++w)
            clk[w] = (double)hb[(size_t)l * 2 * nw + 2 * w] / ((double)hb[(size_t)l * 2 * nw + 2 * w + 1] * 10.0);
        std::sort(clk.begin(), clk.end());
        out[3 * l + 0] = ms;
        out[3 * l + 1] = (double)nw * nm * (shape == 0 ? 2048.0 : 512.0) / (ms * 1e-3) / 1e12;
        out[3 * l + 2] = clk[nw / 2];
    }
    hipFree(dbuf);
    for (auto& e : ev) hipEventDestroy(e);
}

}  // namespace mfgp

extern "C" {
// out[3 l + {0, 1, 2}] = {ms, TFLOP/s, in-kernel clock GHz} of launch l of `launches` back-to-back bare MFMA launches
int32_t mfgp_probe_sustained(int32_t device, int32_t shape, int32_t launches, double* out) {
    if (!out || launches < 1 || launches > 4096 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe_sustained(s, shape, launches, out);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
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
// a, b, c: one fp64 per lane (64 each); out: 7 x 64 results for (CBSZ, ABID) = (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (2,3)
int32_t mfgp_probe_mfma444_layout(int32_t device, const double* a, const double* b, const double* c, double* out7x64) {
    if (!a || !b || !c || !out7x64 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe_444_layout(s, a, b, c, out7x64);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// out[2 g] = HW_ID, out[2 g + 1] = XCC_ID of workgroup g of a grid of G workgroups with lds_bytes of LDS each
int32_t mfgp_probe_placement(int32_t device, int32_t G, int32_t lds_bytes, int32_t hold_us, uint32_t* out) {
    if (!out || hipSetDevice(device) != hipSuccess) return -1;
    unsigned* d = nullptr;
    if (hipMalloc(&d, (size_t)G * 8) != hipSuccess) return -2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(mfgp::mfgp_probe_placement), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(mfgp::mfgp_probe_placement, dim3(G), dim3(256), (size_t)lds_bytes, 0, d, (long long)hold_us * 100);
    hipDeviceSynchronize();
    hipMemcpy(out, d, (size_t)G * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// the same on a stream created with hipExtStreamCreateWithCUMask(nwords x 32 bits): which CUs does a mask leave to the queue?
int32_t mfgp_probe_placement_masked(int32_t device, int32_t G, int32_t lds_bytes, int32_t hold_us, int32_t nwords,
                                    const uint32_t* mask, uint32_t* out, double* ms_out) {
    if (!out || !mask || hipSetDevice(device) != hipSuccess) return -1;
    unsigned* d = nullptr;
    if (hipMalloc(&d, (size_t)G * 8) != hipSuccess) return -2;
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask) != hipSuccess) { hipFree(d); return -3; }
    hipFuncSetAttribute(reinterpret_cast<const void*>(mfgp::mfgp_probe_placement), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfgp::mfgp_probe_placement, dim3(G), dim3(256), (size_t)lds_bytes, s, d, (long long)hold_us * 100);   // warm
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfgp::mfgp_probe_placement, dim3(G), dim3(256), (size_t)lds_bytes, s, d, (long long)hold_us * 100);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms;
    hipMemcpy(out, d, (size_t)G * 8, hipMemcpyDeviceToHost);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipStreamDestroy(s);
    hipFree(d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// out32: see tools/probes/probes.py (fp64_shapes) for the layout
int32_t mfgp_probe_fp64_shapes(int32_t device, double* out32) {
    if (!out32 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe_fp64_shapes(s, out32);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// out24: see tools/probes/probes.py for the layout
int32_t mfgp_probe_detail(int32_t device, double* out24) {
    if (!out24 || hipSetDevice(device) != hipSuccess) return -1;
    hipStream_t s;
    if (hipStreamCreate(&s) != hipSuccess) return -2;
    mfgp::run_probe_detail(s, out24);
    mfgp::run_probe_valu(s, out24 + 12);
    mfgp::run_probe_lowp(s, out24 + 18);
    mfgp::run_probe_bw(s, out24 + 20);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
