// gemm_lab -- experimental variants of the fp64 tile GEMM on v_mfma_f64_4x4x4_4b_f64, measured against the product
// kernel on the same task list.  TOOL CODE: built into tools/gemm_lab/libgemm_lab.so, never linked into libmfgp_hip.so.
// A variant that wins here is moved into csrc/gemm_f64.hip; numbers go to profiles/.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "../../multifidelity_datafusion_gps_amd/csrc/plan.h"

namespace lab {
using mfgp::GemmTask;
using mfgp::TF_A_LOWER; using mfgp::TF_A_UPPER; using mfgp::TF_B_LOWER; using mfgp::TF_B_UPPER;
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// MODE 0: the real kernel.  MODE 1: global loads / LDS stores only for the first K-step (LDS + MFMA bound).
// MODE 2: additionally no fragment reads inside the loop (MFMA issue bound).
template <int BM, int BN, int WM, int WN, int MODE>
__device__ __forceinline__ void gemm444_pipe(const GemmTask t, const double* A, const double* B, double* C, double* C2, int ld) {
    constexpr int KT = 32;
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int CPR = KT / 2, NA = BM * CPR / NT, NBC = BN * CPR / NT, SWM = CPR - 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;
    double* Bs = smem + 2 * BM * KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, q = lane >> 4, cb = (lane >> 2) & 3;
    const double* Ap = A + t.a_off;
    const double* Bp = B + t.b_off;
    const int nk = t.klen / KT;
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER;
    const bool b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const bool any_mask = (t.flags & 15) != 0;
    const int a_lo_shift = t.klen - BM, b_lo_shift = t.klen - BN;

    d2_t ra[NA], rb[NBC];
    double acc[TM][TN][4];
    double* const Cp = C + t.c_off;
    const bool preload = (t.beta != 0.0);
    const double c_scale = preload ? t.beta / t.alpha : 0.0;
    if (preload) {          // one branch around ALL the loads: they are issued together and waited for progressively
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                    const int col = wn * (BN / WN) + ni * 16 + fr;
                    acc[mi][ni][r] = c_scale * Cp[(int64_t)row * ld + col];
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
    }
    // one 32-bit lane offset for all chunks of a K-step; the chunk's row block and the K-step are uniform (scalar base)
    const unsigned lane_goff = (unsigned)((tid / CPR) * ld + 2 * (tid % CPR));
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const double* base = Ap + (int64_t)(u * (NT / CPR)) * ld + kt * KT;
            ra[u] = *reinterpret_cast<const d2_t*>(base + lane_goff);
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const double* base = Bp + (int64_t)(u * (NT / CPR)) * ld + kt * KT;
            rb[u] = *reinterpret_cast<const d2_t*>(base + lane_goff);
        }
    };
    auto mask_tiles = [&](int kt) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, k = kt * KT + 2 * (g % CPR);
            d2_t v = ra[u];
            if (a_lo) { if (k > row + a_lo_shift) v.x = 0.0; if (k + 1 > row + a_lo_shift) v.y = 0.0; }
            if (a_up) { if (k < row) v.x = 0.0; if (k + 1 < row) v.y = 0.0; }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, k = kt * KT + 2 * (g % CPR);
            d2_t v = rb[u];
            if (b_lo) { if (k > row + b_lo_shift) v.x = 0.0; if (k + 1 > row + b_lo_shift) v.y = 0.0; }
            if (b_up) { if (k < row) v.x = 0.0; if (k + 1 < row) v.y = 0.0; }
            rb[u] = v;
        }
    };
    auto store_tiles = [&](int buf, int kt) {
        if (any_mask) mask_tiles(kt);
        double* as = As + buf * (BM * KT);
        double* bs = Bs + buf * (BN * KT);
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            *reinterpret_cast<d2_t*>(as + row * KT + ((c ^ (row & SWM)) << 1)) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            *reinterpret_cast<d2_t*>(bs + row * KT + ((c ^ (row & SWM)) << 1)) = rb[u];
        }
    };
    // Fragment addresses.  chunk(g) = (4 g + q) ^ (row & 15) = chunk(0) ^ 4 g, so the byte offset of group g is the offset of
    // group 0 with bits 6-7 flipped by g: FIVE address registers per lane (four rotations of A, one for B); the row block
    // mi / ni, the LDS buffer and the operand are immediate offsets (all multiples of 256 B, the bank row).
    const char* const smem_b = reinterpret_cast<const char*>(smem);
    int a_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (fr + 4 * r) & 15;
        a_off[r] = (wm * (BM / WM) + row) * (KT * 8) + ((q ^ row) << 4);
    }
    const int b_off = (wn * (BN / WN) + fr) * (KT * 8) + ((q ^ fr) << 4) + 2 * BM * KT * 8;
    auto read_frags = [&](auto BUFC, auto GC, d2_t (&a)[TM][4], d2_t (&b)[TN]) {
        constexpr int buf = decltype(BUFC)::value, g = decltype(GC)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const char* pa = smem_b + (a_off[r] ^ (g << 6));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[mi][r] = *reinterpret_cast<const d2_t*>(pa + buf * (BM * KT * 8) + mi * (16 * KT * 8));
        }
        const char* pb = smem_b + (b_off ^ (g << 6));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            b[ni] = *reinterpret_cast<const d2_t*>(pb + buf * (BN * KT * 8) + ni * (16 * KT * 8));
    };
    auto mma = [&](d2_t (&a)[TM][4], d2_t (&b)[TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mi][ni][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi][r][h], b[ni][h], acc[mi][ni][r], 0, 0, 0);
    };

    d2_t a0[TM][4], b0[TN], a1[TM][4], b1[TN];
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    if (nk > 1 && MODE == 0) load_tiles(1);
    read_frags(I0{}, I0{}, a0, b0);
    // One K-step = four groups of 8 columns.  The fragments of group G + 1 are requested before the MFMAs of group G are
    // issued (two register sets); the next K-step's tile goes from the staging registers to the other LDS buffer during the
    // third group, ONE barrier per K-step follows it, and the fourth group already prefetches from the new buffer.
    auto kstep = [&](int kt, auto BUFC) {
        constexpr int buf = (MODE == 0) ? decltype(BUFC)::value : 0;
        using IB = std::integral_constant<int, buf>;
        using IN = std::integral_constant<int, (MODE == 0) ? (buf ^ 1) : 0>;
        const bool more = kt + 1 < nk;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE < 2) read_frags(IB{}, I1{}, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE < 2) read_frags(IB{}, I2{}, a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE < 2) read_frags(IB{}, I3{}, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
            if (more) store_tiles(buf ^ 1, kt + 1);
            __syncthreads();
            if (kt + 2 < nk) load_tiles(kt + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more && MODE < 2) read_frags(IN{}, I0{}, a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, I0{});
        if (kt + 1 < nk) kstep(kt + 1, I1{});
    }
    const double alpha = t.alpha;
    const bool mirror = (t.c2_off >= 0);
    double* C2p = C2 + (mirror ? t.c2_off : 0);
    // the output addresses are recomputed from laundered lane indices: shared with the pre-load above they would stay live
    // across the whole K loop (64-bit address per output element: 64-128 VGPRs)
    int q2 = q, fr2 = fr, cb2 = cb;
    asm volatile("" : "+v"(q2), "+v"(fr2), "+v"(cb2));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb2 + r) & 3) + q2;
                const int col = wn * (BN / WN) + ni * 16 + fr2;
                const double v = alpha * acc[mi][ni][r];
                Cp[(int64_t)row * ld + col] = v;
                if (mirror) C2p[(int64_t)col * ld + row] = v;
            }
}


// ---- glds variant: the tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass);
// the chunk swizzle sits on the per-lane SOURCE address (the LDS image of one wave-instruction is lane-linear: 4 rows x 256 B);
// triangular masks are a fix-up of the landed chunks (zeros written by the lane that fetched them) before the barrier.
template <int BM, int BN, int WM, int WN, int LDS_SHIFT = 0>
__device__ __forceinline__ void gemm444_glds(const GemmTask t, const double* A, const double* B, double* C, double* C2, int ld) {
    constexpr int KT = 32;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int NA = BM / (4 * NW), NBC = BN / (4 * NW);      // wave-instructions (4 rows each) per wave and K-step
    static_assert(NA >= 1 && NBC >= 1, "tile too small for the wave count");
    extern __shared__ __attribute__((aligned(1024))) double smem[];
    char* const smem_b = reinterpret_cast<char*>(smem) + LDS_SHIFT;      // LDS_SHIFT: probe of the DMA's reach (M0 width)
    constexpr int A_BYTES = BM * KT * 8, B_BYTES = BN * KT * 8;
    constexpr int B_BASE = 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, q = lane >> 4, cb = (lane >> 2) & 3;
    const double* Ap = A + t.a_off;
    const double* Bp = B + t.b_off;
    const int nk = t.klen / KT;
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER;
    const bool b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const bool any_mask = (t.flags & 15) != 0;
    const int a_lo_shift = t.klen - BM, b_lo_shift = t.klen - BN;

    double acc[TM][TN][4];
    double* const Cp = C + t.c_off;
    const bool preload = (t.beta != 0.0);
    const double c_scale = preload ? t.beta / t.alpha : 0.0;
    if (preload) {          // one branch around ALL the loads: they are issued together and waited for progressively
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                    const int col = wn * (BN / WN) + ni * 16 + fr;
                    acc[mi][ni][r] = c_scale * Cp[(int64_t)row * ld + col];
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
    }
    // DMA mapping: wave-instruction u of this wave covers tile rows r0 = 4 (wave + NW u) .. r0 + 3; lane l lands at LDS byte
    // r0 * 256 + 16 l = (row = r0 + (l >> 4), slot = l & 15) and therefore fetches chunk slot ^ (row & 15) of that row.
    // (r0 & 15) = 4 (wave & 3) for every u (NW a multiple of 4), so one lane offset serves all instructions.
    static_assert(NW % 4 == 0, "wave count must keep (r0 & 15) constant per wave");
    const int drow = 4 * (wave & 3) + q;                           // (row & 15) of this lane's DMA rows
    const unsigned dma_goff = (unsigned)(q * ld + 2 * (fr ^ drow));  // element offset from the instruction's base pointer
    auto dma_tiles = [&](int kt, int buf) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r0 = 4 * (wave + NW * u);
            const double* src = Ap + (int64_t)r0 * ld + kt * KT + dma_goff;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(smem_b + buf * A_BYTES + r0 * 256), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int r0 = 4 * (wave + NW * u);
            const double* src = Bp + (int64_t)r0 * ld + kt * KT + dma_goff;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(smem_b + B_BASE + buf * B_BYTES + r0 * 256), 16, 0, 0);
        }
    };
    // masks: after this wave's DMA of K-step kt has landed, the lane that fetched a chunk zeroes its masked halves in LDS
    auto fix_masks = [&](int kt, int buf) {
        const int c = fr ^ drow;                 // chunk this lane fetched (of every one of its rows)
        const int k = kt * KT + 2 * c;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int row = 4 * (wave + NW * u) + q;
            bool zx = false, zy = false;
            if (a_lo) { zx |= (k > row + a_lo_shift); zy |= (k + 1 > row + a_lo_shift); }
            if (a_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + buf * A_BYTES + 4 * (wave + NW * u) * 256 + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int row = 4 * (wave + NW * u) + q;
            bool zx = false, zy = false;
            if (b_lo) { zx |= (k > row + b_lo_shift); zy |= (k + 1 > row + b_lo_shift); }
            if (b_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + B_BASE + buf * B_BYTES + 4 * (wave + NW * u) * 256 + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
    };
    int a_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (fr + 4 * r) & 15;
        a_off[r] = (wm * (BM / WM) + row) * (KT * 8) + ((q ^ row) << 4);
    }
    const int b_off = (wn * (BN / WN) + fr) * (KT * 8) + ((q ^ fr) << 4) + B_BASE;
    auto read_frags = [&](auto BUFC, auto GC, d2_t (&a)[TM][4], d2_t (&b)[TN]) {
        constexpr int buf = decltype(BUFC)::value, g = decltype(GC)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const char* pa = smem_b + (a_off[r] ^ (g << 6));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a[mi][r] = *reinterpret_cast<const d2_t*>(pa + buf * A_BYTES + mi * (16 * KT * 8));
        }
        const char* pb = smem_b + (b_off ^ (g << 6));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            b[ni] = *reinterpret_cast<const d2_t*>(pb + buf * B_BYTES + ni * (16 * KT * 8));
    };
    auto mma = [&](d2_t (&a)[TM][4], d2_t (&b)[TN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mi][ni][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi][r][h], b[ni][h], acc[mi][ni][r], 0, 0, 0);
    };
    d2_t a0[TM][4], b0[TN], a1[TM][4], b1[TN];
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    dma_tiles(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (any_mask) fix_masks(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) dma_tiles(1, 1);
    read_frags(I0{}, I0{}, a0, b0);
    auto kstep = [&](int kt, auto BUFC) {
        constexpr int buf = decltype(BUFC)::value;
        using IB = std::integral_constant<int, buf>;
        using IN = std::integral_constant<int, buf ^ 1>;
        const bool more = kt + 1 < nk;
        __builtin_amdgcn_sched_barrier(0);
        read_frags(IB{}, I1{}, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(IB{}, I2{}, a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(IB{}, I3{}, a1, b1);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        // the DMA of K-step kt + 1 (issued one K-step ago) has landed for this wave; after the barrier for all of them, and
        // every wave's reads of this K-step's buffer are complete, so it can take K-step kt + 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (any_mask && more) fix_masks(kt + 1, buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) dma_tiles(kt + 2, buf);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_frags(IN{}, I0{}, a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, I0{});
        if (kt + 1 < nk) kstep(kt + 1, I1{});
    }
    const double alpha = t.alpha;
    const bool mirror = (t.c2_off >= 0);
    double* C2p = C2 + (mirror ? t.c2_off : 0);
    int q2 = q, fr2 = fr, cb2 = cb;
    asm volatile("" : "+v"(q2), "+v"(fr2), "+v"(cb2));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb2 + r) & 3) + q2;
                const int col = wn * (BN / WN) + ni * 16 + fr2;
                const double v = alpha * acc[mi][ni][r];
                Cp[(int64_t)row * ld + col] = v;
                if (mirror) C2p[(int64_t)col * ld + row] = v;
            }
}

// ---- occupancy-2 variant: 4 waves per workgroup (2 x 2, wave tile 64 x 64), K-steps of 16 columns, two 32 KB LDS stages =
// 64 KB per workgroup, <= 256 VGPRs: TWO workgroups per CU, so that one's barriers, C pre-load and epilogue run under the
// other's MFMAs.  LDS-DMA staging as in gemm444_glds; fragments are read per 8-column group without register double buffering
// (the other workgroup's wave on the same SIMD covers the LDS latency).  SRC_WRAP (probe): take every K-step's operands from
// the first two K-steps (L2-resident) to tell the fabric's share.
template <int BM, int BN, int WM, int WN, int KT, int NST, bool SRC_WRAP>
__device__ __forceinline__ void gemm444_occ2(const GemmTask t, const double* A, const double* B, double* C, double* C2, int ld) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int ROWB = KT * 8;                 // bytes per tile row and stage
    constexpr int CPR = KT / 2, SWM = CPR - 1;   // 16-byte chunks per row
    constexpr int RPI = 1024 / ROWB;             // tile rows per DMA wave-instruction (1 KiB)
    constexpr int NA = BM / (RPI * NW), NBC = BN / (RPI * NW);
    constexpr int NG = KT / 8;
    static_assert(NA >= 1 && NBC >= 1 && (KT == 16 || KT == 32), "shape");
    static_assert((RPI * NW) % 16 == 0 || RPI * NW == 8, "row blocks per wave must keep the swizzle phase constant");
    extern __shared__ __attribute__((aligned(1024))) double smem[];
    char* const smem_b = reinterpret_cast<char*>(smem);
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int B_BASE = NST * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, q = lane >> 4, cb = (lane >> 2) & 3;
    const double* Ap = A + t.a_off;
    const double* Bp = B + t.b_off;
    const int nk = t.klen / KT;
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER;
    const bool b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const bool any_mask = (t.flags & 15) != 0;
    const int a_lo_shift = t.klen - BM, b_lo_shift = t.klen - BN;

    double acc[TM][TN][4];
    double* const Cp = C + t.c_off;
    const bool preload = (t.beta != 0.0);
    const double c_scale = preload ? t.beta / t.alpha : 0.0;
    if (preload) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                    const int col = wn * (BN / WN) + ni * 16 + fr;
                    acc[mi][ni][r] = c_scale * Cp[(int64_t)row * ld + col];
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
    }
    // DMA: wave-instruction u covers rows r0 = RPI (wave + NW u) .. r0 + RPI - 1; lane l -> row r0 + l / CPR, slot l % CPR
    const int dl_row = lane / CPR, dl_slot = lane % CPR;
    const int drow = ((RPI * wave) + dl_row) & SWM;              // (row & SWM) of this lane's DMA rows (RPI NW u = 0 mod CPR)
    const unsigned dma_goff = (unsigned)(dl_row * ld + 2 * (dl_slot ^ drow));
    auto dma_tiles = [&](int kt, int st) {
        const int ks = SRC_WRAP ? (kt & 1) : kt;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r0 = RPI * (wave + NW * u);
            __builtin_amdgcn_global_load_lds(Ap + (int64_t)r0 * ld + ks * KT + dma_goff,
                                             (lds_ptr_t)(smem_b + st * A_BYTES + r0 * ROWB), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int r0 = RPI * (wave + NW * u);
            __builtin_amdgcn_global_load_lds(Bp + (int64_t)r0 * ld + ks * KT + dma_goff,
                                             (lds_ptr_t)(smem_b + B_BASE + st * B_BYTES + r0 * ROWB), 16, 0, 0);
        }
    };
    auto fix_masks = [&](int kt, int st) {
        const int k = kt * KT + 2 * (dl_slot ^ drow);
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r0 = RPI * (wave + NW * u), row = r0 + dl_row;
            bool zx = false, zy = false;
            if (a_lo) { zx |= (k > row + a_lo_shift); zy |= (k + 1 > row + a_lo_shift); }
            if (a_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + st * A_BYTES + r0 * ROWB + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int r0 = RPI * (wave + NW * u), row = r0 + dl_row;
            bool zx = false, zy = false;
            if (b_lo) { zx |= (k > row + b_lo_shift); zy |= (k + 1 > row + b_lo_shift); }
            if (b_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + B_BASE + st * B_BYTES + r0 * ROWB + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
    };
    int a_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (fr + 4 * r) & 15;
        a_off[r] = (wm * (BM / WM) + row) * ROWB + ((q ^ (row & SWM)) << 4);
    }
    const int b_off = (wn * (BN / WN) + fr) * ROWB + ((q ^ (fr & SWM)) << 4) + B_BASE;
    auto group = [&](int st, auto GC) {
        constexpr int g = decltype(GC)::value;
        d2_t a[TM][4], b[TN];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const char* pa = smem_b + st * A_BYTES + (a_off[r] ^ (g << 6));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi][r] = *reinterpret_cast<const d2_t*>(pa + mi * (16 * ROWB));
        }
        const char* pb = smem_b + st * B_BYTES + (b_off ^ (g << 6));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const d2_t*>(pb + ni * (16 * ROWB));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mi][ni][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi][r][h], b[ni][h], acc[mi][ni][r], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // prologue: NST - 1 stages in flight
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nk) dma_tiles(s0, s0);
    int st = 0;                               // stage of K-step kt
    for (int kt = 0; kt < nk; ++kt) {
        // K-step kt's DMA is the oldest outstanding one of this wave: leave the younger NST - 2 stages in flight
        if (NST == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBC) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NBC)) : "memory");
        if (any_mask) fix_masks(kt, st);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // stage st landed for every wave; every wave finished reading stage st - 1
        {
            const int kn = kt + NST - 1;      // refill the stage that K-step kt - 1 used
            int sn = st - 1; if (sn < 0) sn += NST;
            if (kn < nk) dma_tiles(kn, sn);
        }
        group(st, I0{});
        group(st, I1{});
        if (NG == 4) { group(st, I2{}); group(st, I3{}); }
        st = (st + 1 == NST) ? 0 : st + 1;
    }
    const double alpha = t.alpha;
    const bool mirror = (t.c2_off >= 0);
    double* C2p = C2 + (mirror ? t.c2_off : 0);
    int q2 = q, fr2 = fr, cb2 = cb;
    asm volatile("" : "+v"(q2), "+v"(fr2), "+v"(cb2));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb2 + r) & 3) + q2;
                const int col = wn * (BN / WN) + ni * 16 + fr2;
                const double v = alpha * acc[mi][ni][r];
                Cp[(int64_t)row * ld + col] = v;
                if (mirror) C2p[(int64_t)col * ld + row] = v;
            }
}
#define LAB_KERNEL_O(NAME, BM, BN, WM, WN, KT, NST, WRAP, OCC)                                                              \
    __global__ __launch_bounds__(64 * WM * WN, OCC) void NAME(const GemmTask* __restrict__ tasks, const double* A,          \
                                                            const double* B, double* C, double* C2, int ld) {              \
        gemm444_occ2<BM, BN, WM, WN, KT, NST, WRAP>(tasks[blockIdx.x], A, B, C, C2, ld);                                  \
    }
LAB_KERNEL_O(lab_o128_w22_k16s2, 128, 128, 2, 2, 16, 2, false, 2)     // 64 KB, two workgroups per CU
LAB_KERNEL_O(lab_o128_w22_k16s2w, 128, 128, 2, 2, 16, 2, true, 2)     //   ... operands from L2 (probe)
LAB_KERNEL_O(lab_o128_w42_k16s4, 128, 128, 4, 2, 16, 4, false, 1)     // 8 waves, four 32 KB stages: three DMA stages in flight
LAB_KERNEL_O(lab_o128_w42_k32s2, 128, 128, 4, 2, 32, 2, false, 1)     // 8 waves, the g128 shape without fragment prefetch
LAB_KERNEL_O(lab_o128_w42_k32s2w, 128, 128, 4, 2, 32, 2, true, 1)
LAB_KERNEL_O(lab_o64_w22_k32s2, 64, 64, 2, 2, 32, 2, false, 2)

// ---- two independent 4-wave teams in ONE 8-wave workgroup (128 KB), persistent over a task counter ------------------------------
// Why: with two 64 KB workgroups per CU a kernel that needs a whole CU (the leaf: 138 KB) starves until the bulk launch has no
// workgroup left to dispatch -- every slot a retiring workgroup frees is refilled at once (timeline at N = 8192: the leaf waits
// 365 us per macro panel, chain and bulk stream do not overlap any more).  One 128 KB workgroup per CU keeps the leaf's wait at one
// workgroup's retirement, and a grid of fewer workgroups than CUs leaves CUs free for the chain altogether -- if the two halves of
// the workgroup stay as independent as two workgroups were: separate tasks, separate LDS stages, separate (software) barriers.
#ifndef LAB_TEAM_SLEEP
#define LAB_TEAM_SLEEP 0
#endif
__device__ __forceinline__ void team_barrier(volatile unsigned* bar, unsigned& epoch) {
    // every LDS operation of this wave has completed (the callers wait lgkmcnt(0) / vmcnt as their protocol needs) -> arrive, poll
    epoch += 4;
    if ((threadIdx.x & 63) == 0) atomicAdd(const_cast<unsigned*>(bar), 1u);
    unsigned spins = 0;
    while ((int)(*bar - epoch) < 0) {      // wave-uniform broadcast read
        if (LAB_TEAM_SLEEP) __builtin_amdgcn_s_sleep(LAB_TEAM_SLEEP);
        if (++spins > (1u << 22)) break;   // bounded: a lost wave ends the kernel with wrong numbers, never with a hang
    }
}
template <int BM, int BN, int WM, int WN, int KT, int NST>
__device__ __forceinline__ void gemm_team_body(const GemmTask t, const double* A, const double* B, double* C, double* C2, int ld,
                                               char* const smem_b, const int wave, volatile unsigned* bar, unsigned& epoch) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int ROWB = KT * 8;                 // bytes per tile row and stage
    constexpr int CPR = KT / 2, SWM = CPR - 1;   // 16-byte chunks per row; XOR mask of the chunk swizzle
    constexpr int RPI = 1024 / ROWB;             // tile rows per DMA wave-instruction (1 KiB)
    constexpr int NA = BM / (RPI * NW), NBC = BN / (RPI * NW);
    constexpr int NG = KT / 8;                   // 8-column groups per K-step
    static_assert(NA >= 1 && NBC >= 1 && (KT == 16 || KT == 32) && NST >= 2 && NST <= 4, "tile / stage combination");
    static_assert((RPI * NW) % CPR == 0, "a wave's DMA row blocks must share one swizzle phase");
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int B_BASE = NST * A_BYTES;
    const int lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, q = lane >> 4, cb = (lane >> 2) & 3;
    const double* Ap = A + t.a_off;
    const double* Bp = B + t.b_off;
    const int nk = t.klen / KT;
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER;
    const bool b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const bool any_mask = (t.flags & (TF_A_LOWER | TF_A_UPPER | TF_B_LOWER | TF_B_UPPER)) != 0;
    const int a_lo_shift = t.klen - BM, b_lo_shift = t.klen - BN;

    // acc[mi][ni][s]: rotation s of the 16x16 block (mi, ni): row 16 mi + 4 ((cb + s) & 3) + q, column 16 ni + fr
    double acc[TM][TN][4];
    double* const Cp = C + t.c_off;
    const bool preload = (t.beta != 0.0);
    const double c_scale = preload ? t.beta / t.alpha : 0.0;
    if (preload) {   // ONE branch around all the loads: issued together, waited for progressively (inside the element loop the
                     // compiler emits a branch, a load and a vmcnt(0) per element: 32-64 serial round trips per tile)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                    const int col = wn * (BN / WN) + ni * 16 + fr;
                    acc[mi][ni][r] = c_scale * Cp[(int64_t)row * ld + col];
                }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
    }
    // DMA: wave-instruction u of this wave covers tile rows r0 = RPI (wave + NW u) .. r0 + RPI - 1; lane l lands at LDS byte
    // r0 ROWB + 16 l = (row r0 + l / CPR, slot l % CPR) and therefore fetches chunk slot ^ (row & SWM) of that row.
    const int dl_row = lane / CPR, dl_slot = lane % CPR;
    const int drow = ((RPI * wave) + dl_row) & SWM;              // (row & SWM) of this lane's rows: the same for every u
    // ONE 64-bit lane pointer per operand; the row block of instruction u and the K-step are scalar offsets added per
    // instruction (laundered so that they are not folded back into eight loop-invariant vector bases: 12 VGPRs)
    const double* const a_lane = Ap + (dl_row * ld + 2 * (dl_slot ^ drow));
    const double* const b_lane = Bp + (dl_row * ld + 2 * (dl_slot ^ drow));
    auto dma_tiles = [&](int kt, int st) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r0 = RPI * (wave + NW * u);
            int off = r0 * ld + kt * KT;
            asm volatile("" : "+s"(off));
            __builtin_amdgcn_global_load_lds(a_lane + off, (lds_ptr_t)(smem_b + st * A_BYTES + r0 * ROWB), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int r0 = RPI * (wave + NW * u);
            int off = r0 * ld + kt * KT;
            asm volatile("" : "+s"(off));
            __builtin_amdgcn_global_load_lds(b_lane + off, (lds_ptr_t)(smem_b + B_BASE + st * B_BYTES + r0 * ROWB), 16, 0, 0);
        }
    };
    auto fix_masks = [&](int kt, int st) {
        int dr = dl_row, ds = dl_slot ^ drow;
        asm volatile("" : "+v"(dr), "+v"(ds));   // rare path: recompute per call instead of keeping per-chunk rows live
        const int k = kt * KT + 2 * ds;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int r0 = RPI * (wave + NW * u), row = r0 + dr;
            bool zx = false, zy = false;
            if (a_lo) { zx |= (k > row + a_lo_shift); zy |= (k + 1 > row + a_lo_shift); }
            if (a_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + st * A_BYTES + r0 * ROWB + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int r0 = RPI * (wave + NW * u), row = r0 + dr;
            bool zx = false, zy = false;
            if (b_lo) { zx |= (k > row + b_lo_shift); zy |= (k + 1 > row + b_lo_shift); }
            if (b_up) { zx |= (k < row); zy |= (k + 1 < row); }
            double* p = reinterpret_cast<double*>(smem_b + B_BASE + st * B_BYTES + r0 * ROWB + lane * 16);
            if (zx) p[0] = 0.0;
            if (zy) p[1] = 0.0;
        }
    };
    // Fragment addresses: chunk(g) = (4 g + q) ^ (row & SWM) = chunk(0) ^ 4 g, so group g is group 0 with byte-offset bit 6
    // (and 7) flipped: five address registers per lane (four rotations of A, one for B), everything else immediate.
    int a_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (fr + 4 * r) & 15;
        a_off[r] = (wm * (BM / WM) + row) * ROWB + ((q ^ (row & SWM)) << 4);
    }
    const int b_off = (wn * (BN / WN) + fr) * ROWB + ((q ^ (fr & SWM)) << 4) + B_BASE;
    // One 8-column group.  The A fragments are taken one row block at a time (MH) where TM = 4: with all of them in flight the body
    // needs 256 VGPRs, and two such waves per SIMD leave no register for a wave of the serial chain's kernels, which then wait
    // for a bulk workgroup to retire (measured: chain launches 27 -> 46 us at N = 8192).  As written: 206, so a chain wave of
    // up to 96 registers fits beside two bulk waves.  (No sched_barrier inside the group: pinning the order there makes the
    // register allocator ping-pong 40 of the 64 accumulators between two registers -- MFMAs with D != C -- 240 VGPRs.)
    auto group = [&](int st, auto GC) {
        constexpr int g = decltype(GC)::value;
        constexpr int MH = TM > 2 ? TM / 4 : TM;
        d2_t b[TN];
        const char* pb = smem_b + st * B_BYTES + (b_off ^ (g << 6));
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b[ni] = *reinterpret_cast<const d2_t*>(pb + ni * (16 * ROWB));
#pragma unroll
        for (int m0 = 0; m0 < TM; m0 += MH) {
            d2_t a[MH][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const char* pa = smem_b + st * A_BYTES + (a_off[r] ^ (g << 6));
#pragma unroll
                for (int mi = 0; mi < MH; ++mi) a[mi][r] = *reinterpret_cast<const d2_t*>(pa + (m0 + mi) * (16 * ROWB));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mi = 0; mi < MH; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[m0 + mi][ni][r] =
                                __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi][r][h], b[ni][h], acc[m0 + mi][ni][r], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)      // prologue: NST - 1 stages in flight
        if (s0 < nk) dma_tiles(s0, s0);
    int st = 0;                               // stage of K-step kt
    for (int kt = 0; kt < nk; ++kt) {
        // K-step kt's DMA is this wave's oldest outstanding one: the younger NST - 2 stages stay in flight
        if (NST == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBC) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NBC)) : "memory");
        if (any_mask) fix_masks(kt, st);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        team_barrier(bar, epoch);
        {
            const int kn = kt + NST - 1;      // refill the stage K-step kt - 1 used
            int sn = st - 1;
            if (sn < 0) sn += NST;
            if (kn < nk) dma_tiles(kn, sn);
        }
        group(st, I0{});
        group(st, I1{});
        if (NG == 4) {
            group(st, I2{});
            group(st, I3{});
        }
        st = (st + 1 == NST) ? 0 : st + 1;
    }
    const double alpha = t.alpha;
    const bool mirror = (t.c2_off >= 0);
    double* C2p = C2 + (mirror ? t.c2_off : 0);
    // output addresses from laundered lane indices: shared with the pre-load they would stay live across the K loop
    int q2 = q, fr2 = fr, cb2 = cb;
    asm volatile("" : "+v"(q2), "+v"(fr2), "+v"(cb2));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb2 + r) & 3) + q2;
                const int col = wn * (BN / WN) + ni * 16 + fr2;
                const double v = alpha * acc[mi][ni][r];
                Cp[(int64_t)row * ld + col] = v;
                if (mirror) C2p[(int64_t)col * ld + row] = v;
            }
}


template <int RESERVED>
__global__ __launch_bounds__(512, 1) void lab_team_persist(const GemmTask* __restrict__ tasks, int ntasks, unsigned* counter,
                                                           const double* A, const double* B, double* C, double* C2, int ld) {
    extern __shared__ __attribute__((aligned(1024))) double smem[];
    char* const base = reinterpret_cast<char*>(smem);
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int team = wave8 >> 2, wave = wave8 & 3;
    unsigned* const ctl = reinterpret_cast<unsigned*>(base + 2 * 65536) + team * 32;    // [0] barrier counter, [16] next task
    if (threadIdx.x < 64) reinterpret_cast<unsigned*>(base + 2 * 65536)[threadIdx.x] = 0u;
    __syncthreads();
    unsigned epoch = 0;
    for (int round = 0;; ++round) {
        unsigned t;
        if (RESERVED == 2) {               // probe: one task per team, no counter (what the fetch and the lost locality cost)
            if (round) break;
            t = 2 * blockIdx.x + team;
        } else {
            if (wave == 0 && (threadIdx.x & 63) == 0) ctl[16] = atomicAdd(counter, 1u);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            team_barrier(ctl, epoch);
            t = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(ctl + 16));
        }
        if (t >= (unsigned)ntasks) break;
        gemm_team_body<128, 128, 2, 2, 16, 2>(tasks[t], A, B, C, C2, ld, base + team * 65536, wave, ctl, epoch);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        team_barrier(ctl, epoch);        // every wave of the team is through with the task's LDS stages and with ctl[16]
    }
}

#define LAB_KERNEL_G(NAME, BM, BN, WM, WN)                                                                                  \
    __global__ __launch_bounds__(64 * WM * WN, 1) void NAME(const GemmTask* __restrict__ tasks, const double* A,            \
                                                          const double* B, double* C, double* C2, int ld) {                \
        gemm444_glds<BM, BN, WM, WN>(tasks[blockIdx.x], A, B, C, C2, ld);                                                 \
    }
LAB_KERNEL_G(lab_g128_w42_m0, 128, 128, 4, 2)
LAB_KERNEL_G(lab_g128_w22_m0, 128, 128, 2, 2)
LAB_KERNEL_G(lab_g64_w22_m0, 64, 64, 2, 2)
__global__ __launch_bounds__(256, 1) void lab_g64_w22_hi(const GemmTask* __restrict__ tasks, const double* A, const double* B,
                                                         double* C, double* C2, int ld) {
    gemm444_glds<64, 64, 2, 2, 65536>(tasks[blockIdx.x], A, B, C, C2, ld);      // the same tile, its LDS image above 64 KB
}

#define LAB_KERNEL(NAME, BM, BN, WM, WN, MODE, WPS)                                                                         \
    __global__ __launch_bounds__(64 * WM * WN, 1) void NAME(const GemmTask* __restrict__ tasks, const double* A,            \
                                                          const double* B, double* C, double* C2, int ld) {                \
        gemm444_pipe<BM, BN, WM, WN, MODE>(tasks[blockIdx.x], A, B, C, C2, ld);                                           \
    }
LAB_KERNEL(lab_p128_w42_m0, 128, 128, 4, 2, 0, 2)
LAB_KERNEL(lab_p128_w42_m1, 128, 128, 4, 2, 1, 2)
LAB_KERNEL(lab_p128_w42_m2, 128, 128, 4, 2, 2, 2)
LAB_KERNEL(lab_p128_w22_m0, 128, 128, 2, 2, 0, 1)
LAB_KERNEL(lab_p128_w22_m1, 128, 128, 2, 2, 1, 1)
LAB_KERNEL(lab_p128_w22_m2, 128, 128, 2, 2, 2, 1)
LAB_KERNEL(lab_p64_w22_m0, 64, 64, 2, 2, 0, 1)
LAB_KERNEL(lab_p64_w42_m0, 64, 64, 4, 2, 0, 2)   // TM = 1, TN = 2

typedef void (*kern_t)(const GemmTask*, const double*, const double*, double*, double*, int);
struct Variant { const char* name; kern_t k; int tile, threads; int lds = 0; };
static const Variant g_variants[] = {
    {"p128_w42_m0", lab_p128_w42_m0, 128, 512}, {"p128_w42_m1", lab_p128_w42_m1, 128, 512}, {"p128_w42_m2", lab_p128_w42_m2, 128, 512},
    {"p128_w22_m0", lab_p128_w22_m0, 128, 256}, {"p128_w22_m1", lab_p128_w22_m1, 128, 256}, {"p128_w22_m2", lab_p128_w22_m2, 128, 256},
    {"p64_w22_m0", lab_p64_w22_m0, 64, 256}, {"p64_w42_m0", lab_p64_w42_m0, 64, 512},
    {"g128_w42_m0", lab_g128_w42_m0, 128, 512}, {"g128_w22_m0", lab_g128_w22_m0, 128, 256}, {"g64_w22_m0", lab_g64_w22_m0, 64, 256},
    {"g64hi_w22_m0", lab_g64_w22_hi, -64, 256},
    {"team128_persist_m0", nullptr, 128, 512, 2 * 65536 + 1280},
    {"team128_onetask_m0", nullptr, 1128, 512, 2 * 65536 + 1280},
    {"o128_w22_k16s2_m0", lab_o128_w22_k16s2, 128, 256, 2 * 32768}, {"o128_w22_k16s2w", lab_o128_w22_k16s2w, 128, 256, 2 * 32768},
    {"o128_w42_k16s4_m0", lab_o128_w42_k16s4, 128, 512, 4 * 32768}, {"o128_w42_k32s2_m0", lab_o128_w42_k32s2, 128, 512, 2 * 65536},
    {"o128_w42_k32s2w", lab_o128_w42_k32s2w, 128, 512, 2 * 65536}, {"o64_w22_k32s2_m0", lab_o64_w22_k32s2, 64, 256, 2 * 32768},
};
constexpr int NVAR = sizeof(g_variants) / sizeof(g_variants[0]);

__global__ void lab_fill(double* p, int64_t n, unsigned seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        p[i] = ((double)(h >> 11) / 9007199254740992.0 - 0.5) * 2.0;
    }
}
}  // namespace lab

static int g_persist_grid = 250;
extern "C" {
void lab_set_persist_grid(int g) { g_persist_grid = g; }
int lab_num_variants() { return lab::NVAR; }
const char* lab_variant_name(int v) { return (v >= 0 && v < lab::NVAR) ? lab::g_variants[v].name : ""; }

// One launch of (n/tile)^2 tasks C[i][j] (+)= A_i B_j^T, K deep, on an n x n matrix (ld = n): A = rows of S, B = rows of S,
// C into a second matrix.  flags / beta as given (flags masks use the kernel's own convention).  Returns the average launch
// ms over reps in *ms.  If ref != NULL it holds the expected C (n x n) and *maxdiff receives max |C - ref|; if out != NULL
// the result is copied there.
int lab_run(int variant, int n, int K, int flags, double beta, int reps, double* ms, const double* ref, double* out, double* maxdiff) {
    using namespace lab;
    if (variant < 0 || variant >= NVAR) return -1;
    const Variant& v = g_variants[variant];
    const bool onetask = (v.tile == 1128);
    const int tile = onetask ? 128 : (v.tile < 0 ? -v.tile : v.tile);
    if (n % tile || K % 32 || K > n) return -1;
    double *S = nullptr, *Cm = nullptr;
    GemmTask* dt = nullptr;
    const size_t bytes = (size_t)n * n * 8;
    if (hipMalloc(&S, bytes) != hipSuccess || hipMalloc(&Cm, bytes) != hipSuccess) return -2;
    hipLaunchKernelGGL(lab_fill, dim3(2048), dim3(256), 0, 0, S, (int64_t)n * n, 1u);
    hipLaunchKernelGGL(lab_fill, dim3(2048), dim3(256), 0, 0, Cm, (int64_t)n * n, 2u);
    std::vector<GemmTask> ts;
    const int nt = n / tile;
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j < nt; ++j) {
            GemmTask t{};
            t.a_off = (int64_t)i * tile * n;
            t.b_off = (int64_t)j * tile * n;
            t.c_off = (int64_t)i * tile * n + (int64_t)j * tile;
            t.c2_off = -1;
            t.klen = K; t.flags = flags; t.alpha = 1.0; t.beta = beta;
            ts.push_back(t);
        }
    if (hipMalloc(&dt, ts.size() * sizeof(GemmTask)) != hipSuccess) return -2;
    hipMemcpy(dt, ts.data(), ts.size() * sizeof(GemmTask), hipMemcpyHostToDevice);
    const size_t lds = v.lds ? (size_t)v.lds : (size_t)2 * (tile + tile) * 32 * 8 + (v.tile < 0 ? 65536 : 0);
    if (v.k) hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned* ctr = nullptr;
    hipMalloc(&ctr, 64);
    const int grid_p = std::min((int)(ts.size() + 1) / 2, g_persist_grid);
    auto launch = [&]() {
        if (v.k) {
            hipLaunchKernelGGL(v.k, dim3((unsigned)ts.size()), dim3(v.threads), lds, 0, dt, S, S, Cm, nullptr, n);
        } else {
            hipMemsetAsync(ctr, 0, 4, 0);
            if (onetask)
                hipLaunchKernelGGL(lab_team_persist<2>, dim3((unsigned)(ts.size() + 1) / 2), dim3(512), lds, 0, dt, (int)ts.size(), ctr, S, S, Cm, nullptr, n);
            else
                hipLaunchKernelGGL(lab_team_persist<0>, dim3(grid_p), dim3(512), lds, 0, dt, (int)ts.size(), ctr, S, S, Cm, nullptr, n);
        }
    };
    if (!v.k) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(lab_team_persist<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(lab_team_persist<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    launch();   // warm + result
    if (out || ref) {
        std::vector<double> host((size_t)n * n);
        hipMemcpy(host.data(), Cm, bytes, hipMemcpyDeviceToHost);
        if (out) std::copy(host.begin(), host.end(), out);
        if (ref && maxdiff) {
            double md = 0.0;
            for (size_t i = 0; i < host.size(); ++i) md = std::max(md, std::abs(host[i] - ref[i]));
            *maxdiff = md;
        }
    }
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float t_ms = 0.f;
    hipEventElapsedTime(&t_ms, e0, e1);
    *ms = t_ms / reps;
    const hipError_t err = hipGetLastError();
    hipFree(S); hipFree(Cm); hipFree(dt); hipFree(ctr);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return err == hipSuccess ? 0 : -3;
}

// Traffic experiment (round 4): the SAME kernel and the same (n/tile)^2 tasks, enumerated differently.
//   order 0: row-major (i, j)
//   order 1: every task reads operand panels 0 / 0 (C tiles stay distinct): the launch with no operand traffic to speak of
//   order 2: XCD-aware deal of bi x bj super-blocks: workgroup p runs on XCD p mod 8; XCD x walks super-blocks x, x + 8, ...
//   order 3: as 2, but the tiles of a super-block are dealt k-phase-shifted: tile t starts its K loop at chunk (t * K / (bi*bj)) -- not
//            expressible with this task format, so: unused
// Timed over `reps` back-to-back launches (make that >= 1 s for the power-managed steady state).
int lab_run_order(int variant, int n, int K, int order, int bi, int bj, int reps, double* ms) {
    using namespace lab;
    if (variant < 0 || variant >= NVAR) return -1;
    const Variant& v = g_variants[variant];
    if (!v.k || v.tile <= 0 || v.tile > 128) return -1;
    const int tile = v.tile;
    if (n % tile || K % 32 || K > n) return -1;
    const int nt = n / tile;
    if (order == 2 && (bi <= 0 || bj <= 0 || nt % bi || nt % bj)) return -1;
    double *S = nullptr, *Cm = nullptr;
    GemmTask* dt = nullptr;
    const size_t bytes = (size_t)n * n * 8;
    if (hipMalloc(&S, bytes) != hipSuccess || hipMalloc(&Cm, bytes) != hipSuccess) return -2;
    hipLaunchKernelGGL(lab_fill, dim3(2048), dim3(256), 0, 0, S, (int64_t)n * n, 1u);
    hipLaunchKernelGGL(lab_fill, dim3(2048), dim3(256), 0, 0, Cm, (int64_t)n * n, 2u);
    auto mk = [&](int i, int j) {
        GemmTask t{};
        t.a_off = order == 1 ? 0 : (int64_t)i * tile * n;
        t.b_off = order == 1 ? 0 : (int64_t)j * tile * n;
        t.c_off = (int64_t)i * tile * n + (int64_t)j * tile;
        t.c2_off = -1;
        t.klen = K; t.flags = 0; t.alpha = 1.0; t.beta = 1.0;
        return t;
    };
    std::vector<GemmTask> ts;
    if (order != 2) {
        for (int i = 0; i < nt; ++i)
            for (int j = 0; j < nt; ++j) ts.push_back(mk(i, j));
    } else {
        std::vector<std::vector<GemmTask>> per(8);
        int sb = 0;
        for (int I = 0; I < nt / bi; ++I)
            for (int J = 0; J < nt / bj; ++J, ++sb)
                for (int i = 0; i < bi; ++i)
                    for (int j = 0; j < bj; ++j) per[sb % 8].push_back(mk(I * bi + i, J * bj + j));
        size_t longest = 0;
        for (auto& q : per) longest = std::max(longest, q.size());
        for (size_t s2 = 0; s2 < longest; ++s2)
            for (int x = 0; x < 8; ++x) {
                if (s2 < per[x].size()) ts.push_back(per[x][s2]);
                else { ts.clear(); hipFree(S); hipFree(Cm); return -4; }   // uneven deal: the positions would shift XCDs
            }
    }
    if (hipMalloc(&dt, ts.size() * sizeof(GemmTask)) != hipSuccess) return -2;
    hipMemcpy(dt, ts.data(), ts.size() * sizeof(GemmTask), hipMemcpyHostToDevice);
    const size_t lds = v.lds ? (size_t)v.lds : (size_t)2 * (tile + tile) * 32 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL(v.k, dim3((unsigned)ts.size()), dim3(v.threads), lds, 0, dt, S, S, Cm, nullptr, n); };
    launch();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float t_ms = 0.f;
    hipEventElapsedTime(&t_ms, e0, e1);
    *ms = t_ms / reps;
    const hipError_t err = hipGetLastError();
    hipFree(S); hipFree(Cm); hipFree(dt);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return err == hipSuccess ? 0 : -3;
}
}
