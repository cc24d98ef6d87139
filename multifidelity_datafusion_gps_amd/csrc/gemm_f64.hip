// gemm_f64.hip -- the fp64 MFMA tile-GEMM kernel every level-3 step of the GP evaluation runs on.
//
// One workgroup (8 waves, 4x2) computes one BM x BN output tile described by a GemmTask:
//     C = beta*C + alpha * A[i0.., k0..k0+klen) * B[j0.., k0'..)^T          ("NT": both K-contiguous)
// on v_mfma_f64_4x4x4_4b_f64 (round 3; v_mfma_f64_16x16x4_f64 before).  Cholesky panel solves (as products with the
// running inverse), SYRK trailing updates, the triangular inverse, K^-1 = L^-T L^-1 and the predictive-variance product
// are all expressed as lists of such tasks (see plan.cpp); triangular structure = trimmed K ranges
// plus one masked diagonal window.
//
// Why the 4x4x4 shape (profiles/r03_probes.txt, r03_mfma_counters.json): on gfx950 a v_mfma_f64_16x16x4 occupies the
// matrix pipe for 64 cycles (SQ_VALU_MFMA_BUSY_CYCLES / instruction) but issues only every ~100 (1, 2 or 4 waves per
// SIMD alike): 46-49 TFLOP/s, pipe 0.6 busy.  v_mfma_f64_4x4x4_4b (four independent 4x4x4 blocks, 512 flops) takes 16
// busy cycles and issues every 17.5 from ONE wave per SIMD: 71 TFLOP/s bare, 0.9 of the 78.6 TFLOP/s vendor figure.
// Its operands sit in the lanes exactly like the 16x16x4 fragments (lane = 16 k + r: element [row r][k]), and block b
// multiplies rows 4b..4b+3 of the A fragment with rows 4b..4b+3 of the B fragment: the four block DIAGONALS of a
// 16x16x4 product.  Reading the A fragment four times with its row blocks rotated (s = 0..3: lane r takes row
// (r + 4s) mod 16 -- the same addresses as the plain fragment, permuted among the lanes, so the swizzled image is
// conflict-free for them too) gives the whole product in four instructions: the result of rotation s in lane
// (i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3) is C[4 ((b + s) & 3) + i][4 b + j].  (CBSZ / ABID do not broadcast
// blocks for f64 -- measured: they act as |A| / |B| modifiers -- hence the rotated reads.)
//
// Replaces LAPACK dpotrf/dtrtri/dpotri/dgemm behind GPy's pdinv / Posterior (SURVEY.md 8(a) a4,a7,a11).
//
// Data path: global -> registers (16 B per lane, rows of 256 B fully coalesced) -> LDS with a
// 16-byte-chunk XOR swizzle (chunk ^= row & 15) so that the MFMA fragment reads (16 rows x 2 k per
// 32-lane group, ds_read_b64) are bank-conflict free without padding -> MFMA.  LDS is double
// buffered: the global loads of K-step t+1 are in flight while step t is on the matrix cores.
#include <mutex>
#include <type_traits>
#include "mfgp_internal.h"

namespace mfgp {

template <int BM, int BN, int WM, int WN, int NBUF = 2, int KT = BK>
__device__ __forceinline__ void gemm_nt_tile(const GemmTask t, const double* A, const double* B, double* C, double* C2,
                                             int ld) {
    constexpr int NT = 64 * WM * WN;      // threads per workgroup (wave grid WM x WN)
    constexpr int TM = BM / (16 * WM);    // 16-row MFMA blocks per wave along M
    constexpr int TN = BN / (16 * WN);
    constexpr int CPR = KT / 2;           // 16-byte chunks per tile row and K-step (KT = K-step depth: 32, or 16 for the slim variant)
    constexpr int NA = BM * CPR / NT;     // chunks per thread per K-step
    constexpr int NBC = BN * CPR / NT;
    constexpr int SWM = CPR - 1;          // XOR swizzle mask over the chunks of a row
    static_assert(NA >= 1 && NBC >= 1 && (CPR == 16 || CPR == 8), "tile / K-step combination not supported");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* As = smem;                     // [NBUF][BM*KT]
    double* Bs = smem + NBUF * BM * KT;    // [NBUF][BN*KT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15;  // fragment row / output column within a 16-block
    const int q = lane >> 4;   // k sub-index of the fragment / output row group

    const double* Ap = A + t.a_off;
    const double* Bp = B + t.b_off;
    const int nk = t.klen / KT;
    const bool a_lo = t.flags & TF_A_LOWER, a_up = t.flags & TF_A_UPPER;
    const bool b_lo = t.flags & TF_B_LOWER, b_up = t.flags & TF_B_UPPER;
    const int a_lo_shift = t.klen - BM, b_lo_shift = t.klen - BN;

    d2_t ra[NA], rb[NBC];
    // acc[mi][ni][s]: rotation s of the 16x16 block (mi, ni); this lane's element is
    //   row 16 mi + 4 ((cb + s) & 3) + q,  column 16 ni + fr      (q = lane >> 4, cb = (lane >> 2) & 3, fr = lane & 15)
    double acc[TM][TN][4];
    const int cb = (lane >> 2) & 3;
    // Accumulating tasks (beta != 0: the trailing updates C -= A B^T) start from the output tile itself: acc = (beta/alpha) C is
    // loaded HERE, its latency hidden behind the first operand loads, instead of a read-modify-write epilogue that every
    // workgroup pays exposed at the end (a 128x128 tile is 128 KB in and 128 KB out at ~25 GB/s per CU: ~5 us each way
    // of an 86 us K = 512 task).  Exact for the alpha = +-1, beta in {0, 1} tasks the planner emits.
    double* const Cp = C + t.c_off;
    const bool preload = (t.beta != 0.0);
    const double c_scale = preload ? t.beta / t.alpha : 0.0;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if (preload) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                    const int col = wn * (BN / WN) + ni * 16 + fr;
                    acc[mi][ni][r] = c_scale * Cp[(int64_t)row * ld + col];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
            }
        }

    // Global -> registers: raw 16-byte loads only, so that they stay in flight across the MFMA phase; the triangular masks
    // (task-uniform flags; most tasks carry none) are applied when the registers are written to LDS, after the compute
    // phase.  (Rounds 1-2 masked right after the load: every K-step then waited for its NEXT operands before it started.)
    const bool any_mask = (t.flags & (TF_A_LOWER | TF_A_UPPER | TF_B_LOWER | TF_B_UPPER)) != 0;
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            ra[u] = *reinterpret_cast<const d2_t*>(Ap + (int64_t)row * ld + kt * KT + 2 * c);
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            rb[u] = *reinterpret_cast<const d2_t*>(Bp + (int64_t)row * ld + kt * KT + 2 * c);
        }
    };
    auto mask_tiles = [&](int kt) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            const int k = kt * KT + 2 * c;
            d2_t v = ra[u];
            if (a_lo) {
                if (k > row + a_lo_shift) v.x = 0.0;
                if (k + 1 > row + a_lo_shift) v.y = 0.0;
            }
            if (a_up) {
                if (k < row) v.x = 0.0;
                if (k + 1 < row) v.y = 0.0;
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NBC; ++u) {
            const int g = tid + NT * u;
            const int row = g / CPR, c = g % CPR;
            const int k = kt * KT + 2 * c;
            d2_t v = rb[u];
            if (b_lo) {
                if (k > row + b_lo_shift) v.x = 0.0;
                if (k + 1 > row + b_lo_shift) v.y = 0.0;
            }
            if (b_up) {
                if (k < row) v.x = 0.0;
                if (k + 1 < row) v.y = 0.0;
            }
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
    auto compute = [&](int buf) {
        const double* as = As + buf * (BM * KT) + (wm * (BM / WM)) * KT;
        const double* bs = Bs + buf * (BN * KT) + (wn * (BN / WN) + fr) * KT;
        // Eight columns per round: lane (fr, q) fetches the 16-byte chunk 4 g + q of its row -- columns 8 g + 2 q and
        // 8 g + 2 q + 1 -- with ONE ds_read_b128 and feeds the first half to one MFMA step and the second to the next:
        // a step's four k-slots are then columns 8 g + {0, 2, 4, 6} (resp. {1, 3, 5, 7}), the same for both operands.
        // (Bank check, ds_read_b128 lane groups {0-3,12-15,20-27} ...: 8 lanes of slot q on rows R, 8 of slot q + 1 on the
        // complement of R mod 16; chunk ^ row maps them to 16 distinct 16-byte slots for KT = 32 and for KT = 16.)
#pragma unroll
        for (int g = 0; g < KT / 8; ++g) {
            d2_t a[TM][4], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mi * 16 + ((fr + 4 * r) & 15);
                    a[mi][r] = *reinterpret_cast<const d2_t*>(as + row * KT + (((4 * g + q) ^ (row & SWM)) << 1));
                }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                b[ni] = *reinterpret_cast<const d2_t*>(bs + ni * 16 * KT + (((4 * g + q) ^ (fr & SWM)) << 1));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mi][ni][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[mi][r][h], b[ni][h], acc[mi][ni][r], 0, 0, 0);
        }
    };

    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    if constexpr (NBUF == 2) {
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (more) load_tiles(kt + 1);
            __builtin_amdgcn_sched_barrier(0);      // the loads are issued before, and not waited for until after, the MFMAs
            compute(kt & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_tiles((kt + 1) & 1, kt + 1);
            __syncthreads();
        }
    } else {
        // slim variant: the next K-step waits in registers while this one is on the matrix cores, and is written to
        // the single LDS buffer between two barriers
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (more) load_tiles(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (more) store_tiles(0, kt + 1);
            __syncthreads();
        }
    }

    // epilogue (this lane's elements: see acc above)
    const double alpha = t.alpha;
    const bool mirror = (t.c2_off >= 0);
    double* C2p = C2 + (mirror ? t.c2_off : 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (BM / WM) + mi * 16 + 4 * ((cb + r) & 3) + q;
                const int col = wn * (BN / WN) + ni * 16 + fr;
                const double v = alpha * acc[mi][ni][r];
                Cp[(int64_t)row * ld + col] = v;
                if (mirror) C2p[(int64_t)col * ld + row] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Bulk tile body (round 3): LDS-DMA staging, two workgroups per CU.
//
// Measured in tools/gemm_lab (profiles/r03_gemm_lab.txt; n = 8192 square task lists, K = 2048 / 512, beta = 1):
//   round-2 structure on v_mfma_f64_16x16x4                    47-50 TFLOP/s
//   the same on v_mfma_f64_4x4x4_4b (gemm_nt_tile below)        55 / 43
//   + fragment prefetch, one barrier per K-step                 55 / 43     (register staging: the K-step ends in a ds_write
//                                                                            burst every wave waits for at the barrier)
//   + LDS-DMA staging (no staging registers, no ds_write pass)  60 / 46
//   4 waves x (64 x 64), K-steps of 16, 64 KB: TWO workgroups per CU   66 / 56    <- this body
//   bounds: LDS + MFMA only 70.5, MFMA only 72 (the 4x4x4 pipe at 17.5 cycles per instruction)
// What the second workgroup buys is what one workgroup cannot hide from itself: its barrier (the matrix pipe drains while
// eight waves meet), its C pre-load and its epilogue run under the other workgroup's MFMAs.  The operands no longer pass
// through registers: global_load_lds_dwordx4 writes 1 KiB per wave-instruction lane-linearly (8 tile rows x 128 B), the
// chunk swizzle sits on the per-lane SOURCE address, the fragment reads apply the same XOR (guide: "swizzle both sides or
// neither").  Triangular masks are a fix-up of the landed chunks -- the lane that fetched a chunk zeroes its masked halves
// after its own vmcnt(0), before the barrier -- so the K loop itself is mask-free.
// Ordering of one K-step (stage st of NST): s_waitcnt vmcnt (this wave's DMA of the step has landed) -> mask fix-up ->
// lgkmcnt(0) -> s_barrier (every wave's part has landed; every wave has finished reading the previous stage) -> issue the
// DMA that refills the previous stage -> fragments + MFMAs of the step.  Reads of a stage are only issued after the barrier
// that follows the wait retiring its DMA, and a stage is refilled only after the barrier that follows its last read.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int BN, int WM, int WN, int KT, int NST>
__device__ __forceinline__ void gemm_nt_dma(const GemmTask t, const double* A, const double* B, double* C, double* C2, int ld) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int ROWB = KT * 8;                 // bytes per tile row and stage
    constexpr int CPR = KT / 2, SWM = CPR - 1;   // 16-byte chunks per row; XOR mask of the chunk swizzle
    constexpr int RPI = 1024 / ROWB;             // tile rows per DMA wave-instruction (1 KiB)
    constexpr int NA = BM / (RPI * NW), NBC = BN / (RPI * NW);
    constexpr int NG = KT / 8;                   // 8-column groups per K-step
    static_assert(NA >= 1 && NBC >= 1 && (KT == 16 || KT == 32) && NST >= 2 && NST <= 4, "tile / stage combination");
    static_assert((RPI * NW) % CPR == 0, "a wave's DMA row blocks must share one swizzle phase");
    extern __shared__ __attribute__((aligned(1024))) double smem[];
    char* const smem_b = reinterpret_cast<char*>(smem);
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int B_BASE = NST * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform within a wave: as a scalar it keeps every address that depends on it (DMA row blocks, LDS
    // destinations) in SGPRs -- the compiler's divergence analysis only sees threadIdx.x >> 6
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        __builtin_amdgcn_s_barrier();
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

constexpr int GW_M = 4, GW_N = 2;                 // 8 waves per workgroup = 2 per SIMD: one wave alone cannot keep
constexpr int GEMM_THREADS = 64 * GW_M * GW_N;
// two bulk waves per SIMD must leave 64 VGPRs (of 512) for a wave of the chain's kernels: see gemm_nt_dma's `group`
constexpr int BULK_THREADS = 256;                 // bulk tile kernels: 4 waves (2 x 2), two workgroups per CU (gemm_nt_dma)    // the fp64 MFMA pipe busy (probe: 140 vs ~103 cycles per MFMA)

// Batched evaluations (mfgp_eval_batch): blockIdx.y = b selects matrix set b, `bstride` elements further on in the handle's batch
// slab -- the same task list serves every set (all four operand bases move by the same offset); a single evaluation is the
// launch with gridDim.y = 1.  Workgroups are dispatched x-fastest, so set b's tasks keep their XCD-aware order among themselves.
//
// Named entry points over the same tile body, so that a kernel trace separates the roles:
//   mfgp_gemm_nt_f64_t128 / _t64 : the many launches of the recursive Cholesky + inverse
//   mfgp_kinv_syrk_f64           : the ONE launch per evaluation that forms K^-1 = L^-T L^-1 (N^3/3 flops)
//   mfgp_predvar_f64             : the predictive-variance product V = K(X*,X) L^-T
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_gemm_nt_f64_t128(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<128, 128, 2, 2, 16, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_gemm_nt_f64_t64(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<64, 64, 2, 2, 32, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}
// chain variant: the GEMM steps on the serial Cholesky chain (panel, column update).  16 KB of LDS (one buffer, K-steps of
// 16) instead of 64, so a workgroup fits on a CU BESIDE a 128 KB workgroup of the bulk trailing update running on the
// other stream (160 KB per CU, allocated in 1280-byte granules: 32 KB would miss by 768 bytes) instead of waiting
// ~60 us for one to retire; raised wave priority so its MFMAs issue first.
__global__ __launch_bounds__(GEMM_THREADS, 1) void mfgp_gemm_nt_f64_chain(const GemmTask* __restrict__ tasks, const double* A,
                                                                 const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    __builtin_amdgcn_s_setprio(3);
    gemm_nt_tile<64, 64, GW_M, GW_N, 1, 16>(tasks[blockIdx.x], A, B, C, C2, ld);
}
// 32x32 chain variant: 4 waves (2 x 2, one MFMA block each), 16 KB of LDS, K-steps of 32 -- the chain's panel / in-macro update
// launches at chain-bound sizes (planner: MFGP_CHAIN_TILE).  Those launches are latency-bound (a 64x64x128 tile is 128 MFMAs
// per SIMD behind eight serial K-steps); as 32x32 tiles the same work spreads over four times as many workgroups and four
// K-steps: panel launch 12 -> 9.5 us, in-macro update 13-20 -> 10-17 us (N = 4096), one evaluation at N = 2048 1.20 -> 1.03 ms.
// (Measured beside it: K-steps of 16, single- and double-buffered: 1.07 / 1.08 ms.)
__global__ __launch_bounds__(256, 5) void mfgp_gemm_nt_f64_chain32(const GemmTask* __restrict__ tasks, const double* A,
                                                          const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    __builtin_amdgcn_s_setprio(3);
    gemm_nt_tile<32, 32, 2, 2, 1, 32>(tasks[blockIdx.x], A, B, C, C2, ld);
}
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_kinv_syrk_f64(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<128, 128, 2, 2, 16, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_kinv_syrk_f64_t64(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<64, 64, 2, 2, 32, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_predvar_f64(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<128, 128, 2, 2, 16, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}
__global__ __launch_bounds__(BULK_THREADS, 2) void mfgp_predvar_f64_t64(const GemmTask* __restrict__ tasks, const double* A,
        const double* B, double* C, double* C2, int ld, long long bstride,
        const int* __restrict__ flag, int epoch) {
    if (flag && flag[blockIdx.y] == epoch) return;   // this evaluation has already failed (leaf_f64.hip): nothing downstream is used
    A += blockIdx.y * bstride; B += blockIdx.y * bstride; C += blockIdx.y * bstride; C2 += blockIdx.y * bstride;
    gemm_nt_dma<64, 64, 2, 2, 32, 2>(tasks[blockIdx.x], A, B, C, C2, ld);
}

// bulk kernels: two stages of (tile + tile) rows x 16 columns (128-tiles) or x 32 columns (64-tiles): 64 KB / 32 KB
size_t gemm_lds_bytes(int tile) { return (size_t)2 * (tile + tile) * (tile == 128 ? 16 : 32) * sizeof(double); }

typedef void (*gemm_kernel_t)(const GemmTask*, const double*, const double*, double*, double*, int, long long, const int*, int);

// -> 0, or -1 when the planner asked for a (tile, role) pair no kernel exists for (a planner bug: reported through the
// C-ABI's status like every other error, never by terminating the host process)
int launch_gemm(hipStream_t s, int tile, const GemmTask* tasks, int ntasks, const double* A,
                const double* B, double* C, double* C2, int ld, int role, int nbatch, long long bstride, const int* flag, int epoch) {
    if (ntasks <= 0) return 0;
    static const gemm_kernel_t table[3][2] = {{mfgp_gemm_nt_f64_t128, mfgp_gemm_nt_f64_t64},
                                              {mfgp_kinv_syrk_f64, mfgp_kinv_syrk_f64_t64},
                                              {mfgp_predvar_f64, mfgp_predvar_f64_t64}};
    // the opt-in to > 64 KB of dynamic LDS is a per-device attribute of the function; restart threads launch
    // concurrently (engine.py start_background_restarts), so the first launch on each device is serialised
    static std::once_flag attr_once[MFGP_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(attr_once[dev & (MFGP_MAX_DEVICES - 1)], [] {
        for (int r = 0; r < 3; ++r)
            for (int t = 0; t < 2; ++t)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(table[r][t]),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)gemm_lds_bytes(t == 0 ? 128 : 64));
    });
    const dim3 grid(ntasks, nbatch > 0 ? nbatch : 1);
    if (tile == 32) {                // chain step at chain-bound sizes: 32x32 tiles, 4 waves, 16 KB
        hipLaunchKernelGGL(mfgp_gemm_nt_f64_chain32, grid, dim3(256), (size_t)(32 + 32) * 32 * sizeof(double), s, tasks, A, B,
                           C, C2, ld, bstride, flag, epoch);
        return 0;
    }
    if (role == 3 && tile == 64) {   // serial-chain step: slim workgroups that co-reside with the bulk update
        hipLaunchKernelGGL(mfgp_gemm_nt_f64_chain, grid, dim3(GEMM_THREADS), (size_t)(64 + 64) * 16 * sizeof(double), s, tasks, A, B,
                           C, C2, ld, bstride, flag, epoch);
        return 0;
    }
    // roles: 0 bulk, 1 K^-1, 2 predictive variance (distinct symbols over one body); 3 = a chain step -- only its 64-tile
    // form is a kernel of its own (above), a 128-tile chain step runs the bulk kernel; 5 = the 32-tile chain step (handled
    // by tile == 32 above)
    if ((tile != 128 && tile != 64) || role < 0 || (role > 3 && role != 5)) return -1;
    const gemm_kernel_t k = table[role >= 3 ? 0 : role][tile == 128 ? 0 : 1];
    hipLaunchKernelGGL(k, grid, dim3(BULK_THREADS), gemm_lds_bytes(tile), s, tasks, A, B, C, C2, ld, bstride, flag, epoch);
    return 0;
}

}  // namespace mfgp
