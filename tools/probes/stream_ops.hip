// Cost of in-stream synchronisation operations between two dependent tiny kernels (us per pair):
//   plain            k; k
//   record           k; hipEventRecord; k
//   wait(signalled)  k; hipStreamWaitEvent(on an event another stream recorded long ago); k
//   writeValue       k; hipStreamWriteValue64; k
//   waitValue        k; hipStreamWaitValue64 (already satisfied); k
// and the latency of a cross-stream hop A -> B: (record + waitEvent) against (writeValue + waitValue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    // (hipMallocSignalMemory wants exactly 8 bytes per allocation; plain device memory is tried when it is refused)
    uint64_t* sig = nullptr; uint64_t* sig2 = nullptr;
    hipError_t es = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory);
    if (es == hipSuccess) es = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig2), 8, hipMallocSignalMemory);
    if (es != hipSuccess) {
        printf("signal memory: %s -- using hipMalloc\n", hipGetErrorString(es));
        CK(hipMalloc(&sig, 8)); CK(hipMalloc(&sig2, 8));
    }
    CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
    hipEvent_t t0, t1, ev, old;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&old, hipEventDisableTiming));
    CK(hipEventRecord(old, b)); CK(hipStreamSynchronize(b));
    const int R = 200;
    uint64_t epoch = 0;
    for (int mode = 0; mode < 5; ++mode) {
        if (mode >= 3 && !sig) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(t0, a));
            for (int i = 0; i < R; ++i) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
                if (mode == 1) CK(hipEventRecord(ev, a));
                if (mode == 2) CK(hipStreamWaitEvent(a, old, 0));
                if (mode == 3) CK(hipStreamWriteValue64(a, sig, ++epoch, 0));
                if (mode == 4) CK(hipStreamWaitValue64(a, sig, 0, hipStreamWaitValueGte, ~0ull));
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
            }
            CK(hipEventRecord(t1, a)); CK(hipStreamSynchronize(a));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            static const char* names[] = {"plain", "record", "wait(signalled)", "writeValue", "waitValue(satisfied)"};
            if (rep == 1) printf("%-22s %.2f us per pair\n", names[mode], ms * 1e3 / R);
        }
    }
    // cross-stream hop: a: k, signal ; b: wait, k ; a waits for b's completion through the same mechanism
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1 && !sig) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(t0, a));
            for (int i = 0; i < R; ++i) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
                if (mode == 0) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
                else { CK(hipStreamWriteValue64(a, sig, ++epoch, 0)); CK(hipStreamWaitValue64(b, sig, epoch, hipStreamWaitValueGte, ~0ull)); }
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
                if (mode == 0) { CK(hipEventRecord(old, b)); CK(hipStreamWaitEvent(a, old, 0)); }
                else { CK(hipStreamWriteValue64(b, sig2, epoch, 0)); CK(hipStreamWaitValue64(a, sig2, epoch, hipStreamWaitValueGte, ~0ull)); }
            }
            CK(hipEventRecord(t1, a)); CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1) printf("%-22s %.2f us per round trip a -> b -> a (two kernels, two hops)\n", mode == 0 ? "events" : "stream values", ms * 1e3 / R);
        }
    }
    printf("done\n");
    return 0;
}
