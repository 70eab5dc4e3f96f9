// Micro-benchmark: what do the cross-stream hops of the partitioned Arnoldi step cost on MI355X, and which arrangement of
// the halo exchange / boundary rows around the interior SpMV is cheapest?  Kernels are stand-ins of the measured durations
// of rank 0's share of the 8-way 10M-tet partition (profiles/r02_rank_local_step_kernels.txt): update+PC 38 us, interior
// SpMV 77 us, boundary SpMV 9 us, dots stage 1 30 us, stage 2 6 us, pack / unpack 3 us each.
//   V0  one stream, no events (lower bound: no overlap needed when the exchange is empty)
//   V1  round 2: main A, rec e1, B, wait e2, C, D, E        side: wait e1, P, U, rec e2
//   V2  round 3: main A, rec e1, B, wait e2, D, E           side: wait e1, P, U, C, rec e2
//   V3  V2 with the "ready" event attached to kernel A's own dispatch (hipExtLaunchKernelGGL stop event) instead of a marker
//   V4  V2 with events created with hipEventDisableSystemFence off (default flags)
//   V5  V2 with stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory) instead of events
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/hop_cost tools/micro/hop_cost.hip && tools/micro/hop_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// streams `n` doubles `passes` times: duration ~ bytes / bandwidth, like the real kernels (memory-bound)
__global__ void spin_kernel(int us) {
    const unsigned long long t0 = wall_clock64();
    int it = 0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull && it < (1 << 22)) { __builtin_amdgcn_s_sleep(32); ++it; }
}

__global__ void stream_kernel(int n, int passes, const double* __restrict__ x, double* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int p = 0; p < passes; ++p) acc += x[(size_t)p * n + i];
    y[i] = acc;
}

struct K { int n, passes; };
static double* X; static double* Y;
static void launch(const K& k, hipStream_t s) { stream_kernel<<<(k.n + 255) / 256, 256, 0, s>>>(k.n, k.passes, X, Y); }

int main() {
    const int n = 907000;            // 4 * 226744 owned entries
    const size_t big = (size_t)64 * n;
    CK(hipMalloc(&X, big * 8)); CK(hipMalloc(&Y, (size_t)n * 8 * 8));
    CK(hipMemset(X, 0, big * 8));
    hipStream_t m, sd;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    // (round 3, late: two streams may share ONE hardware queue -- then nothing overlaps and every variant measures the same
    // serial chain; take side streams until one overlaps, as host/comm_rccl.c::DflPickConcurrentStream does)
    {
        hipEvent_t em, es; CK(hipEventCreate(&em)); CK(hipEventCreate(&es));
        for (int tries = 0; tries < 6; ++tries) {
            spin_kernel<<<1, 64, 0, m>>>(300); CK(hipEventRecord(em, m));
            spin_kernel<<<1, 64, 0, sd>>>(1); CK(hipEventRecord(es, sd));
            CK(hipEventSynchronize(em)); CK(hipEventSynchronize(es));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, es, em));
            printf("side stream candidate %d: ended %.0f us before the 300-us wave on the main stream%s\n", tries, 1e3 * ms, ms > 0.1f ? " (overlaps)" : " (same queue?)");
            if (ms > 0.1f) break;
            CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));   // (the rejected stream is left alive on purpose)
        }
    }
    unsigned int* sig = nullptr;   // V5: stream memory operations instead of events
    bool have_sig = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) {   // no signal memory on this stack: pinned host memory is the other kind the value operations accept
        (void)hipGetLastError();
        have_sig = hipHostMalloc((void**)&sig, 64, hipHostMallocDefault) == hipSuccess;
        if (have_sig) { memset(sig, 0, 64); printf("V5 on pinned host memory (hipMallocSignalMemory refused)\n"); }
        else { (void)hipGetLastError(); printf("no signal memory: V5 skipped\n"); }
    } else CK(hipMemset(sig, 0, 64));
    unsigned int epoch = 0;
    // calibrate `passes` so that one pass-set lasts about the target (bandwidth ~5 TB/s: 7.26 MB per pass = 1.45 us)
    auto time_one = [&](K k) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int r = 0; r < 5; ++r) launch(k, m);
        hipEventRecord(a, m); for (int r = 0; r < 50; ++r) launch(k, m); hipEventRecord(b, m); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b);
        hipEventDestroy(a); hipEventDestroy(b); return 1e3f * t / 50; };
    K A{n, 24}, B{n, 56}, Cb{45000, 24}, D{n, 20}, E{2048, 8}, P{45000, 1}, U{45000, 1};
    printf("stand-alone (back to back, us): A %.1f  B %.1f  C %.1f  D %.1f  E %.1f  P %.1f\n", time_one(A), time_one(B), time_one(Cb), time_one(D), time_one(E), time_one(P));
    const int REP = 400;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int variant = 0; variant <= 5; ++variant) {
        if (variant == 5 && !have_sig) break;
        unsigned flags = hipEventDisableTiming | (variant == 4 ? 0u : (unsigned)hipEventDisableSystemFence);
        hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, flags)); CK(hipEventCreateWithFlags(&e2, flags));
        auto iter = [&]() -> int {
            if (variant == 0) { launch(A, m); launch(P, m); launch(U, m); launch(B, m); launch(Cb, m); launch(D, m); launch(E, m); return 0; }
            if (variant == 5) {   // V2's arrangement with hipStreamWriteValue32 / hipStreamWaitValue32 (>=) on signal memory
                ++epoch;
                launch(A, m);
                CK(hipStreamWriteValue32(m, sig, epoch, 0));
                CK(hipStreamWaitValue32(sd, sig, epoch, hipStreamWaitValueGte, 0xffffffffu));
                launch(P, sd); launch(U, sd);
                launch(B, m);
                launch(Cb, sd);
                CK(hipStreamWriteValue32(sd, sig + 8, epoch, 0));
                CK(hipStreamWaitValue32(m, sig + 8, epoch, hipStreamWaitValueGte, 0xffffffffu));
                launch(D, m); launch(E, m);
                return 0;
            }
            if (variant == 3) {
                hipExtLaunchKernelGGL(stream_kernel, dim3((A.n + 255) / 256), dim3(256), 0, m, nullptr, e1, 0, A.n, A.passes, (const double*)X, Y);
            } else {
                launch(A, m);
                CK(hipEventRecord(e1, m));
            }
            CK(hipStreamWaitEvent(sd, e1, 0));
            launch(P, sd); launch(U, sd);
            launch(B, m);
            if (variant == 1) {
                CK(hipEventRecord(e2, sd));
                CK(hipStreamWaitEvent(m, e2, 0));
                launch(Cb, m);
            } else {
                launch(Cb, sd);
                CK(hipEventRecord(e2, sd));
                CK(hipStreamWaitEvent(m, e2, 0));
            }
            launch(D, m); launch(E, m);
            return 0;
        };
        for (int r = 0; r < 20; ++r) if (iter()) return 1;
        CK(hipStreamSynchronize(m)); CK(hipStreamSynchronize(sd));
        CK(hipEventRecord(t0, m));
        for (int r = 0; r < REP; ++r) if (iter()) return 1;
        CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1));
        CK(hipStreamSynchronize(sd));
        float t; CK(hipEventElapsedTime(&t, t0, t1));
        printf("V%d: %.2f us per Arnoldi step\n", variant, 1e3 * t / REP);
        CK(hipEventDestroy(e1)); CK(hipEventDestroy(e2));
    }
    return 0;
}
