// Micro-benchmark: what do the cross-stream hops of the partitioned Arnoldi step cost on MI355X, and which arrangement of
// the halo exchange / boundary rows around the interior SpMV is cheapest?  Kernels are stand-ins of the measured durations
// of rank 0's share of the 8-way 10M-tet partition (profiles/r02_rank_local_step_kernels.txt): update+PC 38 us, interior
// SpMV 77 us, boundary SpMV 9 us, dots stage 1 30 us, stage 2 6 us, pack / unpack 3 us each.
//   V0  one stream, no events (lower bound: no overlap needed when the exchange is empty)
//   V1  round 2: main A, rec e1, B, wait e2, C, D, E        side: wait e1, P, U, rec e2
//   V2  round 3: main A, rec e1, B, wait e2, D, E           side: wait e1, P, U, C, rec e2
//   V3  V2 with the "ready" event attached to kernel A's own dispatch (hipExtLaunchKernelGGL stop event) instead of a marker
//   V4  V2 with events created with hipEventDisableSystemFence off (default flags)
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/hop_cost tools/micro/hop_cost.hip && tools/micro/hop_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// streams `n` doubles `passes` times: duration ~ bytes / bandwidth, like the real kernels (memory-bound)
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
    // calibrate `passes` so that one pass-set lasts about the target (bandwidth ~5 TB/s: 7.26 MB per pass = 1.45 us)
    auto time_one = [&](K k) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int r = 0; r < 5; ++r) launch(k, m);
        hipEventRecord(a, m); for (int r = 0; r < 50; ++r) launch(k, m); hipEventRecord(b, m); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b);
        hipEventDestroy(a); hipEventDestroy(b); return 1e3f * t / 50; };
    K A{n, 24}, B{n, 56}, Cb{45000, 24}, D{n, 20}, E{2048, 8}, P{45000, 1}, U{45000, 1};
    printf("stand-alone (back to back, us): A %.1f  B %.1f  C %.1f  D %.1f  E %.1f  P %.1f\n", time_one(A), time_one(B), time_one(Cb), time_one(D), time_one(E), time_one(P));
    const int REP = 400;
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int variant = 0; variant <= 4; ++variant) {
        unsigned flags = hipEventDisableTiming | (variant == 4 ? 0u : (unsigned)hipEventDisableSystemFence);
        hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, flags)); CK(hipEventCreateWithFlags(&e2, flags));
        auto iter = [&]() -> int {
            if (variant == 0) { launch(A, m); launch(P, m); launch(U, m); launch(B, m); launch(Cb, m); launch(D, m); launch(E, m); return 0; }
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
