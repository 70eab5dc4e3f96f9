// Micro-benchmark: does a hipGraph shrink the gaps between small dependent kernels on MI355X?
// chain of K dependent kernels (each streams n doubles), launched normally vs replayed from a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void axpy(int n, double a, const double* x, double* y) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a * x[i] + y[i];
}
int main(int argc, char** argv) {
    const int K = 8, REP = 200;
    for (int n : {1 << 12, 1 << 18, 1 << 22, 6912000}) {
        double *x, *y;
        CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
        CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
        hipStream_t s; CK(hipStreamCreate(&s));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto chain = [&]() { for (int k = 0; k < K; ++k) axpy<<<(n + 255) / 256, 256, 0, s>>>(n, 1.0, (k & 1) ? y : x, (k & 1) ? x : y); };
        for (int r = 0; r < 10; ++r) chain();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < REP; ++r) chain();
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float t_plain; CK(hipEventElapsedTime(&t_plain, a, b));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        chain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float t_graph; CK(hipEventElapsedTime(&t_graph, a, b));
        printf("n=%8d: %d-kernel chain  plain %.2f us per kernel   graph %.2f us per kernel\n", n, K, 1e3 * t_plain / (REP * K), 1e3 * t_graph / (REP * K));
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
