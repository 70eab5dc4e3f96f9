// Developer probe (tools/probe_clocks.py): a BOUNDED background kernel that keeps a trickle of HBM reads going on its own
// stream while the solver runs, to test whether the in-loop SpMV's slow mode is a power-management state the memory system
// drops into during the few microseconds of near-idle between the small kernels of an Arnoldi step.  Every workgroup reads
// 1 KiB per iteration from a 256 MiB buffer (beyond L2, inside nothing anybody else uses) and sleeps in between; the loop
// count is fixed at launch, so the grid always drains.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/keepalive.hip -o /tmp/libkeepalive.so
#include <hip/hip_runtime.h>
#include <cstdio>
static uint4* g_buf = nullptr;
static unsigned* g_sink = nullptr;
static hipStream_t g_stream = nullptr;
static const size_t kBytes = 256ull << 20;
__global__ void keepalive_kernel(const uint4* buf, size_t n16, long iters, int sleep_reps, unsigned* sink) {
    unsigned acc = 0;
    size_t i = ((size_t)blockIdx.x * 977 * 64 + threadIdx.x) % n16;
    for (long it = 0; it < iters; ++it) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(buf + i));
        acc ^= v.x ^ v.w;
        i += (size_t)gridDim.x * 64 * 131;
        if (i >= n16) i -= n16;
        for (int s = 0; s < sleep_reps; ++s) __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}
extern "C" int keepalive_start(long iters, int blocks, int sleep_reps) {
    if (!g_buf) {
        if (hipMalloc(&g_buf, kBytes) != hipSuccess || hipMalloc(&g_sink, 64) != hipSuccess) return 1;
        if (hipMemset(g_buf, 1, kBytes) != hipSuccess) return 2;
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 3;
    }
    keepalive_kernel<<<blocks, 64, 0, g_stream>>>(g_buf, kBytes / 16, iters, sleep_reps, g_sink);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
extern "C" int keepalive_wait(void) { return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 1; }
