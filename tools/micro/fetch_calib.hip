// Calibration of rocprofv3's FETCH_SIZE on gfx950 per access width (MI355X_MICROARCH.md, HBM: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams the same 1 GiB buffer
// once (far beyond the 256 MiB Infinity Cache) with 4-, 8- or 16-byte loads per lane, or gathers 48-byte pieces of random
// 128-byte records the way phase 1 of the Jacobian kernel reads node records (16 B per lane, three loads per record).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
//   cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc --output-format csv -- /tmp/fetch_calib
// FETCH_SIZE (KB) x 1024 / bytes streamed = the factor to divide by.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void read_b32(size_t n, const unsigned* p, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b64(size_t n, const uint2* p, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint2 v = p[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_b128(size_t n, const uint4* p, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// every lane reads bytes 0..47 of one 128-byte record, records visited once each in a scrambled order
__global__ void gather_rec48(size_t nrec, const uint4* p, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = (i * 2654435761ull) % nrec;  // nrec is a power of two and the multiplier odd: a permutation
        const uint4* q = p + r * 8;
        uint4 a = q[0], b = q[1], c = q[2];
        acc ^= a.x ^ b.y ^ c.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30;
    void* buf; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        read_b32<<<grid, 256>>>(bytes / 4, (const unsigned*)buf, out);
        read_b64<<<grid, 256>>>(bytes / 8, (const uint2*)buf, out);
        read_b128<<<grid, 256>>>(bytes / 16, (const uint4*)buf, out);
        gather_rec48<<<grid, 256>>>(bytes / 128, (const uint4*)buf, out);
    }
    CK(hipDeviceSynchronize());
    printf("streamed %zu bytes per kernel (gather_rec48: 48 of every 128)\n", bytes);
    return 0;
}
