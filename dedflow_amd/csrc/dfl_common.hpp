// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../include/dedflow_kernels.h"

typedef dfl_index I;
typedef dfl_value T;

void dfl_record_error(hipError_t e, const char* file, int line);

#define DFL_GUARD(expr)                                        \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) dfl_record_error(_e, __FILE__, __LINE__); \
    } while (0)
#define DFL_LAUNCH_CHECK() DFL_GUARD(hipGetLastError())

static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

// sum over the 64 lanes of a wave; every lane returns the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// block-wide sum for 256-thread blocks; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* lds4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return r;
}
