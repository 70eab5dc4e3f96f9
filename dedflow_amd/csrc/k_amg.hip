// Kernels of the two-level preconditioner (host/pc_twolevel.c): Galerkin coarse matrix over node aggregates with a
// piecewise-constant prolongation, restriction of a residual, prolongation of a coarse correction.
// Vectors keep the layout [u: Nx3 AoS | p: N | ...] on both levels; matrices are block CSR (4x4 per nodal nonzero).
// All sums run over host-built lists in a fixed order: bitwise reproducible.
#include "dfl_common.hpp"

namespace {

constexpr int BLK = 256;

// Ac[cz] = sum of the fine blocks A[f], f in idx[off[cz] .. off[cz+1]) (P^T A P for piecewise-constant P: plain block sums).
// One wave per coarse nonzero: 4 groups of 16 lanes (lane = block entry) take every 4th list item, combined by two shuffles.
__global__ __launch_bounds__(BLK) void galerkin_kernel(I nnzc, const I* __restrict__ off, const I* __restrict__ idx,
                                                      const T* __restrict__ vf, T* __restrict__ vc) {
    const long long w = ((long long)blockIdx.x * BLK + threadIdx.x) >> 6;
    if (w >= nnzc) return;
    const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
    double acc = 0.0;
    const int lo = off[w], hi = off[w + 1];
    for (int k = lo + g; k < hi; k += 4) acc += vf[(long long)idx[k] * 16 + e];
    acc += __shfl_xor(acc, 16, WAVE);
    acc += __shfl_xor(acc, 32, WAVE);
    if (g == 0) vc[w * 16 + e] = acc;
}

// rc[I] = sum of r over the nodes of aggregate I (node lists: anode[aoff[I] .. aoff[I+1])), 4 components.
// One wave per aggregate: 16 nodes x 4 components per trip.
// DIFF: the restricted vector is r - s (the residual r - A z with s = A z left by a plain matvec: the general
// y = b y + a A x form of the matvec is 0.70 ms at 10M tets against 0.59 ms, and the copy of r goes as well)
template <bool DIFF>
__global__ __launch_bounds__(BLK) void restrict_kernel(I Nc, const I* __restrict__ aoff, const I* __restrict__ anode, I N,
                                                      const T* __restrict__ r, const T* __restrict__ sub, T* __restrict__ rc) {
    const long long w = ((long long)blockIdx.x * BLK + threadIdx.x) >> 6;
    if (w >= Nc) return;
    const int lane = threadIdx.x & 63, c = lane & 3, g = lane >> 2;
    double acc = 0.0;
    const int lo = aoff[w], hi = aoff[w + 1];
    for (int k = lo + g; k < hi; k += 16) {
        const long long n = anode[k];
        const long long i = c < 3 ? 3 * n + c : 3LL * N + n;
        acc += DIFF ? r[i] - sub[i] : r[i];
    }
#pragma unroll
    for (int s = 4; s < 64; s <<= 1) acc += __shfl_xor(acc, s, WAVE);
    if (g == 0) rc[c < 3 ? 3 * w + c : 3LL * Nc + w] = acc;
}

// z += P xc: every node of rows [0, nrows) adds the correction of its aggregate (partitioned runs: the owned nodes)
__global__ __launch_bounds__(BLK) void prolong_add_kernel(I nrows, I N, const I* __restrict__ agg, I Nc, const T* __restrict__ xc,
                                                         T* __restrict__ z) {
    const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i >= nrows) return;
    const long long a = agg[i];
    z[3 * i] += xc[3 * a];
    z[3 * i + 1] += xc[3 * a + 1];
    z[3 * i + 2] += xc[3 * a + 2];
    z[3LL * N + i] += xc[3LL * Nc + a];
}

}  // namespace

extern "C" {

void dfl_amg_galerkin(I nnzc, const I* off, const I* idx, const T* val_fine, T* val_coarse, void* stream) {
    if (nnzc <= 0) return;
    galerkin_kernel<<<ceil_div((long long)nnzc * 64, BLK), BLK, 0, S(stream)>>>(nnzc, off, idx, val_fine, val_coarse);
    DFL_LAUNCH_CHECK();
}
void dfl_amg_restrict(I Nc, const I* aoff, const I* anode, I N, const T* r, T* rc, void* stream) {
    if (Nc <= 0) return;
    restrict_kernel<false><<<ceil_div((long long)Nc * 64, BLK), BLK, 0, S(stream)>>>(Nc, aoff, anode, N, r, nullptr, rc);
    DFL_LAUNCH_CHECK();
}
void dfl_amg_restrict_diff(I Nc, const I* aoff, const I* anode, I N, const T* r, const T* sub, T* rc, void* stream) {
    if (Nc <= 0) return;
    restrict_kernel<true><<<ceil_div((long long)Nc * 64, BLK), BLK, 0, S(stream)>>>(Nc, aoff, anode, N, r, sub, rc);
    DFL_LAUNCH_CHECK();
}
void dfl_amg_prolong_add_rows(I nrows, I N, const I* agg, I Nc, const T* xc, T* z, void* stream) {
    if (nrows <= 0) return;
    prolong_add_kernel<<<ceil_div(nrows, BLK), BLK, 0, S(stream)>>>(nrows, N, agg, Nc, xc, z);
    DFL_LAUNCH_CHECK();
}
void dfl_amg_prolong_add(I N, const I* agg, I Nc, const T* xc, T* z, void* stream) {
    dfl_amg_prolong_add_rows(N, N, agg, Nc, xc, z, stream);
}

}  // extern "C"
