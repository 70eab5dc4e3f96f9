// Multicolor block-DILU preconditioner on the (u,p) block-CSR matrix ("ILU0-style" PC of BASELINE
// config 5 / SURVEY 8(f)-4; the reference only has the PC_ILU0 enum value, pc.h, no implementation).
//   M = (E + L) E^-1 (E + U),   E_i = A_ii - sum_{j ~ i, color(j) < color(i)} A_ij E_j^-1 A_ji
// with L / U the blocks below / above in the node-color order.  Nodes of one color are mutually
// non-adjacent, so a color is one launch; the two triangular sweeps together read every off-diagonal
// 128-byte block line once (about the traffic of one SpMV) plus E^-1.
// Mapping as in the SpMV (k_matrix.hip): 8 lanes per node row, lane l owns block entries (2l, 2l+1).
#include "dfl_common.hpp"

namespace {

constexpr int BLK = 256;
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long xidx(int col, int c, long long N3) { return c < 3 ? 3LL * col + c : N3 + col; }

// C = A * B, 4x4 row-major
__device__ __forceinline__ void mm4(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}

// in-place 4x4 inverse, Gauss-Jordan with partial pivoting; static indexing only (stays in registers)
__device__ __forceinline__ void inv4(double* a) {
    double b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = (i % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {  // bring the largest |a[r][c]| of rows c.. up to row c
            if (fabs(a[r * 4 + c]) > fabs(a[c * 4 + c])) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double t = a[r * 4 + k]; a[r * 4 + k] = a[c * 4 + k]; a[c * 4 + k] = t;
                    t = b[r * 4 + k]; b[r * 4 + k] = b[c * 4 + k]; b[c * 4 + k] = t;
                }
            }
        }
        const double ip = 1.0 / a[c * 4 + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[c * 4 + k] *= ip; b[c * 4 + k] *= ip; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r * 4 + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[r * 4 + k] -= f * a[c * 4 + k]; b[r * 4 + k] -= f * b[c * 4 + k]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = b[i];
}

__device__ __forceinline__ int find_col(const I* __restrict__ ci, int lo, int hi, int col) {  // ci ascending in [lo, hi)
    --hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ci[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// E^-1 of the rows of one color (all lower colors done): one thread per row
__global__ __launch_bounds__(BLK) void dilu_setup_kernel(I nrows_c, const I* __restrict__ rows, I nown, const I* __restrict__ rp,
                                                        const I* __restrict__ ci, const T* __restrict__ val,
                                                        const unsigned char* __restrict__ color, T* __restrict__ Einv) {
    const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
    if (t >= nrows_c) return;
    const int row = rows[t];
    const int cr = color[row];
    double E[16];
    const int s = rp[row], e = rp[row + 1];
    const int kd = find_col(ci, s, e, row);
#pragma unroll
    for (int i = 0; i < 16; ++i) E[i] = val[(long long)kd * 16 + i];
    for (int k = s; k < e; ++k) {
        const int j = ci[k];
        if (j >= nown || color[j] >= cr) continue;
        const int kk = find_col(ci, rp[j], rp[j + 1], row);  // symmetric pattern: (j, row) exists
        double Aij[16], Ej[16], Aji[16], P[16], Q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            Aij[i] = val[(long long)k * 16 + i];
            Ej[i] = Einv[(long long)j * 16 + i];
            Aji[i] = val[(long long)kk * 16 + i];
        }
        mm4(Aij, Ej, P);
        mm4(P, Aji, Q);
#pragma unroll
        for (int i = 0; i < 16; ++i) E[i] -= Q[i];
    }
    inv4(E);
#pragma unroll
    for (int i = 0; i < 16; ++i) Einv[(long long)row * 16 + i] = E[i];
}

// one color of a triangular sweep.  FWD: z_i = E_i^-1 (r_i - sum_{color(j)<color(i)} A_ij z_j)
//                                   BWD: z_i -= E_i^-1 sum_{color(j)>color(i)} A_ij z_j
// The L / U neighbours of every row are precomputed lists (eptr over the color-ordered row slots, enz = nodal nonzero,
// ecol = column node), so the loop carries no color lookup and no predicate and issues two independent block loads per trip.
// VT = float: the off-diagonal blocks come from the single-precision copy PC_TWOLEVEL keeps for its smoother (E^-1 and all
// sums stay double)
template <bool FWD, typename VT = T>
__global__ __launch_bounds__(BLK) void dilu_sweep_kernel(I slot0, I nrows_c, const I* __restrict__ rows, I N,
                                                        const I* __restrict__ eptr, const I* __restrict__ enz,
                                                        const I* __restrict__ ecol, const VT* __restrict__ val,
                                                        const T* __restrict__ Einv, const T* __restrict__ r, T* __restrict__ z) {
    const long long gid = (long long)blockIdx.x * BLK + threadIdx.x;
    const long long lslot = gid >> 3;
    const int l = threadIdx.x & 7;
    if (lslot >= nrows_c) return;  // whole 8-lane groups leave together
    const long long slot = slot0 + lslot;
    const int row = rows[slot];
    const long long N3 = 3LL * N;
    const int br = l >> 1;
    const bool hi = (l & 1);
    typedef VT v2t __attribute__((ext_vector_type(2)));
    const v2t* __restrict__ v2 = reinterpret_cast<const v2t*>(val) + l;
    double acc0 = 0.0, acc1 = 0.0;
    int q = eptr[slot];
    const int qe = eptr[slot + 1];
    // (a triangle of a nodal row holds ~7 neighbours: four block lines in flight first, then the tail in twos and ones; the
    // association of the sums is that of the two-at-a-time loop: entries q, q+2, ... into acc0, the others into acc1)
    for (; q + 4 <= qe; q += 4) {
        const int k0 = enz[q], k1 = enz[q + 1], k2 = enz[q + 2], k3 = enz[q + 3];
        const int c0 = ecol[q], c1 = ecol[q + 1], c2 = ecol[q + 2], c3 = ecol[q + 3];
        const v2t a0 = v2[(long long)k0 * 8], a1 = v2[(long long)k1 * 8], a2 = v2[(long long)k2 * 8], a3 = v2[(long long)k3 * 8];
        const double x0a = z[hi ? 3LL * c0 + 2 : 3LL * c0], x0b = z[hi ? N3 + c0 : 3LL * c0 + 1];
        const double x1a = z[hi ? 3LL * c1 + 2 : 3LL * c1], x1b = z[hi ? N3 + c1 : 3LL * c1 + 1];
        const double x2a = z[hi ? 3LL * c2 + 2 : 3LL * c2], x2b = z[hi ? N3 + c2 : 3LL * c2 + 1];
        const double x3a = z[hi ? 3LL * c3 + 2 : 3LL * c3], x3b = z[hi ? N3 + c3 : 3LL * c3 + 1];
        acc0 += (double)a0.x * x0a + (double)a0.y * x0b;
        acc1 += (double)a1.x * x1a + (double)a1.y * x1b;
        acc0 += (double)a2.x * x2a + (double)a2.y * x2b;
        acc1 += (double)a3.x * x3a + (double)a3.y * x3b;
    }
    for (; q + 2 <= qe; q += 2) {
        const int k0 = enz[q], k1 = enz[q + 1], c0 = ecol[q], c1 = ecol[q + 1];
        const v2t a0 = v2[(long long)k0 * 8], a1 = v2[(long long)k1 * 8];
        const double x0a = z[hi ? 3LL * c0 + 2 : 3LL * c0], x0b = z[hi ? N3 + c0 : 3LL * c0 + 1];
        const double x1a = z[hi ? 3LL * c1 + 2 : 3LL * c1], x1b = z[hi ? N3 + c1 : 3LL * c1 + 1];
        acc0 += (double)a0.x * x0a + (double)a0.y * x0b;
        acc1 += (double)a1.x * x1a + (double)a1.y * x1b;
    }
    if (q < qe) {
        const int k0 = enz[q], c0 = ecol[q];
        const v2t a0 = v2[(long long)k0 * 8];
        acc0 += (double)a0.x * z[hi ? 3LL * c0 + 2 : 3LL * c0] + (double)a0.y * z[hi ? N3 + c0 : 3LL * c0 + 1];
    }
    double acc = acc0 + acc1;
    acc += __shfl_xor(acc, 1, WAVE);  // both lanes of block row `br` hold the row sum
    const long long yi = xidx(row, br, N3);
    const double t = FWD ? r[yi] - acc : acc;
    const int base = (threadIdx.x & (WAVE - 1)) & ~7;
    const double t0 = __shfl(t, base + (hi ? 4 : 0), WAVE);  // component 2*hi   lives on lanes 4*hi, 4*hi+1
    const double t1 = __shfl(t, base + (hi ? 6 : 2), WAVE);  // component 2*hi+1 lives on lanes 4*hi+2, ...
    const d2v ev = reinterpret_cast<const d2v*>(Einv + (long long)row * 16)[l];
    double part = ev.x * t0 + ev.y * t1;
    part += __shfl_xor(part, 1, WAVE);
    if (!hi) {
        if (FWD) z[yi] = part;
        else z[yi] -= part;
    }
}

__global__ __launch_bounds__(BLK) void copy_tail_kernel(long long begin, long long end, const T* __restrict__ x, T* __restrict__ y) {
    const long long i = begin + (long long)blockIdx.x * BLK + threadIdx.x;
    if (i < end) y[i] = x[i];
}

}  // namespace

extern "C" {

void dfl_dilu_setup_color(I nrows_c, const I* rows, I nown, const I* rp, const I* ci, const T* val, const unsigned char* color,
                          T* Einv, void* stream) {
    if (nrows_c <= 0) return;
    dilu_setup_kernel<<<ceil_div(nrows_c, BLK), BLK, 0, S(stream)>>>(nrows_c, rows, nown, rp, ci, val, color, Einv);
    DFL_LAUNCH_CHECK();
}

void dfl_dilu_sweep_color(int forward, I slot0, I nrows_c, const I* rows, I N, const I* eptr, const I* enz, const I* ecol,
                          const T* val, const T* Einv, const T* r, T* z, void* stream) {
    if (nrows_c <= 0) return;
    const int grid = ceil_div((long long)nrows_c * 8, BLK);
    if (forward) dilu_sweep_kernel<true><<<grid, BLK, 0, S(stream)>>>(slot0, nrows_c, rows, N, eptr, enz, ecol, val, Einv, r, z);
    else dilu_sweep_kernel<false><<<grid, BLK, 0, S(stream)>>>(slot0, nrows_c, rows, N, eptr, enz, ecol, val, Einv, r, z);
    DFL_LAUNCH_CHECK();
}

void dfl_dilu_sweep_color_f32(int forward, I slot0, I nrows_c, const I* rows, I N, const I* eptr, const I* enz, const I* ecol,
                              const float* valf, const T* Einv, const T* r, T* z, void* stream) {
    if (nrows_c <= 0) return;
    const int grid = ceil_div((long long)nrows_c * 8, BLK);
    if (forward) dilu_sweep_kernel<true, float><<<grid, BLK, 0, S(stream)>>>(slot0, nrows_c, rows, N, eptr, enz, ecol, valf, Einv, r, z);
    else dilu_sweep_kernel<false, float><<<grid, BLK, 0, S(stream)>>>(slot0, nrows_c, rows, N, eptr, enz, ecol, valf, Einv, r, z);
    DFL_LAUNCH_CHECK();
}

void dfl_copy_range(int64_t begin, int64_t end, const T* x, T* y, void* stream) {
    if (end <= begin) return;
    copy_tail_kernel<<<ceil_div(end - begin, BLK), BLK, 0, S(stream)>>>(begin, end, x, y);
    DFL_LAUNCH_CHECK();
}

}  // extern "C"
