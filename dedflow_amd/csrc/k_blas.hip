// BLAS-1, deterministic reductions, fused classical Gram-Schmidt and the GMRES
// small recurrences -- replaces the cuBLAS calls of src/krylov.c:114-319,
// src/main.c:107-130 and the elementwise kernels of src/vec.cu:14-76.
// All of these are HBM-bound streams: 16-byte accesses per lane, grid sized to
// a few blocks per CU, two-stage (order-fixed) reductions instead of atomics so
// that Krylov residual histories are bitwise reproducible run to run.
#include "dfl_common.hpp"
#include <cmath>

static char g_err[512] = "";
void dfl_record_error(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof g_err, "HIP error: %s at %s:%d", hipGetErrorString(e), file, line);
    fprintf(stderr, "GPUAssert: %s\n", g_err);
}

namespace {

constexpr int BLK = 256;
constexpr int MAX_PART = 1024;  // stage-1 partials of a reduction

template <class F>
__global__ __launch_bounds__(BLK) void map1(I n, T* x, F f) {
    long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    long long n2 = n >> 1;
    if (i < n2) {
        double2 v = reinterpret_cast<double2*>(x)[i];
        v.x = f(v.x);
        v.y = f(v.y);
        reinterpret_cast<double2*>(x)[i] = v;
    }
    if (i == 0 && (n & 1)) x[n - 1] = f(x[n - 1]);
}

template <class F>
__global__ __launch_bounds__(BLK) void map3(I n, const T* a, const T* b, T* c, F f) {
    long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    long long n2 = n >> 1;
    if (i < n2) {
        double2 u = reinterpret_cast<const double2*>(a)[i];
        double2 v = reinterpret_cast<const double2*>(b)[i];
        double2 r;
        r.x = f(u.x, v.x);
        r.y = f(u.y, v.y);
        reinterpret_cast<double2*>(c)[i] = r;
    }
    if (i == 0 && (n & 1)) c[n - 1] = f(a[n - 1], b[n - 1]);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class F>
__global__ __launch_bounds__(BLK) void map1_scalar(I n, T* x, F f) {
    long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i < n) x[i] = f(x[i]);
}
template <class F>
__global__ __launch_bounds__(BLK) void map3_scalar(I n, const T* a, const T* b, T* c, F f) {
    long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i < n) c[i] = f(a[i], b[i]);
}

template <class F>
void launch_map1(I n, T* x, F f, void* stream) {
    if (n <= 0) return;
    if (aligned16(x)) map1<<<ceil_div((n >> 1) + 1, BLK), BLK, 0, S(stream)>>>(n, x, f);
    else map1_scalar<<<ceil_div(n, BLK), BLK, 0, S(stream)>>>(n, x, f);
    DFL_LAUNCH_CHECK();
}
template <class F>
void launch_map3(I n, const T* a, const T* b, T* c, F f, void* stream) {
    if (n <= 0) return;
    if (aligned16(a) && aligned16(b) && aligned16(c)) map3<<<ceil_div((n >> 1) + 1, BLK), BLK, 0, S(stream)>>>(n, a, b, c, f);
    else map3_scalar<<<ceil_div(n, BLK), BLK, 0, S(stream)>>>(n, a, b, c, f);
    DFL_LAUNCH_CHECK();
}

// ---- reductions ----------------------------------------------------------------
template <bool DOT>
__global__ __launch_bounds__(BLK) void reduce_stage1(I n, const T* x, const T* y, T* part) {
    __shared__ double lds[4];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * BLK;
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        double a = x[i];
        acc += DOT ? a * y[i] : a * a;
    }
    double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

template <bool SQRT>
__global__ __launch_bounds__(BLK) void reduce_stage2(int npart, const T* part, T* out) {
    __shared__ double lds[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npart; i += BLK) acc += part[i];
    double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[0] = SQRT ? sqrt(r) : r;
}

__global__ __launch_bounds__(BLK) void scal_inv_dev(I n, const T* d_scale, T* x) {
    const double s = 1.0 / d_scale[0];
    long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    long long n2 = n >> 1;
    if (i < n2) {
        double2 v = reinterpret_cast<double2*>(x)[i];
        v.x *= s;
        v.y *= s;
        reinterpret_cast<double2*>(x)[i] = v;
    }
    if (i == 0 && (n & 1)) x[n - 1] *= s;
}

// Krylov basis columns are streamed once per pass (2.3 GB at 10M tets, m = 40): nontemporal
// loads keep them from displacing w and the partial sums in L2 / MALL.
typedef double d2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const T* p) {
    const d2s v = __builtin_nontemporal_load(reinterpret_cast<const d2s*>(p));
    return make_double2(v.x, v.y);
}

// ---- fused CGS -------------------------------------------------------------------
// d_h[j] = Q[:,j].w : grid (row blocks, column tiles of CT); each thread owns
// RPT double2 row slots, keeps w in registers across the CT columns of its tile.
constexpr int CT = 8;
constexpr int RPT = 4;                      // double2 slots per thread
constexpr int ROWS_PER_BLOCK = BLK * RPT * 2;  // 2048 rows

template <int MODE = 0>  // bit 1 = plain loads (developer A/B)
__global__ __launch_bounds__(BLK) void cgs_dots_stage1(I n, I ncol, const T* __restrict__ Q, long long ldq,
                                                      const T* __restrict__ w, T* __restrict__ part, int nrb, int ct) {
    __shared__ double lds[4];
    const long long r0 = (long long)blockIdx.x * ROWS_PER_BLOCK;
    const int c0 = blockIdx.y * ct;
    double2 wv[RPT];
    long long row[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        row[i] = r0 + 2LL * threadIdx.x + (long long)i * (2 * BLK);
        if (row[i] + 1 < n) wv[i] = *reinterpret_cast<const double2*>(w + row[i]);
        else { wv[i].x = (row[i] < n) ? w[row[i]] : 0.0; wv[i].y = 0.0; }
    }
    for (int c = 0; c < ct; ++c) {
        const int col = c0 + c;
        if (col >= ncol) break;  // uniform
        const T* q = Q + (long long)col * ldq;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            double2 qv;
            if (row[i] + 1 < n) qv = (MODE & 2) ? *reinterpret_cast<const double2*>(q + row[i]) : ld_stream(q + row[i]);
            else { qv.x = (row[i] < n) ? q[row[i]] : 0.0; qv.y = 0.0; }
            acc += qv.x * wv[i].x + qv.y * wv[i].y;
        }
        double r = block_sum_256(acc, lds);
        if (threadIdx.x == 0) part[(long long)col * nrb + blockIdx.x] = r;
    }
}

// The same partial sums with ALL loads of a column tile issued before the first reduction: the kernel above runs a
// workgroup reduction (two barriers) after every column, so only the 4 loads of one column are in flight per thread; here
// the CT * RPT loads of the tile are, and one barrier serves the eight sums.  Same additions in the same order per column:
// bitwise the same partial sums.
template <int MODE = 0>
__global__ __launch_bounds__(BLK) void cgs_dots_stage1_tile(I n, I ncol, const T* __restrict__ Q, long long ldq,
                                                           const T* __restrict__ w, T* __restrict__ part, int nrb) {
    __shared__ double lds8[4][CT];
    const long long r0 = (long long)blockIdx.x * ROWS_PER_BLOCK;
    const int c0 = blockIdx.y * CT;
    double2 wv[RPT];
    long long row[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        row[i] = r0 + 2LL * threadIdx.x + (long long)i * (2 * BLK);
        if (row[i] + 1 < n) wv[i] = *reinterpret_cast<const double2*>(w + row[i]);
        else { wv[i].x = (row[i] < n) ? w[row[i]] : 0.0; wv[i].y = 0.0; }
    }
    double acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        acc[c] = 0.0;
        const int col = c0 + c;
        if (col < ncol) {  // uniform
            const T* q = Q + (long long)col * ldq;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                double2 qv;
                if (row[i] + 1 < n) qv = (MODE & 2) ? *reinterpret_cast<const double2*>(q + row[i]) : ld_stream(q + row[i]);
                else { qv.x = (row[i] < n) ? q[row[i]] : 0.0; qv.y = 0.0; }
                acc[c] += qv.x * wv[i].x + qv.y * wv[i].y;
            }
        }
    }
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const double v = wave_sum(acc[c]);
        if (lane == 0) lds8[wv_id][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < CT && c0 + (int)threadIdx.x < ncol)
        part[(long long)(c0 + threadIdx.x) * nrb + blockIdx.x] =
            (lds8[0][threadIdx.x] + lds8[1][threadIdx.x]) + (lds8[2][threadIdx.x] + lds8[3][threadIdx.x]);
}

__global__ __launch_bounds__(BLK) void cgs_dots_stage2(int nrb, const T* part, T* d_h) {
    __shared__ double lds[4];
    const T* p = part + (long long)blockIdx.x * nrb;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nrb; i += BLK) acc += p[i];
    double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) d_h[blockIdx.x] = r;
}

// w -= Q h (SUB) or y = Q c (!SUB); optional partial ||w||^2 per block
constexpr int UPT = 2;  // double2 slots per thread (1 and 4 measured: 0.2093 against 0.2028 ms per CGS kernel)
constexpr int UROWS = BLK * UPT * 2;
template <bool SUB, int MODE = 0>  // MODE (developer A/B, DFL_CGS_MODE): bit 0 = columns walked last to first, bit 1 = plain loads
__global__ __launch_bounds__(BLK) void cgs_update_kernel(I n, I ncol, const T* __restrict__ Q, long long ldq,
                                                        const T* __restrict__ d_h, T* __restrict__ w, T* __restrict__ part) {
    __shared__ double lds[4];
    __shared__ double sh[128];
    for (int j = threadIdx.x; j < ncol && j < 128; j += BLK) sh[j] = d_h[j];
    __syncthreads();
    long long blk = blockIdx.x;
    if (MODE & 4) {  // developer A/B: one contiguous slab of row chunks per XCD (workgroup b runs on XCD b % 8), as the SpMV
        const long long per = (gridDim.x + 7) >> 3;
        blk = (blk & 7) * per + (blk >> 3);
    }
    const long long r0 = blk * UROWS;
    long long row[UPT];
    double2 acc[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        row[i] = r0 + 2LL * threadIdx.x + (long long)i * (2 * BLK);
        if (SUB) {
            if (row[i] + 1 < n) acc[i] = *reinterpret_cast<const double2*>(w + row[i]);
            else { acc[i].x = (row[i] < n) ? w[row[i]] : 0.0; acc[i].y = 0.0; }
        } else { acc[i].x = 0.0; acc[i].y = 0.0; }
    }
#pragma unroll 4  // (8 columns in flight: 0.212 against 0.205 ms per CGS kernel -- slower)
    for (int jj = 0; jj < ncol; ++jj) {
        const int j = (MODE & 1) ? ncol - 1 - jj : jj;
        const double h = (j < 128) ? sh[j] : d_h[j];
        const T* q = Q + (long long)j * ldq;
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            double2 qv;
            if (row[i] + 1 < n) qv = (MODE & 2) ? *reinterpret_cast<const double2*>(q + row[i]) : ld_stream(q + row[i]);
            else { qv.x = (row[i] < n) ? q[row[i]] : 0.0; qv.y = 0.0; }
            if (SUB) { acc[i].x -= qv.x * h; acc[i].y -= qv.y * h; }
            else { acc[i].x += qv.x * h; acc[i].y += qv.y * h; }
        }
    }
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        if (row[i] + 1 < n) *reinterpret_cast<double2*>(w + row[i]) = acc[i];
        else if (row[i] < n) w[row[i]] = acc[i].x;
        ss += acc[i].x * acc[i].x + acc[i].y * acc[i].y;
    }
    if (part) {
        double r = block_sum_256(ss, lds);
        if (threadIdx.x == 0 && blk < gridDim.x) part[blk] = r;  // (slab order: the partial of row chunk blk, as without the remap)
    }
}

// ---- GMRES recurrences (single lane; O(k) flops) ----------------------------------
__device__ void drotg_dev(double* a, double* b, double* c, double* s) {
    double roe = *b;
    if (fabs(*a) > fabs(*b)) roe = *a;
    double scale = fabs(*a) + fabs(*b), r, z;
    if (scale == 0.0) { *c = 1.0; *s = 0.0; r = 0.0; z = 0.0; }
    else {
        double ta = *a / scale, tb = *b / scale;
        r = scale * sqrt(ta * ta + tb * tb);
        r = (roe < 0.0 ? -1.0 : 1.0) * r;
        *c = *a / r; *s = *b / r; z = 1.0;
        if (fabs(*a) > fabs(*b)) z = *s;
        if (fabs(*b) >= fabs(*a) && *c != 0.0) z = 1.0 / *c;
    }
    *a = r; *b = z;
}

// The Givens step of column `iter` (krylov.c:256-277, krylov_util.cu:5-19), called by a whole workgroup: the column and
// the earlier rotations are staged in LDS with coalesced loads first (a single thread walking them in global memory
// pays one dependent load latency per entry), thread 0 runs the O(iter) recurrence on the staged copy, the column is
// written back by all threads.  `nrm` = ||w||.
constexpr int GIV_MAX = 1024;  // columns longer than this fall back to the in-place walk
__device__ void givens_step_block(I iter, double nrm, T* H, I ldh, T* gv, T* beta, T* res_hist, double* s_col, double* s_gv) {
    T* col = H + (long long)iter * ldh;
    const int t = threadIdx.x, nt = blockDim.x;
    const bool staged = iter + 2 <= GIV_MAX;
    if (staged) {
        for (int i = t; i <= iter; i += nt) s_col[i] = col[i];
        for (int i = t; i < 2 * iter; i += nt) s_gv[i] = gv[i];
        __syncthreads();
    }
    if (t == 0) {
        double* c_ = staged ? s_col : col;
        const double* g_ = staged ? s_gv : gv;
        c_[iter + 1] = nrm;  // H[iter+1, iter] = ||w||, krylov.c:228-230
        for (I i = 0; i < iter; ++i) {  // cublasDrot(n=1), krylov.c:258-263
            const double c = g_[2 * i], s = g_[2 * i + 1];
            const double x = c_[i], y = c_[i + 1];
            c_[i] = c * x + s * y;
            c_[i + 1] = c * y - s * x;
        }
        double gc, gs;
        drotg_dev(&c_[iter], &c_[iter + 1], &gc, &gs);  // :266
        gv[2 * iter] = gc;
        gv[2 * iter + 1] = gs;
        c_[iter + 1] = 0.0;  // :267
        const double b0 = beta[iter];  // krylov_util.cu:5-19
        beta[iter + 1] = -gs * b0;
        beta[iter] = b0 * gc;
        if (res_hist) res_hist[iter] = fabs(beta[iter + 1]);
    }
    if (staged) {
        __syncthreads();
        for (int i = t; i <= iter + 1; i += nt) col[i] = s_col[i];
    }
}

// SQUARED: *d_nrm holds the (all-reduced) squared norm; it is replaced by its square root first
template <bool SQUARED>
__global__ __launch_bounds__(BLK) void gmres_givens_kernel(I iter, T* d_nrm, T* H, I ldh, T* gv, T* beta, T* res_hist) {
    __shared__ double s_col[GIV_MAX], s_gv[2 * GIV_MAX];
    __shared__ double s_nrm;
    if (threadIdx.x == 0) {
        s_nrm = SQUARED ? sqrt(d_nrm[0]) : d_nrm[0];
        if (SQUARED) d_nrm[0] = s_nrm;
    }
    __syncthreads();
    givens_step_block(iter, s_nrm, H, ldh, gv, beta, res_hist, s_col, s_gv);
}

// Partitioned runs, fused-norm option: column `iter` of H holds the all-reduced h_0..h_iter and, in slot iter+1, the
// all-reduced w.w from the SAME reduction; ||w - Q h||^2 = w.w - sum h_j^2 (Q orthonormal), so the second all-reduce of an
// Arnoldi step disappears.  Cancellation makes this unsafe when ||w - Qh|| << ||w||: *d_flag is raised when less than
// 1e-6 of w.w is left (the caller then knows the history is unreliable and can switch the option off).
__global__ __launch_bounds__(BLK) void gmres_givens_pythagoras_kernel(I iter, T* d_nrm, T* H, I ldh, T* gv, T* beta, T* res_hist,
                                                                     int* d_flag) {
    __shared__ double s_col[GIV_MAX], s_gv[2 * GIV_MAX];
    __shared__ double s_nrm;
    if (threadIdx.x == 0) {
        const T* col = H + (long long)iter * ldh;
        const double ww = col[iter + 1];
        double hh = 0.0;
        for (I j = 0; j <= iter; ++j) hh += col[j] * col[j];
        double r = ww - hh;
        if (r < 1e-6 * ww) {
            if (d_flag) *d_flag = 1;
            if (r < 0.0) r = 0.0;
        }
        s_nrm = sqrt(r);
        d_nrm[0] = s_nrm;
    }
    __syncthreads();
    givens_step_block(iter, s_nrm, H, ldh, gv, beta, res_hist, s_col, s_gv);
}

// second stage of ||w||^2 (partials of cgs_update_kernel, same fixed order as reduce_stage2) + square root + the Givens
// step in one launch: one kernel boundary less per Arnoldi step
__global__ __launch_bounds__(BLK) void norm_givens_kernel(int npart, const T* part, T* d_nrm, I iter, T* H, I ldh, T* gv, T* beta,
                                                         T* res_hist) {
    __shared__ double lds[4];
    __shared__ double s_nrm;
    __shared__ double s_col[GIV_MAX], s_gv[2 * GIV_MAX];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npart; i += BLK) acc += part[i];
    const double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) {
        s_nrm = sqrt(r);
        d_nrm[0] = s_nrm;
    }
    __syncthreads();
    givens_step_block(iter, s_nrm, H, ldh, gv, beta, res_hist, s_col, s_gv);
}

// Partitioned runs with the fused norm and the Jacobi tree: the raw column hraw = [h_0 .. h_iter, w.w] is complete (all-
// reduced) BEFORE the update, so ||w - Q h||^2 = w.w - sum h_j^2 is known up front and three launches collapse into one:
//   cgs_update (w -= Q h)  +  gmres_givens_pythagoras (block 0, on a copy of the column in H)  +  the preconditioner
//   application that opens the NEXT Arnoldi step (q = w / nrm stored as the new basis column, z = M^-1 q).
// One owned node per thread (its 3 velocity rows and its pressure row); every block recomputes the norm from hraw in the
// order of gmres_givens_pythagoras_kernel, so all blocks scale by the same bits.  Ghost rows are never touched (zero).
__global__ __launch_bounds__(BLK) void cgs_update_pc_kernel(I nrows, I N, I ncol, const T* __restrict__ Q, long long ldq,
                                                           const T* __restrict__ hraw, T* __restrict__ w,
                                                           const T* __restrict__ dinv33, const T* __restrict__ dinv1,
                                                           T* __restrict__ z, I iter, T* H, I ldh, T* gv, T* beta, T* res_hist,
                                                           T* d_nrm, int* d_flag, T* __restrict__ z4) {
    __shared__ double sh[GIV_MAX + 2];
    __shared__ double s_col[GIV_MAX], s_gv[2 * GIV_MAX];
    __shared__ double s_nrm;
    for (int j = threadIdx.x; j < ncol + 1 && j < GIV_MAX + 2; j += BLK) sh[j] = hraw[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ww = sh[ncol];
        double hh = 0.0;
        for (I j = 0; j < ncol; ++j) hh += sh[j] * sh[j];
        double r = ww - hh;
        if (r < 1e-6 * ww) {
            if (d_flag && blockIdx.x == 0) *d_flag = 1;
            if (r < 0.0) r = 0.0;
        }
        s_nrm = sqrt(r);
    }
    __syncthreads();
    const double nrm = s_nrm;
    const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i < nrows) {
        double a0 = w[3 * i], a1 = w[3 * i + 1], a2 = w[3 * i + 2], ap = w[3LL * N + i];
#pragma unroll 4
        for (int j = 0; j < ncol; ++j) {
            const double h = sh[j];
            const T* q = Q + (long long)j * ldq;
            a0 -= __builtin_nontemporal_load(q + 3 * i) * h;
            a1 -= __builtin_nontemporal_load(q + 3 * i + 1) * h;
            a2 -= __builtin_nontemporal_load(q + 3 * i + 2) * h;
            ap -= __builtin_nontemporal_load(q + 3LL * N + i) * h;
        }
        const double s = 1.0 / nrm;
        a0 *= s; a1 *= s; a2 *= s; ap *= s;
        w[3 * i] = a0; w[3 * i + 1] = a1; w[3 * i + 2] = a2; w[3LL * N + i] = ap;
        const T* A = dinv33 + i * 9;
        const double z0 = A[0] * a0 + A[3] * a1 + A[6] * a2, z1 = A[1] * a0 + A[4] * a1 + A[7] * a2;
        const double z2 = A[2] * a0 + A[5] * a1 + A[8] * a2, zp = ap * dinv1[i];
        z[3 * i + 0] = z0;
        z[3 * i + 1] = z1;
        z[3 * i + 2] = z2;
        z[3LL * N + i] = zp;
        if (z4) {  // the interleaved copy the matvec gathers from (dfl_bcsr_spmv_x4)
            double2* o = reinterpret_cast<double2*>(z4 + 4 * i);
            o[0] = make_double2(z0, z1);
            o[1] = make_double2(z2, zp);
        }
    }
    if (blockIdx.x == 0) {  // uniform per block: the barriers inside givens_step_block are safe
        T* col = H + (long long)iter * ldh;
        for (int j = threadIdx.x; j < ncol; j += BLK) col[j] = sh[j];
        if (threadIdx.x == 0) d_nrm[0] = nrm;
        __syncthreads();
        givens_step_block(iter, nrm, H, ldh, gv, beta, res_hist, s_col, s_gv);
    }
}

// The same step with TWO owned nodes per thread: nodes 2t and 2t + 1 are six consecutive velocity rows (three 16-byte
// loads per basis column, 16-byte aligned: the column stride 4N doubles and 6t doubles are both multiples of two) and two
// consecutive pressure rows (one 16-byte load, 8-byte aligned when N is odd) -- the node-per-thread form above streams the
// basis with 8-byte loads (5.0 TB/s at 227k owned nodes).  REVERSE walks the basis columns from the newest to the oldest:
// the dots pass before it read them oldest to newest, so the columns it touched last (still in the 256 MiB Infinity Cache
// when the whole basis no longer fits) are read first.  Same arithmetic per row except for the order of the column sum.
typedef double d2a8 __attribute__((ext_vector_type(2), aligned(8)));
template <bool REVERSE>
__global__ __launch_bounds__(BLK) void cgs_update_pc2_kernel(I nrows, I N, I ncol, const T* __restrict__ Q, long long ldq,
                                                            const T* __restrict__ hraw, T* __restrict__ w,
                                                            const T* __restrict__ dinv33, const T* __restrict__ dinv1,
                                                            T* __restrict__ z, I iter, T* H, I ldh, T* gv, T* beta, T* res_hist,
                                                            T* d_nrm, int* d_flag) {
    __shared__ double sh[GIV_MAX + 2];
    __shared__ double s_col[GIV_MAX], s_gv[2 * GIV_MAX];
    __shared__ double s_nrm;
    for (int j = threadIdx.x; j < ncol + 1 && j < GIV_MAX + 2; j += BLK) sh[j] = hraw[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ww = sh[ncol];
        double hh = 0.0;
        for (I j = 0; j < ncol; ++j) hh += sh[j] * sh[j];
        double r = ww - hh;
        if (r < 1e-6 * ww) {
            if (d_flag && blockIdx.x == 0) *d_flag = 1;
            if (r < 0.0) r = 0.0;
        }
        s_nrm = sqrt(r);
    }
    __syncthreads();
    const double nrm = s_nrm;
    const long long i0 = 2 * ((long long)blockIdx.x * BLK + threadIdx.x);
    if (i0 + 1 < nrows) {
        const T* wu = w + 3 * i0;
        const T* wp = w + 3LL * N + i0;
        d2s u0 = *reinterpret_cast<const d2s*>(wu), u1 = *reinterpret_cast<const d2s*>(wu + 2), u2 = *reinterpret_cast<const d2s*>(wu + 4);
        d2a8 pp = *reinterpret_cast<const d2a8*>(wp);
#pragma unroll 4
        for (int jj = 0; jj < ncol; ++jj) {
            const int j = REVERSE ? ncol - 1 - jj : jj;
            const double h = sh[j];
            const T* q = Q + (long long)j * ldq;
            const d2s q0 = __builtin_nontemporal_load(reinterpret_cast<const d2s*>(q + 3 * i0));
            const d2s q1 = __builtin_nontemporal_load(reinterpret_cast<const d2s*>(q + 3 * i0 + 2));
            const d2s q2 = __builtin_nontemporal_load(reinterpret_cast<const d2s*>(q + 3 * i0 + 4));
            const d2a8 qp = __builtin_nontemporal_load(reinterpret_cast<const d2a8*>(q + 3LL * N + i0));
            u0 -= q0 * h; u1 -= q1 * h; u2 -= q2 * h;
            pp.x -= qp.x * h; pp.y -= qp.y * h;
        }
        const double s = 1.0 / nrm;
        u0 *= s; u1 *= s; u2 *= s; pp.x *= s; pp.y *= s;
        *reinterpret_cast<d2s*>(w + 3 * i0) = u0;
        *reinterpret_cast<d2s*>(w + 3 * i0 + 2) = u1;
        *reinterpret_cast<d2s*>(w + 3 * i0 + 4) = u2;
        *reinterpret_cast<d2a8*>(w + 3LL * N + i0) = pp;
        const T* A = dinv33 + i0 * 9;  // node 2t: (u0.x, u0.y, u1.x); node 2t + 1: (u1.y, u2.x, u2.y)
        d2s z0, z1, z2;
        z0.x = A[0] * u0.x + A[3] * u0.y + A[6] * u1.x;
        z0.y = A[1] * u0.x + A[4] * u0.y + A[7] * u1.x;
        z1.x = A[2] * u0.x + A[5] * u0.y + A[8] * u1.x;
        z1.y = A[9] * u1.y + A[12] * u2.x + A[15] * u2.y;
        z2.x = A[10] * u1.y + A[13] * u2.x + A[16] * u2.y;
        z2.y = A[11] * u1.y + A[14] * u2.x + A[17] * u2.y;
        *reinterpret_cast<d2s*>(z + 3 * i0) = z0;
        *reinterpret_cast<d2s*>(z + 3 * i0 + 2) = z1;
        *reinterpret_cast<d2s*>(z + 3 * i0 + 4) = z2;
        d2a8 zp;
        zp.x = pp.x * dinv1[i0];
        zp.y = pp.y * dinv1[i0 + 1];
        *reinterpret_cast<d2a8*>(z + 3LL * N + i0) = zp;
    } else if (i0 < nrows) {  // the odd last node
        const long long i = i0;
        double a0 = w[3 * i], a1 = w[3 * i + 1], a2 = w[3 * i + 2], ap = w[3LL * N + i];
        for (int jj = 0; jj < ncol; ++jj) {
            const int j = REVERSE ? ncol - 1 - jj : jj;
            const double h = sh[j];
            const T* q = Q + (long long)j * ldq;
            a0 -= q[3 * i] * h; a1 -= q[3 * i + 1] * h; a2 -= q[3 * i + 2] * h; ap -= q[3LL * N + i] * h;
        }
        const double s = 1.0 / nrm;
        a0 *= s; a1 *= s; a2 *= s; ap *= s;
        w[3 * i] = a0; w[3 * i + 1] = a1; w[3 * i + 2] = a2; w[3LL * N + i] = ap;
        const T* A = dinv33 + i * 9;
        z[3 * i + 0] = A[0] * a0 + A[3] * a1 + A[6] * a2;
        z[3 * i + 1] = A[1] * a0 + A[4] * a1 + A[7] * a2;
        z[3 * i + 2] = A[2] * a0 + A[5] * a1 + A[8] * a2;
        z[3LL * N + i] = ap * dinv1[i];
    }
    if (blockIdx.x == 0) {  // uniform per block: the barriers inside givens_step_block are safe
        T* col = H + (long long)iter * ldh;
        for (int j = threadIdx.x; j < ncol; j += BLK) col[j] = sh[j];
        if (threadIdx.x == 0) d_nrm[0] = nrm;
        __syncthreads();
        givens_step_block(iter, nrm, H, ldh, gv, beta, res_hist, s_col, s_gv);
    }
}

// H[0:m,0:m] y = beta by back substitution (cublasDtrsv, krylov.c:297-301), one workgroup: column-oriented so that every
// step is one coalesced column update (a single thread walking rows pays a dependent global load per entry: 85 us at m = 40)
template <bool STAGED>
__global__ __launch_bounds__(BLK) void gmres_trsv_kernel(I m, const T* __restrict__ H, I ldh, T* __restrict__ beta) {
    extern __shared__ double s_mem[];  // [m] right-hand side, then (STAGED) the m x m upper triangle, column-major
    __shared__ double s_y;
    double* s_b = s_mem;
    double* s_H = s_mem + m;
    const int t = threadIdx.x;
    for (int i = t; i < m; i += BLK) s_b[i] = beta[i];
    if (STAGED)
        for (int k = t; k < m * m; k += BLK) {
            const int c = k / m, r = k - c * m;
            s_H[k] = H[(long long)c * ldh + r];
        }
    __syncthreads();
    for (I i = m - 1; i >= 0; --i) {
        const T* col = STAGED ? s_H + (long long)i * m : H + (long long)i * ldh;  // H(r, i), r <= i: contiguous
        if (t == 0) {
            s_y = s_b[i] / col[i];
            s_b[i] = s_y;
        }
        __syncthreads();
        const double y = s_y;
        for (int r = t; r < i; r += BLK) s_b[r] -= col[r] * y;
        __syncthreads();
    }
    for (int i = t; i < m; i += BLK) beta[i] = s_b[i];
}

__global__ void sqrt_kernel(T* v) { v[0] = sqrt(v[0]); }

// one wave that stays resident for about `us` microseconds of the constant 100 MHz clock (bounded by an iteration count as
// well, so that it ends whatever the clock does): lets a host routine find out whether two streams run CONCURRENTLY
__global__ void spin_kernel(int us, int* sink) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long ticks = (unsigned long long)us * 100ull;
    int it = 0;
    while (wall_clock64() - t0 < ticks && it < (1 << 22)) {
        __builtin_amdgcn_s_sleep(32);
        ++it;
    }
    if (sink && it < 0) *sink = it;
}

__global__ void residual_update_kernel(T* beta, T* gv) {
    double b0 = beta[0];
    beta[1] = -gv[1] * b0;
    beta[0] = b0 * gv[0];
}

// halo pack / unpack: out[i] = x[idx[i]] and x[idx[i]] = in[i]
__global__ __launch_bounds__(256) void gather_idx_kernel(I n, const I* __restrict__ idx, const T* __restrict__ x,
                                                        T* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[idx[i]];
}
__global__ __launch_bounds__(256) void scatter_idx_kernel(I n, const I* __restrict__ idx, const T* __restrict__ in,
                                                         T* __restrict__ x) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[idx[i]] = in[i];
}


// ---- generalized-alpha state algebra of the Newton driver, fused (src/main.c:107-118, 242-253, 535-565) -----------------
// One thread per node.  The reference runs this as memset + 6 cublasDaxpy/Dcopy passes over 6N-vectors; the arithmetic
// per entry is kept (zero + a*x as a product, every axpy as one fused multiply-add, in the reference's order):
//   dwgalpha = f1_0 * dwgold + f1_1 * dwg           (pressure slot: dwgalpha = dwg, not alpha-interpolated)
//   wgalpha  = wgold + f2_0 * dwgold + f2_1 * dwg   (pressure slot: 0)
// nodep != NULL: the packed gather record of the node (pack_nodes_kernel layout) is written in the same pass.
__global__ __launch_bounds__(BLK) void alpha_states_kernel(I N, const T* __restrict__ wgold, const T* __restrict__ dwgold,
                                                          const T* __restrict__ dwg, T f1_0, T f1_1, T f2_0, T f2_1,
                                                          const T* __restrict__ xg, T* __restrict__ wga, T* __restrict__ dwga,
                                                          T* __restrict__ nodep, T* __restrict__ nodexu) {
    const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i >= N) return;
    const long long idx[6] = {3 * i, 3 * i + 1, 3 * i + 2, 3LL * N + i, 4LL * N + i, 5LL * N + i};
    double w[6], d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double o = wgold[idx[k]], d0 = dwgold[idx[k]], d1 = dwg[idx[k]];
        if (k == 3) { d[k] = d1; w[k] = 0.0; }
        else {
            d[k] = fma(f1_1, d1, f1_0 * d0);
            w[k] = fma(f2_1, d1, fma(f2_0, d0, o));
        }
        wga[idx[k]] = w[k];
        dwga[idx[k]] = d[k];
    }
    if (nodep) {  // x[3] u[3] phi T du[3] p(rate vector, Q9) dphi dT pad pad
        double2* o = reinterpret_cast<double2*>(nodep + i * 16);
        const double x0 = xg[3 * i], x1 = xg[3 * i + 1], x2 = xg[3 * i + 2];
        o[0] = make_double2(x0, x1);
        o[1] = make_double2(x2, w[0]);
        o[2] = make_double2(w[1], w[2]);
        o[3] = make_double2(w[4], w[5]);
        o[4] = make_double2(d[0], d[1]);
        o[5] = make_double2(d[2], d[3]);
        o[6] = make_double2(d[4], d[5]);
        o[7] = make_double2(0.0, 0.0);
        if (nodexu) {  // the compact (x, u) records of the Jacobian kernel
            double2* c = reinterpret_cast<double2*>(nodexu + i * 8);
            c[0] = make_double2(x0, x1);
            c[1] = make_double2(x2, w[0]);
            c[2] = make_double2(w[1], w[2]);
            c[3] = make_double2(0.0, 0.0);
        }
    }
}

// predictor (main.c:544-546): dwg *= fac on the velocity and phi / T slots (the pressure slot is left alone)
__global__ __launch_bounds__(BLK) void alpha_predict_kernel(I N, T fac, T* __restrict__ dwg) {
    const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i >= 6LL * N) return;
    if (i >= 3LL * N && i < 4LL * N) return;
    dwg[i] *= fac;
}
// corrector (main.c:556-565): wgold += c0 * dwgold + c1 * dwg (not the pressure slot), then dwgold = dwg (all slots)
__global__ __launch_bounds__(BLK) void alpha_correct_kernel(I N, T c0, T c1, T* __restrict__ wgold, T* __restrict__ dwgold,
                                                           const T* __restrict__ dwg) {
    const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
    if (i >= 6LL * N) return;
    const double d1 = dwg[i];
    if (!(i >= 3LL * N && i < 4LL * N)) wgold[i] = fma(c1, d1, fma(c0, dwgold[i], wgold[i]));
    dwgold[i] = d1;
}

// four segment sums of squares in one pass (Newton norms of u, p, phi, T; main.c:125-130): grid (blocks, 4)
__global__ __launch_bounds__(BLK) void norms4_stage1(I N, const T* __restrict__ F, T* __restrict__ part, int nblk) {
    __shared__ double lds[4];
    const int seg = blockIdx.y;
    const long long begin = seg == 0 ? 0 : (2LL + seg) * N, len = seg == 0 ? 3LL * N : N;
    double acc = 0.0;
    for (long long k = (long long)blockIdx.x * BLK + threadIdx.x; k < len; k += (long long)nblk * BLK) {
        const double v = F[begin + k];
        acc += v * v;
    }
    const double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) part[seg * nblk + blockIdx.x] = r;
}
template <bool SQRT>
__global__ __launch_bounds__(BLK) void norms4_stage2(int nblk, const T* __restrict__ part, T* __restrict__ out) {
    __shared__ double lds[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += BLK) acc += part[blockIdx.x * nblk + i];
    const double r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = SQRT ? sqrt(r) : r;
}

}  // namespace

extern "C" {
void dfl_gather_idx(I n, const I* idx, const T* x, T* out, void* stream) {
    if (n <= 0) return;
    gather_idx_kernel<<<ceil_div(n, 256), 256, 0, S(stream)>>>(n, idx, x, out);
    DFL_LAUNCH_CHECK();
}
void dfl_scatter_idx(I n, const I* idx, const T* in, T* x, void* stream) {
    if (n <= 0) return;
    scatter_idx_kernel<<<ceil_div(n, 256), 256, 0, S(stream)>>>(n, idx, in, x);
    DFL_LAUNCH_CHECK();
}



void dfl_alpha_states2(I N, const T* wgold, const T* dwgold, const T* dwg, T f1_0, T f1_1, T f2_0, T f2_1, const T* xg, T* wgalpha,
                       T* dwgalpha, T* nodep, T* nodexu, void* stream) {
    if (N <= 0) return;
    alpha_states_kernel<<<ceil_div(N, BLK), BLK, 0, S(stream)>>>(N, wgold, dwgold, dwg, f1_0, f1_1, f2_0, f2_1, xg, wgalpha, dwgalpha, nodep,
                                                                 nodexu);
    DFL_LAUNCH_CHECK();
}
void dfl_alpha_states(I N, const T* wgold, const T* dwgold, const T* dwg, T f1_0, T f1_1, T f2_0, T f2_1, const T* xg, T* wgalpha,
                      T* dwgalpha, T* nodep, void* stream) {
    dfl_alpha_states2(N, wgold, dwgold, dwg, f1_0, f1_1, f2_0, f2_1, xg, wgalpha, dwgalpha, nodep, nullptr, stream);
}
void dfl_alpha_predict(I N, T fac, T* dwg, void* stream) {
    if (N <= 0) return;
    alpha_predict_kernel<<<ceil_div(6LL * N, BLK), BLK, 0, S(stream)>>>(N, fac, dwg);
    DFL_LAUNCH_CHECK();
}
void dfl_alpha_correct(I N, T c0, T c1, T* wgold, T* dwgold, const T* dwg, void* stream) {
    if (N <= 0) return;
    alpha_correct_kernel<<<ceil_div(6LL * N, BLK), BLK, 0, S(stream)>>>(N, c0, c1, wgold, dwgold, dwg);
    DFL_LAUNCH_CHECK();
}
void dfl_norms4(I N, const T* F, T* d_out4, int take_sqrt, T* work, void* stream) {
    if (N <= 0) return;
    int nblk = ceil_div(3LL * N, BLK * 8);
    if (nblk < 1) nblk = 1;
    if (nblk > MAX_PART / 4) nblk = MAX_PART / 4;
    norms4_stage1<<<dim3(nblk, 4), BLK, 0, S(stream)>>>(N, F, work, nblk);
    if (take_sqrt) norms4_stage2<true><<<4, BLK, 0, S(stream)>>>(nblk, work, d_out4);
    else norms4_stage2<false><<<4, BLK, 0, S(stream)>>>(nblk, work, d_out4);
    DFL_LAUNCH_CHECK();
}

int dfl_abi_version(void) { return 1; }
const char* dfl_last_error(void) { return g_err; }

void dfl_daxpy(I n, T alpha, const T* x, T* y, void* stream) {
    launch_map3(n, x, y, y, [alpha] __device__(double a, double b) { return alpha * a + b; }, stream);
}
void dfl_dscal(I n, T alpha, T* x, void* stream) {
    launch_map1(n, x, [alpha] __device__(double a) { return alpha * a; }, stream);
}
void dfl_dcopy(I n, const T* x, T* y, void* stream) {
    if (n > 0) DFL_GUARD(hipMemcpyAsync(y, x, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, S(stream)));
}
void dfl_dset(I n, T alpha, T* x, void* stream) {
    launch_map1(n, x, [alpha] __device__(double) { return alpha; }, stream);
}
void SetValGPU(T* val, I n, T alpha) { dfl_dset(n, alpha, val, nullptr); }
void dfl_pointwise_mult(I n, const T* x, const T* y, T* z, void* stream) {
    launch_map3(n, x, y, z, [] __device__(double a, double b) { return a * b; }, stream);
}
void dfl_pointwise_div(I n, const T* x, const T* y, T* z, void* stream) {
    launch_map3(n, x, y, z, [] __device__(double a, double b) { return a / b; }, stream);
}
void dfl_pointwise_inv(I n, T* x, void* stream) {
    launch_map1(n, x, [] __device__(double a) { return 1.0 / a; }, stream);
}

I dfl_reduce_work_size(void) { return MAX_PART; }

static int reduce_grid(I n) {
    int g = ceil_div(n, BLK * 8);
    if (g < 1) g = 1;
    if (g > MAX_PART) g = MAX_PART;
    return g;
}
void dfl_ddot(I n, const T* x, const T* y, T* d_out, T* work, void* stream) {
    int g = reduce_grid(n);
    reduce_stage1<true><<<g, BLK, 0, S(stream)>>>(n, x, y, work);
    reduce_stage2<false><<<1, BLK, 0, S(stream)>>>(g, work, d_out);
    DFL_LAUNCH_CHECK();
}
void dfl_dnrm2(I n, const T* x, T* d_out, T* work, void* stream) {
    int g = reduce_grid(n);
    reduce_stage1<false><<<g, BLK, 0, S(stream)>>>(n, x, x, work);
    reduce_stage2<true><<<1, BLK, 0, S(stream)>>>(g, work, d_out);
    DFL_LAUNCH_CHECK();
}
void dfl_dscal_inv_dev(I n, const T* d_scale, T* x, void* stream) {
    if (n <= 0) return;
    scal_inv_dev<<<ceil_div((n >> 1) + 1, BLK), BLK, 0, S(stream)>>>(n, d_scale, x);
    DFL_LAUNCH_CHECK();
}

static int cgs_mode() {
    static int m = -1;
    if (m < 0) m = getenv("DFL_CGS_MODE") ? atoi(getenv("DFL_CGS_MODE")) & 7 : 0;
    return m;
}
#define CGS_UPDATE_LAUNCH(...)                                                                     \
    do {                                                                                           \
        switch (cgs_mode()) {                                                                      \
            case 1: cgs_update_kernel<true, 1><<<g, BLK, 0, S(stream)>>>(__VA_ARGS__); break;      \
            case 2: cgs_update_kernel<true, 2><<<g, BLK, 0, S(stream)>>>(__VA_ARGS__); break;      \
            case 3: cgs_update_kernel<true, 3><<<g, BLK, 0, S(stream)>>>(__VA_ARGS__); break;      \
            case 4: cgs_update_kernel<true, 4><<<(g + 7) / 8 * 8, BLK, 0, S(stream)>>>(__VA_ARGS__); break; \
            default: cgs_update_kernel<true, 0><<<g, BLK, 0, S(stream)>>>(__VA_ARGS__); break;     \
        }                                                                                          \
    } while (0)
int64_t dfl_cgs_work_size(I n, I ncol) {
    int64_t nrb = ceil_div(n, ROWS_PER_BLOCK);
    int64_t a = nrb * (int64_t)(ncol > 0 ? ncol : 1);
    int64_t b = ceil_div(n, UROWS);
    return (a > b ? a : b) + 16;
}
void dfl_cgs_dots(I n, I ncol, const T* Q, int64_t ldq, const T* w, T* d_h, T* work, void* stream) {
    if (ncol <= 0) return;
    int nrb = ceil_div(n, ROWS_PER_BLOCK);
    // column tile of a workgroup: w is read once per (row block, tile), so with tiles of 8 a 41-column step re-read it six
    // times (15 % of the kernel's bytes over a 40-iteration solve); tiles of 64 read it once while the grid still has
    // thousands of row blocks.  Below 64k rows the grid would get too small: tiles of 8 there (DFL_CGS_TILE overrides).
    static int tile_env = -1;
    if (tile_env < 0) tile_env = getenv("DFL_CGS_TILE") ? atoi(getenv("DFL_CGS_TILE")) : 0;
    const int ct = tile_env > 0 ? tile_env : CT;  // (measured: 0.2013-0.2017 ms per CGS kernel with 64, 0.2021-0.2026 with 8: the re-reads of w come out of the caches; not worth a change of the default)
    dim3 grid(nrb, ceil_div(ncol, ct));
    // developer A/B, DFL_CGS_DOTS_TILE=1: all loads of a column tile before the reductions -- 0.2144-0.216 against 0.212-0.2138 ms:
    // not faster (occupancy already hides the per-column barrier), so the per-column kernel stays
    static const bool per_column = !(getenv("DFL_CGS_DOTS_TILE") && atoi(getenv("DFL_CGS_DOTS_TILE")) == 1);
    if (cgs_mode() & 2) cgs_dots_stage1<2><<<grid, BLK, 0, S(stream)>>>(n, ncol, Q, ldq, w, work, nrb, ct);
    else if (per_column) cgs_dots_stage1<0><<<grid, BLK, 0, S(stream)>>>(n, ncol, Q, ldq, w, work, nrb, ct);
    else { dim3 grid8(nrb, ceil_div(ncol, CT)); cgs_dots_stage1_tile<0><<<grid8, BLK, 0, S(stream)>>>(n, ncol, Q, ldq, w, work, nrb); }
    cgs_dots_stage2<<<ncol, BLK, 0, S(stream)>>>(nrb, work, d_h);
    DFL_LAUNCH_CHECK();
}
void dfl_cgs_update(I n, I ncol, const T* Q, int64_t ldq, const T* d_h, T* w, T* d_nrm, int take_sqrt, T* work, void* stream) {
    int g = ceil_div(n, UROWS);
    CGS_UPDATE_LAUNCH(n, ncol, Q, ldq, d_h, w, d_nrm ? work : nullptr);
    if (d_nrm) {
        if (take_sqrt) reduce_stage2<true><<<1, BLK, 0, S(stream)>>>(g, work, d_nrm);
        else reduce_stage2<false><<<1, BLK, 0, S(stream)>>>(g, work, d_nrm);
    }
    DFL_LAUNCH_CHECK();
}
void dfl_cgs_update_givens(I n, I ncol, const T* Q, int64_t ldq, const T* d_h, T* w, T* d_nrm, T* work, I iter, T* d_H, I ldh,
                           T* d_gv, T* d_beta, T* d_res_hist, void* stream) {
    int g = ceil_div(n, UROWS);
    CGS_UPDATE_LAUNCH(n, ncol, Q, ldq, d_h, w, work);
    norm_givens_kernel<<<1, BLK, 0, S(stream)>>>(g, work, d_nrm, iter, d_H, ldh, d_gv, d_beta, d_res_hist);
    DFL_LAUNCH_CHECK();
}
void dfl_spin_us(int us, void* stream) {
    if (us <= 0) return;
    if (us > 20000) us = 20000;
    spin_kernel<<<1, 64, 0, S(stream)>>>(us, nullptr);
    DFL_LAUNCH_CHECK();
}
void dfl_dsqrt_dev(T* d_val, void* stream) {
    sqrt_kernel<<<1, 1, 0, S(stream)>>>(d_val);
    DFL_LAUNCH_CHECK();
}
void dfl_gemv_n(I n, I ncol, const T* Q, int64_t ldq, const T* d_c, T* y, void* stream) {
    int g = ceil_div(n, UROWS);
    cgs_update_kernel<false><<<g, BLK, 0, S(stream)>>>(n, ncol, Q, ldq, d_c, y, nullptr);
    DFL_LAUNCH_CHECK();
}

void dfl_gmres_givens(I iter, const T* d_nrm, T* d_H, I ldh, T* d_gv, T* d_beta, T* d_res_hist, void* stream) {
    gmres_givens_kernel<false><<<1, BLK, 0, S(stream)>>>(iter, const_cast<T*>(d_nrm), d_H, ldh, d_gv, d_beta, d_res_hist);
    DFL_LAUNCH_CHECK();
}
void dfl_gmres_givens_sq(I iter, T* d_nrm_sq, T* d_H, I ldh, T* d_gv, T* d_beta, T* d_res_hist, void* stream) {
    gmres_givens_kernel<true><<<1, BLK, 0, S(stream)>>>(iter, d_nrm_sq, d_H, ldh, d_gv, d_beta, d_res_hist);
    DFL_LAUNCH_CHECK();
}
void dfl_cgs_update_pc_givens(I nrows, I N, I ncol, const T* Q, int64_t ldq, const T* d_hraw, T* w, const T* dinv33, const T* dinv1,
                              T* z, I iter, T* d_H, I ldh, T* d_gv, T* d_beta, T* d_res_hist, T* d_nrm, int* d_flag, void* stream) {
    dfl_cgs_update_pc_givens_x4(nrows, N, ncol, Q, ldq, d_hraw, w, dinv33, dinv1, z, nullptr, iter, d_H, ldh, d_gv, d_beta, d_res_hist,
                                d_nrm, d_flag, stream);
}
void dfl_cgs_update_pc_givens_x4(I nrows, I N, I ncol, const T* Q, int64_t ldq, const T* d_hraw, T* w, const T* dinv33,
                                 const T* dinv1, T* z, T* z4, I iter, T* d_H, I ldh, T* d_gv, T* d_beta, T* d_res_hist, T* d_nrm,
                                 int* d_flag, void* stream) {
    if (ncol + 1 > GIV_MAX + 2 || nrows <= 0) abort();  // the caller falls back to the separate kernels beyond GIV_MAX columns
    // DFL_UPDATE_PC: 0 = one node per thread (default), 1 = two nodes per thread (16-byte loads), 2 = + newest column first.
    // Measured on rank 0 / rank 4 of the 8-way 10M-tet partition (gpurun_out/r3c, profiles/r03_rank_local_*): 7.99 / 7.39 ms
    // per step with 0, 8.09 / 7.51 with 1, 8.11 / 7.51 with 2 -- at 227k owned nodes the two-node form leaves 1.7 waves per
    // SIMD, and the wider loads do not make up for the lost memory-level parallelism; the column order changes nothing (the
    // rank's basis streams from HBM either way).  The wide forms stay for A/B.
    static int variant = -1;
    if (variant < 0) { const char* e = getenv("DFL_UPDATE_PC"); variant = e ? atoi(e) : 0; }
    const int grid2 = (int)ceil_div((nrows + 1) / 2, BLK);
    if (variant == 0 || z4)
        cgs_update_pc_kernel<<<ceil_div(nrows, BLK), BLK, 0, S(stream)>>>(nrows, N, ncol, Q, ldq, d_hraw, w, dinv33, dinv1, z, iter, d_H, ldh,
                                                                          d_gv, d_beta, d_res_hist, d_nrm, d_flag, z4);
    else if (variant == 1)
        cgs_update_pc2_kernel<false><<<grid2, BLK, 0, S(stream)>>>(nrows, N, ncol, Q, ldq, d_hraw, w, dinv33, dinv1, z, iter, d_H, ldh, d_gv,
                                                                   d_beta, d_res_hist, d_nrm, d_flag);
    else
        cgs_update_pc2_kernel<true><<<grid2, BLK, 0, S(stream)>>>(nrows, N, ncol, Q, ldq, d_hraw, w, dinv33, dinv1, z, iter, d_H, ldh, d_gv,
                                                                  d_beta, d_res_hist, d_nrm, d_flag);
    DFL_LAUNCH_CHECK();
}
void dfl_gmres_givens_pythagoras(I iter, T* d_nrm, T* d_H, I ldh, T* d_gv, T* d_beta, T* d_res_hist, int* d_flag, void* stream) {
    gmres_givens_pythagoras_kernel<<<1, BLK, 0, S(stream)>>>(iter, d_nrm, d_H, ldh, d_gv, d_beta, d_res_hist, d_flag);
    DFL_LAUNCH_CHECK();
}
void dfl_gmres_trsv(I m, const T* d_H, I ldh, T* d_beta, void* stream) {
    if (m <= 0) return;
    const size_t staged = ((size_t)m * m + m) * sizeof(double);
    if (staged <= 60 * 1024) gmres_trsv_kernel<true><<<1, BLK, staged, S(stream)>>>(m, d_H, ldh, d_beta);
    else gmres_trsv_kernel<false><<<1, BLK, (size_t)m * sizeof(double), S(stream)>>>(m, d_H, ldh, d_beta);
    DFL_LAUNCH_CHECK();
}
void GMRESResidualUpdatePrivate(T* beta, T* gv) {
    residual_update_kernel<<<1, 1>>>(beta, gv);
    DFL_LAUNCH_CHECK();
}

}  // extern "C"
