// DEM particle contact-force sweep for gfx950.
//
// BUILD-DEFINED: the reference's Particle.c is a storage container (coord/vel/acc arrays,
// mass 1.0, radius 0.1) whose Add/Update/Remove hooks are empty and whose calls in the time
// loop are commented out (src/Particle.c:120-130, src/main.c:547-569).
// BASELINE.json nevertheless names "the DEM particle contact-force sweep", so the behaviour
// is specified here, with the simplest standard model:
//   monodisperse spheres (radius R, mass m), linear spring-dashpot normal contact
//       overlap d = 2R - |xi - xj| > 0,  n = (xi - xj)/|xi - xj|
//       F_ij = (kn * d - gn * ((vi - vj) . n)) n          acc_i = sum_j F_ij / m
//   plus the same law against the six walls of the unit box (d = R - distance to wall).
// Neighbour search: uniform cell list, cell edge >= 4R (a particle's interaction range [x - 2R, x + 2R] then covers at most
// two cells per axis), particles sorted by (cell, particle id)
// (counting sort + per-cell ordering => fixed summation order => bitwise reproducible forces).
// HBM-bound: (48 read + 24 write) B per particle + 24 B per tested neighbour.
#include "dfl_common.hpp"

namespace {

constexpr int BLK = 256;
constexpr int SCAN_CHUNK = 1024;  // cells per scan block

__device__ __forceinline__ int cell_coord(double x, double inv_cell, int ncell) {
    int c = (int)floor(x * inv_cell);
    return c < 0 ? 0 : (c >= ncell ? ncell - 1 : c);
}

// The whole sweep is six small dependent launches on the library stream, no allocation, no host round trip:
//   bin    cell of every particle; count[cell]++ (integer atomics: the COUNTS are deterministic, the returned ranks are
//          not -- the segment sort below removes that freedom).  The grid is sized for a few particles per cell, so the
//          atomics are nearly uncontended (a per-chunk counter bumped by every particle cost 18 us of serialised atomics)
//   chunk  total of every 1024-cell chunk
//   scan   cell_start = exclusive scan of count (every block adds up the chunk totals before it), count zeroed
//   place  slot[cell_start[cell] + rank] = particle (arrival order)
//   sort   every particle finds its place among the few members of its cell (number of members with a smaller id => the
//          neighbour loops visit particles in a FIXED order: bitwise reproducible forces) and writes its own sorted copy of
//          position / velocity, so that the force kernel streams contiguous runs instead of gathering through order[]
//   force  one thread per sorted slot over the 9 contiguous 3-cell runs around it
__global__ __launch_bounds__(BLK) void dem_bin_kernel(I P, const T* __restrict__ coord, T inv_cell, I ncell, I* __restrict__ cell_of,
                                                     I* __restrict__ rank, I* __restrict__ count) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= P) return;
    const int cx = cell_coord(coord[3 * i], inv_cell, ncell);
    const int cy = cell_coord(coord[3 * i + 1], inv_cell, ncell);
    const int cz = cell_coord(coord[3 * i + 2], inv_cell, ncell);
    const int c = cx + ncell * (cy + ncell * cz);
    cell_of[i] = c;
    rank[i] = atomicAdd(&count[c], 1);
}

// cell_start[0 .. n] = exclusive scan of count[0 .. n) (cell_start[n] = total); block b owns cells [b*1024, (b+1)*1024)
__global__ __launch_bounds__(BLK) void dem_scan_kernel(I n, I* __restrict__ count, const I* __restrict__ chunk_sum,
                                                      I* __restrict__ cell_start) {
    __shared__ int s_part[BLK];
    __shared__ int s_base;
    const int t = threadIdx.x, b = blockIdx.x;
    int acc = 0;
    for (int k = t; k < b; k += BLK) acc += chunk_sum[k];
    s_part[t] = acc;
    __syncthreads();
    for (int off = BLK / 2; off > 0; off >>= 1) {
        if (t < off) s_part[t] += s_part[t + off];
        __syncthreads();
    }
    if (t == 0) s_base = s_part[0];
    __syncthreads();
    const int base = s_base;
    __syncthreads();
    // 4 consecutive cells per thread
    const long long c0 = (long long)b * SCAN_CHUNK + 4 * t;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (c0 + k < n) ? count[c0 + k] : 0;
        if (c0 + k < n) count[c0 + k] = 0;  // ready for the next sweep
    }
    const int mine = v[0] + v[1] + v[2] + v[3];
    s_part[t] = mine;
    __syncthreads();
    for (int off = 1; off < BLK; off <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread sums
        const int add = t >= off ? s_part[t - off] : 0;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    int run = base + s_part[t] - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c0 + k <= n) cell_start[c0 + k] = run;
        run += v[k];
    }
}

// total of every 1024-cell chunk (large grids only: feeds dem_scan_kernel)
__global__ __launch_bounds__(BLK) void dem_chunk_sum_kernel(I n, const I* __restrict__ count, I* __restrict__ chunk_sum) {
    __shared__ int s_part[BLK];
    const int t = threadIdx.x;
    const long long c0 = (long long)blockIdx.x * SCAN_CHUNK + 4 * t;
    int v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += (c0 + k < n) ? count[c0 + k] : 0;
    s_part[t] = v;
    __syncthreads();
    for (int off = BLK / 2; off > 0; off >>= 1) {
        if (t < off) s_part[t] += s_part[t + off];
        __syncthreads();
    }
    if (t == 0) chunk_sum[blockIdx.x] = s_part[0];
}

__global__ __launch_bounds__(BLK) void dem_place_kernel(I P, const I* __restrict__ cell_of, const I* __restrict__ rank,
                                                       const I* __restrict__ cell_start, I* __restrict__ slot) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= P) return;
    slot[cell_start[cell_of[i]] + rank[i]] = i;
}

__global__ __launch_bounds__(BLK) void dem_sort_cells_kernel(I P, const I* __restrict__ cell_of, const I* __restrict__ cell_start,
                                                            const I* __restrict__ slot, I* __restrict__ order,
                                                            const T* __restrict__ coord, const T* __restrict__ vel,
                                                            T* __restrict__ sorted /*[P][6]*/) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= P) return;
    const int c = cell_of[i];
    const int lo = cell_start[c], hi = cell_start[c + 1];
    int r = 0;
    for (int a = lo; a < hi; ++a) r += slot[a] < i;  // a handful of members per cell
    const long long pos = lo + r;
    order[pos] = i;
    T* o = sorted + pos * 6;
    o[0] = coord[3 * i]; o[1] = coord[3 * i + 1]; o[2] = coord[3 * i + 2];
    o[3] = vel[3 * i]; o[4] = vel[3 * i + 1]; o[5] = vel[3 * i + 2];
}

__global__ __launch_bounds__(BLK) void dem_force_kernel(I P, const T* __restrict__ sorted, T R, T mass, T kn, T gn, T inv_cell,
                                                       I ncell, const I* __restrict__ order, const I* __restrict__ cell_start,
                                                       T* __restrict__ acc) {
    const int s = blockIdx.x * BLK + threadIdx.x;
    if (s >= P) return;
    const T* me = sorted + (long long)s * 6;
    const double xi = me[0], yi = me[1], zi = me[2], ui = me[3], vi = me[4], wi = me[5];
    // cells the interaction range [x - 2R, x + 2R] touches: at most two per axis (cell edge >= 4R)
    const double rng = 2.0 * R;
    const int x0 = cell_coord(xi - rng, inv_cell, ncell), x1 = cell_coord(xi + rng, inv_cell, ncell);
    const int y0 = cell_coord(yi - rng, inv_cell, ncell), y1 = cell_coord(yi + rng, inv_cell, ncell);
    const int z0 = cell_coord(zi - rng, inv_cell, ncell), z1 = cell_coord(zi + rng, inv_cell, ncell);
    double fx = 0.0, fy = 0.0, fz = 0.0;
    const double d2max = 4.0 * R * R;
    for (int z = z0; z <= z1; ++z) {
        for (int y = y0; y <= y1; ++y) {
            const int c0 = x0 + ncell * (y + ncell * z), c1 = x1 + ncell * (y + ncell * z);
            // the x-neighbour cells are one contiguous run of the sorted copies (ascending cell, then particle id)
            for (int t = cell_start[c0]; t < cell_start[c1 + 1]; ++t) {
                if (t == s) continue;
                const T* o = sorted + (long long)t * 6;
                const double rx = xi - o[0], ry = yi - o[1], rz = zi - o[2];
                const double d2 = rx * rx + ry * ry + rz * rz;
                if (d2 >= d2max || d2 == 0.0) continue;
                const double dist = sqrt(d2), inv = 1.0 / dist;
                const double nx = rx * inv, ny = ry * inv, nz = rz * inv;
                const double vn = (ui - o[3]) * nx + (vi - o[4]) * ny + (wi - o[5]) * nz;
                const double f = kn * (2.0 * R - dist) - gn * vn;
                fx += f * nx; fy += f * ny; fz += f * nz;
            }
        }
    }
    // walls of the unit box
    const double p[3] = {xi, yi, zi}, v[3] = {ui, vi, wi};
    double fw[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double lo = R - p[d];          // overlap with the wall x_d = 0 (normal +e_d)
        if (lo > 0.0) fw[d] += kn * lo - gn * v[d];
        const double hi = p[d] + R - 1.0;    // overlap with the wall x_d = 1 (normal -e_d)
        if (hi > 0.0) fw[d] -= kn * hi + gn * v[d];
    }
    const double im = 1.0 / mass;
    const long long i = order[s];
    acc[3 * i] = (fx + fw[0]) * im;
    acc[3 * i + 1] = (fy + fw[1]) * im;
    acc[3 * i + 2] = (fz + fw[2]) * im;
}

// semi-implicit Euler: v += dt a ; x += dt v
__global__ void dem_integrate_kernel(I n3, T dt, T* __restrict__ coord, T* __restrict__ vel, const T* __restrict__ acc) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n3) return;
    const double v = vel[i] + dt * acc[i];
    vel[i] = v;
    coord[i] += dt * v;
}

}  // namespace

extern "C" {

I dfl_dem_num_chunks(I ncell3) { return (I)(((long long)ncell3 + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK); }

// count[ncell3 + 1] must be zero on entry (it is again on return); chunk_sum[dfl_dem_num_chunks] and slot[P] are scratch
void dfl_dem_build_cells(I P, const T* coord, const T* vel, T cell, I ncell, I* cell_of, I* rank, I* count, I* chunk_sum,
                         I* cell_start, I* slot, I* order, T* sorted, void* stream) {
    if (P <= 0) return;
    const I ncell3 = ncell * ncell * ncell;
    const I nchunk = dfl_dem_num_chunks(ncell3);
    dem_bin_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, coord, 1.0 / cell, ncell, cell_of, rank, count);
    dem_chunk_sum_kernel<<<nchunk, BLK, 0, S(stream)>>>(ncell3, count, chunk_sum);
    dem_scan_kernel<<<nchunk, BLK, 0, S(stream)>>>(ncell3, count, chunk_sum, cell_start);
    dem_place_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, cell_of, rank, cell_start, slot);
    dem_sort_cells_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, cell_of, cell_start, slot, order, coord, vel, sorted);
    DFL_LAUNCH_CHECK();
}

void dfl_dem_forces(I P, const T* sorted, T radius, T mass, T kn, T gamma_n, T cell, I ncell, const I* order, const I* cell_start,
                    T* acc, void* stream) {
    if (P <= 0) return;
    dem_force_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, sorted, radius, mass, kn, gamma_n, 1.0 / cell, ncell, order, cell_start,
                                                            acc);
    DFL_LAUNCH_CHECK();
}

void dfl_dem_integrate(I P, T dt, T* coord, T* vel, const T* acc, void* stream) {
    if (P <= 0) return;
    dem_integrate_kernel<<<ceil_div((long long)P * 3, BLK), BLK, 0, S(stream)>>>(3 * P, dt, coord, vel, acc);
    DFL_LAUNCH_CHECK();
}

}  // extern "C"
