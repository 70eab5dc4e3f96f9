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
// Neighbour search: uniform cell list, cell edge >= 2R, particles sorted by cell
// (stable radix sort => fixed summation order => bitwise reproducible forces).
// HBM-bound: (48 read + 24 write) B per particle + 24 B per tested neighbour.
#include "dfl_common.hpp"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int BLK = 256;

__device__ __forceinline__ int cell_coord(double x, double inv_cell, int ncell) {
    int c = (int)floor(x * inv_cell);
    return c < 0 ? 0 : (c >= ncell ? ncell - 1 : c);
}

__global__ void cell_index_kernel(I P, const T* __restrict__ coord, T inv_cell, I ncell, I* __restrict__ cell_id,
                                  I* __restrict__ order) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= P) return;
    const int cx = cell_coord(coord[3 * i], inv_cell, ncell);
    const int cy = cell_coord(coord[3 * i + 1], inv_cell, ncell);
    const int cz = cell_coord(coord[3 * i + 2], inv_cell, ncell);
    cell_id[i] = cx + ncell * (cy + ncell * cz);
    order[i] = i;
}

// cell_start[c] = first sorted slot of cell c, cell_start[ncell3] = P ; empty cells get the next start
__global__ void cell_bounds_kernel(I P, const I* __restrict__ sorted_cell, I ncell3, I* __restrict__ cell_start) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i > P) return;
    const int cur = (i < P) ? sorted_cell[i] : ncell3;
    const int prev = (i > 0) ? sorted_cell[i - 1] : -1;
    for (int c = prev + 1; c <= cur; ++c) cell_start[c] = i;
}

__global__ __launch_bounds__(BLK) void dem_force_kernel(I P, const T* __restrict__ coord, const T* __restrict__ vel, T R, T mass,
                                                       T kn, T gn, T inv_cell, I ncell, const I* __restrict__ order,
                                                       const I* __restrict__ cell_start, T* __restrict__ acc) {
    const int s = blockIdx.x * BLK + threadIdx.x;
    if (s >= P) return;
    const int i = order[s];
    const double xi = coord[3 * i], yi = coord[3 * i + 1], zi = coord[3 * i + 2];
    const double ui = vel[3 * i], vi = vel[3 * i + 1], wi = vel[3 * i + 2];
    const int cx = cell_coord(xi, inv_cell, ncell), cy = cell_coord(yi, inv_cell, ncell), cz = cell_coord(zi, inv_cell, ncell);
    double fx = 0.0, fy = 0.0, fz = 0.0;
    const double d2max = 4.0 * R * R;
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= ncell) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= ncell) continue;
            const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < ncell ? cx + 1 : ncell - 1;
            const int c0 = x0 + ncell * (y + ncell * z), c1 = x1 + ncell * (y + ncell * z);
            // the three x-neighbour cells are contiguous in the sorted order
            for (int t = cell_start[c0]; t < cell_start[c1 + 1]; ++t) {
                const int j = order[t];
                if (j == i) continue;
                const double rx = xi - coord[3 * j], ry = yi - coord[3 * j + 1], rz = zi - coord[3 * j + 2];
                const double d2 = rx * rx + ry * ry + rz * rz;
                if (d2 >= d2max || d2 == 0.0) continue;
                const double dist = sqrt(d2), inv = 1.0 / dist;
                const double nx = rx * inv, ny = ry * inv, nz = rz * inv;
                const double vn = (ui - vel[3 * j]) * nx + (vi - vel[3 * j + 1]) * ny + (wi - vel[3 * j + 2]) * nz;
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

void dfl_dem_cell_index(I P, const T* coord, T cell, I ncell, I* cell_id, I* order, void* stream) {
    if (P <= 0) return;
    cell_index_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, coord, 1.0 / cell, ncell, cell_id, order);
    DFL_LAUNCH_CHECK();
}

// sorts (cell_id, order) by cell (stable) and builds cell_start[ncell3+1]; synchronises; allocates temp storage
void dfl_dem_sort_by_cell(I P, I* cell_id, I* order, I ncell3, I* cell_start) {
    if (P <= 0) return;
    I *k2 = nullptr, *v2 = nullptr;
    DFL_GUARD(hipMalloc((void**)&k2, sizeof(I) * (size_t)P));
    DFL_GUARD(hipMalloc((void**)&v2, sizeof(I) * (size_t)P));
    int bits = 1;
    while ((1LL << bits) < (long long)ncell3 + 1 && bits < 31) ++bits;
    size_t bytes = 0;
    unsigned int* kin = reinterpret_cast<unsigned int*>(cell_id);
    unsigned int* kout = reinterpret_cast<unsigned int*>(k2);
    (void)rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, order, v2, (size_t)P, 0, bits);
    void* tmp = nullptr;
    DFL_GUARD(hipMalloc(&tmp, bytes + 16));
    DFL_GUARD(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, order, v2, (size_t)P, 0, bits));
    DFL_GUARD(hipMemcpyAsync(cell_id, k2, sizeof(I) * (size_t)P, hipMemcpyDeviceToDevice, 0));
    DFL_GUARD(hipMemcpyAsync(order, v2, sizeof(I) * (size_t)P, hipMemcpyDeviceToDevice, 0));
    cell_bounds_kernel<<<ceil_div((long long)P + 1, BLK), BLK>>>(P, cell_id, ncell3, cell_start);
    DFL_GUARD(hipDeviceSynchronize());
    DFL_GUARD(hipFree(tmp));
    DFL_GUARD(hipFree(k2));
    DFL_GUARD(hipFree(v2));
}

void dfl_dem_forces(I P, const T* coord, const T* vel, T radius, T mass, T kn, T gamma_n, T cell, I ncell, const I* order,
                    const I* cell_start, T* acc, void* stream) {
    if (P <= 0) return;
    dem_force_kernel<<<ceil_div(P, BLK), BLK, 0, S(stream)>>>(P, coord, vel, radius, mass, kn, gamma_n, 1.0 / cell, ncell, order,
                                                            cell_start, acc);
    DFL_LAUNCH_CHECK();
}

void dfl_dem_integrate(I P, T dt, T* coord, T* vel, const T* acc, void* stream) {
    if (P <= 0) return;
    dem_integrate_kernel<<<ceil_div((long long)P * 3, BLK), BLK, 0, S(stream)>>>(3 * P, dt, coord, vel, acc);
    DFL_LAUNCH_CHECK();
}

}  // extern "C"
