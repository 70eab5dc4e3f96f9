// =============================================================================
//  ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.cpp header).
//
//  Build-defined extensions: pieces BASELINE.json names but the reference does
//  not contain -- CG, ILU0-style preconditioner, DEM
//  contact sweep -- plus OpenMP variants used only by bench.py's cpu_baseline
//  leg.  All "parity unpinned": there is no reference behaviour to match; the
//  tests pin them with analytic / scipy cross-checks.
// =============================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <omp.h>

typedef int32_t i32;
typedef double f64;

extern "C" {

int orc_num_threads() { return omp_get_max_threads(); }
int orc_num_procs() { return omp_get_num_procs(); }
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

// OpenMP scalar-CSR SpMV (rows in parallel) -- cpu_baseline only
void orc_csr_spmv_omp(i32 nrow, const i32* rp, const i32* ci, const f64* val, f64 alpha, const f64* x, f64 beta, f64* y) {
#pragma omp parallel for schedule(static)
    for (i32 i = 0; i < nrow; ++i) {
        f64 s = 0.0;
        for (i32 j = rp[i]; j < rp[i + 1]; ++j) s += val[j] * x[ci[j]];
        y[i] = alpha * s + beta * y[i];
    }
}


// ---- DEM contact sweep (BUILD-DEFINED; the reference has none) -----------------
// linear spring-dashpot normal contact between equal spheres and against the unit-box walls.
// `brute` != 0: all pairs O(P^2) (independent cross-check of the cell list for small P).
static void pair_force(const f64* x, const f64* v, i32 i, i32 j, f64 R, f64 kn, f64 gn, f64* f) {
    f64 rx = x[3 * i] - x[3 * j], ry = x[3 * i + 1] - x[3 * j + 1], rz = x[3 * i + 2] - x[3 * j + 2];
    f64 d2 = rx * rx + ry * ry + rz * rz;
    if (d2 >= 4.0 * R * R || d2 == 0.0) return;
    f64 dist = std::sqrt(d2), inv = 1.0 / dist;
    f64 nx = rx * inv, ny = ry * inv, nz = rz * inv;
    f64 vn = (v[3 * i] - v[3 * j]) * nx + (v[3 * i + 1] - v[3 * j + 1]) * ny + (v[3 * i + 2] - v[3 * j + 2]) * nz;
    f64 fm = kn * (2.0 * R - dist) - gn * vn;
    f[0] += fm * nx; f[1] += fm * ny; f[2] += fm * nz;
}

void orc_dem_forces(i32 P, const f64* x, const f64* v, f64 R, f64 mass, f64 kn, f64 gn, i32 brute, f64* acc,
                    long long* tested_pairs) {
    i32 ncell = (i32)std::floor(1.0 / (2.0 * R));
    if (ncell < 1) ncell = 1;
    if (ncell > 1024) ncell = 1024;
    const f64 inv_cell = (f64)ncell;
    auto cc = [&](f64 p) { i32 c = (i32)std::floor(p * inv_cell); return c < 0 ? 0 : (c >= ncell ? ncell - 1 : c); };
    std::vector<i32> cell(P), order(P);
    for (i32 i = 0; i < P; ++i) { cell[i] = cc(x[3 * i]) + ncell * (cc(x[3 * i + 1]) + ncell * cc(x[3 * i + 2])); order[i] = i; }
    std::stable_sort(order.begin(), order.end(), [&](i32 a, i32 b) { return cell[a] < cell[b]; });
    const long long nc3 = (long long)ncell * ncell * ncell;
    std::vector<i32> start(nc3 + 1, 0);
    for (i32 i = 0; i < P; ++i) start[cell[i] + 1]++;
    for (long long c = 0; c < nc3; ++c) start[c + 1] += start[c];
    long long tested = 0;
    for (i32 i = 0; i < P; ++i) {
        f64 f[3] = {0.0, 0.0, 0.0};
        if (brute) {
            for (i32 j = 0; j < P; ++j) if (j != i) { pair_force(x, v, i, j, R, kn, gn, f); ++tested; }
        } else {
            i32 cx = cc(x[3 * i]), cy = cc(x[3 * i + 1]), cz = cc(x[3 * i + 2]);
            for (i32 dz = -1; dz <= 1; ++dz) for (i32 dy = -1; dy <= 1; ++dy) for (i32 dx = -1; dx <= 1; ++dx) {
                i32 X = cx + dx, Y = cy + dy, Z = cz + dz;
                if (X < 0 || Y < 0 || Z < 0 || X >= ncell || Y >= ncell || Z >= ncell) continue;
                long long c = X + (long long)ncell * (Y + (long long)ncell * Z);
                for (i32 t = start[c]; t < start[c + 1]; ++t) { i32 j = order[t]; if (j != i) { pair_force(x, v, i, j, R, kn, gn, f); ++tested; } }
            }
        }
        for (int d = 0; d < 3; ++d) {
            f64 lo = R - x[3 * i + d];
            if (lo > 0.0) f[d] += kn * lo - gn * v[3 * i + d];
            f64 hi = x[3 * i + d] + R - 1.0;
            if (hi > 0.0) f[d] -= kn * hi + gn * v[3 * i + d];
        }
        acc[3 * i] = f[0] / mass; acc[3 * i + 1] = f[1] / mass; acc[3 * i + 2] = f[2] / mass;
    }
    if (tested_pairs) *tested_pairs = tested;
}

}  // extern "C"
