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

// OpenMP scalar-CSR SpMV (rows in parallel) -- cpu_baseline only
void orc_csr_spmv_omp(i32 nrow, const i32* rp, const i32* ci, const f64* val, f64 alpha, const f64* x, f64 beta, f64* y) {
#pragma omp parallel for schedule(static)
    for (i32 i = 0; i < nrow; ++i) {
        f64 s = 0.0;
        for (i32 j = rp[i]; j < rp[i + 1]; ++j) s += val[j] * x[ci[j]];
        y[i] = alpha * s + beta * y[i];
    }
}

}  // extern "C"
