/* Element partitioning for multi-GPU runs.  The reference has no working partitioner
 * (src/partition.c:16-77 wraps METIS_PartMeshNodal behind USE_METIS and writes to a struct
 * field that is commented out, src/Mesh.h:23-25); METIS is not available offline.
 * Recursive coordinate bisection of tet centroids: split the longest axis of the current
 * bounding box at the weighted median so that part sizes follow the requested counts
 * (works for any num_part, not only powers of two). */
#include <string.h>
#include "dedflow.h"
#include "rcb.h"

static void centroid_bbox(const f64* c, const index_type* idx, index_type n, f64* lo, f64* hi) {
    for (int d = 0; d < 3; ++d) { lo[d] = 1e300; hi[d] = -1e300; }
    for (index_type i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            f64 v = c[(size_t)idx[i] * 3 + d];
            if (v < lo[d]) lo[d] = v;
            if (v > hi[d]) hi[d] = v;
        }
}

/* quickselect on (coordinate, id) keys so that ties are broken deterministically */

static void rcb(const f64* c, index_type* idx, index_type n, index_type part0, index_type nparts, index_type* epart) {
    if (nparts <= 1 || n == 0) {
        for (index_type i = 0; i < n; ++i) epart[idx[i]] = part0;
        return;
    }
    f64 lo[3], hi[3];
    centroid_bbox(c, idx, n, lo, hi);
    int ax = 0;
    if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
    if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
    index_type pl = nparts / 2;
    index_type nl = (index_type)(((int64_t)n * pl) / nparts);
    if (nl > 0 && nl < n) select_kth(c, ax, idx, n, nl);
    rcb(c, idx, nl, part0, pl, epart);
    rcb(c, idx + nl, n - nl, part0 + pl, nparts - pl, epart);
}

void DflPartitionRCB(index_type T, const index_type* ien, const f64* xg, index_type num_part, index_type* epart) {
    f64* c = (f64*)malloc((size_t)T * 3 * sizeof(f64));
    index_type* idx = (index_type*)malloc((size_t)T * sizeof(index_type));
    for (index_type e = 0; e < T; ++e) {
        for (int d = 0; d < 3; ++d) {
            f64 s = 0.0;
            for (int a = 0; a < 4; ++a) s += xg[(size_t)ien[(size_t)e * 4 + a] * 3 + d];
            c[(size_t)e * 3 + d] = 0.25 * s;
        }
        idx[e] = e;
    }
    rcb(c, idx, T, 0, num_part, epart);
    free(idx);
    free(c);
}
