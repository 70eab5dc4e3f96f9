/* Recursive-coordinate-bisection helpers shared by the partitioner and the patch / aggregate builders (host/partition.c,
 * patch.c, rowpatch.c, slotpatch.c, pc_twolevel.c): the total order on points along one axis (ties broken by index, so
 * that every build of a schedule is deterministic), quickselect on an index array, the longest axis of a point set. */
#ifndef DFL_HOST_RCB_H
#define DFL_HOST_RCB_H
#include "dedflow.h"

static inline int key_less(const f64* c, int ax, index_type a, index_type b) {
    f64 va = c[(size_t)a * 3 + ax], vb = c[(size_t)b * 3 + ax];
    return va < vb || (va == vb && a < b);
}
/* permutes idx[0..n) so that idx[k] is the k-th point along axis ax, smaller ones before it, larger ones after */
static inline void select_kth(const f64* c, int ax, index_type* idx, index_type n, index_type k) {
    index_type lo = 0, hi = n - 1;
    while (lo < hi) {
        index_type p = idx[lo + (hi - lo) / 2], i = lo, j = hi;
        while (i <= j) {
            while (key_less(c, ax, idx[i], p)) ++i;
            while (key_less(c, ax, p, idx[j])) --j;
            if (i <= j) { index_type t = idx[i]; idx[i] = idx[j]; idx[j] = t; ++i; --j; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}

#endif
