/* Row-owner patch schedule for the Jacobian assembly (assembly schedule mode 3).
 *
 * Modes 0-2 scatter element blocks into the CSR values, so every 128-byte block line is
 * read-modify-written once per contributing tet (modes 0/1, 6.3 x per block) or once per
 * contributing patch (mode 2, 2.3 x).  Here the NODES are clustered into spatial patches
 * (recursive coordinate bisection down to `leaf` nodes / `slot_cap` nodal nonzeros) and a
 * workgroup owns every matrix row of its nodes: it evaluates the (a, .) block row of each tet
 * touching an owned node a ("items" = (tet, a) pairs, 4 lanes each), sums them in an LDS table
 * that mirrors the owned CSR rows, and streams each row out ONCE -- no colors, no atomics in
 * HBM, one launch, and `J = contributions` can overwrite (beta = 0) so the separate
 * MatrixZero pass of src/main.c:44 folds into the write.
 * Work is still exactly 16 blocks per tet; only the (tet, a) prologue loads are repeated.
 * Tets are addressed by their position in the execution schedule, so the kernel shares the
 * schedule-ordered connectivity and geometry cache (ien_b, egeo_b) with the other kernels.
 */
#include <string.h>
#include <omp.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"
#include "rcb.h"

typedef struct { index_type lo, hi; } Range;
typedef struct {
    const f64* c;         /* node coordinates */
    index_type* idx;      /* node permutation (RCB order) */
    const index_type* rp; /* host nodal row pointer */
    index_type leaf, cap;
    Range* out;
    index_type nout, capout;
} Ctx;

static void emit(Ctx* x, index_type lo, index_type hi) {
#pragma omp critical(dfl_rowpatch_emit)
    {
        if (x->nout == x->capout) {
            x->capout *= 2;
            x->out = (Range*)realloc(x->out, sizeof(Range) * (size_t)x->capout);
        }
        x->out[x->nout].lo = lo;
        x->out[x->nout].hi = hi;
        x->nout++;
    }
}
static void split(Ctx* x, index_type lo, index_type hi) {
    const index_type n = hi - lo;
    if (n <= x->leaf) {
        int64_t slots = 0;
        for (index_type i = lo; i < hi; ++i) slots += x->rp[x->idx[i] + 1] - x->rp[x->idx[i]];
        if (slots <= x->cap || n <= 1) { emit(x, lo, hi); return; }
    }
    f64 bl[3] = {1e300, 1e300, 1e300}, bh[3] = {-1e300, -1e300, -1e300};
    for (index_type i = lo; i < hi; ++i)
        for (int d = 0; d < 3; ++d) {
            f64 v = x->c[(size_t)x->idx[i] * 3 + d];
            if (v < bl[d]) bl[d] = v;
            if (v > bh[d]) bh[d] = v;
        }
    int ax = 0;
    if (bh[1] - bl[1] > bh[ax] - bl[ax]) ax = 1;
    if (bh[2] - bl[2] > bh[ax] - bl[ax]) ax = 2;
    const index_type half = n / 2;
    select_kth(x->c, ax, x->idx + lo, n, half);
    if (n > 4096) {
#pragma omp task
        split(x, lo, lo + half);
#pragma omp task
        split(x, lo + half, hi);
#pragma omp taskwait
    } else {
        split(x, lo, lo + half);
        split(x, lo + half, hi);
    }
}
static int cmp_range(const void* a, const void* b) {
    index_type x = ((const Range*)a)->lo, y = ((const Range*)b)->lo;
    return (x > y) - (x < y);
}
static int cmp_i32(const void* a, const void* b) {
    index_type x = *(const index_type*)a, y = *(const index_type*)b;
    return (x > y) - (x < y);
}
typedef struct { index_type ea; index_type k; } Item; /* ea = tet*4 + a, k = local node of the patch */
static int cmp_item(const void* a, const void* b) {
    index_type x = ((const Item*)a)->ea, y = ((const Item*)b)->ea;
    return (x > y) - (x < y);
}
static index_type find_nz(const index_type* rp, const index_type* ci, index_type row, index_type col) {
    index_type lo = rp[row], hi = rp[row + 1] - 1;
    while (lo < hi) {
        index_type mid = (lo + hi) >> 1;
        if (ci[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo;
}

RowPatchSched* DflBuildRowPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap) {
    const index_type T = mesh->num_tet, N = mesh->num_node;
    const index_type* ien = mesh->host->ien;
    const f64* xg = mesh->host->xg;
    ASSERT((int64_t)T * 4 < 2147483647LL && "row-patch item ids are 32-bit");
    RowPatchSched* ps = (RowPatchSched*)CdamMallocHost(SIZE_OF(RowPatchSched));
    memset(ps, 0, sizeof *ps);
    ps->attr = spy;

    index_type* rp = (index_type*)malloc(sizeof(index_type) * ((size_t)N + 1));
    index_type* ci = (index_type*)malloc(sizeof(index_type) * (size_t)spy->nnz);
    HIPGUARD(hipMemcpy(rp, spy->row_ptr, sizeof(index_type) * ((size_t)N + 1), D2H));
    HIPGUARD(hipMemcpy(ci, spy->col_ind, sizeof(index_type) * (size_t)spy->nnz, D2H));
    int nt = omp_get_max_threads();
    if (getenv("DFL_HOST_THREADS")) nt = atoi(getenv("DFL_HOST_THREADS"));
    if (nt > 16) nt = 16; /* a GPU box grants one rank a ~16-core share of a much wider host */
    if (nt < 1) nt = 1;
    const int verbose = getenv("DFL_PATCH_VERBOSE") != NULL;
    double t0 = omp_get_wtime();

    /* node -> tets (counting sort; tets ascending inside a node) */
    index_type* vp = (index_type*)calloc((size_t)N + 1, sizeof(index_type));
    for (size_t i = 0; i < (size_t)T * 4; ++i) vp[ien[i] + 1]++;
    for (index_type n = 0; n < N; ++n) vp[n + 1] += vp[n];
    const index_type* sched_elem = ((MeshExt*)mesh->ext)->h_sched_elem;
    ASSERT(sched_elem && "Mesh3DGenerateColorBatch must run before the row-owner schedule is built");
    index_type* ve = (index_type*)malloc(sizeof(index_type) * (size_t)T * 4); /* (schedule position)*4 + a */
    {
        index_type* cur = (index_type*)malloc(sizeof(index_type) * (size_t)N);
        memcpy(cur, vp, sizeof(index_type) * (size_t)N);
        for (index_type pos = 0; pos < T; ++pos) { /* ve holds schedule positions, ascending inside a node */
            const index_type e = sched_elem[pos];
            for (int a = 0; a < 4; ++a) ve[cur[ien[(size_t)e * 4 + a]]++] = pos * 4 + a;
        }
        free(cur);
    }

    index_type* idx = (index_type*)malloc(sizeof(index_type) * (size_t)N);
    for (index_type n = 0; n < N; ++n) idx[n] = n;
    Ctx x = {xg, idx, rp, leaf, slot_cap, NULL, 0, 1024};
    x.out = (Range*)malloc(sizeof(Range) * (size_t)x.capout);
#pragma omp parallel num_threads(nt)
#pragma omp single
    split(&x, 0, N);
    qsort(x.out, (size_t)x.nout, sizeof(Range), cmp_range);
    const index_type P = x.nout;
    if (verbose) fprintf(stderr, "[rowpatch] %d patches of <= %d nodes / %d slots in %.2f s\n", P, leaf, slot_cap, omp_get_wtime() - t0);

    /* offsets: items (tet,a pairs) and slots (nodal nonzeros) per patch, RCB order */
    index_type* ioff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    index_type* soff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    ioff[0] = soff[0] = 0;
    index_type maxs = 0;
    for (index_type p = 0; p < P; ++p) {
        int64_t ni = 0, ns = 0;
        for (index_type i = x.out[p].lo; i < x.out[p].hi; ++i) {
            const index_type n = idx[i];
            ni += vp[n + 1] - vp[n];
            ns += rp[n + 1] - rp[n];
        }
        if (ns > 4095) { /* a single node with a row of > 4095 nonzeros: the LDS table holds 12-bit slot ids */
            fprintf(stderr, "row-owner schedule (assembly schedule 3) cannot hold this mesh: %lld nodal nonzeros in one node patch "
                            "(limit 4095), around node %d\n", (long long)ns, idx[x.out[p].lo]);
            free(soff); free(ioff); free(x.out); free(idx); free(ve); free(vp); free(ci); free(rp);
            CdamFreeHost(ps, SIZE_OF(RowPatchSched));
            return NULL;
        }
        ioff[p + 1] = (index_type)(ioff[p] + ni);
        soff[p + 1] = (index_type)(soff[p] + ns);
        if (ns > maxs) maxs = (index_type)ns;
    }
    ASSERT(ioff[P] == T * 4 && soff[P] == spy->nnz);
    index_type* item_ea = (index_type*)malloc(sizeof(index_type) * (size_t)T * 4);
    uint16_t* item_slot = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)T * 16);
    index_type* slot_nz = (index_type*)malloc(sizeof(index_type) * (size_t)spy->nnz);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, nn = x.out[p].hi - lo;
        index_type* nodes = idx + lo;
        qsort(nodes, (size_t)nn, sizeof(index_type), cmp_i32); /* rows of a patch in ascending node order */
        const index_type ni = ioff[p + 1] - ioff[p];
        Item* it = (Item*)malloc(sizeof(Item) * (size_t)(ni > 0 ? ni : 1));
        index_type* rowbase = (index_type*)malloc(sizeof(index_type) * (size_t)nn);
        index_type m = 0, sb = 0;
        for (index_type k = 0; k < nn; ++k) {
            const index_type n = nodes[k];
            rowbase[k] = sb;
            for (index_type z = rp[n]; z < rp[n + 1]; ++z) slot_nz[soff[p] + sb++] = z;
            for (index_type j = vp[n]; j < vp[n + 1]; ++j) { it[m].ea = ve[j]; it[m].k = k; ++m; }
        }
        qsort(it, (size_t)ni, sizeof(Item), cmp_item); /* the items of one tet adjacent: shared prologue lines */
        for (index_type j = 0; j < ni; ++j) {
            const index_type e = sched_elem[it[j].ea >> 2], n = nodes[it[j].k];
            const index_type* nd = ien + (size_t)e * 4;
            item_ea[ioff[p] + j] = it[j].ea;
            for (int b = 0; b < 4; ++b)
                item_slot[((size_t)ioff[p] + j) * 4 + b] = (uint16_t)(rowbase[it[j].k] + find_nz(rp, ci, n, nd[b]) - rp[n]);
        }
        free(rowbase);
        free(it);
    }
    ps->num_patch = P;
    ps->max_slots = maxs;
    ps->d_ioff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_soff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_item_ea = (index_type*)CdamMallocDevice((ptrdiff_t)T * 4 * SIZE_OF(index_type));
    ps->d_item_slot = (uint16_t*)CdamMallocDevice((ptrdiff_t)T * 16 * (ptrdiff_t)sizeof(uint16_t));
    ps->d_slot_nz = (index_type*)CdamMallocDevice((ptrdiff_t)spy->nnz * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(ps->d_ioff, ioff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_soff, soff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_item_ea, item_ea, sizeof(index_type) * (size_t)T * 4, H2D));
    HIPGUARD(hipMemcpy(ps->d_item_slot, item_slot, sizeof(uint16_t) * (size_t)T * 16, H2D));
    HIPGUARD(hipMemcpy(ps->d_slot_nz, slot_nz, sizeof(index_type) * (size_t)spy->nnz, H2D));
    if (verbose) fprintf(stderr, "[rowpatch] uploaded at %.2f s (max slots %d)\n", omp_get_wtime() - t0, maxs);
    free(slot_nz); free(item_slot); free(item_ea); free(soff); free(ioff); free(x.out); free(idx);
    free(ve); free(vp); free(ci); free(rp);
    return ps;
}

void DflFreeRowPatchSchedule(RowPatchSched* ps) {
    if (!ps) return;
    CdamFreeDevice(ps->d_ioff, 0); CdamFreeDevice(ps->d_soff, 0); CdamFreeDevice(ps->d_item_ea, 0);
    CdamFreeDevice(ps->d_item_slot, 0); CdamFreeDevice(ps->d_slot_nz, 0);
    CdamFreeHost(ps, SIZE_OF(RowPatchSched));
}
