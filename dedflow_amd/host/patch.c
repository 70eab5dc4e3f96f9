/* Patch schedule for the Jacobian assembly (assembly schedule mode 2).
 *
 * The colored scatter of the reference (and of modes 0/1 here) read-modify-writes every 4x4
 * block once per contributing tet: 16 blocks x 256 B per tet = 41 GB at 10M tets, although the
 * matrix has only 25.6M blocks (3.3 GB): every block is touched 6.3 times on average.  Here the
 * tets are clustered into spatial patches (recursive coordinate bisection down to `leaf` tets,
 * split further until the patch's distinct blocks fit `slot_cap` LDS slots); a workgroup sums all
 * contributions of its patch in an LDS table and read-modify-writes each of the patch's blocks
 * ONCE.  Patches are colored (balanced greedy over 64-bit node masks) so that concurrently
 * running patches share no node -- still a color-partitioned, lock-free scatter into the CSR
 * values, one level up.  Host-side, OpenMP tasks; deterministic (patches are identified by their
 * position in the RCB order).
 */
#include <string.h>
#include <omp.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"
#include "rcb.h"

typedef struct { index_type lo, hi; } Range;
typedef struct {
    const f64* c;          /* centroids */
    index_type* idx;       /* element permutation (RCB order) */
    const index_type* ien; /* host connectivity */
    const index_type *rp, *ci; /* host nodal pattern */
    index_type leaf, cap;
    int cap_on_nodes;      /* 0: cap bounds the distinct (row,col) blocks of a patch, 1: its distinct nodes */
    Range* out;            /* emitted patches */
    index_type nout, capout;
} Ctx;

static int cmp_i32(const void* a, const void* b) {
    index_type x = *(const index_type*)a, y = *(const index_type*)b;
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

/* distinct blocks touched by elements idx[lo..hi); keys (sorted unique) returned in `keys` if non-NULL */
static index_type patch_blocks(const Ctx* x, index_type lo, index_type hi, index_type* keys) {
    index_type n = 0;
    for (index_type e = lo; e < hi; ++e) {
        const index_type* nd = x->ien + (size_t)x->idx[e] * 4;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) keys[n++] = find_nz(x->rp, x->ci, nd[a], nd[b]);
    }
    qsort(keys, (size_t)n, sizeof(index_type), cmp_i32);
    index_type m = 0;
    for (index_type i = 0; i < n; ++i)
        if (i == 0 || keys[i] != keys[i - 1]) keys[m++] = keys[i];
    return m;
}

/* distinct nodes of elements idx[lo..hi); sorted unique ids returned in `keys` (room for 4 per element) */
static index_type patch_nodes(const Ctx* x, index_type lo, index_type hi, index_type* keys) {
    index_type n = 0;
    for (index_type e = lo; e < hi; ++e) {
        const index_type* nd = x->ien + (size_t)x->idx[e] * 4;
        for (int a = 0; a < 4; ++a) keys[n++] = nd[a];
    }
    qsort(keys, (size_t)n, sizeof(index_type), cmp_i32);
    index_type m = 0;
    for (index_type i = 0; i < n; ++i)
        if (i == 0 || keys[i] != keys[i - 1]) keys[m++] = keys[i];
    return m;
}


static void emit(Ctx* x, index_type lo, index_type hi) {
#pragma omp critical(dfl_patch_emit)
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
        index_type* keys = (index_type*)malloc(sizeof(index_type) * (size_t)n * 16);
        index_type nb = x->cap_on_nodes ? patch_nodes(x, lo, hi, keys) : patch_blocks(x, lo, hi, keys);
        free(keys);
        if (nb <= x->cap || n <= 1) { emit(x, lo, hi); return; }
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
    /* cut at a multiple of the leaf size so that leaves come out full (a 64-tet patch fills a 256-thread
       workgroup; plain halving leaves them 60 % full on average) */
    index_type half = n / 2;
    if (x->cap_on_nodes && n > x->leaf) {
        const index_type nleaf = (n + x->leaf - 1) / x->leaf;
        half = (nleaf / 2) * x->leaf;
    }
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

PatchSched* DflBuildPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap) {
    const index_type T = mesh->num_tet, N = mesh->num_node;
    const index_type* ien = mesh->host->ien;
    const f64* xg = mesh->host->xg;
    PatchSched* ps = (PatchSched*)CdamMallocHost(SIZE_OF(PatchSched));
    memset(ps, 0, sizeof *ps);
    ps->attr = spy;

    index_type* rp = (index_type*)malloc(sizeof(index_type) * ((size_t)N + 1));
    index_type* ci = (index_type*)malloc(sizeof(index_type) * (size_t)spy->nnz);
    HIPGUARD(hipMemcpy(rp, spy->row_ptr, sizeof(index_type) * ((size_t)N + 1), D2H));
    HIPGUARD(hipMemcpy(ci, spy->col_ind, sizeof(index_type) * (size_t)spy->nnz, D2H));
    f64* c = (f64*)malloc(sizeof(f64) * (size_t)T * 3);
    index_type* idx = (index_type*)malloc(sizeof(index_type) * (size_t)T);
#pragma omp parallel for schedule(static) num_threads(8)
    for (index_type e = 0; e < T; ++e) {
        for (int d = 0; d < 3; ++d) {
            f64 s = 0.0;
            for (int a = 0; a < 4; ++a) s += xg[(size_t)ien[(size_t)e * 4 + a] * 3 + d];
            c[(size_t)e * 3 + d] = 0.25 * s;
        }
        idx[e] = e;
    }
    /* a GPU box exposes every hardware thread of the host but grants one rank a ~16-core share:
       an uncapped OpenMP team (256 spinning threads) made the task tree 30x slower */
    int nt = omp_get_max_threads();
    if (getenv("DFL_HOST_THREADS")) nt = atoi(getenv("DFL_HOST_THREADS"));
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    const int verbose = getenv("DFL_PATCH_VERBOSE") != NULL;
    double t0 = omp_get_wtime();
    if (verbose) fprintf(stderr, "[patch] threads=%d T=%d leaf=%d cap=%d\n", nt, T, leaf, slot_cap);
    Ctx x = {c, idx, ien, rp, ci, leaf, slot_cap, 0, NULL, 0, 1024};
    x.out = (Range*)malloc(sizeof(Range) * (size_t)x.capout);
#pragma omp parallel num_threads(nt)
#pragma omp single
    split(&x, 0, T);
    qsort(x.out, (size_t)x.nout, sizeof(Range), cmp_range);
    const index_type P = x.nout;
    if (verbose) fprintf(stderr, "[patch] clustering: %d patches in %.2f s\n", P, omp_get_wtime() - t0);

    /* per patch: sorted unique block list + local slot of every (elem,a,b) */
    index_type* nblk = (index_type*)malloc(sizeof(index_type) * (size_t)P);
    index_type** keys_of = (index_type**)malloc(sizeof(index_type*) * (size_t)P);
    uint16_t* lslot_rcb = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)T * 16); /* indexed by RCB position */
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, hi = x.out[p].hi;
        index_type* keys = (index_type*)malloc(sizeof(index_type) * (size_t)(hi - lo) * 16);
        const index_type nb = patch_blocks(&x, lo, hi, keys);
        nblk[p] = nb;
        keys_of[p] = keys;
        for (index_type e = lo; e < hi; ++e) {
            const index_type* nd = ien + (size_t)idx[e] * 4;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    index_type nz = find_nz(rp, ci, nd[a], nd[b]);
                    index_type l = 0, h = nb - 1; /* position in the sorted unique list */
                    while (l < h) {
                        index_type mid = (l + h) >> 1;
                        if (keys[mid] < nz) l = mid + 1; else h = mid;
                    }
                    lslot_rcb[(size_t)e * 16 + a * 4 + b] = (uint16_t)l;
                }
        }
    }

    if (verbose) fprintf(stderr, "[patch] slot maps at %.2f s\n", omp_get_wtime() - t0);
    /* balanced greedy coloring of the patches over node masks */
    u64* node_mask = (u64*)calloc((size_t)N, sizeof(u64));
    u8* pcolor = (u8*)malloc((size_t)P);
    int64_t load[64];
    memset(load, 0, sizeof load);
    int nopen = 1;
    for (index_type p = 0; p < P; ++p) {
        u64 used = 0;
        for (index_type e = x.out[p].lo; e < x.out[p].hi; ++e) {
            const index_type* nd = ien + (size_t)idx[e] * 4;
            used |= node_mask[nd[0]] | node_mask[nd[1]] | node_mask[nd[2]] | node_mask[nd[3]];
        }
        int best = -1;
        for (int k = 0; k < nopen; ++k)
            if (!((used >> k) & 1ULL) && (best < 0 || load[k] < load[best])) best = k;
        if (best < 0) {
            if (nopen == 64) { /* patches around a vertex shared by very many tets: more conflict classes than a 64-bit mask */
                fprintf(stderr, "tet-patch schedule (assembly schedule 2) cannot hold this mesh: more than 64 patch colors (patch %d)\n", p);
                for (index_type q = 0; q < P; ++q) free(keys_of[q]);
                free(pcolor); free(node_mask); free(lslot_rcb); free(keys_of); free(nblk); free(x.out); free(idx); free(c);
                free(ci); free(rp);
                CdamFreeHost(ps, SIZE_OF(PatchSched));
                return NULL;
            }
            best = nopen++;
        }
        pcolor[p] = (u8)best;
        load[best] += x.out[p].hi - x.out[p].lo;
        const u64 bit = 1ULL << best;
        for (index_type e = x.out[p].lo; e < x.out[p].hi; ++e) {
            const index_type* nd = ien + (size_t)idx[e] * 4;
            node_mask[nd[0]] |= bit; node_mask[nd[1]] |= bit; node_mask[nd[2]] |= bit; node_mask[nd[3]] |= bit;
        }
    }

    if (verbose) fprintf(stderr, "[patch] %d colors at %.2f s\n", nopen, omp_get_wtime() - t0);
    /* final order: color-major, then RCB position */
    ps->num_patch = P;
    ps->num_color = nopen;
    ps->color_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nopen + 1));
    memset(ps->color_offset, 0, sizeof(index_type) * (size_t)(nopen + 1));
    for (index_type p = 0; p < P; ++p) ps->color_offset[pcolor[p] + 1]++;
    for (int k = 0; k < nopen; ++k) ps->color_offset[k + 1] += ps->color_offset[k];
    index_type* order = (index_type*)malloc(sizeof(index_type) * (size_t)P);
    index_type cur[65];
    memcpy(cur, ps->color_offset, sizeof(index_type) * (size_t)(nopen + 1));
    for (index_type p = 0; p < P; ++p) order[cur[pcolor[p]]++] = p;

    index_type* eoff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    index_type* boff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    eoff[0] = 0;
    boff[0] = 0;
    int64_t tot = 0;
    index_type maxs = 0;
    for (index_type q = 0; q < P; ++q) {
        const index_type p = order[q];
        eoff[q + 1] = eoff[q] + (x.out[p].hi - x.out[p].lo);
        tot += nblk[p];
        ASSERT(tot < 2147483647LL);
        boff[q + 1] = (index_type)tot;
        if (nblk[p] > maxs) maxs = nblk[p];
    }
    ps->max_slots = maxs;
    ps->total_slots = tot;
    index_type* ien_p = (index_type*)malloc(sizeof(index_type) * (size_t)T * 4);
    uint16_t* lslot = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)T * 16);
    index_type* blk_nz = (index_type*)malloc(sizeof(index_type) * (size_t)(tot > 0 ? tot : 1));
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type q = 0; q < P; ++q) {
        const index_type p = order[q];
        const index_type lo = x.out[p].lo, n = x.out[p].hi - lo;
        for (index_type k = 0; k < n; ++k) {
            memcpy(ien_p + ((size_t)eoff[q] + k) * 4, ien + (size_t)idx[lo + k] * 4, 4 * sizeof(index_type));
            memcpy(lslot + ((size_t)eoff[q] + k) * 16, lslot_rcb + ((size_t)lo + k) * 16, 16 * sizeof(uint16_t));
        }
        memcpy(blk_nz + boff[q], keys_of[p], sizeof(index_type) * (size_t)nblk[p]);
        free(keys_of[p]);
    }

    ps->d_eoff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_boff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_ien = (index_type*)CdamMallocDevice((ptrdiff_t)T * 4 * SIZE_OF(index_type));
    ps->d_lslot = (uint16_t*)CdamMallocDevice((ptrdiff_t)T * 16 * (ptrdiff_t)sizeof(uint16_t));
    ps->d_blk_nz = (index_type*)CdamMallocDevice((ptrdiff_t)(tot > 0 ? tot : 1) * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(ps->d_eoff, eoff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_boff, boff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_ien, ien_p, sizeof(index_type) * (size_t)T * 4, H2D));
    HIPGUARD(hipMemcpy(ps->d_lslot, lslot, sizeof(uint16_t) * (size_t)T * 16, H2D));
    HIPGUARD(hipMemcpy(ps->d_blk_nz, blk_nz, sizeof(index_type) * (size_t)tot, H2D));

    if (verbose) fprintf(stderr, "[patch] uploaded at %.2f s (max slots %d, total %lld)\n", omp_get_wtime() - t0, maxs, (long long)tot);
    free(blk_nz); free(lslot); free(ien_p); free(boff); free(eoff); free(order); free(pcolor); free(node_mask);
    free(lslot_rcb); free(keys_of); free(nblk); free(x.out); free(idx); free(c); free(ci); free(rp);
    return ps;
}

void DflFreePatchSchedule(PatchSched* ps) {
    if (!ps) return;
    CdamFreeDevice(ps->d_eoff, 0); CdamFreeDevice(ps->d_boff, 0); CdamFreeDevice(ps->d_ien, 0);
    CdamFreeDevice(ps->d_lslot, 0); CdamFreeDevice(ps->d_blk_nz, 0); CdamFreeDevice(ps->d_egeo, 0);
    if (ps->color_offset) CdamFreeHost(ps->color_offset, 0);
    CdamFreeHost(ps, SIZE_OF(PatchSched));
}


/* ---- RHS patches -------------------------------------------------------------------------
 * The colored RHS scatter fetches 4 node records and read-modify-writes 4 residual records per
 * tet, and inside a color no two tets share a node, so none of it is reused on chip (~1 KB/tet of
 * HBM traffic for 404 B/tet of algorithmic bytes).  Here spatial patches of <= 64 tets stage their
 * <= node_cap node records in LDS once, sum the residual of every patch node in a FIXED order
 * (per-patch adjacency lists) and write one partial record per patch node; a second kernel adds,
 * again in fixed order, the partials of every node into F.  No colors, no atomics, two launches,
 * bitwise reproducible. */
/* pad_tets / pad_nodes > 0: fixed-stride ("padded") layout for the wave-per-patch kernel -- patch p owns tet slots
 * [p*pad_tets, (p+1)*pad_tets) and node slots [p*pad_nodes, (p+1)*pad_nodes) (unused node slots hold -1), so every index
 * array of a patch is addressed from the patch id alone and the loads of a patch form a two-hop chain
 * (lists -> node records) instead of three (offsets -> lists -> records); d_cnt[p] = num_tets | num_nodes << 16. */
RhsPatchSched* DflBuildRhsPatchSchedule(Mesh3D* mesh, index_type leaf, index_type node_cap, index_type pad_tets, index_type pad_nodes) {
    const index_type T = mesh->num_tet, N = mesh->num_node;
    const index_type* ien = mesh->host->ien;
    const f64* xg = mesh->host->xg;
    ASSERT(leaf >= 1 && leaf <= dfl_rhs_patch_max_tets() && node_cap >= 4 && node_cap <= dfl_rhs_patch_max_nodes());
    const int padded = pad_tets > 0 && pad_nodes > 0;
    ASSERT(!padded || (leaf <= pad_tets && node_cap <= pad_nodes));
    RhsPatchSched* ps = (RhsPatchSched*)CdamMallocHost(SIZE_OF(RhsPatchSched));
    memset(ps, 0, sizeof *ps);
    f64* c = (f64*)malloc(sizeof(f64) * (size_t)T * 3);
    index_type* idx = (index_type*)malloc(sizeof(index_type) * (size_t)T);
#pragma omp parallel for schedule(static) num_threads(8)
    for (index_type e = 0; e < T; ++e) {
        for (int d = 0; d < 3; ++d) {
            f64 s = 0.0;
            for (int a = 0; a < 4; ++a) s += xg[(size_t)ien[(size_t)e * 4 + a] * 3 + d];
            c[(size_t)e * 3 + d] = 0.25 * s;
        }
        idx[e] = e;
    }
    int nt = omp_get_max_threads();
    if (getenv("DFL_HOST_THREADS")) nt = atoi(getenv("DFL_HOST_THREADS"));
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    const int verbose = getenv("DFL_PATCH_VERBOSE") != NULL;
    double t0 = omp_get_wtime();
    Ctx x = {c, idx, ien, NULL, NULL, leaf, node_cap, 1, NULL, 0, 1024};
    x.out = (Range*)malloc(sizeof(Range) * (size_t)x.capout);
#pragma omp parallel num_threads(nt)
#pragma omp single
    split(&x, 0, T);
    qsort(x.out, (size_t)x.nout, sizeof(Range), cmp_range);
    const index_type P = x.nout;

    /* per patch: node list, local connectivity, adjacency (node -> (tet,a) in ascending tet order) */
    index_type* eoff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    index_type* noff = (index_type*)malloc(sizeof(index_type) * ((size_t)P + 1));
    index_type* nn_of = (index_type*)malloc(sizeof(index_type) * (size_t)P);
    index_type** nodes_of = (index_type**)malloc(sizeof(index_type*) * (size_t)P);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, hi = x.out[p].hi;
        index_type* keys = (index_type*)malloc(sizeof(index_type) * (size_t)(hi - lo) * 4);
        nn_of[p] = patch_nodes(&x, lo, hi, keys);
        nodes_of[p] = keys;
    }
    eoff[0] = noff[0] = 0;
    int64_t totn = 0;
    for (index_type p = 0; p < P; ++p) {
        eoff[p + 1] = padded ? (p + 1) * pad_tets : eoff[p] + (x.out[p].hi - x.out[p].lo);
        totn += padded ? pad_nodes : nn_of[p];
        ASSERT(totn < 2147483647LL && (int64_t)eoff[p + 1] * 4 < 2147483647LL);
        noff[p + 1] = (index_type)totn;
    }
    const size_t tslots = padded ? (size_t)P * (size_t)pad_tets : (size_t)T;
    index_type* pnode = (index_type*)malloc(sizeof(index_type) * (size_t)(totn > 0 ? totn : 1));
    memset(pnode, 0xff, sizeof(index_type) * (size_t)(totn > 0 ? totn : 1)); /* -1 = unused slot (padded layout) */
    u8* lien = (u8*)calloc(tslots * 4 + 4, 1);
    uint16_t* adj = (uint16_t*)calloc(tslots * 4 + 4, sizeof(uint16_t));
    uint16_t* adj_start = (uint16_t*)calloc((size_t)totn + (size_t)P + 1, sizeof(uint16_t));
    index_type* cnt_of = (index_type*)malloc(sizeof(index_type) * (size_t)(P > 0 ? P : 1));
    /* lane-per-tet kernel (64-tet patches): two-level ordered sum -- every sub-list has exactly 4 entries (padding = slot
       256, which holds 0.0), so all lanes of the first level do the same work whatever the valence of their node */
    const int sublists = padded && pad_tets == 64 && pad_nodes == 64;
    uint16_t* sub4 = NULL;
    uint16_t* sub_start = NULL;
    if (sublists) {
        sub4 = (uint16_t*)malloc(sizeof(uint16_t) * 512 * (size_t)(P > 0 ? P : 1));
        for (size_t i = 0; i < 512 * (size_t)P; ++i) sub4[i] = 256;
        sub_start = (uint16_t*)calloc((size_t)(pad_nodes + 1) * (size_t)(P > 0 ? P : 1), sizeof(uint16_t));
    }
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, ne = x.out[p].hi - lo, nn = nn_of[p];
        const index_type* keys = nodes_of[p];
        cnt_of[p] = ne | (nn << 16);
        memcpy(pnode + noff[p], keys, sizeof(index_type) * (size_t)nn);
        uint16_t cnt[256]; /* node_cap <= 255: local node ids are bytes */
        memset(cnt, 0, sizeof cnt);
        for (index_type k = 0; k < ne; ++k) {
            const index_type* nd = ien + (size_t)idx[lo + k] * 4;
            for (int a = 0; a < 4; ++a) {
                index_type l = 0, h = nn - 1;
                while (l < h) {
                    index_type mid = (l + h) >> 1;
                    if (keys[mid] < nd[a]) l = mid + 1; else h = mid;
                }
                lien[((size_t)eoff[p] + k) * 4 + a] = (u8)l;
                cnt[l]++;
            }
        }
        uint16_t* st = adj_start + (size_t)noff[p] + p; /* nn + 1 entries */
        st[0] = 0;
        for (index_type k = 0; k < nn; ++k) st[k + 1] = (uint16_t)(st[k] + cnt[k]);
        uint16_t cur[256];
        memcpy(cur, st, sizeof(uint16_t) * (size_t)nn);
        for (index_type k = 0; k < ne; ++k)
            for (int a = 0; a < 4; ++a) {
                const u8 l = lien[((size_t)eoff[p] + k) * 4 + a];
                adj[(size_t)eoff[p] * 4 + cur[l]++] = (uint16_t)(k * 4 + a);
            }
        if (sublists) {
            uint16_t* s4 = sub4 + (size_t)p * 512;
            uint16_t* ss = sub_start + (size_t)p * (size_t)(pad_nodes + 1);
            index_type sidx = 0;
            for (index_type k = 0; k < nn; ++k) {
                ss[k] = (uint16_t)sidx;
                for (index_type q = st[k]; q < st[k + 1]; q += 4, ++sidx) {
                    ASSERT(sidx < 128);
                    for (index_type i = 0; i < 4 && q + i < st[k + 1]; ++i) s4[sidx * 4 + i] = adj[(size_t)eoff[p] * 4 + q + i];
                }
            }
            for (index_type k = nn; k <= pad_nodes; ++k) ss[k] = (uint16_t)sidx;
        }
    }
    /* node -> its partial records (ascending patch order) */
    index_type* goff = (index_type*)calloc((size_t)N + 1, sizeof(index_type));
    for (int64_t i = 0; i < totn; ++i) if (pnode[i] >= 0) goff[pnode[i] + 1]++;
    for (index_type n = 0; n < N; ++n) goff[n + 1] += goff[n];
    index_type* gidx = (index_type*)malloc(sizeof(index_type) * (size_t)(totn > 0 ? totn : 1));
    {
        index_type* cur = (index_type*)malloc(sizeof(index_type) * (size_t)N);
        memcpy(cur, goff, sizeof(index_type) * (size_t)N);
        for (int64_t i = 0; i < totn; ++i) if (pnode[i] >= 0) gidx[cur[pnode[i]]++] = (index_type)i;
        free(cur);
    }
    ps->num_patch = P;
    ps->total_nodes = (index_type)totn;
    ps->d_eoff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_noff = (index_type*)CdamMallocDevice(((ptrdiff_t)P + 1) * SIZE_OF(index_type));
    ps->d_pnode = (index_type*)CdamMallocDevice((ptrdiff_t)(totn > 0 ? totn : 1) * SIZE_OF(index_type));
    ps->pad_tets = padded ? pad_tets : 0;
    ps->pad_nodes = padded ? pad_nodes : 0;
    ps->d_cnt = (index_type*)CdamMallocDevice((ptrdiff_t)(P > 0 ? P : 1) * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(ps->d_cnt, cnt_of, sizeof(index_type) * (size_t)P, H2D));
    ps->d_lien = (u8*)CdamMallocDevice((ptrdiff_t)tslots * 4 + 4);
    ps->d_adj = (uint16_t*)CdamMallocDevice(((ptrdiff_t)tslots * 4 + 4) * (ptrdiff_t)sizeof(uint16_t));
    ps->d_adj_start = (uint16_t*)CdamMallocDevice(((ptrdiff_t)totn + P) * (ptrdiff_t)sizeof(uint16_t));
    ps->d_goff = (index_type*)CdamMallocDevice(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    ps->d_gidx = (index_type*)CdamMallocDevice((ptrdiff_t)(totn > 0 ? totn : 1) * SIZE_OF(index_type));
    ps->d_partial = (f64*)CdamMallocDevice((ptrdiff_t)(totn > 0 ? totn : 1) * 6 * SIZE_OF(f64));
    if (sublists) {
        ps->d_sub4 = (uint16_t*)CdamMallocDevice((ptrdiff_t)512 * (P > 0 ? P : 1) * (ptrdiff_t)sizeof(uint16_t));
        ps->d_sub_start = (uint16_t*)CdamMallocDevice((ptrdiff_t)(pad_nodes + 1) * (P > 0 ? P : 1) * (ptrdiff_t)sizeof(uint16_t));
        HIPGUARD(hipMemcpy(ps->d_sub4, sub4, sizeof(uint16_t) * 512 * (size_t)P, H2D));
        HIPGUARD(hipMemcpy(ps->d_sub_start, sub_start, sizeof(uint16_t) * (size_t)(pad_nodes + 1) * (size_t)P, H2D));
    }
    HIPGUARD(hipMemcpy(ps->d_eoff, eoff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_noff, noff, sizeof(index_type) * ((size_t)P + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_pnode, pnode, sizeof(index_type) * (size_t)totn, H2D));
    HIPGUARD(hipMemcpy(ps->d_lien, lien, tslots * 4, H2D));
    HIPGUARD(hipMemcpy(ps->d_adj, adj, sizeof(uint16_t) * tslots * 4, H2D));
    HIPGUARD(hipMemcpy(ps->d_adj_start, adj_start, sizeof(uint16_t) * ((size_t)totn + (size_t)P), H2D));
    HIPGUARD(hipMemcpy(ps->d_goff, goff, sizeof(index_type) * ((size_t)N + 1), H2D));
    HIPGUARD(hipMemcpy(ps->d_gidx, gidx, sizeof(index_type) * (size_t)totn, H2D));
    if (verbose) fprintf(stderr, "[rhspatch] %d patches, %lld patch nodes (%.2f per node) in %.2f s\n", P, (long long)totn,
                         (double)totn / (double)(N > 0 ? N : 1), omp_get_wtime() - t0);
    for (index_type p = 0; p < P; ++p) free(nodes_of[p]);
    free(cnt_of); free(sub4); free(sub_start);
    free(gidx); free(goff); free(adj_start); free(adj); free(lien); free(pnode);
    free(nodes_of); free(nn_of); free(noff); free(eoff); free(x.out); free(idx); free(c);
    return ps;
}

void DflFreeRhsPatchSchedule(RhsPatchSched* ps) {
    if (!ps) return;
    CdamFreeDevice(ps->d_eoff, 0); CdamFreeDevice(ps->d_noff, 0); CdamFreeDevice(ps->d_pnode, 0);
    CdamFreeDevice(ps->d_lien, 0); CdamFreeDevice(ps->d_adj, 0); CdamFreeDevice(ps->d_adj_start, 0);
    CdamFreeDevice(ps->d_goff, 0); CdamFreeDevice(ps->d_gidx, 0); CdamFreeDevice(ps->d_cnt, 0);
    CdamFreeDevice(ps->d_partial, 0); CdamFreeDevice(ps->d_sub4, 0); CdamFreeDevice(ps->d_sub_start, 0);
    CdamFreeHost(ps, SIZE_OF(RhsPatchSched));
}
