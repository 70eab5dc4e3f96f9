/* Slot-owner patch schedule for the Jacobian assembly (assembly schedule mode 4, default).
 *
 * The colored scatter of the reference (src/assemble.cu:1362-1371 -> src/matrix_impl.cu:370-453) read-modify-writes
 * a 128-byte block line once per contributing tet.  Here the NODES are clustered into spatial patches (recursive
 * coordinate bisection) and one workgroup owns every CSR row of its nodes -- like schedule 3 -- but the sum of a
 * nodal nonzero ("slot") is formed in REGISTERS by the lane group that owns the slot:
 *   phase 1  one lane per tet touching the patch: geometry, convective shape derivatives and stabilisation
 *            parameters at the 4 quadrature points -> a 304-byte record in LDS;
 *   phase 2  one lane PAIR per slot walks the slot's list of (tet, a, b) contributions (this file builds it), each lane
 *            every other entry: evaluates the (a, b) block of each from the LDS records, adds them in list order, the
 *            two lanes exchange halves and write the 128-byte line once.  (Lane quads were the first form: with 4-6
 *            contributions per edge slot a quad idles a quarter of its lanes on the second trip; pairs: 2.31 -> 2.06 ms.)
 * No LDS or HBM atomics, one launch, fixed summation order (bitwise reproducible), no per-element geometry cache
 * (geometry is recomputed from the node records, so moving meshes need no invalidation).
 *
 * Layout produced here (all device arrays):
 *   hdr[p]      = {tet_off, num_tet | num_node << 16, pos_off, num_pos, group_off, trips_lo, trips_hi, node_off}
 *   pnode       [sum num_node]    global ids of the distinct nodes of the tets touching a patch, ascending: the kernel stages
 *                                 their (x, u) records in LDS ONCE per patch -- one lane per node, LDS-DMA one patch ahead --
 *                                 instead of gathering 4 records per (patch, tet) pair through L2 (~55 nodes against ~90 x 4)
 *   ptet_lid    [sum num_tet]     the four patch-LOCAL node ids (one byte each) of every (patch, tet) pair, ascending element
 *                                 id inside a patch (4 B instead of 16 B of global ids per pair: -0.36 GB per assembly at 10M tets)
 *   slot_nz     [positions]       nodal nonzero of every slot position; inside a patch the slots are sorted by contribution
 *                                 count (descending) and dealt to the waves in snake order, so that the 32 lane
 *                                 pairs of a wave see equal trip counts
 *   ldesc       lane-major contribution descriptors, (local tet << 4) | (a << 2) | b, 0xFFFF = none.  Position q of a patch
 *               belongs to pass q / 128, wave (q % 128) / 32, lane pair q % 32; lane 2 pair + j walks contributions j, j + 2,
 *               ... of the position's list (ascending local tet).  For every (pass, wave) of the patch, pass-major, with T =
 *               the largest trip count among the wave's 32 pairs (trips_lo / trips_hi: one byte per (pass, wave)):
 *               floor(T / 2) groups of [64 lanes] x {trip 2g, trip 2g + 1} (one 32-bit word per lane and group) and, for odd T, a
 *               TAIL of [64 lanes] x {trip T - 1} (one u16 per lane: 32 words; round 2 padded it to a whole group, a quarter of
 *               the descriptor bytes at the usual 3 trips).  Offsets count in units of one trip (32 words).  The kernel
 *               reads a lane's descriptors with coalesced word loads one patch ahead -- no offset list, no LDS staging.
 */
#include <string.h>
#include <omp.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"
#include "rcb.h"

#define SPLIT_MIN 12              /* contributions from which a (diagonal) slot is cut into four parts (one lane pair each) */
#define SLOT_LEADER 0x40000000    /* slot_nz flag: first pair of a split slot (sums the four parts, stores the line) */
#define SLOT_FOLLOWER 0x80000000u /* slot_nz flag: pairs 2-4 of a split slot (no store) */

typedef struct { index_type lo, hi; } Range;
typedef struct {
    const f64* c;          /* node coordinates */
    index_type* idx;       /* node permutation (RCB order) */
    const index_type* rp;  /* host nodal row pointer */
    const index_type* vp;  /* node -> tet list offsets */
    const index_type* ve;  /* tet*4 + a, ascending tet inside a node */
    const index_type* ien; /* connectivity (distinct nodes of a candidate patch) */
    index_type leaf, cap, tcap, ncap;
    Range* out;
    index_type nout, capout;
} Ctx;

static int cmp_i32(const void* a, const void* b) {
    index_type x = *(const index_type*)a, y = *(const index_type*)b;
    return (x > y) - (x < y);
}
static int cmp_range(const void* a, const void* b) {
    index_type x = ((const Range*)a)->lo, y = ((const Range*)b)->lo;
    return (x > y) - (x < y);
}
static void emit(Ctx* x, index_type lo, index_type hi) {
#pragma omp critical(dfl_slotpatch_emit)
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
/* distinct tets touching the nodes idx[lo..hi), and the distinct nodes of those tets (4096 / 16384 = "too many") */
static index_type count_tets(const Ctx* x, index_type lo, index_type hi, index_type* num_nodes) {
    index_type tmp[4096];
    index_type m = 0;
    *num_nodes = 16384;
    for (index_type i = lo; i < hi; ++i) {
        const index_type n = x->idx[i];
        for (index_type j = x->vp[n]; j < x->vp[n + 1]; ++j) {
            if (m == 4096) return 4096;
            tmp[m++] = x->ve[j] >> 2;
        }
    }
    qsort(tmp, (size_t)m, sizeof(index_type), cmp_i32);
    index_type u = 0;
    for (index_type i = 0; i < m; ++i)
        if (i == 0 || tmp[i] != tmp[i - 1]) tmp[u++] = tmp[i];
    if (u > 1024) return u;
    index_type nd[4096];
    for (index_type i = 0; i < u; ++i)
        for (int a = 0; a < 4; ++a) nd[4 * i + a] = x->ien[(size_t)tmp[i] * 4 + a];
    qsort(nd, (size_t)u * 4, sizeof(index_type), cmp_i32);
    index_type v = 0;
    for (index_type i = 0; i < 4 * u; ++i)
        if (i == 0 || nd[i] != nd[i - 1]) ++v;
    *num_nodes = v;
    return u;
}
static void split(Ctx* x, index_type lo, index_type hi) {
    const index_type n = hi - lo;
    if (n <= x->leaf) {
        int64_t slots = 0;
        for (index_type i = lo; i < hi; ++i) slots += x->rp[x->idx[i] + 1] - x->rp[x->idx[i]];
        int64_t items = 0; /* (tet, owned node) pairs: 4 contributions each, at most 8 per lane of the workgroup */
        for (index_type i = lo; i < hi; ++i) items += x->vp[x->idx[i] + 1] - x->vp[x->idx[i]];
        /* + 3 positions per node: a diagonal slot (>= SPLIT_MIN contributions) is walked by four lane pairs */
        index_type nnode = 0;
        if (n <= 1 || (slots + 3 * (int64_t)n <= x->cap && items <= 2 * DFL_SLOT_BLOCK && count_tets(x, lo, hi, &nnode) <= x->tcap &&
                       nnode <= x->ncap)) { emit(x, lo, hi); return; }
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
    /* Proportional cuts: a range of n nodes becomes ceil(n / leaf) leaves of (nearly) equal size, so that the leaves come out
       at the size asked for whatever n is (plain halving gives n / 2^k: 6.6 nodes = 119 slot positions at 10M tets, but 8.1
       nodes = 146 positions at 50M tets -- a second, nearly empty pass over the positions in every patch).  A leaf that
       breaks a cap is halved. */
    index_type half = n / 2;
    if (n > x->leaf) {
        const int64_t L = ((int64_t)n + x->leaf - 1) / x->leaf;
        half = (index_type)((int64_t)n * ((L + 1) / 2) / L);
        if (half < 1) half = 1;
        if (half >= n) half = n - 1;
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
static index_type rank_to_pos(index_type r, index_type np, index_type pass, index_type nwave, int heavy_last) {
    const index_type ps = r / pass, base = ps * pass;
    const index_type in_pass = (np - base < pass) ? np - base : pass; /* positions of this pass */
    const index_type nfull = in_pass / 32;                            /* its full chunks */
    const index_type chunk = (r - base) >> 5;
    if (!heavy_last) { /* round 2: full passes snake (even passes forward), the partial pass in rank order */
        if (in_pass < pass) return r;
        return base + ((ps & 1) ? nwave - 1 - chunk : chunk) * 32 + (r & 31);
    }
    if (chunk >= nfull) return r; /* the trailing partial chunk stays behind the full ones */
    const index_type wv = (ps & 1) ? chunk : nfull - 1 - chunk;
    return base + wv * 32 + (r & 31);
}
static index_type find_nz(const index_type* rp, const index_type* ci, index_type row, index_type col) {
    index_type lo = rp[row], hi = rp[row + 1] - 1;
    while (lo < hi) {
        index_type mid = (lo + hi) >> 1;
        if (ci[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* The shape limits of tet_lhs_slot_kernel (csrc/k_assemble2.hip) for ONE node patch: slot positions (one lane pair each, two
 * passes of DFL_SLOT_BLOCK / 2 pairs), tets touching the patch (one lane each in phase 1), trips of a lane pair (a header
 * byte), distinct nodes of those tets (one lane each, staged in LDS).  0 = fits; otherwise the reason is written to `why`.  The
 * recursive bisection stops at one node, so a single node of very high valence (> 256 tets, a row of > 252 nonzeros, > 64
 * nodes around it) is what can break them -- no cube mesh does, a mesh with a
 * fan of hundreds of tets around one vertex does.  Host arithmetic only (tests/test_abi_cpu.py calls it without a GPU). */
static int g_test_pos_limit = 0, g_test_tet_limit = 0;
/* test hook: lower the limits (0 = the kernel's own) so that an ordinary mesh exercises the refusal and the fall-back */
void DflSlotPatchSetTestLimits(int positions, int tets) {
    g_test_pos_limit = positions;
    g_test_tet_limit = tets;
}
int DflSlotPatchLimitCheck(int64_t num_positions, int64_t num_tets, int64_t num_nodes, int64_t max_contributions_of_a_position, char* why,
                           size_t why_len) {
    const int64_t pos_limit = g_test_pos_limit > 0 ? g_test_pos_limit : DFL_SLOT_BLOCK - 1;
    const int64_t tet_limit = g_test_tet_limit > 0 ? g_test_tet_limit : DFL_SLOT_BLOCK;
    if (num_positions > pos_limit) {
        if (why) snprintf(why, why_len, "%lld slot positions in one node patch (limit %lld)", (long long)num_positions, (long long)pos_limit);
        return 1;
    }
    if (num_tets > tet_limit) {
        if (why) snprintf(why, why_len, "%lld tets touch one node patch (limit %lld)", (long long)num_tets, (long long)tet_limit);
        return 2;
    }
    if (num_nodes > DFL_SLOT_NODES) {
        if (why) snprintf(why, why_len, "the tets touching one node patch have %lld distinct nodes (limit %d)", (long long)num_nodes, DFL_SLOT_NODES);
        return 4;
    }
    if ((max_contributions_of_a_position + 1) / 2 > 254) {
        if (why) snprintf(why, why_len, "%lld contributions to one slot position (limit 508)", (long long)max_contributions_of_a_position);
        return 3;
    }
    return 0;
}

/* NULL (after a message on stderr) when the mesh breaks a limit of the kernel: the caller falls back to the colored schedule */
SlotPatchSched* DflBuildSlotPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap, index_type tet_cap) {
    const index_type T = mesh->num_tet, N = mesh->num_node;
    const index_type* ien = mesh->host->ien;
    const f64* xg = mesh->host->xg;
    if ((int64_t)T * 16 >= 2147483647LL) {
        fprintf(stderr, "slot-owner schedule: %d tets need 64-bit contribution offsets (limit 134M tets per mesh / rank)\n", T);
        return NULL;
    }
    if (tet_cap > DFL_SLOT_BLOCK) tet_cap = DFL_SLOT_BLOCK;       /* one tet per lane in phase 1 (and 12-bit local tet ids in the descriptors) */
    if (slot_cap > DFL_SLOT_BLOCK - 1) slot_cap = DFL_SLOT_BLOCK - 1; /* one slot offset per lane */
    SlotPatchSched* ps = (SlotPatchSched*)CdamMallocHost(SIZE_OF(SlotPatchSched));
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
    const int heavy_last_g = getenv("DFL_SLOT_HEAVY_LAST") ? 1 : 0; /* A/B only: measured equal (2.17 against 2.16 ms, gpurun_out/r3h) */
    double t0 = omp_get_wtime();

    /* node -> (tet, a) lists, ascending tet inside a node */
    index_type* vp = (index_type*)calloc((size_t)N + 1, sizeof(index_type));
    for (size_t i = 0; i < (size_t)T * 4; ++i) vp[ien[i] + 1]++;
    for (index_type n = 0; n < N; ++n) vp[n + 1] += vp[n];
    index_type* ve = (index_type*)malloc(sizeof(index_type) * (size_t)T * 4);
    {
        index_type* cur = (index_type*)malloc(sizeof(index_type) * (size_t)N);
        memcpy(cur, vp, sizeof(index_type) * (size_t)N);
        for (index_type e = 0; e < T; ++e)
            for (int a = 0; a < 4; ++a) ve[cur[ien[(size_t)e * 4 + a]]++] = e * 4 + a;
        free(cur);
    }
    index_type* idx = (index_type*)malloc(sizeof(index_type) * (size_t)N);
    for (index_type n = 0; n < N; ++n) idx[n] = n;
    Ctx x = {xg, idx, rp, vp, ve, ien, leaf, slot_cap, tet_cap, DFL_SLOT_NODES, NULL, 0, 1024};
    x.out = (Range*)malloc(sizeof(Range) * (size_t)x.capout);
#pragma omp parallel num_threads(nt)
#pragma omp single
    split(&x, 0, N);
    qsort(x.out, (size_t)x.nout, sizeof(Range), cmp_range);
    const index_type P = x.nout;

    /* pass 1: per-patch tet lists and sizes */
    index_type** tets_of = (index_type**)malloc(sizeof(index_type*) * (size_t)P);
    index_type* nt_of = (index_type*)malloc(sizeof(index_type) * (size_t)P);
    index_type** nodes_of = (index_type**)malloc(sizeof(index_type*) * (size_t)P); /* distinct nodes of the patch's tets, ascending */
    index_type* nn_of = (index_type*)malloc(sizeof(index_type) * (size_t)P);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, nn = x.out[p].hi - lo;
        index_type* nodes = idx + lo;
        qsort(nodes, (size_t)nn, sizeof(index_type), cmp_i32); /* rows of a patch in ascending node order */
        index_type m = 0;
        for (index_type k = 0; k < nn; ++k) m += vp[nodes[k] + 1] - vp[nodes[k]];
        index_type* tl = (index_type*)malloc(sizeof(index_type) * (size_t)(m > 0 ? m : 1));
        m = 0;
        for (index_type k = 0; k < nn; ++k)
            for (index_type j = vp[nodes[k]]; j < vp[nodes[k] + 1]; ++j) tl[m++] = ve[j] >> 2;
        qsort(tl, (size_t)m, sizeof(index_type), cmp_i32);
        index_type u = 0;
        for (index_type i = 0; i < m; ++i)
            if (i == 0 || tl[i] != tl[i - 1]) tl[u++] = tl[i];
        tets_of[p] = tl;
        nt_of[p] = u;
        index_type* nl = (index_type*)malloc(sizeof(index_type) * 4 * (size_t)(u > 0 ? u : 1));
        for (index_type i = 0; i < u; ++i) memcpy(nl + 4 * (size_t)i, ien + (size_t)tl[i] * 4, 4 * sizeof(index_type));
        qsort(nl, (size_t)u * 4, sizeof(index_type), cmp_i32);
        index_type v = 0;
        for (index_type i = 0; i < 4 * u; ++i)
            if (i == 0 || nl[i] != nl[i - 1]) nl[v++] = nl[i];
        nodes_of[p] = nl;
        nn_of[p] = v;
    }
    int32_t* hdr = (int32_t*)calloc((size_t)P * 8 + 8, sizeof(int32_t));
    int64_t tot_t = 0, tot_s = 0, tot_n = 0;
    index_type maxt = 0, maxs = 0, maxn = 0;
    int broken = 0;
    char why[160] = {0};
    for (index_type p = 0; p < P && !broken; ++p) {
        int64_t ns = 0; /* slot POSITIONS: one per nodal nonzero + 3 more for every split diagonal slot */
        int64_t maxc = 0; /* most contributions any position of the patch can receive: its node's valence */
        for (index_type i = x.out[p].lo; i < x.out[p].hi; ++i) {
            ns += rp[idx[i] + 1] - rp[idx[i]];
            const index_type valence = vp[idx[i] + 1] - vp[idx[i]];
            if (valence >= SPLIT_MIN) ns += 3;
            const int64_t c = valence >= SPLIT_MIN ? (valence + 3) / 4 : valence;
            if (c > maxc) maxc = c;
        }
        /* (the bisection honours the caps down to single nodes; a single node can still break the kernel's own limits) */
        broken = DflSlotPatchLimitCheck(ns, nt_of[p], nn_of[p], maxc, why, sizeof why);
        if (broken) {
            fprintf(stderr, "slot-owner schedule (assembly schedule 4) cannot hold this mesh: %s, around node %d\n", why, idx[x.out[p].lo]);
            break;
        }
        hdr[8 * p + 0] = (int32_t)tot_t;
        hdr[8 * p + 1] = nt_of[p] | (nn_of[p] << 16);
        hdr[8 * p + 2] = (int32_t)tot_s;
        hdr[8 * p + 3] = (int32_t)ns;
        hdr[8 * p + 7] = (int32_t)tot_n;
        tot_t += nt_of[p];
        tot_s += ns;
        tot_n += nn_of[p];
        if (nn_of[p] > maxn) maxn = nn_of[p];
        if (nt_of[p] > maxt) maxt = nt_of[p];
        if (ns > maxs) maxs = (index_type)ns;
    }
    if (!broken && (tot_t >= 2147483647LL || tot_s >= 2147483647LL)) {
        fprintf(stderr, "slot-owner schedule: %lld patch-tet entries / %lld slot positions exceed 32-bit offsets\n", (long long)tot_t, (long long)tot_s);
        broken = 4;
    }
    if (broken) {
        for (index_type p = 0; p < P; ++p) { free(tets_of[p]); free(nodes_of[p]); }
        free(hdr); free(nt_of); free(tets_of); free(nn_of); free(nodes_of); free(x.out); free(idx); free(ve); free(vp); free(ci); free(rp);
        CdamFreeHost(ps, SIZE_OF(SlotPatchSched));
        return NULL;
    }
    ASSERT(tot_s >= spy->nnz);
    if (verbose)
        fprintf(stderr, "[slotpatch] %d patches (<= %d nodes / %d slots / %d tets): %.2f tets per patch-tet list entry per tet, max tets %d, "
                        "max slots %d, distinct nodes per patch mean %.1f max %d, %.2f s\n", P, leaf, slot_cap, tet_cap,
                (double)tot_t / (double)(T > 0 ? T : 1), maxt, maxs, (double)tot_n / (double)(P > 0 ? P : 1), maxn, omp_get_wtime() - t0);

    uint32_t* ptet_lid = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(tot_t > 0 ? tot_t : 1));
    index_type* pnode = (index_type*)malloc(sizeof(index_type) * (size_t)(tot_n > 0 ? tot_n : 1));
    index_type* slot_nz = (index_type*)malloc(sizeof(index_type) * (size_t)tot_s);
    index_type* coff = (index_type*)malloc(sizeof(index_type) * ((size_t)tot_s + 1));
    uint16_t* desc = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)T * 16);
    /* contribution counts per patch first (prefix over patches), then the fill */
    int64_t* cbase = (int64_t*)malloc(sizeof(int64_t) * ((size_t)P + 1));
    index_type maxc_all = 0;
    cbase[0] = 0;
    for (index_type p = 0; p < P; ++p) {
        int64_t items = 0;
        for (index_type i = x.out[p].lo; i < x.out[p].hi; ++i) items += vp[idx[i] + 1] - vp[idx[i]];
        cbase[p + 1] = cbase[p] + items * 4;
        hdr[8 * p + 4] = (int32_t)cbase[p];
        hdr[8 * p + 5] = (int32_t)(items * 4);
        if (items * 4 > maxc_all) maxc_all = (index_type)(items * 4);
    }
    ASSERT(cbase[P] == (int64_t)T * 16);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type lo = x.out[p].lo, nn = x.out[p].hi - lo;
        const index_type* nodes = idx + lo;
        const index_type ntp = nt_of[p], s0 = hdr[8 * p + 2];
        index_type ns = 0; /* real slots (nodal nonzeros of the patch rows); hdr[8 p + 3] counts positions */
        for (index_type k = 0; k < nn; ++k) ns += rp[nodes[k] + 1] - rp[nodes[k]];
        const index_type* tl = tets_of[p];
        {   /* node table of the patch and the patch-local ids of its tets' vertices */
            const index_type* nl = nodes_of[p];
            const index_type nnp = nn_of[p];
            memcpy(pnode + hdr[8 * p + 7], nl, sizeof(index_type) * (size_t)nnp);
            for (index_type k = 0; k < ntp; ++k) {
                uint32_t packed = 0;
                for (int b = 0; b < 4; ++b) {
                    const index_type g = ien[(size_t)tl[k] * 4 + b];
                    index_type l = 0, h = nnp - 1;
                    while (l < h) {
                        const index_type mid = (l + h) >> 1;
                        if (nl[mid] < g) l = mid + 1; else h = mid;
                    }
                    packed |= (uint32_t)l << (8 * b);
                }
                ptet_lid[(size_t)hdr[8 * p] + k] = packed;
            }
        }
        index_type* rowbase = (index_type*)malloc(sizeof(index_type) * (size_t)nn);
        index_type* cnt = (index_type*)calloc((size_t)ns, sizeof(index_type));
        index_type sb = 0;
        for (index_type k = 0; k < nn; ++k) { rowbase[k] = sb; sb += rp[nodes[k] + 1] - rp[nodes[k]]; }
        /* items of the patch in ascending (tet, a) order: merge of the per-node lists */
        index_type ni = (index_type)((cbase[p + 1] - cbase[p]) / 4);
        int64_t* it = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ni > 0 ? ni : 1)); /* (ea << 16) | k */
        index_type m = 0;
        for (index_type k = 0; k < nn; ++k)
            for (index_type j = vp[nodes[k]]; j < vp[nodes[k] + 1]; ++j) it[m++] = ((int64_t)ve[j] << 16) | k;
        /* int64 keys order by (tet, a) first: shell sort (a few hundred entries) */
        {
            for (index_type gap = m / 2; gap > 0; gap /= 2)
                for (index_type i = gap; i < m; ++i) {
                    int64_t v = it[i];
                    index_type j = i;
                    for (; j >= gap && it[j - gap] > v; j -= gap) it[j] = it[j - gap];
                    it[j] = v;
                }
        }
        index_type* islot = (index_type*)malloc(sizeof(index_type) * 4 * (size_t)(ni > 0 ? ni : 1));
        index_type* ilt = (index_type*)malloc(sizeof(index_type) * (size_t)(ni > 0 ? ni : 1));
        index_type cur_lt = 0;
        for (index_type i = 0; i < m; ++i) {
            const index_type ea = (index_type)(it[i] >> 16), k = (index_type)(it[i] & 0xffff);
            const index_type e = ea >> 2, n = nodes[k];
            while (tl[cur_lt] < e) ++cur_lt; /* tl ascending, items ascending */
            ilt[i] = cur_lt;
            for (int b = 0; b < 4; ++b) {
                const index_type s = rowbase[k] + find_nz(rp, ci, n, ien[(size_t)e * 4 + b]) - rp[n];
                islot[4 * i + b] = s;
                cnt[s]++;
            }
        }
        /* Positions.  A diagonal slot (>= SPLIT_MIN contributions; 24 in the interior of a Kuhn mesh against 4-6 for an
           edge slot) is cut into four consecutive parts, each walked by its own lane pair -- four adjacent pairs = half a
           16-lane DPP row, summed by two row shifts in the kernel -- so that every pair of a wave pass has 2-3 trips
           instead of the whole pass waiting 12 trips for its diagonal slots.  Ranks: the split slots first (4 ranks each,
           descending count), then the others by descending count; ranks are dealt to the waves in snake order in chunks
           of 32 pairs (a multiple of 4: the groups stay aligned). */
        const index_type np = hdr[8 * p + 3];
        index_type* first_rank = (index_type*)malloc(sizeof(index_type) * (size_t)(ns > 0 ? ns : 1));
        unsigned char* is_split = (unsigned char*)calloc((size_t)(ns > 0 ? ns : 1), 1);
        {
            index_type sidx = 0;
            for (index_type k = 0; k < nn; ++k) {
                const index_type dz = find_nz(rp, ci, nodes[k], nodes[k]) - rp[nodes[k]];
                if (vp[nodes[k] + 1] - vp[nodes[k]] >= SPLIT_MIN) is_split[sidx + dz] = 1;
                sidx += rp[nodes[k] + 1] - rp[nodes[k]];
            }
        }
        index_type maxc = 0;
        for (index_type s = 0; s < ns; ++s) if (cnt[s] > maxc) maxc = cnt[s];
        index_type* bucket = (index_type*)calloc(2 * ((size_t)maxc + 2), sizeof(index_type));
        index_type* bsplit = bucket, *bsingle = bucket + maxc + 2;
        index_type nsplit = 0;
        for (index_type s = 0; s < ns; ++s) {
            if (is_split[s]) { bsplit[maxc - cnt[s] + 1]++; nsplit++; }
            else bsingle[maxc - cnt[s] + 1]++;
        }
        for (index_type c = 0; c <= maxc; ++c) { bsplit[c + 1] += bsplit[c]; bsingle[c + 1] += bsingle[c]; }
        ASSERT(np == ns + 3 * nsplit);
        for (index_type s = 0; s < ns; ++s)
            first_rank[s] = is_split[s] ? 4 * bsplit[maxc - cnt[s]]++ : 4 * nsplit + bsingle[maxc - cnt[s]]++;
        const index_type pass = DFL_SLOT_BLOCK / 2, nwave = DFL_SLOT_BLOCK / 64; /* lane pairs per pass over the positions; waves */
        const int heavy_last = heavy_last_g;
        /* rank -> position.  Ranks are dealt to the waves in chunks of 32 lane pairs: full passes in snake order, the partial
           pass in rank order (heaviest chunk -- split slots, most contributions -- on wave 0).  DFL_SLOT_HEAVY_LAST=1 lays the
           full chunks of a pass out in reverse wave order instead, so that the heaviest chunk does not share a wave with
           phase 1 (waves 0 / 1): measured equal, kept for A/B -- the kernel is bound by none of its waves' instruction
           streams (profiles/r03_pmc_lhs.txt). */
#define RANK_TO_POS(r) rank_to_pos((r), np, pass, nwave, heavy_last)
        /* contribution counts in position order, then offsets */
        index_type* cpos = (index_type*)calloc((size_t)np + 1, sizeof(index_type));
        for (index_type s = 0; s < ns; ++s) {
            if (is_split[s])
                for (index_type v = 0; v < 4; ++v)
                    cpos[RANK_TO_POS(first_rank[s] + v) + 1] = (index_type)((int64_t)cnt[s] * (v + 1) / 4 - (int64_t)cnt[s] * v / 4);
            else cpos[RANK_TO_POS(first_rank[s]) + 1] = cnt[s];
        }
        for (index_type q = 0; q < np; ++q) cpos[q + 1] += cpos[q];
        for (index_type q = 0; q < np; ++q) coff[s0 + q] = (index_type)(cbase[p] + cpos[q]);
        {
            index_type sidx = 0;
            for (index_type k = 0; k < nn; ++k)
                for (index_type z = rp[nodes[k]]; z < rp[nodes[k] + 1]; ++z, ++sidx) {
                    if (is_split[sidx]) {
                        slot_nz[s0 + RANK_TO_POS(first_rank[sidx])] = z | SLOT_LEADER;
                        for (index_type v = 1; v < 4; ++v) slot_nz[s0 + RANK_TO_POS(first_rank[sidx] + v)] = z | SLOT_FOLLOWER;
                    } else slot_nz[s0 + RANK_TO_POS(first_rank[sidx])] = z;
                }
        }
        index_type* seen = (index_type*)calloc((size_t)(ns > 0 ? ns : 1), sizeof(index_type));
        for (index_type i = 0; i < m; ++i) {
            const index_type a = (index_type)((it[i] >> 16) & 3);
            for (int b = 0; b < 4; ++b) {
                const index_type s = islot[4 * i + b];
                index_type r = first_rank[s];
                if (is_split[s]) { /* part v holds contributions [c v / 4, c (v + 1) / 4) of the slot's ascending list */
                    const index_type k = seen[s]++;
                    index_type v = (index_type)(((int64_t)k * 4) / cnt[s]);
                    while ((int64_t)cnt[s] * (v + 1) / 4 <= k) ++v;
                    while ((int64_t)cnt[s] * v / 4 > k) --v;
                    r += v;
                }
                desc[cbase[p] + cpos[RANK_TO_POS(r)]++] = (uint16_t)((ilt[i] << 4) | (a << 2) | b);
            }
        }
#undef RANK_TO_POS
        free(seen); free(is_split); free(first_rank);
        index_type* pos_of = NULL;
        free(cpos); free(pos_of); free(bucket); free(ilt); free(islot); free(it); free(cnt); free(rowbase);
        free(tets_of[p]);
        free(nodes_of[p]);
    }
    coff[tot_s] = (index_type)cbase[P];
    /* lane-major descriptor groups (see the layout comment): sizes, offsets, fill */
    const index_type pass_pairs = DFL_SLOT_BLOCK / 2, nwave_blk = DFL_SLOT_BLOCK / 64;
    ASSERT(2 * nwave_blk <= 8 && "one trip byte per (pass, wave): trips_lo / trips_hi hold eight");
    int64_t* gbase = (int64_t*)malloc(sizeof(int64_t) * ((size_t)P + 1));
    gbase[0] = 0;
    for (index_type p = 0; p < P; ++p) gbase[p + 1] = 0;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type s0 = hdr[8 * p + 2], np = hdr[8 * p + 3];
        uint32_t tb[2] = {0, 0};
        int64_t groups = 0;
        for (index_type q = 0; q < 2 * nwave_blk; ++q) {
            const index_type first = (q / nwave_blk) * pass_pairs + (q % nwave_blk) * 32;
            index_type maxc = 0;
            for (index_type k = first; k < first + 32 && k < np; ++k) {
                const index_type c = coff[s0 + k + 1] - coff[s0 + k];
                if (c > maxc) maxc = c;
            }
            const index_type trips = (maxc + 1) / 2;
            ASSERT(trips <= 254 && "slot-patch trip counts are bytes");
            tb[q >> 2] |= (uint32_t)trips << (8 * (q & 3));
            groups += trips; /* in UNITS of one trip = 64 lanes x u16 = 32 words */
        }
        hdr[8 * p + 5] = (int32_t)tb[0];
        hdr[8 * p + 6] = (int32_t)tb[1];
        gbase[p + 1] = groups;
    }
    for (index_type p = 0; p < P; ++p) gbase[p + 1] += gbase[p];
    ASSERT(gbase[P] < 2147483647LL);
    const size_t ld_words = ((size_t)gbase[P] + 8) * 32; /* + 8 units: the kernel's clamped prefetch may read past the last group */
    uint32_t* ldesc = (uint32_t*)malloc(sizeof(uint32_t) * ld_words);
    memset(ldesc, 0xff, sizeof(uint32_t) * ld_words);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (index_type p = 0; p < P; ++p) {
        const index_type s0 = hdr[8 * p + 2], np = hdr[8 * p + 3];
        const uint32_t tb[2] = {(uint32_t)hdr[8 * p + 5], (uint32_t)hdr[8 * p + 6]};
        hdr[8 * p + 4] = (int32_t)gbase[p];
        int64_t g0 = gbase[p];
        for (index_type q = 0; q < 2 * nwave_blk; ++q) {
            const index_type trips = (index_type)((tb[q >> 2] >> (8 * (q & 3))) & 255u);
            const index_type first = (q / nwave_blk) * pass_pairs + (q % nwave_blk) * 32;
            uint16_t* base = (uint16_t*)(ldesc + (size_t)g0 * 32); /* u16 view: a full group = 128 entries, a tail = 64 */
            for (index_type k = first; k < first + 32 && k < np; ++k) {
                const index_type c0 = coff[s0 + k], c = coff[s0 + k + 1] - c0;
                for (index_type i = 0; i < c; ++i) { /* contribution i: lane 2 (k - first) + (i & 1), trip i / 2 */
                    const index_type lane = 2 * (k - first) + (i & 1), trip = i >> 1, g = trip >> 1;
                    if (2 * g + 1 == trips) base[(size_t)g * 128 + lane] = desc[c0 + i];               /* tail: one u16 per lane */
                    else base[((size_t)g * 64 + lane) * 2 + (trip & 1)] = desc[c0 + i];                  /* word per lane: two trips */
                }
            }
            g0 += trips;
        }
    }
    ps->num_patch = P;
    ps->max_tets = maxt;
    ps->max_slots = maxs;
    ps->max_contrib = maxc_all;
    ps->total_tets = tot_t;
    ps->d_hdr = (int32_t*)CdamMallocDevice((ptrdiff_t)P * 8 * (ptrdiff_t)sizeof(int32_t) + 32);
    ps->max_nodes = maxn;
    ps->total_nodes = tot_n;
    ps->d_ptet_lid = (uint32_t*)CdamMallocDevice((ptrdiff_t)(tot_t > 0 ? tot_t : 1) * (ptrdiff_t)sizeof(uint32_t) + 1024);
    ps->d_pnode = (index_type*)CdamMallocDevice(((ptrdiff_t)tot_n + DFL_SLOT_BLOCK) * SIZE_OF(index_type));
    ps->d_slot_nz = (index_type*)CdamMallocDevice(((ptrdiff_t)tot_s + DFL_SLOT_BLOCK) * SIZE_OF(index_type));
    ps->d_ldesc = (uint32_t*)CdamMallocDevice((ptrdiff_t)ld_words * (ptrdiff_t)sizeof(uint32_t));
    HIPGUARD(hipMemcpy(ps->d_hdr, hdr, sizeof(int32_t) * 8 * (size_t)P, H2D));
    HIPGUARD(hipMemcpy(ps->d_ptet_lid, ptet_lid, sizeof(uint32_t) * (size_t)tot_t, H2D));
    HIPGUARD(hipMemcpy(ps->d_pnode, pnode, sizeof(index_type) * (size_t)tot_n, H2D));
    HIPGUARD(hipMemcpy(ps->d_slot_nz, slot_nz, sizeof(index_type) * (size_t)tot_s, H2D));
    HIPGUARD(hipMemcpy(ps->d_ldesc, ldesc, sizeof(uint32_t) * ld_words, H2D));
    if (verbose)
        fprintf(stderr, "[slotpatch] %.1f descriptor trips per patch, %.0f %% of their entries used\n", (double)gbase[P] / (double)(P > 0 ? P : 1),
                100.0 * 16.0 * (double)T / (64.0 * (double)(gbase[P] > 0 ? gbase[P] : 1)));
    free(ldesc); free(gbase);
    if (verbose) fprintf(stderr, "[slotpatch] uploaded at %.2f s\n", omp_get_wtime() - t0);
    free(cbase); free(desc); free(coff); free(slot_nz); free(ptet_lid); free(pnode); free(hdr); free(nt_of); free(tets_of); free(nn_of); free(nodes_of);
    free(x.out); free(idx); free(ve); free(vp); free(ci); free(rp);
    return ps;
}

void DflFreeSlotPatchSchedule(SlotPatchSched* ps) {
    if (!ps) return;
    CdamFreeDevice(ps->d_hdr, 0); CdamFreeDevice(ps->d_ptet_lid, 0); CdamFreeDevice(ps->d_pnode, 0); CdamFreeDevice(ps->d_slot_nz, 0);
    CdamFreeDevice(ps->d_ldesc, 0);
    CdamFreeHost(ps, SIZE_OF(SlotPatchSched));
}
