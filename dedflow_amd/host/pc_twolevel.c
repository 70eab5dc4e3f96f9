/* PC_TWOLEVEL: block-DILU smoothing + aggregation coarse-grid correction for the block-mode (u,p) matrix.
 *
 * Not in the reference (its only multilevel hook is the commented AMGX configuration of src/krylov.c:409-437, an
 * AGGREGATION AMG with a MULTICOLOR_DILU smoother); BASELINE config 5 (50M tets, 100-step transient) is out of reach of
 * one-level preconditioners: DILU-GMRES needs ~600 iterations per solve there and stalls on later steps, because the
 * pressure part of the system is a Poisson-like problem whose conditioning grows like h^-2.  One coarse level removes that
 * dependence (measured iteration counts: DESIGN.md section 3):
 *
 *     z  = S r                                  S = multicolor block-DILU (host/pc_dilu.c)
 *     z += P Ac^-1 P^T (r - A z)                P = piecewise constant over node aggregates, Ac = P^T A P
 *
 * Aggregates: recursive coordinate bisection of the mesh nodes down to `agg_size` nodes (host, once per mesh + pattern).
 * P^T A P with a piecewise-constant P is a sum of fine 4x4 blocks per coarse nonzero: host-built lists, summed on the
 * device at every PCSetup in list order (csrc/k_amg.hip) -- deterministic.  The coarse matrix is an ordinary block-mode
 * MatrixFS, and Ac^-1 is an inner KrylovSolve on it to rtol 0.1, so the outer solver has to be flexible (KrylovSolve
 * switches FGMRES on when it builds this PC).  While the coarse level has more than DFL_TL_COARSEST (262144) nodes its inner
 * solver is FGMRES preconditioned by another PCTwoLevel on the aggregate centroids (a K-cycle: a handful of inner
 * iterations per level, every level 1/agg_size of the one above); the last level is DILU-GMRES (Jacobi-GMRES when it has
 * <= 131072 nodes: see tl_build).  With a single coarse level
 * the inner Jacobi/DILU iteration count grows with the coarse grid and the 50M-tet case (232k aggregates) ran into its cap.
 * No post-smoothing: a second DILU sweep after the correction (z += S (r - A z)) stalls the 50M-tet solve at 1e-3 -- the
 * block-DILU iteration is not a convergent smoother for this stabilised (u,p) system on fine meshes, which is also why the
 * outer count still grows slowly with the mesh (16 / 17 / 28 / 38 iterations to rtol 1e-4 at M = 60 / 119 / 160 / 203;
 * DILU alone: 120 at M = 119, ~600 at M = 203), independent of the aggregate size (8 or 64), of the inner tolerance and of a
 * damping factor on the smoother step (0.5 - 1.3).  With the time step fixed the CFL number grows with the mesh (10 at M = 203
 * for |u| = 1): the momentum block becomes advection-dominated, which a piecewise-constant coarse space does not follow.
 *
 * Element-partitioned matrices (PCCreateTwoLevelDist; one process per GPU, owned rows first): every rank cuts aggregates out
 * of the nodes it OWNS and numbers them globally (rank order); ghost nodes learn their aggregate through one halo exchange.
 * A rank's owned rows give complete rows of Ac = P^T A P for its own aggregates; pattern (once) and values (every PCSetup)
 * are then REPLICATED -- every rank fills its rows of a zero buffer and the buffers are summed by DflComm.allreduce_sum,
 * which is an exact all-gather (x + 0 = x) -- so the coarse problem (<= 130k nodes at 50M tets) is solved redundantly and
 * identically on every rank with no collective inside the inner iteration.  Per application the partitioned form adds one
 * halo exchange (ghost entries of z for t = r - A z) and one all-reduce of the restricted residual (4 Nc doubles); the
 * smoother is the rank-local DILU (block-diagonal across ranks, host/pc_dilu.c).  The K-cycle levels below work on the
 * replicated matrices and need no communication at all.
 */
#include <string.h>
#include <omp.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"
#include "solver_private.h"
#include "rcb.h"

typedef struct PCTwoLevel {
    Matrix* A;
    PC* smoother;
    const CSRAttr* spy;          /* fine pattern the hierarchy was built for */
    index_type N, Nc, n, n_active, agg_size;
    index_type *d_agg;           /* [N] aggregate of every node */
    index_type *d_aoff, *d_anode; /* aggregate -> nodes (ascending node id) */
    CSRAttr *c1x1, *c1x1b, *c3x3, *c3x1, *c1x3; /* coarse nodal pattern, its copy for A11, expanded patterns */
    Matrix* Ac;
    index_type *d_goff, *d_gidx; /* coarse nonzero -> fine nonzeros (ascending) */
    Krylov* cksp;
    f64 *d_t, *d_rc, *d_xc;
    index_type inner_maxit;
    f64 inner_rtol;
    float* d_valf;        /* single-precision copy of the fine block values: smoother sweeps and residual matvec (tl_setup) */
    int64_t valf_len;
    b32 inner_rtol_given; /* set through PCTwoLevelSetInner: the per-solver default does not replace it */
    b32 inner_jacobi;     /* last level solved by Jacobi-GMRES (small coarse levels) instead of DILU-GMRES */
    int64_t inner_iterations;
    const f64* xyz;              /* [N][3] host coordinates the aggregates are cut from */
    f64* xyz_owned;              /* coarse levels own theirs (aggregate centroids) */
    int level;
    /* element-partitioned fine level */
    b32 dist;                    /* comm below is valid and world > 1 */
    DflComm comm;
    index_type n_owned;          /* rows this rank aggregates (== N on one GPU) */
    index_type agg_base, Nc_local; /* this rank's aggregates are [agg_base, agg_base + Nc_local) of the Nc global ones */
} PCTwoLevel;
static PC* tl_create(Matrix* mat, const f64* xyz, f64* xyz_owned, index_type agg_size, int level);

/* ---- aggregation: RCB over node coordinates ------------------------------------------------------------------- */
typedef struct { index_type lo, hi; } Range;
typedef struct { const f64* c; index_type* idx; index_type leaf; Range* out; index_type nout, capout; } Rcb;

static void rcb_emit(Rcb* x, index_type lo, index_type hi) {
#pragma omp critical(dfl_twolevel_emit)
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
static void rcb_split(Rcb* x, index_type lo, index_type hi) {
    const index_type n = hi - lo;
    if (n <= x->leaf) { rcb_emit(x, lo, hi); return; }
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
    if (n > 8192) {
#pragma omp task
        rcb_split(x, lo, lo + half);
#pragma omp task
        rcb_split(x, lo + half, hi);
#pragma omp taskwait
    } else {
        rcb_split(x, lo, lo + half);
        rcb_split(x, lo + half, hi);
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

static void tl_release(PCTwoLevel* d) {
    CdamFreeDevice(d->d_agg, 0); CdamFreeDevice(d->d_aoff, 0); CdamFreeDevice(d->d_anode, 0);
    CdamFreeDevice(d->d_goff, 0); CdamFreeDevice(d->d_gidx, 0);
    CdamFreeDevice(d->d_t, 0); CdamFreeDevice(d->d_rc, 0); CdamFreeDevice(d->d_xc, 0);
    d->d_agg = d->d_aoff = d->d_anode = d->d_goff = d->d_gidx = NULL;
    d->d_t = d->d_rc = d->d_xc = NULL;
    if (d->smoother) PCDILUSetF32Values(d->smoother, NULL);
    CdamFreeDevice(d->d_valf, 0);
    d->d_valf = NULL;
    d->valf_len = 0;
    if (d->cksp) KrylovDestroy(d->cksp);
    d->cksp = NULL;
    if (d->Ac) MatrixDestroy(d->Ac);
    d->Ac = NULL;
    CSRAttrDestroy(d->c3x3); CSRAttrDestroy(d->c3x1); CSRAttrDestroy(d->c1x3); CSRAttrDestroy(d->c1x1b); CSRAttrDestroy(d->c1x1);
    d->c3x3 = d->c3x1 = d->c1x3 = d->c1x1b = d->c1x1 = NULL;
}

/* ---- replication helpers (partitioned fine level) -------------------------------------------------------------
 * Small integer tables travel as f64 (exact below 2^53) through the one reduction DflComm offers: every rank writes its
 * own entries into a zero buffer, the sum over ranks is the concatenation. */
static void dist_sum_f64(const DflComm* c, f64* d_buf, int64_t n) {
    const int64_t chunk = (int64_t)1 << 30;
    for (int64_t o = 0; o < n; o += chunk) c->allreduce_sum(c->ctx, d_buf + o, (index_type)(n - o < chunk ? n - o : chunk));
}
/* host array of `total` doubles, this rank's `count` entries at `at`, zero elsewhere -> the sum over all ranks */
static void dist_replicate(const DflComm* c, f64* host, int64_t total) {
    if (total <= 0) return;
    hipStream_t s = DflStream();
    f64* d = (f64*)CdamMallocDevice((ptrdiff_t)total * SIZE_OF(f64));
    HIPGUARD(hipMemcpyAsync(d, host, sizeof(f64) * (size_t)total, H2D, s));
    dist_sum_f64(c, d, total);
    HIPGUARD(hipMemcpyAsync(host, d, sizeof(f64) * (size_t)total, D2H, s));
    HIPGUARD(hipStreamSynchronize(s));
    CdamFreeDevice(d, 0);
}

/* aggregates, coarse pattern, Galerkin lists, coarse matrix + solver: once per (mesh, fine pattern) */
static void tl_build(PCTwoLevel* d) {
    const CSRAttr* spy = d->spy;
    const index_type N = d->N;
    const index_type no = d->dist ? d->n_owned : N; /* rows aggregated here */
    const f64* xg = d->xyz;
    int nt = omp_get_max_threads();
    if (getenv("DFL_HOST_THREADS")) nt = atoi(getenv("DFL_HOST_THREADS"));
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    const int verbose = getenv("DFL_PATCH_VERBOSE") != NULL;
    double t0 = omp_get_wtime();
    index_type* rp = (index_type*)malloc(sizeof(index_type) * ((size_t)N + 1));
    index_type* ci = (index_type*)malloc(sizeof(index_type) * (size_t)spy->nnz);
    HIPGUARD(hipMemcpy(rp, spy->row_ptr, sizeof(index_type) * ((size_t)N + 1), D2H));
    HIPGUARD(hipMemcpy(ci, spy->col_ind, sizeof(index_type) * (size_t)spy->nnz, D2H));
    d->d_t = (f64*)CdamMallocDevice((ptrdiff_t)d->n * SIZE_OF(f64));
    /* 1. aggregates of the nodes this rank owns */
    index_type* idx = (index_type*)malloc(sizeof(index_type) * (size_t)(no > 0 ? no : 1));
    for (index_type i = 0; i < no; ++i) idx[i] = i;
    Rcb x = {xg, idx, d->agg_size, NULL, 0, 1024};
    x.out = (Range*)malloc(sizeof(Range) * (size_t)x.capout);
    if (no > 0) {
#pragma omp parallel num_threads(nt)
#pragma omp single
        rcb_split(&x, 0, no);
    }
    qsort(x.out, (size_t)x.nout, sizeof(Range), cmp_range);
    const index_type Ncl = x.nout;
    /* global numbering of the aggregates: rank order */
    index_type base = 0, Nc = Ncl;
    int64_t N_global = N;
    if (d->dist) {
        const int W = d->comm.world;
        f64* cnt = (f64*)calloc((size_t)2 * W, sizeof(f64));
        cnt[d->comm.rank] = (f64)Ncl;
        cnt[W + d->comm.rank] = (f64)no;
        dist_replicate(&d->comm, cnt, 2 * (int64_t)W);
        int64_t tot = 0;
        N_global = 0;
        for (int q = 0; q < W; ++q) {
            if (q == d->comm.rank) base = (index_type)tot;
            tot += (int64_t)cnt[q];
            N_global += (int64_t)cnt[W + q];
        }
        ASSERT(tot < 2147483647LL / 8);
        Nc = (index_type)tot;
        free(cnt);
    }
    index_type* agg = (index_type*)malloc(sizeof(index_type) * (size_t)N); /* GLOBAL aggregate of every local node */
    index_type* aoff = (index_type*)malloc(sizeof(index_type) * ((size_t)Nc + 1));
    for (index_type g = 0; g <= base; ++g) aoff[g] = 0;
    for (index_type I = 0; I < Ncl; ++I) {
        aoff[base + I] = x.out[I].lo;
        qsort(idx + x.out[I].lo, (size_t)(x.out[I].hi - x.out[I].lo), sizeof(index_type), cmp_i32); /* ascending node id */
        for (index_type k = x.out[I].lo; k < x.out[I].hi; ++k) agg[idx[k]] = base + I;
    }
    for (index_type g = base + Ncl; g <= Nc; ++g) aoff[g] = no; /* the other ranks' aggregates hold none of our nodes */
    if (d->dist && N > no) {
        /* ghost nodes: the aggregate their owner gave them, through the solver's own halo exchange (ids ride in the u0 slot) */
        hipStream_t s = DflStream();
        f64* h = (f64*)calloc((size_t)d->n, sizeof(f64));
        for (index_type i = 0; i < no; ++i) h[(size_t)3 * i] = (f64)agg[i];
        HIPGUARD(hipMemcpyAsync(d->d_t, h, sizeof(f64) * (size_t)d->n, H2D, s));
        d->comm.halo_exchange(d->comm.ctx, d->d_t);
        HIPGUARD(hipMemcpyAsync(h, d->d_t, sizeof(f64) * (size_t)d->n, D2H, s));
        HIPGUARD(hipStreamSynchronize(s));
        for (index_type i = no; i < N; ++i) {
            agg[i] = (index_type)h[(size_t)3 * i];
            ASSERT(agg[i] >= 0 && agg[i] < Nc && (agg[i] < base || agg[i] >= base + Ncl));
        }
        free(h);
    } else if (d->dist) {
        d->comm.halo_exchange(d->comm.ctx, d->d_t); /* a rank without ghosts still takes part in the exchange */
    }
    /* 2. coarse pattern: row I = sorted unique aggregates of the neighbours of its nodes (own rows here, then replicated) */
    index_type** rows = (index_type**)malloc(sizeof(index_type*) * (size_t)(Ncl > 0 ? Ncl : 1));
    index_type* rlen = (index_type*)malloc(sizeof(index_type) * (size_t)(Ncl > 0 ? Ncl : 1));
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (index_type I = 0; I < Ncl; ++I) {
        size_t cap = 0;
        for (index_type k = aoff[base + I]; k < aoff[base + I + 1]; ++k) cap += (size_t)(rp[idx[k] + 1] - rp[idx[k]]);
        index_type* tmp = (index_type*)malloc(sizeof(index_type) * (cap > 0 ? cap : 1));
        size_t m = 0;
        for (index_type k = aoff[base + I]; k < aoff[base + I + 1]; ++k)
            for (index_type z = rp[idx[k]]; z < rp[idx[k] + 1]; ++z) tmp[m++] = agg[ci[z]];
        qsort(tmp, m, sizeof(index_type), cmp_i32);
        index_type u = 0;
        for (size_t q = 0; q < m; ++q)
            if (q == 0 || tmp[q] != tmp[q - 1]) tmp[u++] = tmp[q];
        rows[I] = tmp;
        rlen[I] = u;
    }
    index_type* crp = (index_type*)malloc(sizeof(index_type) * ((size_t)Nc + 1));
    if (d->dist) {
        f64* len = (f64*)calloc((size_t)(Nc > 0 ? Nc : 1), sizeof(f64));
        for (index_type I = 0; I < Ncl; ++I) len[base + I] = (f64)rlen[I];
        dist_replicate(&d->comm, len, Nc);
        crp[0] = 0;
        for (index_type g = 0; g < Nc; ++g) {
            ASSERT((int64_t)crp[g] + (int64_t)len[g] < 2147483647LL / 16);
            crp[g + 1] = crp[g] + (index_type)len[g];
        }
        free(len);
    } else {
        crp[0] = 0;
        for (index_type I = 0; I < Nc; ++I) {
            ASSERT((int64_t)crp[I] + rlen[I] < 2147483647LL);
            crp[I + 1] = crp[I] + rlen[I];
        }
    }
    const index_type nnzc = crp[Nc];
    index_type* cci = (index_type*)malloc(sizeof(index_type) * (size_t)(nnzc > 0 ? nnzc : 1));
    for (index_type I = 0; I < Ncl; ++I) {
        memcpy(cci + crp[base + I], rows[I], sizeof(index_type) * (size_t)rlen[I]);
        free(rows[I]);
    }
    if (d->dist) {
        f64* col = (f64*)calloc((size_t)(nnzc > 0 ? nnzc : 1), sizeof(f64));
        for (index_type z = crp[base]; z < crp[base + Ncl]; ++z) col[z] = (f64)cci[z];
        dist_replicate(&d->comm, col, nnzc);
        for (index_type z = 0; z < nnzc; ++z) cci[z] = (index_type)col[z];
        free(col);
    }
    /* 3. Galerkin lists: coarse nonzero -> fine nonzeros of the OWNED fine rows (they come first), ascending fine index */
    const index_type fnz = rp[no];
    index_type* goff = (index_type*)calloc((size_t)nnzc + 1, sizeof(index_type));
    index_type* fz2cz = (index_type*)malloc(sizeof(index_type) * (size_t)(fnz > 0 ? fnz : 1));
#pragma omp parallel for schedule(static) num_threads(nt)
    for (index_type i = 0; i < no; ++i) {
        const index_type I = agg[i];
        for (index_type z = rp[i]; z < rp[i + 1]; ++z) {
            const index_type J = agg[ci[z]];
            index_type lo = crp[I], hi = crp[I + 1] - 1;
            while (lo < hi) {
                index_type mid = (lo + hi) >> 1;
                if (cci[mid] < J) lo = mid + 1; else hi = mid;
            }
            fz2cz[z] = lo;
        }
    }
    for (index_type z = 0; z < fnz; ++z) goff[fz2cz[z] + 1]++;
    for (index_type c = 0; c < nnzc; ++c) goff[c + 1] += goff[c];
    index_type* gidx = (index_type*)malloc(sizeof(index_type) * (size_t)(fnz > 0 ? fnz : 1));
    {
        index_type* cur = (index_type*)malloc(sizeof(index_type) * (size_t)(nnzc > 0 ? nnzc : 1));
        memcpy(cur, goff, sizeof(index_type) * (size_t)nnzc);
        for (index_type z = 0; z < fnz; ++z) gidx[cur[fz2cz[z]]++] = z;
        free(cur);
    }
    /* 4. upload; coarse matrix objects in the reference's own shapes */
#define UP(dst, src, cnt)                                                                             \
    dst = (index_type*)CdamMallocDevice((ptrdiff_t)((cnt) > 0 ? (cnt) : 1) * SIZE_OF(index_type));    \
    HIPGUARD(hipMemcpy(dst, src, sizeof(index_type) * (size_t)(cnt), H2D));
    UP(d->d_agg, agg, N) UP(d->d_aoff, aoff, Nc + 1) UP(d->d_anode, idx, no)
    UP(d->d_goff, goff, nnzc + 1) UP(d->d_gidx, gidx, fnz)
    CSRAttr* c = (CSRAttr*)CdamMallocHost(SIZE_OF(CSRAttr));
    memset(c, 0, sizeof *c);
    c->num_row = c->num_col = Nc;
    c->nnz = nnzc;
    UP(c->row_ptr, crp, Nc + 1) UP(c->col_ind, cci, nnzc)
#undef UP
    d->c1x1 = c;
    d->c3x3 = CSRAttrCreateBlock(c, 3, 3);
    d->c3x1 = CSRAttrCreateBlock(c, 3, 1);
    d->c1x3 = CSRAttrCreateBlock(c, 1, 3);
    const index_type offset[] = {0, 3, 4, 5, 6};
    d->Ac = MatrixCreateTypeFS(4, offset, NULL);
    MatrixFS* fs = (MatrixFS*)d->Ac->data;
    fs->spy1x1 = c;
    fs->mat[0] = MatrixCreateTypeCSR(d->c3x3, NULL);
    fs->mat[1] = MatrixCreateTypeCSR(d->c3x1, NULL);
    fs->mat[4] = MatrixCreateTypeCSR(d->c1x3, NULL);
    d->c1x1b = CSRAttrCreateBlock(c, 1, 1);
    fs->mat[5] = MatrixCreateTypeCSR(d->c1x1b, NULL);
    MatrixSetup(d->Ac);
    ASSERT(MatrixFSBlockValues(d->Ac));
    index_type coarsest = 262144;
    if (getenv("DFL_TL_COARSEST")) coarsest = atoi(getenv("DFL_TL_COARSEST"));
    const int recurse = Nc > coarsest && (int64_t)Nc * 2 < N_global;
    index_type kcycle = 8;
    if (getenv("DFL_TL_KCYCLE")) kcycle = atoi(getenv("DFL_TL_KCYCLE"));
    if (getenv("DFL_TL_INNER_RTOL")) d->inner_rtol = atof(getenv("DFL_TL_INNER_RTOL"));
    d->cksp = KrylovCreateGMRES(recurse ? kcycle : d->inner_maxit, 0.0, d->inner_rtol, NULL);
    KrylovSetVerbose(d->cksp, FALSE);
    DflKrylovMarkInner(d->cksp); /* a coarse solver never times basis placements in the middle of an outer solve */
    if (recurse) {
        /* centroids of ALL aggregates (the level below works on the replicated coarse matrix) */
        f64* cx = (f64*)calloc(3 * (size_t)Nc, sizeof(f64));
        for (index_type I = base; I < base + Ncl; ++I) {
            f64 sx = 0, sy = 0, sz = 0;
            for (index_type k = aoff[I]; k < aoff[I + 1]; ++k) {
                sx += xg[(size_t)idx[k] * 3];
                sy += xg[(size_t)idx[k] * 3 + 1];
                sz += xg[(size_t)idx[k] * 3 + 2];
            }
            const f64 w = 1.0 / (f64)(aoff[I + 1] - aoff[I]);
            cx[(size_t)I * 3] = sx * w;
            cx[(size_t)I * 3 + 1] = sy * w;
            cx[(size_t)I * 3 + 2] = sz * w;
        }
        if (d->dist) dist_replicate(&d->comm, cx, 3 * (int64_t)Nc);
        d->cksp->pc = tl_create(d->Ac, cx, cx, d->agg_size, d->level + 1); /* destroyed with the inner solver */
        KrylovSetFlexible(d->cksp, TRUE);
        KrylovSetCheckInterval(d->cksp, 2);
    } else {
        /* The last level's solver.  A coarse level of <= 131072 nodes is LATENCY-bound: a DILU application is 2 x 12 colour
           launches of 8-10 us each (a chain of four dependent loads per launch) for 18 us of matvec, 340 us per inner
           iteration against 70 us with the reference's Jacobi tree, which needs only twice the iterations (to rtol 0.05,
           where the outer count is the same: 20 at 1.3M and 10M tets): 10M tets 116 -> 88 ms per solve, 1.3M tets 46 -> 27 ms.
           Above that size DILU wins (50M tets, 232k aggregates: 713 against 782 ms).  DFL_TL_INNER_PC=jacobi|dilu forces
           one; an inner tolerance set through PCTwoLevelSetInner / DFL_TL_INNER_RTOL is kept as given.
           DFL_TL_INNER_CHECK = convergence test every k inner iterations (default 4). */
        index_type jacobi_max = 131072;
        if (getenv("DFL_TL_INNER_JACOBI_MAX")) jacobi_max = atoi(getenv("DFL_TL_INNER_JACOBI_MAX"));
        b32 jacobi = Nc <= jacobi_max;
        if (getenv("DFL_TL_INNER_PC")) jacobi = !strcmp(getenv("DFL_TL_INNER_PC"), "jacobi");
        if (jacobi) {
            d->cksp->pc = DflKrylovBuildPC(d->cksp, d->Ac);
            /* 4 launches per inner iteration instead of 7 (the norm by Pythagoras: exact enough for a solve to rtol 0.05) */
            if (!getenv("DFL_TL_INNER_FUSED") || atoi(getenv("DFL_TL_INNER_FUSED"))) KrylovSetFusedNorm(d->cksp, TRUE);
            if (!d->inner_rtol_given && !getenv("DFL_TL_INNER_RTOL")) d->cksp->rtol = d->inner_rtol = 0.05;
        } else {
            d->cksp->pc = PCCreateDILU(d->Ac);
        }
        d->inner_jacobi = jacobi;
        KrylovSetCheckInterval(d->cksp, getenv("DFL_TL_INNER_CHECK") ? atoi(getenv("DFL_TL_INNER_CHECK")) : 4);
    }
    d->Nc = Nc;
    d->Nc_local = Ncl;
    d->agg_base = base;
    d->d_rc = (f64*)CdamMallocDevice((ptrdiff_t)Nc * 6 * SIZE_OF(f64));
    d->d_xc = (f64*)CdamMallocDevice((ptrdiff_t)Nc * 6 * SIZE_OF(f64));
    if (verbose)
        fprintf(stderr, "[twolevel] level %d%s: %d nodes -> %d aggregates (<= %d nodes)%s, coarse nnz %d (%.1f per row), %s below, %.2f s\n",
                d->level, d->dist ? " (partitioned)" : "", no, Ncl, d->agg_size, d->dist ? " of this rank" : "", nnzc,
                (double)nnzc / (double)(Nc > 0 ? Nc : 1), recurse ? "another level" : "DILU-GMRES", omp_get_wtime() - t0);
    free(gidx); free(fz2cz); free(goff); free(cci); free(rlen); free(rows); free(crp); free(aoff); free(agg); free(x.out); free(idx);
    free(ci); free(rp);
}

static void tl_setup(PC* pc) {
    PCTwoLevel* d = (PCTwoLevel*)pc->data;
    Matrix* A = (Matrix*)pc->mat;
    MatrixFS* fs = (MatrixFS*)A->data;
    if (d->spy != fs->spy1x1 || (d->dist && d->n_owned != MatrixFSOwnedRows(A))) {
        tl_release(d);
        d->spy = fs->spy1x1;
        d->N = fs->spy1x1->num_row;
        d->n_owned = MatrixFSOwnedRows(A);
        tl_build(d);
    }
    PCSetup(d->smoother);
    /* Measured alternative, OFF by default (DFL_TL_F32=1): mixed precision INSIDE the preconditioner -- the smoother's sweeps
       and the residual matvec of the coarse correction read a single-precision copy of the block values (E^-1, all sums, the
       Galerkin matrix and everything outside the preconditioner stay double).  The outer solver is flexible and the iteration
       counts do not move (20 / 20 / 40 at 1.3M / 10M / 50M tets), but neither does the time much: 82 -> 78 ms and 715 -> 688 ms
       per solve -- the matvec on HALF the value bytes takes 0.55 ms against 0.59 ms with the SAME number of L2 requests
       (3.5e7, tools/pmc_spmv_f32_vs_f64.sh): these kernels are paced by requests, not by bytes.  Not worth 8 GB at 50M tets
       and a precision caveat. */
    {
        static int f32 = -1;
        if (f32 < 0) f32 = getenv("DFL_TL_F32") && atoi(getenv("DFL_TL_F32")) == 1;
        MatrixFS* fsf = (MatrixFS*)A->data;
        const int64_t len = (int64_t)fsf->spy1x1->nnz * 16;
        if (f32) {
            if (d->valf_len != len) {
                CdamFreeDevice(d->d_valf, 0);
                d->d_valf = (float*)CdamMallocDevice((ptrdiff_t)len * (ptrdiff_t)sizeof(float));
                d->valf_len = len;
            }
            dfl_bcsr_values_to_f32(len, MatrixFSBlockValues(A), d->d_valf, DflStream());
            PCDILUSetF32Values(d->smoother, d->d_valf);
        } else {
            PCDILUSetF32Values(d->smoother, NULL);
        }
    }
    dfl_amg_galerkin(d->c1x1->nnz, d->d_goff, d->d_gidx, MatrixFSBlockValues(A), MatrixFSBlockValues(d->Ac), DflStream());
    /* partitioned: every rank summed the coarse rows of its own aggregates (the lists of the others are empty: zeros) */
    if (d->dist) dist_sum_f64(&d->comm, MatrixFSBlockValues(d->Ac), (int64_t)d->c1x1->nnz * 16);
    PCSetup((PC*)d->cksp->pc); /* the levels below: their coarse matrices change here, not between applications */
}

static void tl_apply(PC* pc, value_type* r, value_type* z) {
    PCTwoLevel* d = (PCTwoLevel*)pc->data;
    Matrix* A = (Matrix*)pc->mat;
    hipStream_t s = DflStream();
    const index_type N = d->N, Nc = d->Nc, n = d->n_active > 0 ? d->n_active : d->n;
    PCDILUSetActiveLength(d->smoother, n);
    PCApply(d->smoother, r, z);                 /* z = S r (copies the phi / T tail when n > 4N) */
    if (d->dist) d->comm.halo_exchange(d->comm.ctx, z); /* the owned rows of A read ghost entries of z */
    if (d->d_valf && d->valf_len) {             /* t = A z on the (u,p) part (owned rows); the restriction forms r - t */
        MatrixFS* fsf = (MatrixFS*)A->data;
        dfl_bcsr_spmv_f32(MatrixFSOwnedRows(A), N, fsf->spy1x1->row_ptr, fsf->spy1x1->col_ind, d->d_valf, z, d->d_t, s);
    } else {
        MatrixMatVec(A, z, d->d_t);
    }
    dfl_amg_restrict_diff(Nc, d->d_aoff, d->d_anode, N, r, d->d_t, d->d_rc, s);
    /* partitioned: rc is zero outside this rank's aggregates; the sum over ranks is the whole coarse residual, on every rank */
    if (d->dist) dist_sum_f64(&d->comm, d->d_rc, 4 * (int64_t)Nc);
    HIPGUARD(hipMemsetAsync(d->d_xc, 0, (size_t)Nc * 6 * sizeof(f64), s));
    DflKrylovSolvePrepared(d->cksp, d->Ac, d->d_xc, d->d_rc); /* the phi / T tail of rc is zero: the solve runs on 4 Nc */
    d->inner_iterations += KrylovGetStats(d->cksp)->iterations;
    dfl_amg_prolong_add_rows(d->dist ? d->n_owned : N, N, d->d_agg, Nc, d->d_xc, z, s);
}

static void tl_destroy(PC* pc) {
    PCTwoLevel* d = (PCTwoLevel*)pc->data;
    tl_release(d);
    PCDestroy(d->smoother);
    free(d->xyz_owned);
    CdamFreeHost(d, SIZE_OF(PCTwoLevel));
}

PC* PCCreateTwoLevelDist(Matrix* mat, const Mesh3D* mesh, index_type agg_size, const DflComm* comm) {
    if (!mat || !MatrixFSBlockValues(mat) || !mesh || !mesh->host || mesh->num_node != ((MatrixFS*)mat->data)->spy1x1->num_row) {
        fprintf(stderr, "PCCreateTwoLevel: needs the block-mode (u,p) field-split matrix and the mesh it was built on\n");
        return NULL;
    }
    const b32 partitioned = comm && comm->world > 1;
    if (!partitioned && MatrixFSOwnedRows(mat) != mesh->num_node) {
        fprintf(stderr, "PCCreateTwoLevel: an element-partitioned matrix needs the partition's DflComm with rank / world set "
                        "(PCCreateTwoLevelDist)\n");
        return NULL;
    }
    if (partitioned && (comm->rank < 0 || comm->rank >= comm->world || !comm->allreduce_sum || !comm->halo_exchange)) {
        fprintf(stderr, "PCCreateTwoLevelDist: the communicator lacks rank / world or a callback\n");
        return NULL;
    }
    PC* pc = tl_create(mat, mesh->host->xg, NULL, agg_size, 0);
    if (partitioned) {
        PCTwoLevel* d = (PCTwoLevel*)pc->data;
        d->dist = TRUE;
        d->comm = *comm;
    }
    return pc;
}
PC* PCCreateTwoLevel(Matrix* mat, const Mesh3D* mesh, index_type agg_size) { return PCCreateTwoLevelDist(mat, mesh, agg_size, NULL); }

static PC* tl_create(Matrix* mat, const f64* xyz, f64* xyz_owned, index_type agg_size, int level) {
    PC* pc = (PC*)CdamMallocHost(SIZE_OF(PC));
    memset(pc, 0, sizeof *pc);
    PCTwoLevel* d = (PCTwoLevel*)CdamMallocHost(SIZE_OF(PCTwoLevel));
    memset(d, 0, sizeof *d);
    d->A = mat;
    d->xyz = xyz;
    d->xyz_owned = xyz_owned;
    d->level = level;
    d->n = MatrixNumRow(mat);
    d->agg_size = agg_size > 0 ? agg_size : 64;
    d->inner_maxit = 40;
    d->inner_rtol = 0.1;
    d->smoother = PCCreateDILU(mat);
    pc->type = PC_TWOLEVEL;
    pc->mat = mat;
    pc->data = d;
    pc->op->setup = tl_setup;
    pc->op->apply = tl_apply;
    pc->op->destroy = tl_destroy;
    return pc;
}

void PCTwoLevelSetActiveLength(PC* pc, index_type n_active) {
    if (pc && pc->type == PC_TWOLEVEL) ((PCTwoLevel*)pc->data)->n_active = n_active;
}
void PCTwoLevelSetInner(PC* pc, index_type max_iter, f64 rtol) {
    if (!pc || pc->type != PC_TWOLEVEL) return;
    PCTwoLevel* d = (PCTwoLevel*)pc->data;
    d->inner_maxit = max_iter;
    d->inner_rtol = rtol;
    d->inner_rtol_given = TRUE;
    if (d->cksp) {
        d->cksp->max_iter = max_iter;
        d->cksp->rtol = rtol;
    }
}
void PCTwoLevelInfo(PC* pc, index_type* num_aggregate, index_type* coarse_nnz, int64_t* inner_iterations) {
    PCTwoLevel* d = (PCTwoLevel*)pc->data;
    if (num_aggregate) *num_aggregate = d->Nc;
    if (coarse_nnz) *coarse_nnz = d->c1x1 ? d->c1x1->nnz : 0;
    if (inner_iterations) *inner_iterations = d->inner_iterations;
}
/* introspection for tests: device pointers of the aggregate map and of the coarse block values */
const index_type* PCTwoLevelAggregates(PC* pc) { return ((PCTwoLevel*)pc->data)->d_agg; }
Matrix* PCTwoLevelCoarseMatrix(PC* pc) { return ((PCTwoLevel*)pc->data)->Ac; }
