/* PC_ILU0: multicolor block-DILU preconditioner for the block-mode (u,p) field-split matrix
 * (kernels and formula: csrc/k_dilu.hip).  The reference declares PC_ILU0 in its PCType enum (pc.h) but
 * implements only Jacobi / decomposition / AMGX; BASELINE config 5 asks for an "ILU0-style" PC, so this is
 * build-defined.  Node colors: greedy first-fit in node order over the nodal pattern (host, once per pattern);
 * E^-1 is recomputed at every PCSetup (the matrix values change every Newton iteration).
 * Partitioned runs: only owned rows are preconditioned and ghost columns are dropped from L and U, i.e. the
 * DILU is block-diagonal across ranks (additive-Schwarz flavour, no communication in the apply). */
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

typedef struct PCDilu {
    index_type N, nown, n; /* nodes, owned nodes, vector length of the operator (4N or 6N) */
    index_type n_active;   /* length the next apply works on (GMRES runs on 4N when the tail of b is zero) */
    const CSRAttr* spy;
    index_type num_color;
    index_type* color_offset; /* host [num_color+1] */
    index_type* d_rows;       /* device [nown] rows grouped by color */
    u8* d_color;              /* device [N] */
    index_type *d_lptr, *d_lnz, *d_lcol; /* device: strictly-lower (by color) neighbours of each row slot */
    index_type *d_uptr, *d_unz, *d_ucol; /* device: strictly-upper neighbours */
    f64* d_Einv;              /* device [N][16] */
    const float* valf;        /* PCDILUSetF32Values: the sweeps read the off-diagonal blocks from this single-precision copy */
} PCDilu;

static void dilu_build_colors(PCDilu* d) {
    const CSRAttr* spy = d->spy;
    const index_type N = d->N, nown = d->nown;
    index_type* rp = (index_type*)CdamMallocHost(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    index_type* ci = (index_type*)CdamMallocHost((ptrdiff_t)spy->nnz * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(rp, spy->row_ptr, sizeof(index_type) * ((size_t)N + 1), D2H));
    HIPGUARD(hipMemcpy(ci, spy->col_ind, sizeof(index_type) * (size_t)spy->nnz, D2H));
    u8* color = (u8*)CdamMallocHost((ptrdiff_t)N);
    memset(color, 255, (size_t)N); /* 255 = not a preconditioned row (ghost) */
    index_type count[64];
    memset(count, 0, sizeof count);
    index_type nc = 0;
    for (index_type i = 0; i < nown; ++i) {
        u64 used = 0;
        for (index_type k = rp[i]; k < rp[i + 1]; ++k) {
            const index_type j = ci[k];
            if (j != i && j < nown && color[j] != 255) used |= 1ULL << color[j];
        }
        index_type c = 0;
        while ((used >> c) & 1ULL) ++c;
        ASSERT(c < 64 && "node coloring needs more than 64 colors");
        color[i] = (u8)c;
        count[c]++;
        if (c + 1 > nc) nc = c + 1;
    }
    d->num_color = nc;
    d->color_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nc + 1));
    d->color_offset[0] = 0;
    for (index_type c = 0; c < nc; ++c) d->color_offset[c + 1] = d->color_offset[c] + count[c];
    index_type* rows = (index_type*)CdamMallocHost((ptrdiff_t)(nown > 0 ? nown : 1) * SIZE_OF(index_type));
    index_type cur[64];
    memcpy(cur, d->color_offset, sizeof(index_type) * (size_t)nc);
    for (index_type i = 0; i < nown; ++i) rows[cur[color[i]]++] = i;
    /* per row slot (color order): lists of the lower- and higher-colored neighbours (nonzero index, column) */
    {
        index_type* lptr = (index_type*)CdamMallocHost(((ptrdiff_t)nown + 1) * SIZE_OF(index_type));
        index_type* uptr = (index_type*)CdamMallocHost(((ptrdiff_t)nown + 1) * SIZE_OF(index_type));
        lptr[0] = uptr[0] = 0;
        for (index_type sl = 0; sl < nown; ++sl) {
            const index_type i = rows[sl];
            index_type nl = 0, nu = 0;
            for (index_type k = rp[i]; k < rp[i + 1]; ++k) {
                const index_type j = ci[k];
                if (j >= nown || j == i) continue;
                if (color[j] < color[i]) ++nl; else if (color[j] > color[i]) ++nu;
            }
            lptr[sl + 1] = lptr[sl] + nl;
            uptr[sl + 1] = uptr[sl] + nu;
        }
        const index_type nL = lptr[nown], nU = uptr[nown];
        index_type* lnz = (index_type*)CdamMallocHost((ptrdiff_t)(nL > 0 ? nL : 1) * SIZE_OF(index_type));
        index_type* lcol = (index_type*)CdamMallocHost((ptrdiff_t)(nL > 0 ? nL : 1) * SIZE_OF(index_type));
        index_type* unz = (index_type*)CdamMallocHost((ptrdiff_t)(nU > 0 ? nU : 1) * SIZE_OF(index_type));
        index_type* ucol = (index_type*)CdamMallocHost((ptrdiff_t)(nU > 0 ? nU : 1) * SIZE_OF(index_type));
        for (index_type sl = 0; sl < nown; ++sl) {
            const index_type i = rows[sl];
            index_type pl = lptr[sl], pu = uptr[sl];
            for (index_type k = rp[i]; k < rp[i + 1]; ++k) {
                const index_type j = ci[k];
                if (j >= nown || j == i) continue;
                if (color[j] < color[i]) { lnz[pl] = k; lcol[pl++] = j; }
                else if (color[j] > color[i]) { unz[pu] = k; ucol[pu++] = j; }
            }
        }
#define UP(dst, src, cnt)                                                                         \
        dst = (index_type*)CdamMallocDevice((ptrdiff_t)((cnt) > 0 ? (cnt) : 1) * SIZE_OF(index_type)); \
        HIPGUARD(hipMemcpy(dst, src, sizeof(index_type) * (size_t)(cnt), H2D));
        UP(d->d_lptr, lptr, nown + 1) UP(d->d_uptr, uptr, nown + 1)
        UP(d->d_lnz, lnz, nL) UP(d->d_lcol, lcol, nL) UP(d->d_unz, unz, nU) UP(d->d_ucol, ucol, nU)
#undef UP
        CdamFreeHost(ucol, 0); CdamFreeHost(unz, 0); CdamFreeHost(lcol, 0); CdamFreeHost(lnz, 0);
        CdamFreeHost(uptr, 0); CdamFreeHost(lptr, 0);
    }
    d->d_rows = (index_type*)CdamMallocDevice((ptrdiff_t)(nown > 0 ? nown : 1) * SIZE_OF(index_type));
    d->d_color = (u8*)CdamMallocDevice((ptrdiff_t)(N > 0 ? N : 1));
    HIPGUARD(hipMemcpy(d->d_rows, rows, sizeof(index_type) * (size_t)nown, H2D));
    HIPGUARD(hipMemcpy(d->d_color, color, (size_t)N, H2D));
    CdamFreeHost(rows, 0);
    CdamFreeHost(color, 0);
    CdamFreeHost(ci, 0);
    CdamFreeHost(rp, 0);
}

static void dilu_release_colors(PCDilu* d) {
    if (d->color_offset) CdamFreeHost(d->color_offset, 0);
    CdamFreeDevice(d->d_rows, 0);
    CdamFreeDevice(d->d_color, 0);
    CdamFreeDevice(d->d_lptr, 0); CdamFreeDevice(d->d_lnz, 0); CdamFreeDevice(d->d_lcol, 0);
    CdamFreeDevice(d->d_uptr, 0); CdamFreeDevice(d->d_unz, 0); CdamFreeDevice(d->d_ucol, 0);
    d->d_lptr = d->d_lnz = d->d_lcol = d->d_uptr = d->d_unz = d->d_ucol = NULL;
    d->color_offset = NULL;
    d->d_rows = NULL;
    d->d_color = NULL;
}

static void dilu_setup(PC* pc) {
    PCDilu* d = (PCDilu*)pc->data;
    Matrix* A = (Matrix*)pc->mat;
    MatrixFS* fs = (MatrixFS*)A->data;
    const index_type nown = MatrixFSOwnedRows(A);
    if (d->spy != fs->spy1x1 || d->nown != nown || !d->d_rows) { /* new pattern / partition: recolor */
        dilu_release_colors(d);
        d->spy = fs->spy1x1;
        d->N = fs->spy1x1->num_row;
        d->nown = nown;
        CdamFreeDevice(d->d_Einv, 0);
        d->d_Einv = (f64*)CdamMallocDevice((ptrdiff_t)(d->N > 0 ? d->N : 1) * 16 * SIZE_OF(f64));
        dilu_build_colors(d);
    }
    const f64* val = MatrixFSBlockValues(A);
    hipStream_t s = DflStream();
    for (index_type c = 0; c < d->num_color; ++c) /* ascending: a color needs E^-1 of all lower colors */
        dfl_dilu_setup_color(d->color_offset[c + 1] - d->color_offset[c], d->d_rows + d->color_offset[c], d->nown, d->spy->row_ptr,
                             d->spy->col_ind, val, d->d_color, d->d_Einv, s);
}

static void dilu_apply(PC* pc, value_type* x, value_type* y) {
    PCDilu* d = (PCDilu*)pc->data;
    const f64* val = MatrixFSBlockValues((Matrix*)pc->mat);
    hipStream_t s = DflStream();
    const index_type n = d->n_active > 0 ? d->n_active : d->n;
    if (d->valf) {
        for (index_type c = 0; c < d->num_color; ++c)
            dfl_dilu_sweep_color_f32(1, d->color_offset[c], d->color_offset[c + 1] - d->color_offset[c], d->d_rows, d->N, d->d_lptr,
                                     d->d_lnz, d->d_lcol, d->valf, d->d_Einv, x, y, s);
        for (index_type c = d->num_color - 1; c >= 0; --c)
            dfl_dilu_sweep_color_f32(0, d->color_offset[c], d->color_offset[c + 1] - d->color_offset[c], d->d_rows, d->N, d->d_uptr,
                                     d->d_unz, d->d_ucol, d->valf, d->d_Einv, x, y, s);
    } else {
        for (index_type c = 0; c < d->num_color; ++c)
            dfl_dilu_sweep_color(1, d->color_offset[c], d->color_offset[c + 1] - d->color_offset[c], d->d_rows, d->N, d->d_lptr,
                                 d->d_lnz, d->d_lcol, val, d->d_Einv, x, y, s);
        for (index_type c = d->num_color - 1; c >= 0; --c)
            dfl_dilu_sweep_color(0, d->color_offset[c], d->color_offset[c + 1] - d->color_offset[c], d->d_rows, d->N, d->d_uptr,
                                 d->d_unz, d->d_ucol, val, d->d_Einv, x, y, s);
    }
    if (d->nown < d->N) { /* ghost entries of a Krylov vector stay zero */
        HIPGUARD(hipMemsetAsync(y + 3 * (size_t)d->nown, 0, sizeof(f64) * 3 * (size_t)(d->N - d->nown), s));
        HIPGUARD(hipMemsetAsync(y + 3 * (size_t)d->N + d->nown, 0, sizeof(f64) * (size_t)(d->N - d->nown), s));
    }
    if (n > 4 * d->N) dfl_copy_range(4 * (int64_t)d->N, n, x, y, s); /* phi / T sections: PCNone */
}

static void dilu_destroy(PC* pc) {
    PCDilu* d = (PCDilu*)pc->data;
    dilu_release_colors(d);
    CdamFreeDevice(d->d_Einv, 0);
    CdamFreeHost(d, SIZE_OF(PCDilu));
}

PC* PCCreateDILU(Matrix* mat) {
    if (!mat || !MatrixFSBlockValues(mat)) {
        fprintf(stderr, "PCCreateDILU: needs the block-mode (u,p) field-split matrix\n");
        return NULL;
    }
    PC* pc = (PC*)CdamMallocHost(SIZE_OF(PC));
    memset(pc, 0, sizeof *pc);
    PCDilu* d = (PCDilu*)CdamMallocHost(SIZE_OF(PCDilu));
    memset(d, 0, sizeof *d);
    d->n = MatrixNumRow(mat);
    pc->type = PC_ILU0;
    pc->mat = mat;
    pc->data = d;
    pc->op->setup = dilu_setup;
    pc->op->apply = dilu_apply;
    pc->op->destroy = dilu_destroy;
    return pc;
}

/* The sweeps of the following applications read the off-diagonal blocks from `valf`, a single-precision copy of the matrix's
 * block values the CALLER keeps current (dfl_bcsr_values_to_f32 after every change of the matrix) and owns; NULL = back to
 * the matrix's own double-precision values.  E^-1 (PCSetup) is always formed and applied in double precision.  For a DILU
 * used as a smoother under a flexible solver (PC_TWOLEVEL); the stand-alone PC_ILU0 never sets it. */
void PCDILUSetF32Values(PC* pc, const float* valf) {
    if (pc && pc->type == PC_ILU0) ((PCDilu*)pc->data)->valf = valf;
}
void PCDILUSetActiveLength(PC* pc, index_type n_active) {
    if (pc && pc->type == PC_ILU0) ((PCDilu*)pc->data)->n_active = n_active;
}

/* introspection for tests: number of node colors; colors copied to host `color_out[N]` (255 = ghost row) */
index_type PCDILUGetColors(PC* pc, u8* color_out) {
    PCDilu* d = (PCDilu*)pc->data;
    if (color_out && d->d_color) HIPGUARD(hipMemcpy(color_out, d->d_color, (size_t)d->N, D2H));
    return d->num_color;
}
const f64* PCDILUGetInverseBlocks(PC* pc) { return ((PCDilu*)pc->data)->d_Einv; }
