/* Matrix objects behind the reference's Matrix / MatrixOp vtable (src/matrix.h:27-147).
 *
 * MAT_TYPE_CSR : scalar CSR in the reference layout, hand-written SpMV instead of
 *                cusparseSpMV (src/matrix.c:101-165).
 * MAT_TYPE_FS  : when the grid is the (u,p) 2x2 layout the reference driver builds
 *                (src/main.c:374-391: offsets {0,3,4,..}, A00 3x3 / A01 3x1 / A10 1x3 /
 *                A11 1x1 over one nodal pattern) the four sub-matrices are stored as ONE
 *                array of 4x4 blocks ("block mode") and every operation is a single
 *                launch; otherwise the reference's per-sub-matrix loop is used.
 */
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

#define MATRIX_CALL(mat, func, ...)                                                                          \
    do {                                                                                                     \
        if ((mat)->op->func) (mat)->op->func(mat, ##__VA_ARGS__);                                            \
        else fprintf(stderr, "Matrix operation %s is not implemented for type: %d\n", #func, (mat)->type);   \
    } while (0)

/* =============================== CSR ================================================== */
static MatrixCSR* csr_of(Matrix* m) { return (MatrixCSR*)m->data; }

static value_type* csr_alloc_values(MatrixCSR* c) {
    if (!c->val) c->val = (value_type*)CdamMallocDevice((ptrdiff_t)c->attr->nnz * SIZE_OF(value_type));
    return c->val;
}
static b32 csr_is_view(MatrixCSR* c) { return c->owner && c->owner->block_mode; }

/* values in the reference layout.  A sub-matrix of a block-mode MatrixFS is a VIEW of the parent's 4x4 block array:
 * reading it materialises all four sub-matrix arrays from the block storage first (MatrixFSExportSubmatrices), and a
 * mutating operation writes them back afterwards (csr_view_commit) -- correct, not fast; the hot path never does this. */
static value_type* csr_values(Matrix* m) {
    MatrixCSR* c = csr_of(m);
    if (csr_is_view(c)) {
        MatrixFS* fs = c->owner;
        index_type n = fs->n_offset;
        dfl_block_export_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_alloc_values(csr_of(fs->mat[0])),
                            csr_alloc_values(csr_of(fs->mat[1])), csr_alloc_values(csr_of(fs->mat[n])),
                            csr_alloc_values(csr_of(fs->mat[n + 1])), DflStream());
        return c->val;
    }
    return csr_alloc_values(c);
}
static void csr_view_commit(Matrix* m) {
    MatrixCSR* c = csr_of(m);
    if (!csr_is_view(c)) return;
    MatrixFS* fs = c->owner;
    index_type n = fs->n_offset;
    dfl_block_import_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_of(fs->mat[0])->val, csr_of(fs->mat[1])->val,
                        csr_of(fs->mat[n])->val, csr_of(fs->mat[n + 1])->val, DflStream());
}
/* nodal pattern behind a (possibly row-expanded) CSR pattern */
static const CSRAttr* csr_nodal(const CSRAttr* a) { return a->parent ? a->parent : a; }

static void csr_setup(Matrix* m) { UNUSED(m); }

static void csr_zero(Matrix* m) {
    MatrixCSR* c = csr_of(m);
    HIPGUARD(hipMemsetAsync(csr_values(m), 0, (size_t)c->attr->nnz * sizeof(value_type), DflStream()));
    csr_view_commit(m);
}

static void csr_zero_row(Matrix* m, index_type n, const index_type* row, index_type shift, value_type diag) {
    MatrixCSR* c = csr_of(m);
    MatrixCSRZeroRowGPU(csr_values(m), c->attr->num_row, c->attr->num_col, c->attr->row_ptr, c->attr->col_ind, n, row, shift,
                        diag);
    csr_view_commit(m);
}

static void csr_amvpby(Matrix* m, value_type alpha, value_type* x, value_type beta, value_type* y) {
    MatrixCSR* c = csr_of(m);
    dfl_csr_spmv(c->attr->num_row, c->attr->row_ptr, c->attr->col_ind, csr_values(m), alpha, x, beta, y, DflStream());
}

static void csr_amvpby_mask(Matrix* m, value_type alpha, value_type* x, value_type beta, value_type* y, value_type* lm,
                            value_type* rm) {
    MatrixCSR* c = csr_of(m);
    value_type* input = x;
    if (rm) {
        input = (value_type*)CdamMallocDevice((ptrdiff_t)c->attr->num_col * SIZE_OF(value_type));
        VecPointwiseMult(x, rm, input, c->attr->num_col);
    }
    csr_amvpby(m, alpha, input, beta, y);
    if (lm) VecPointwiseMult(y, lm, y, c->attr->num_row);
    if (rm) {
        HIPGUARD(hipStreamSynchronize(DflStream()));
        CdamFreeDevice(input, 0);
    }
}

static void csr_matvec(Matrix* m, value_type* x, value_type* y) { csr_amvpby(m, 1.0, x, 0.0, y); }
static void csr_matvec_mask(Matrix* m, value_type* x, value_type* y, value_type* lm, value_type* rm) {
    csr_amvpby_mask(m, 1.0, x, 0.0, y, lm, rm);
}

static void csr_get_diag(Matrix* m, value_type* diag, index_type bs) {
    MatrixCSR* c = csr_of(m);
    const CSRAttr* attr = c->attr;
    ASSERT(attr->num_row == attr->num_col && "Matrix is not square");
    if (csr_is_view(c)) { /* view into the block storage */
        const CSRAttr* spy = c->owner->spy1x1;
        index_type i = c->owner_slot / c->owner->n_offset;
        if (i == 0 && bs == 3) dfl_bcsr_get_diag(spy->num_row, spy->row_ptr, spy->col_ind, c->owner->block_val, diag, NULL, NULL, DflStream());
        else if (i == 0 && bs == 1) dfl_bcsr_get_diag(spy->num_row, spy->row_ptr, spy->col_ind, c->owner->block_val, NULL, NULL, diag, DflStream());
        else if (i == 1 && bs == 1) dfl_bcsr_get_diag(spy->num_row, spy->row_ptr, spy->col_ind, c->owner->block_val, NULL, diag, NULL, DflStream());
        else { fprintf(stderr, "MatrixGetDiag: block size %d is not compatible with this sub-matrix\n", bs); ASSERT(0); }
        return;
    }
    if (bs == 1) MatrixCSRGetDiagGPU(csr_values(m), attr->row_ptr, attr->col_ind, diag, attr->num_row);
    else if (bs > 1) {
        ASSERT(attr->parent && attr->num_row == attr->parent->num_row * bs && "Block size is not compatible with the matrix size");
        const CSRAttr* p = attr->parent;
        MatrixGetDiagBlockGPU(csr_values(m), bs, p->num_row, p->num_col, p->row_ptr, p->col_ind, diag, bs, bs * bs);
    } else ASSERT(0 && "Block size should be greater than 0");
}

/* MatrixCSRSetValuesCOO / Ind, matrix.c:243-260 (set_values_ind is left NULL by the reference, wired here) */
static void csr_set_values_coo(Matrix* m, value_type alpha, index_type n, const index_type* row, const index_type* col,
                               const value_type* val, value_type beta) {
    MatrixCSR* c = csr_of(m);
    MatrixCSRSetValuesCOOGPU(csr_values(m), alpha, c->attr->num_row, c->attr->num_col, c->attr->row_ptr, c->attr->col_ind, n, row, col,
                             val, beta);
    csr_view_commit(m);
}
static void csr_set_values_ind(Matrix* m, value_type alpha, index_type n, const index_type* ind, const value_type* val,
                               value_type beta) {
    MatrixCSRSetValuesIndGPU(csr_values(m), alpha, n, ind, val, beta);
    csr_view_commit(m);
}
/* MatrixCSRAddElemValueBatched / BlockedBatched, matrix.c:262-297: one value (one block_row x block_col block) per
 * (element, a, b); the pattern handed to the launcher is the nodal one behind this matrix */
static void csr_add_elem_value_batched(Matrix* m, index_type nshl, index_type nb, const index_type* batch_ptr, const index_type* ien,
                                       const value_type* val, const index_type* mask) {
    const CSRAttr* a = csr_of(m)->attr;
    MatrixCSRAddElemValueBatchedGPU(csr_values(m), 1.0, nb, batch_ptr, ien, nshl, a->num_row, a->num_col, a->row_ptr, a->col_ind, val,
                                    1.0, mask);
    csr_view_commit(m);
}
static void csr_add_elem_value_blocked_batched(Matrix* m, index_type nshl, index_type nb, const index_type* batch_ptr,
                                               const index_type* ien, index_type br, index_type bc, const value_type* val, int lda,
                                               int stride, const index_type* mask) {
    const CSRAttr* a = csr_of(m)->attr;
    const CSRAttr* nodal = csr_nodal(a);
    if (a->num_row != nodal->num_row * br || a->num_col != nodal->num_col * bc) {
        fprintf(stderr, "MatrixAddElemValueBlockedBatched: block %d x %d does not match this CSR matrix\n", br, bc);
        ASSERT(0);
        return;
    }
    MatrixCSRAddElemValueBlockedBatchedGPU(csr_values(m), 1.0, nb, batch_ptr, ien, nshl, nodal->num_row, nodal->num_col,
                                           nodal->row_ptr, nodal->col_ind, br, bc, val, lda, stride, 1.0, mask);
    csr_view_commit(m);
}
static void csr_add_value_batched(Matrix* m, index_type nb, const index_type* brow, const index_type* bcol, const value_type* A) {
    const CSRAttr* a = csr_of(m)->attr;
    MatrixCSRSetValueBatchedGPU(csr_values(m), 1.0, a->num_row, a->num_col, a->row_ptr, a->col_ind, nb, brow, bcol, A, 1.0);
    csr_view_commit(m);
}
static void csr_add_value_blocked_batched(Matrix* m, index_type nb, const index_type* brow, const index_type* bcol, index_type br,
                                          index_type bc, const value_type* A, int lda, int stride) {
    const CSRAttr* nodal = csr_nodal(csr_of(m)->attr);
    MatrixCSRSetValueBlockedBatchedGPU(csr_values(m), 1.0, nodal->num_row, nodal->num_col, nodal->row_ptr, nodal->col_ind, nb, brow,
                                       bcol, br, bc, A, 1.0, lda, stride);
    csr_view_commit(m);
}

/* MatrixCSRCreate / MatrixCSRDestroy, matrix.c:20-60 (values are allocated on first use here) */
MatrixCSR* MatrixCSRCreate(const CSRAttr* attr, void* ctx) {
    UNUSED(ctx);
    MatrixCSR* c = (MatrixCSR*)CdamMallocHost(SIZE_OF(MatrixCSR));
    memset(c, 0, sizeof *c);
    c->attr = attr;
    return c;
}
void MatrixCSRDestroy(Matrix* m) {
    MatrixCSR* c = csr_of(m);
    CdamFreeDevice(c->val, 0);
    CdamFreeHost(c, SIZE_OF(MatrixCSR));
    CdamFreeHost(m, SIZE_OF(Matrix));
}
static void csr_destroy(Matrix* m) { MatrixCSRDestroy(m); }

Matrix* MatrixCreateTypeCSR(const CSRAttr* attr, void* ctx) {
    Matrix* m = (Matrix*)CdamMallocHost(SIZE_OF(Matrix));
    memset(m, 0, sizeof *m);
    m->size[0] = attr->num_row;
    m->size[1] = attr->num_col;
    m->type = MAT_TYPE_CSR;
    m->data = MatrixCSRCreate(attr, ctx);
    m->op->setup = csr_setup;
    m->op->zero = csr_zero;
    m->op->zero_row = csr_zero_row;
    m->op->amvpby = csr_amvpby;
    m->op->amvpby_mask = csr_amvpby_mask;
    m->op->matvec = csr_matvec;
    m->op->matvec_mask = csr_matvec_mask;
    m->op->get_diag = csr_get_diag;
    m->op->set_values_coo = csr_set_values_coo;
    m->op->set_values_ind = csr_set_values_ind;
    m->op->add_elem_value_batched = csr_add_elem_value_batched;
    m->op->add_elem_value_blocked_batched = csr_add_elem_value_blocked_batched;
    m->op->add_value_batched = csr_add_value_batched;
    m->op->add_value_blocked_batched = csr_add_value_blocked_batched;
    m->op->destroy = csr_destroy;
    return m;
}

/* =============================== FS =================================================== */
static MatrixFS* fs_of(Matrix* m) { return (MatrixFS*)m->data; }

static b32 fs_is_up_layout(MatrixFS* fs) {
    index_type n = fs->n_offset;
    if (n < 2 || !fs->spy1x1) return FALSE;
    if (fs->offset[0] != 0 || fs->offset[1] != 3 || fs->offset[2] != 4) return FALSE;
    for (index_type i = 0; i < n; ++i)
        for (index_type j = 0; j < n; ++j) {
            Matrix* s = fs->mat[i * n + j];
            b32 expect = (i < 2 && j < 2);
            if (!expect) { if (s) return FALSE; continue; }
            if (!s || s->type != MAT_TYPE_CSR) return FALSE;
            const CSRAttr* a = csr_of(s)->attr;
            index_type br = i == 0 ? 3 : 1, bc = j == 0 ? 3 : 1;
            if (a->num_row != fs->spy1x1->num_row * br || a->num_col != fs->spy1x1->num_col * bc ||
                a->nnz != fs->spy1x1->nnz * br * bc) return FALSE;
        }
    return TRUE;
}

static void fs_setup(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    index_type n = fs->n_offset;
    index_type num_row = fs->spy1x1->num_row, num_col = fs->spy1x1->num_col;
    m->size[0] = fs->offset[n] * num_row; /* matrix.c:408-409 */
    m->size[1] = fs->offset[n] * num_col;
    for (index_type i = 0; i < n * n; ++i)
        if (fs->mat[i]) MatrixSetup(fs->mat[i]);
    fs->block_mode = !fs->reference_layout && fs_is_up_layout(fs);
    if (fs->owned_rows <= 0 || fs->owned_rows > num_row) fs->owned_rows = num_row;
    if (fs->block_mode) {
        if (!fs->block_val) fs->block_val = (value_type*)CdamMallocDevice((ptrdiff_t)fs->spy1x1->nnz * 16 * SIZE_OF(value_type));
        for (index_type i = 0; i < 2; ++i)
            for (index_type j = 0; j < 2; ++j) {
                MatrixCSR* c = csr_of(fs->mat[i * n + j]);
                c->owner = fs;
                c->owner_slot = i * n + j;
            }
    } else {
        value_type** matval = (value_type**)CdamMallocHost(SIZE_OF(value_type*) * n * n);
        for (index_type i = 0; i < n * n; ++i) matval[i] = fs->mat[i] ? csr_values(fs->mat[i]) : NULL;
        HIPGUARD(hipMemcpy(fs->d_matval, matval, sizeof(value_type*) * (size_t)(n * n), H2D));
        CdamFreeHost(matval, 0);
    }
}

static void fs_zero(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    if (fs->block_mode) {
        HIPGUARD(hipMemsetAsync(fs->block_val, 0, (size_t)fs->spy1x1->nnz * 16 * sizeof(value_type), DflStream()));
        return;
    }
    for (index_type i = 0; i < fs->n_offset * fs->n_offset; ++i)
        if (fs->mat[i]) MatrixZero(fs->mat[i]);
}

/* MatrixFSZeroRow, matrix.c:449-469.  `row` holds scalar rows node*3+ic of the velocity
 * block; the pressure block-row call degenerates to a no-op in the reference (negative
 * count) -- same here. */
static void fs_zero_row(Matrix* m, index_type n, const index_type* row, index_type shift, value_type diag) {
    MatrixFS* fs = fs_of(m);
    index_type no = fs->n_offset, num_row = fs->spy1x1->num_row;
    if (fs->block_mode) { /* rows node*3+comp of the velocity block-row; the pressure block-row call is a no-op as in the reference */
        dfl_bcsr_zero_scalar_rows(num_row, fs->spy1x1->row_ptr, fs->spy1x1->col_ind, fs->block_val, n, row, shift, diag, DflStream());
        return;
    }
    for (index_type i = 0; i < no; ++i)
        for (index_type j = 0; j < no; ++j) {
            if (!fs->mat[i * no + j]) continue;
            MatrixZeroRow(fs->mat[i * no + j], n - fs->offset[i] * num_row, row + fs->offset[i] * num_row,
                          -num_row * fs->offset[i], i == j ? diag : 0.0);
        }
}

static void fs_amvpby(Matrix* m, value_type alpha, value_type* x, value_type beta, value_type* y) {
    MatrixFS* fs = fs_of(m);
    index_type no = fs->n_offset, num_row = fs->spy1x1->num_row, num_col = fs->spy1x1->num_col;
    if (fs->block_mode) { /* scal(4N) + 4 SpMV of matrix.c:471-497 in one launch */
        static int use_x4 = -1, x4_min = 4096; /* DFL_SPMV_X4=0: the matvec gathers from the reference-layout vector (the A/B) */
        if (use_x4 < 0) {
            use_x4 = !(getenv("DFL_SPMV_X4") && atoi(getenv("DFL_SPMV_X4")) == 0);
            if (getenv("DFL_SPMV_X4_MIN")) x4_min = atoi(getenv("DFL_SPMV_X4_MIN"));
        }
        const index_type rows = fs->owned_rows > 0 ? fs->owned_rows : num_row;
        if (use_x4 && beta == 0.0 && num_row >= x4_min) {
            /* one pass writes x interleaved (2 x 32 B per node), the matvec gathers 16 bytes per lane from it: 0.02 + 0.50 ms
               against 0.57 ms at 10M tets, bitwise the same result */
            value_type* x4 = DflMatrixFSInterleavedScratch(m);
            dfl_interleave4(0, num_row, num_row, x, x4, DflStream());
            dfl_bcsr_spmv_x4(0, rows, num_row, fs->spy1x1->row_ptr, fs->spy1x1->col_ind, fs->block_val, alpha, x4, y, DflStream());
            return;
        }
        dfl_bcsr_spmv_rows(fs->owned_rows, num_row, fs->spy1x1->row_ptr, fs->spy1x1->col_ind, fs->block_val, alpha, x, beta, y,
                           DflStream());
        return;
    }
    dfl_dscal(no * num_row, beta, y, DflStream());
    for (index_type i = 0; i < no; ++i)
        for (index_type j = 0; j < no; ++j)
            if (fs->mat[i * no + j])
                MatrixAMVPBY(fs->mat[i * no + j], alpha, x + fs->offset[j] * num_col, 1.0, y + fs->offset[i] * num_row);
}

static void fs_matvec(Matrix* m, value_type* x, value_type* y) { fs_amvpby(m, 1.0, x, 0.0, y); }

/* y = lm .* (alpha A (rm .* x) + beta y) over the stored block rows.  The reference's MatrixFSAMVPBYWithMask
 * (matrix.c:499-517) hands `beta` to every sub-matrix product of a block row, so that only the last block column of a row
 * survives beta = 0 -- dead code there; this is the operation its comment (matrix.c:167) describes. */
static void fs_amvpby_mask(Matrix* m, value_type alpha, value_type* x, value_type beta, value_type* y, value_type* lm,
                           value_type* rm) {
    MatrixFS* fs = fs_of(m);
    const index_type len = fs->n_offset * fs->spy1x1->num_col;
    value_type* input = x;
    if (rm) {
        input = (value_type*)CdamMallocDevice((ptrdiff_t)len * SIZE_OF(value_type));
        VecPointwiseMult(x, rm, input, len);
    }
    fs_amvpby(m, alpha, input, beta, y);
    if (lm) VecPointwiseMult(y, lm, y, fs->n_offset * fs->spy1x1->num_row);
    if (rm) {
        HIPGUARD(hipStreamSynchronize(DflStream()));
        CdamFreeDevice(input, 0);
    }
}
static void fs_matvec_mask(Matrix* m, value_type* x, value_type* y, value_type* lm, value_type* rm) {
    fs_amvpby_mask(m, 1.0, x, 0.0, y, lm, rm);
}

/* MatrixFSAddElemValueBlockedBatched, matrix.c:574-592: the live LHS scatter entry point of the reference
 * (assemble.cu:253-271 -> here -> SetBlockValueToSubmatGPU).  One batch = one color (conflict-free). */
static void fs_add_elem_value_blocked_batched(Matrix* m, index_type nshl, index_type nb, const index_type* batch_ptr,
                                              const index_type* ien, index_type br, index_type bc, const value_type* val, int lda,
                                              int stride, const index_type* mask) {
    MatrixFS* fs = fs_of(m);
    const CSRAttr* spy = fs->spy1x1;
    UNUSED(br); UNUSED(bc);
    if (fs->block_mode)
        dfl_bcsr_add_elem_blocked(fs->block_val, 1.0, nshl, nb, batch_ptr, ien, spy->row_ptr, spy->col_ind, val, lda, stride, 1.0, mask,
                                  DflStream());
    else
        SetBlockValueToSubmatGPU(fs->d_matval, 1.0, fs->n_offset, fs->d_offset, nshl, nb, batch_ptr, ien, spy->num_row, spy->num_col,
                                 spy->row_ptr, spy->col_ind, val, lda, stride, 1.0, mask);
}
/* MatrixFSAddValueBlockedBatched, matrix.c:620-648 */
static void fs_add_value_blocked_batched(Matrix* m, index_type nb, const index_type* brow, const index_type* bcol, index_type br,
                                         index_type bc, const value_type* A, int lda, int stride) {
    MatrixFS* fs = fs_of(m);
    index_type no = fs->n_offset;
    UNUSED(br); UNUSED(bc);
    for (index_type i = 0; i < no; ++i)
        for (index_type j = 0; j < no; ++j)
            if (fs->mat[i * no + j])
                MatrixAddValueBlockedBatched(fs->mat[i * no + j], nb, brow, bcol, fs->offset[i + 1] - fs->offset[i],
                                             fs->offset[j + 1] - fs->offset[j], A + fs->offset[i] * lda + fs->offset[j], lda, stride);
}

static void fs_get_diag(Matrix* m, value_type* diag, index_type bs) {
    MatrixFS* fs = fs_of(m);
    index_type no = fs->n_offset, num_row = fs->spy1x1->num_row;
    UNUSED(bs);
    HIPGUARD(hipMemsetAsync(diag, 0, sizeof(value_type) * (size_t)fs->offset[no] * (size_t)num_row, DflStream()));
    for (index_type i = 0; i < no; ++i)
        if (fs->mat[i * no + i]) MatrixGetDiag(fs->mat[i * no + i], diag + fs->offset[i] * num_row, 1);
}

void MatrixFSDestroy(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    index_type n = fs->n_offset;
    for (index_type i = 0; i < n * n; ++i)
        if (fs->mat[i]) MatrixDestroy(fs->mat[i]);
    if (fs->block_val_heap) { if (fs->block_val) HIPGUARD(hipFree(fs->block_val)); }
    else CdamFreeDevice(fs->block_val, 0);
    if (fs->x4) { if (fs->x4_pool) CdamFreeDevice(fs->x4, 0); else HIPGUARD(hipFree(fs->x4)); }
    fs->x4 = NULL;
    CdamFreeHost(fs->offset, 0);
    CdamFreeDevice(fs->d_offset, 0);
    CdamFreeDevice(fs->d_matval, 0);
    CdamFreeHost(fs->mat, 0);
    CdamFreeHost(fs->stream, 0);
    CdamFreeHost(fs, SIZE_OF(MatrixFS));
    CdamFreeHost(m, SIZE_OF(Matrix));
}

static void fs_destroy(Matrix* m) { MatrixFSDestroy(m); }

/* MatrixFSCreate, matrix.c:336-363 */
MatrixFS* MatrixFSCreate(index_type n_offset, const index_type* offset, void* ctx) {
    UNUSED(ctx);
    MatrixFS* fs = (MatrixFS*)CdamMallocHost(SIZE_OF(MatrixFS));
    memset(fs, 0, sizeof *fs);
    fs->n_offset = n_offset;
    fs->offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (n_offset + 1));
    memcpy(fs->offset, offset, sizeof(index_type) * (size_t)(n_offset + 1));
    fs->d_offset = (index_type*)CdamMallocDevice((n_offset + 1) * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(fs->d_offset, offset, sizeof(index_type) * (size_t)(n_offset + 1), H2D));
    fs->d_matval = (value_type**)CdamMallocDevice(SIZE_OF(value_type*) * n_offset * n_offset);
    fs->mat = (Matrix**)CdamMallocHost(SIZE_OF(Matrix*) * n_offset * n_offset);
    memset(fs->mat, 0, sizeof(Matrix*) * (size_t)(n_offset * n_offset));
    fs->stream = (hipStream_t*)CdamMallocHost(SIZE_OF(hipStream_t) * n_offset); /* never used for launches (matrix.c:489) */
    memset(fs->stream, 0, sizeof(hipStream_t) * (size_t)n_offset);
    return fs;
}

Matrix* MatrixCreateTypeFS(index_type n_offset, const index_type* offset, void* ctx) {
    Matrix* m = (Matrix*)CdamMallocHost(SIZE_OF(Matrix));
    memset(m, 0, sizeof *m);
    m->size[0] = offset[n_offset];
    m->size[1] = offset[n_offset];
    m->type = MAT_TYPE_FS;
    m->data = MatrixFSCreate(n_offset, offset, ctx);
    m->op->setup = fs_setup;
    m->op->zero = fs_zero;
    m->op->zero_row = fs_zero_row;
    m->op->amvpby = fs_amvpby;
    m->op->amvpby_mask = fs_amvpby_mask;
    m->op->matvec = fs_matvec;
    m->op->matvec_mask = fs_matvec_mask;
    m->op->get_diag = fs_get_diag;
    m->op->add_elem_value_blocked_batched = fs_add_elem_value_blocked_batched;
    m->op->add_value_blocked_batched = fs_add_value_blocked_batched;
    m->op->destroy = fs_destroy;
    return m;
}

value_type* DflMatrixFSInterleavedScratch(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    ASSERT(fs->block_mode && "the interleaved matvec needs the block-mode (u,p) matrix");
    if (!fs->x4) { /* a plain hipMalloc block, like the Krylov work space: vectors in the pool chunk next to the value array are
                      the slow operands of this kernel ("SpMV placement"); DFL_X4_POOL=1 for the A/B */
        const size_t bytes = (size_t)fs->spy1x1->num_row * 4 * sizeof(value_type);
        if (getenv("DFL_X4_POOL") && atoi(getenv("DFL_X4_POOL")) == 1) {
            fs->x4 = (value_type*)CdamMallocDevice((ptrdiff_t)bytes);
            fs->x4_pool = TRUE;
        } else {
            HIPGUARD(hipMalloc((void**)&fs->x4, bytes));
            HIPGUARD(hipMemsetAsync(fs->x4, 0, bytes, DflStream()));
        }
    }
    return fs->x4;
}
void DflMatrixFSMatVecX4Range(Matrix* m, const value_type* x4, value_type* y, index_type row0, index_type row1) {
    MatrixFS* fs = fs_of(m);
    ASSERT(fs->block_mode && "the interleaved matvec needs the block-mode (u,p) matrix");
    dfl_bcsr_spmv_x4(row0, row1, fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->spy1x1->col_ind, fs->block_val, 1.0, x4, y,
                     DflStream());
}
void MatrixFSMatVecRange(Matrix* m, value_type* x, value_type* y, index_type row0, index_type row1) {
    MatrixFS* fs = fs_of(m);
    ASSERT(fs->block_mode && "MatrixFSMatVecRange needs the block-mode (u,p) matrix");
    dfl_bcsr_spmv_range(row0, row1, fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->spy1x1->col_ind, fs->block_val, 1.0, x, 0.0, y,
                        DflStream());
}
void MatrixFSSetOwnedRows(Matrix* m, index_type n) {
    if (m && m->type == MAT_TYPE_FS) fs_of(m)->owned_rows = n;
}
index_type MatrixFSOwnedRows(Matrix* m) {
    if (!m || m->type != MAT_TYPE_FS) return m ? MatrixNumRow(m) : 0;
    MatrixFS* fs = fs_of(m);
    return fs->owned_rows > 0 ? fs->owned_rows : fs->spy1x1->num_row;
}

void MatrixFSUseReferenceLayout(Matrix* m, b32 on) {
    if (m && m->type == MAT_TYPE_FS) fs_of(m)->reference_layout = on;
}

value_type* MatrixFSBlockValues(Matrix* m) {
    if (!m || m->type != MAT_TYPE_FS) return NULL;
    MatrixFS* fs = fs_of(m);
    return fs->block_mode ? fs->block_val : NULL;
}

void MatrixFSExportSubmatrices(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    if (!fs->block_mode) return;
    index_type n = fs->n_offset;
    dfl_block_export_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_alloc_values(csr_of(fs->mat[0])),
                        csr_alloc_values(csr_of(fs->mat[1])), csr_alloc_values(csr_of(fs->mat[n])),
                        csr_alloc_values(csr_of(fs->mat[n + 1])), DflStream());
}

/* A (u,p) MatrixFS kept in the reference's layout (MatrixFSUseReferenceLayout) handed to the library's own assembly: the
 * kernels work on a scratch 4x4-block array -- Begin fills it from the four sub-matrix arrays, End writes it back.  Two
 * conversion passes per assembly call: the price of Level-B compatibility, not the hot path.  NULL when the
 * matrix does not hold the (u,p) 2x2 layout of main.c:374-391. */
value_type* DflMatrixFSScratchBlockBegin(Matrix* m) {
    if (!m || m->type != MAT_TYPE_FS) return NULL;
    MatrixFS* fs = fs_of(m);
    if (fs->block_mode) return fs->block_val;
    if (!fs->spy1x1 || !fs_is_up_layout(fs)) return NULL;
    const index_type n = fs->n_offset;
    if (!fs->block_val) fs->block_val = (value_type*)CdamMallocDevice((ptrdiff_t)fs->spy1x1->nnz * 16 * SIZE_OF(value_type));
    dfl_block_import_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_alloc_values(csr_of(fs->mat[0])),
                        csr_alloc_values(csr_of(fs->mat[1])), csr_alloc_values(csr_of(fs->mat[n])),
                        csr_alloc_values(csr_of(fs->mat[n + 1])), DflStream());
    return fs->block_val;
}
void DflMatrixFSScratchBlockEnd(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    if (fs->block_mode || !fs->block_val) return;
    const index_type n = fs->n_offset;
    dfl_block_export_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_alloc_values(csr_of(fs->mat[0])),
                        csr_alloc_values(csr_of(fs->mat[1])), csr_alloc_values(csr_of(fs->mat[n])),
                        csr_alloc_values(csr_of(fs->mat[n + 1])), DflStream());
}

void MatrixFSImportSubmatrices(Matrix* m) {
    MatrixFS* fs = fs_of(m);
    if (!fs->block_mode) return;
    index_type n = fs->n_offset;
    dfl_block_import_fs(fs->spy1x1->num_row, fs->spy1x1->row_ptr, fs->block_val, csr_alloc_values(csr_of(fs->mat[0])),
                        csr_alloc_values(csr_of(fs->mat[1])), csr_alloc_values(csr_of(fs->mat[n])),
                        csr_alloc_values(csr_of(fs->mat[n + 1])), DflStream());
}

/* =============================== dispatch (matrix.c:730-864) ============================ */
void MatrixDestroy(Matrix* mat) { if (mat) MATRIX_CALL(mat, destroy); }
void MatrixSetup(Matrix* mat) { ASSERT(mat && "Matrix is NULL"); MATRIX_CALL(mat, setup); }
void MatrixZero(Matrix* mat) { ASSERT(mat && "Matrix is NULL"); MATRIX_CALL(mat, zero); }
void MatrixZeroRow(Matrix* mat, index_type n, const index_type* row, index_type shift, value_type diag) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, zero_row, n, row, shift, diag);
}
void MatrixAMVPBY(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y) {
    ASSERT(A && x && y);
    MATRIX_CALL(A, amvpby, alpha, x, beta, y);
}
void MatrixAMVPBYWithMask(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y, value_type* lm,
                          value_type* rm) {
    ASSERT(A && x && y);
    MATRIX_CALL(A, amvpby_mask, alpha, x, beta, y, lm, rm);
}
void MatrixMatVec(Matrix* mat, value_type* x, value_type* y) { ASSERT(mat && x && y); MATRIX_CALL(mat, matvec, x, y); }
void MatrixMatVecWithMask(Matrix* mat, value_type* x, value_type* y, value_type* lm, value_type* rm) {
    ASSERT(mat && x && y);
    MATRIX_CALL(mat, matvec_mask, x, y, lm, rm);
}
void MatrixGetDiag(Matrix* mat, value_type* diag, index_type bs) { ASSERT(mat && diag); MATRIX_CALL(mat, get_diag, diag, bs); }
void MatrixSetValuesCOO(Matrix* mat, value_type alpha, index_type n, const index_type* row, const index_type* col,
                        const value_type* val, value_type beta) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, set_values_coo, alpha, n, row, col, val, beta);
}
void MatrixSetValuesInd(Matrix* mat, value_type alpha, index_type n, const index_type* ind, const value_type* val, value_type beta) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, set_values_ind, alpha, n, ind, val, beta);
}
void MatrixAddElemValueBatched(Matrix* mat, index_type nshl, index_type nb, const index_type* batch_ptr, const index_type* ien,
                               const value_type* val, const index_type* mask) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, add_elem_value_batched, nshl, nb, batch_ptr, ien, val, mask);
}
/* the reference skips the three dispatchers below silently when the slot is empty (matrix.c:819-864); an empty slot here
 * is reported like every other one, so that a caller never assembles nothing without a message */
void MatrixAddElemValueBlockedBatched(Matrix* mat, index_type nshl, index_type nb, const index_type* batch_ptr,
                                      const index_type* ien, index_type br, index_type bc, const value_type* val, int lda,
                                      int stride, const index_type* mask) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, add_elem_value_blocked_batched, nshl, nb, batch_ptr, ien, br, bc, val, lda, stride, mask);
}
void MatrixAddValueBatched(Matrix* mat, index_type nb, const index_type* brow, const index_type* bcol, const value_type* A) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, add_value_batched, nb, brow, bcol, A);
}
void MatrixAddValueBlockedBatched(Matrix* mat, index_type nb, const index_type* brow, const index_type* bcol, index_type br,
                                  index_type bc, const value_type* A, int lda, int stride) {
    ASSERT(mat && "Matrix is NULL");
    MATRIX_CALL(mat, add_value_blocked_batched, nb, brow, bcol, br, bc, A, lda, stride);
}

/* Move the 4x4-block value array of a block-mode MatrixFS into `new_val` (a plain hipMalloc block the matrix then owns);
   the values are copied on the library stream, the old array is released.  Used by the placement calibration of the
   Krylov solver: every consumer reads fs->block_val at call time, nothing caches the pointer. */
void DflMatrixFSRelocateBlockValues(Matrix* m, value_type* new_val) {
    MatrixFS* fs = fs_of(m);
    ASSERT(fs->block_mode && fs->block_val && new_val);
    HIPGUARD(hipMemcpyAsync(new_val, fs->block_val, (size_t)fs->spy1x1->nnz * 16 * sizeof(value_type), D2D, DflStream()));
    HIPGUARD(hipStreamSynchronize(DflStream()));
    if (fs->block_val_heap) HIPGUARD(hipFree(fs->block_val));
    else CdamFreeDevice(fs->block_val, 0);
    fs->block_val = new_val;
    fs->block_val_heap = TRUE;
}
