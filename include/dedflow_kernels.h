/* dedflow_kernels.h -- thin C ABI of the hand-written gfx950 (MI355X) kernels.
 *
 * Plain pointers and sizes only (no torch / C++ types).  Every pointer is a
 * DEVICE pointer unless the name starts with h_.  `stream` is a hipStream_t
 * passed as void* (NULL = the null stream).  Launchers are asynchronous and
 * never allocate, free or synchronise unless their comment says so; the caller
 * owns every buffer (same ownership rule as the reference).
 *
 * Each entry cites the reference interface it replaces (paths relative to
 * zexxzhao/DEDFlow @ 2024-10-16).  Where the reference called a vendor library
 * from host C (cuBLAS / cuSPARSE / cuRAND / Thrust / CUB) the replacement is a
 * dfl_* launcher here.
 *
 * Native matrix layout ("block CSR"): one nodal pattern (row_ptr[N+1],
 * col_ind[nnz1], sorted ascending per row) shared by all sub-matrices and one
 * 4x4 block of f64 per nodal nonzero, val[k*16 + r*4 + c] with r,c in
 * (u0,u1,u2,p).  It replaces the reference's four row-expanded scalar CSR
 * arrays A00/A01/A10/A11 (src/main.c:385-391, src/csr_impl.cu:24-59); the
 * phi/T rows and columns are not stored, exactly as the reference drops them
 * (NULL sub-matrices, src/matrix_impl.cu:424-426).  dfl_block_export_fs /
 * dfl_block_import_fs convert to/from the reference layout.
 * Global vectors keep the reference layout [u: Nx3 AoS | p: N | phi: N | T: N]
 * (src/main.c:108-118).
 */
#ifndef DEDFLOW_KERNELS_H
#define DEDFLOW_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t dfl_index;   /* index_type  = i32 (config/config.mk:51) */
typedef double dfl_value;    /* value_type  = f64 */

/* ---- library / device ------------------------------------------------------- */
int dfl_abi_version(void);
/* last HIP error text seen by a launcher (thread-unsafe, like the reference's CUGUARD printf) */
const char* dfl_last_error(void);

/* ---- BLAS-1 on device vectors (replaces cublasD{axpy,copy,scal,nrm2,dot},
 *      src/krylov.c:114-319, src/main.c:107-130,226,242-265,544-565; VecAXPY etc. src/vec.cu:14-76) */
void dfl_daxpy(dfl_index n, dfl_value alpha, const dfl_value* x, dfl_value* y, void* stream);
void dfl_dscal(dfl_index n, dfl_value alpha, dfl_value* x, void* stream);
void dfl_dcopy(dfl_index n, const dfl_value* x, dfl_value* y, void* stream);
void dfl_dset(dfl_index n, dfl_value alpha, dfl_value* x, void* stream);               /* SetValGPU, matrix_impl.cu:467-471 */
void dfl_pointwise_mult(dfl_index n, const dfl_value* x, const dfl_value* y, dfl_value* z, void* stream); /* VecPointwiseMult */
void dfl_pointwise_div(dfl_index n, const dfl_value* x, const dfl_value* y, dfl_value* z, void* stream);  /* VecPointwiseDiv  */
void dfl_pointwise_inv(dfl_index n, dfl_value* x, void* stream);                                          /* VecPointwiseInv  */
/* halo pack / unpack for element-partitioned runs: out[i] = x[idx[i]];  x[idx[i]] = in[i] (idx unique) */
void dfl_gather_idx(dfl_index n, const dfl_index* idx, const dfl_value* x, dfl_value* out, void* stream);
void dfl_scatter_idx(dfl_index n, const dfl_index* idx, const dfl_value* in, dfl_value* x, void* stream);
/* deterministic two-stage reductions; result written to *d_out (device).  `work`
 * holds at least dfl_reduce_work_size() doubles. */
dfl_index dfl_reduce_work_size(void);
void dfl_ddot(dfl_index n, const dfl_value* x, const dfl_value* y, dfl_value* d_out, dfl_value* work, void* stream);
void dfl_dnrm2(dfl_index n, const dfl_value* x, dfl_value* d_out, dfl_value* work, void* stream);
/* one resident wave for about `us` microseconds (<= 20000; bounded whatever the clock does): a probe for whether two streams
 * of this process run concurrently (host/comm_rccl.c picks its halo stream with it) */
void dfl_spin_us(int us, void* stream);
/* x *= 1 / *d_scale  (cublasDscal with the reciprocal of a device-resident norm, krylov.c:130-131,235-237) */
void dfl_dscal_inv_dev(dfl_index n, const dfl_value* d_scale, dfl_value* x, void* stream);

/* ---- generalized-alpha state algebra of the Newton driver, one pass each (replaces the cublasDaxpy / Dcopy / Dscal /
 *      Dnrm2 sequences of src/main.c:107-130, 242-265, 544-565; vectors are [u: Nx3 | p | phi | T]):
 *  dfl_alpha_states  : dwgalpha = f1_0 dwgold + f1_1 dwg (p slot: dwg), wgalpha = wgold + f2_0 dwgold + f2_1 dwg (p slot: 0);
 *                      nodep != NULL also writes the packed gather records of dfl_pack_nodes from these states and xg
 *  dfl_alpha_predict : dwg *= fac except the p slot;   dfl_alpha_correct: wgold += c0 dwgold + c1 dwg (except p), dwgold = dwg
 *  dfl_norms4        : d_out4[k] = ||F_u||, ||F_p||, ||F_phi||, ||F_T|| (take_sqrt = 0: sums of squares, for partitioned
 *                      runs that all-reduce first); work >= dfl_reduce_work_size() doubles */
void dfl_alpha_states(dfl_index N, const dfl_value* wgold, const dfl_value* dwgold, const dfl_value* dwg, dfl_value f1_0,
                      dfl_value f1_1, dfl_value f2_0, dfl_value f2_1, const dfl_value* xg, dfl_value* wgalpha, dfl_value* dwgalpha,
                      dfl_value* nodep, void* stream);
void dfl_alpha_states2(dfl_index N, const dfl_value* wgold, const dfl_value* dwgold, const dfl_value* dwg, dfl_value f1_0,
                       dfl_value f1_1, dfl_value f2_0, dfl_value f2_1, const dfl_value* xg, dfl_value* wgalpha, dfl_value* dwgalpha,
                       dfl_value* nodep /*or NULL*/, dfl_value* nodexu /*or NULL: compact (x, u) records*/, void* stream);
void dfl_alpha_predict(dfl_index N, dfl_value fac, dfl_value* dwg, void* stream);
void dfl_alpha_correct(dfl_index N, dfl_value c0, dfl_value c1, dfl_value* wgold, dfl_value* dwgold, const dfl_value* dwg, void* stream);
void dfl_norms4(dfl_index N, const dfl_value* F, dfl_value* d_out4, int take_sqrt, dfl_value* work, void* stream);

/* ---- fused classical Gram-Schmidt (replaces the two cublasDgemv of krylov.c:166-183
 *      and the Dnrm2 of :230).
 *   dfl_cgs_dots   : d_h[j] = Q[:,j] . w, j < ncol  (one pass over Q[:,0:ncol] and ~ncol/16 passes over w)
 *   dfl_cgs_update : w -= Q[:,0:ncol] h ; *d_nrm = ||w||_2   (one pass over Q, w read+written once)
 * Q is column-major with leading dimension ldq.  `work` >= dfl_cgs_work_size(n, ncol) doubles. */
int64_t dfl_cgs_work_size(dfl_index n, dfl_index ncol);
void dfl_cgs_dots(dfl_index n, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* w, dfl_value* d_h,
                  dfl_value* work, void* stream);
void dfl_cgs_update(dfl_index n, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* d_h, dfl_value* w,
                    dfl_value* d_nrm, int take_sqrt, dfl_value* work, void* stream);
/* dfl_cgs_update (with the square root) followed by the Givens step of column `iter`; the second stage of the norm
 * and the Givens recurrence share one launch (2 launches instead of 3) */
void dfl_cgs_update_givens(dfl_index n, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* d_h, dfl_value* w,
                           dfl_value* d_nrm, dfl_value* work, dfl_index iter, dfl_value* d_H, dfl_index ldh, dfl_value* d_gv,
                           dfl_value* d_beta, dfl_value* d_res_hist, void* stream);
void dfl_dsqrt_dev(dfl_value* d_val, void* stream); /* *d_val = sqrt(*d_val) */
/* y = Q[:,0:ncol] c  (cublasDgemv OP_N of krylov.c:304-311) */
void dfl_gemv_n(dfl_index n, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* d_c, dfl_value* y, void* stream);

/* ---- GMRES small recurrences, device resident (replaces cublasDrot x k, Drotg,
 *      cudaMemset(8B), GMRESResidualUpdatePrivate of krylov.c:256-277, krylov_util.cu:5-19).
 * Column `iter` of H (leading dimension ldh) holds h[0..iter]; *d_nrm = ||w|| becomes h[iter+1].
 * d_res_hist[iter] = |beta[iter+1]|. */
void dfl_gmres_givens(dfl_index iter, const dfl_value* d_nrm, dfl_value* d_H, dfl_index ldh, dfl_value* d_gv,
                      dfl_value* d_beta, dfl_value* d_res_hist, void* stream);
/* same, for partitioned runs: *d_nrm_sq holds the all-reduced squared norm and is replaced by its square root first */
void dfl_gmres_givens_sq(dfl_index iter, dfl_value* d_nrm_sq, dfl_value* d_H, dfl_index ldh, dfl_value* d_gv,
                         dfl_value* d_beta, dfl_value* d_res_hist, void* stream);
/* fused-norm option of partitioned runs: H[0..iter, iter] = all-reduced h, H[iter+1, iter] = all-reduced w.w from the same
 * reduction; the norm of the orthogonalised vector comes from w.w - sum h_j^2 (written to *d_nrm); *d_flag (int, may be
 * NULL) is raised when cancellation leaves less than 1e-6 of w.w */
/* partitioned runs, fused norm + Jacobi tree: w -= Q h, the Pythagorean norm and Givens step of column `iter` (d_hraw =
 * [h_0..h_iter, w.w] all-reduced, copied into H), q = w / nrm in place and z = M^-1 q for the next Arnoldi step: one launch */
void dfl_cgs_update_pc_givens(dfl_index nrows, dfl_index N, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* d_hraw,
                              dfl_value* w, const dfl_value* dinv33, const dfl_value* dinv1, dfl_value* z, dfl_index iter,
                              dfl_value* d_H, dfl_index ldh, dfl_value* d_gv, dfl_value* d_beta, dfl_value* d_res_hist,
                              dfl_value* d_nrm, int* d_flag, void* stream);
/* the same, also writing z interleaved into z4[node][4] (owned rows) for dfl_bcsr_spmv_x4; z4 == NULL: as above */
void dfl_cgs_update_pc_givens_x4(dfl_index nrows, dfl_index N, dfl_index ncol, const dfl_value* Q, int64_t ldq, const dfl_value* d_hraw,
                                 dfl_value* w, const dfl_value* dinv33, const dfl_value* dinv1, dfl_value* z, dfl_value* z4,
                                 dfl_index iter, dfl_value* d_H, dfl_index ldh, dfl_value* d_gv, dfl_value* d_beta,
                                 dfl_value* d_res_hist, dfl_value* d_nrm, int* d_flag, void* stream);
void dfl_gmres_givens_pythagoras(dfl_index iter, dfl_value* d_nrm, dfl_value* d_H, dfl_index ldh, dfl_value* d_gv,
                                 dfl_value* d_beta, dfl_value* d_res_hist, int* d_flag, void* stream);
/* back substitution H[0:m,0:m] y = beta[0:m] in place on beta (cublasDtrsv, krylov.c:297-301) */
void dfl_gmres_trsv(dfl_index m, const dfl_value* d_H, dfl_index ldh, dfl_value* d_beta, void* stream);
void GMRESResidualUpdatePrivate(dfl_value* beta, dfl_value* gv); /* same symbol as krylov_util.cu:22-24 */

/* ---- block-CSR SpMV (replaces scal + 4 x cusparseSpMV, src/matrix.c:101-165,471-497):
 *      y[0:4N] = alpha * A * x[0:4N] + beta * y[0:4N] */
void dfl_bcsr_spmv(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val, dfl_value alpha,
                   const dfl_value* x, dfl_value beta, dfl_value* y, void* stream);
/* element-partitioned runs: only the first `nrows` node rows (the nodes this rank owns) are
 * computed; x / y keep the local layout with N = owned + ghost nodes */
void dfl_bcsr_spmv_rows(dfl_index nrows, dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val,
                        dfl_value alpha, const dfl_value* x, dfl_value beta, dfl_value* y, void* stream);
/* node rows [row0, row1) only (interior / boundary split that overlaps the halo exchange with the matvec) */
void dfl_bcsr_spmv_range(dfl_index row0, dfl_index row1, dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind,
                         const dfl_value* val, dfl_value alpha, const dfl_value* x, dfl_value beta, dfl_value* y, void* stream);
void dfl_pc_jacobi_setup_rows(dfl_index nrows, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val,
                              dfl_value* dinv33, dfl_value* dinv1, void* stream);
void dfl_pc_jacobi_apply_rows(dfl_index nrows, dfl_index N, dfl_index n, const dfl_value* dinv33, const dfl_value* dinv1,
                              const dfl_value* x, dfl_value* y, void* stream);
void dfl_pc_jacobi_apply_scaled_rows(dfl_index nrows, dfl_index N, dfl_index n, const dfl_value* dinv33, const dfl_value* dinv1,
                                     const dfl_value* w, const dfl_value* d_nrm, dfl_value* q_out, dfl_value* y, void* stream);
/* multicolor block-DILU (csrc/k_dilu.hip): rows[0:nrows_c] = node rows of one color; color[N] u8; Einv[N][16].
 * setup: E_i^-1 of one color (all lower colors already done; columns >= nown (ghosts) are ignored).
 * sweep: one color of the forward (z = E^-1(r - L z)) or backward (z -= E^-1 U z) substitution over rows[slot0 ..
 * slot0+nrows_c); the strictly lower / upper neighbours of the row at slot s are enz / ecol [eptr[s], eptr[s+1])
 * (nodal nonzero index and column node), built once per coloring by the host */
void dfl_dilu_setup_color(dfl_index nrows_c, const dfl_index* rows, dfl_index nown, const dfl_index* row_ptr,
                          const dfl_index* col_ind, const dfl_value* val, const unsigned char* color, dfl_value* Einv, void* stream);
void dfl_dilu_sweep_color(int forward, dfl_index slot0, dfl_index nrows_c, const dfl_index* rows, dfl_index N,
                          const dfl_index* eptr, const dfl_index* enz, const dfl_index* ecol, const dfl_value* val,
                          const dfl_value* Einv, const dfl_value* r, dfl_value* z, void* stream);
void dfl_copy_range(int64_t begin, int64_t end, const dfl_value* x, dfl_value* y, void* stream); /* y[begin:end] = x[begin:end] */
/* scalar CSR SpMV for the reference-layout sub-matrices (cusparseSpMV, matrix.c:151-162) */
void dfl_csr_spmv(dfl_index nrow, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val, dfl_value alpha,
                  const dfl_value* x, dfl_value beta, dfl_value* y, void* stream);

/* ---- preconditioner (src/pc.c:44-147, src/krylov.c:439-453):
 *  setup : dinv33[9N] = image of inv(D_uu)^T as the reference stores it (row-major extract,
 *          column-major inverse, Q7), dinv1[N] = 1 / A_pp diagonal
 *  apply : y[0:3N] = inv(D)^T x, y[3N:4N] = x * dinv1, y[4N:n] = x (PCNone sections) */
void dfl_pc_jacobi_setup(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val,
                         dfl_value* dinv33, dfl_value* dinv1, void* stream);
void dfl_pc_jacobi_apply(dfl_index N, dfl_index n, const dfl_value* dinv33, const dfl_value* dinv1, const dfl_value* x,
                         dfl_value* y, void* stream);
/* same, fused with the normalisation of the incoming Krylov vector:
 *   q = w / *d_nrm (stored to q_out), y = M^{-1} q */
void dfl_pc_jacobi_apply_scaled(dfl_index N, dfl_index n, const dfl_value* dinv33, const dfl_value* dinv1, const dfl_value* w,
                                const dfl_value* d_nrm, dfl_value* q_out, dfl_value* y, void* stream);
/* stand-alone pieces of the same preconditioner (generic PC tree):
 *  dfl_block3_invert : in place, row-major 3x3 blocks -> the reference's post-getri memory image (pc.c:75-77)
 *  dfl_block3_apply  : cublasDgemvStridedBatched(OP_N) on that image (pc.c:104-112) */
void dfl_block3_invert(dfl_index N, dfl_value* diag33, void* stream);
void dfl_block3_apply(dfl_index N, const dfl_value* dinv33, const dfl_value* x, dfl_value* y, void* stream);
/* diagonal extraction with the reference's semantics (matrix_impl.cu:25-44, 642-683) */
void dfl_bcsr_get_diag(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val,
                       dfl_value* diag33_rowmajor /*9N or NULL*/, dfl_value* diag_p /*N or NULL*/,
                       dfl_value* diag_u_scalar /*3N or NULL*/, void* stream);

/* ---- layout conversion block CSR <-> reference FS layout (parity tests, export) */
void dfl_block_export_fs(dfl_index N, const dfl_index* row_ptr, const dfl_value* val, dfl_value* A00, dfl_value* A01,
                         dfl_value* A10, dfl_value* A11, void* stream);
void dfl_block_import_fs(dfl_index N, const dfl_index* row_ptr, dfl_value* val, const dfl_value* A00, const dfl_value* A01,
                         const dfl_value* A10, const dfl_value* A11, void* stream);

/* ---- Dirichlet (src/dirichlet_impl.cu:15-36, src/matrix_impl.cu:6-23, src/matrix.c:449-469) */
void ApplyBCVecNodalGPU(dfl_value* b, dfl_index n_bc_node, const dfl_index* bc_node, dfl_index shape, dfl_index init);
void GetRowFromNodeGPU(dfl_index n, dfl_index* row, dfl_index shape, dfl_index init);
void GetNodeFromRowGPU(dfl_index n, dfl_index* node, dfl_index shape);
void dfl_dirichlet_vec(dfl_value* b, dfl_index n_bnode, const dfl_index* bnode, dfl_index shape, dfl_index comp, void* stream);
/* rows (node*3+comp) of the block matrix <- diag * unit row (velocity part), pressure column part <- 0 */
void dfl_bcsr_zero_rows(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, dfl_value* val, dfl_index n_bnode,
                        const dfl_index* bnode, dfl_index comp, dfl_value diag, void* stream);
/* reference-layout launcher, same symbol/signature as matrix_impl.h:10-12 (Q3-safe) */
void MatrixCSRZeroRowGPU(dfl_value* matval, dfl_index num_row, dfl_index num_col, const dfl_index* row_ptr,
                         const dfl_index* col_ind, dfl_index n, const dfl_index* row, dfl_index shift, dfl_value diag);
void MatrixCSRGetDiagGPU(const dfl_value* val, const dfl_index* row_ptr, const dfl_index* col_ind, dfl_value* diag,
                         dfl_index num_row);
void MatrixGetDiagBlockGPU(const dfl_value* matval, dfl_index block_size, dfl_index num_row, dfl_index num_col,
                           const dfl_index* row_ptr, const dfl_index* col_idx, dfl_value* diag_block, int lda, int stride);
void SetValGPU(dfl_value* val, dfl_index n, dfl_value alpha);

/* ---- sparsity pattern (host algorithm in the reference: src/csr.c:81-190; expansion src/csr_impl.cu:24-59) */
/* counts per-row unique neighbours into row_len[N]; *d_overflow != 0 if a row exceeds 64 (csr.c:63 ASSERT) */
void dfl_pattern_count(dfl_index N, const dfl_index* ien, const dfl_index* v2e_row, const dfl_index* v2e_col,
                       dfl_index* row_len, dfl_index* d_overflow, void* stream);
void dfl_pattern_fill(dfl_index N, const dfl_index* ien, const dfl_index* v2e_row, const dfl_index* v2e_col,
                      const dfl_index* row_ptr, dfl_index* col_ind, void* stream);
/* exclusive scan of len[n] into ptr[n+1] (thrust::inclusive_scan, color_impl.cu:35); temp from dfl_scan_temp_bytes */
int64_t dfl_scan_temp_bytes(dfl_index n);
void dfl_exclusive_scan_i32(dfl_index n, const dfl_index* len, dfl_index* ptr, void* temp, int64_t temp_bytes, void* stream);
/* ExpandCSRByBlockSize on raw arrays (csr_impl.cu:126-156), last row_ptr entry written (Q3 fix) */
void dfl_csr_expand(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, dfl_index br, dfl_index bc,
                    dfl_index* new_row_ptr, dfl_index* new_col_ind, void* stream);
/* (elem,a,b) -> nodal nonzero index, replaces the per-thread linear col_ind search of matrix_impl.cu:407-411 */
void dfl_elem_nzmap(dfl_index T, const dfl_index* ien_b, const dfl_index* row_ptr, const dfl_index* col_ind,
                    dfl_index* nzmap_b, void* stream);

/* ---- coloring + batching (src/color_impl.cu:17-255, src/indexing.cu:92-102, src/Mesh.c:165-206) */
void GenerateV2EMapRowTetGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, dfl_index* row_ptr);
void GenerateV2EMapColTetGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, const dfl_index* row_ptr,
                             dfl_index* col_idx);
/* prism (6 vertices) and hex (8) flavours of the same map (color_impl.h:12-16; those element types are otherwise empty
 * in the reference and out of scope here) */
void GenerateV2EMapRowPrismGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, dfl_index* row_ptr);
void GenerateV2EMapColPrismGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, const dfl_index* row_ptr, dfl_index* col_idx);
void GenerateV2EMapRowHexGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, dfl_index* row_ptr);
void GenerateV2EMapColHexGPU(const dfl_index* ien, dfl_index num_elem, dfl_index num_node, const dfl_index* row_ptr, dfl_index* col_idx);
void GenerateRandomColor(dfl_index* color, dfl_index num_elem, dfl_index max_color); /* XORWOW(1234), LEGACY ordering */
void ColorElementJPLTetGPU(const dfl_index* ien, const dfl_index* row_ptr, const dfl_index* col_ind, dfl_index max_color,
                           dfl_index* color, dfl_index num_elem);
void GetMaxColorGPU(const dfl_index* color, dfl_index num_elem, dfl_index* h_max_color);
dfl_index CountValueColorLegacy(const dfl_index* data, dfl_index n, dfl_index value);
void FindValueColor(const dfl_index* data, dfl_index n, dfl_index value, dfl_index* result);
dfl_index CountValueI(const dfl_index* data, dfl_index n, dfl_index value);                      /* indexing.h:9 */
void FindValueI(const dfl_index* data, dfl_index n, dfl_index value, dfl_index* result);          /* indexing.h:10 */
dfl_index CountValueColor(const dfl_index* data, dfl_index n, dfl_index value, void* buffer);     /* indexing.h:13, buffer unused */
/* positions of (row[i], col[i]) in a CSR pattern, -1 if absent (kernel behind CSRAttrGetNZIndBatchedGPU, csr_impl.h:7-9) */
void dfl_csr_find_nz(dfl_index batch_size, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_index* row,
                     const dfl_index* col, dfl_index* ind, void* stream);
/* pc_impl.h:6-7: point Jacobi on a scalar CSR matrix, y = x / diag(A) (in place: x /= diag(A)) */
void PCJacobiDevice(dfl_index n, dfl_index nnz, dfl_value* data, dfl_index* row_ptr, dfl_index* col_idx, dfl_value* x, dfl_value* y);
void PCJacobiInplaceDevice(dfl_index n, dfl_index nnz, dfl_value* data, dfl_index* row_ptr, dfl_index* col_idx, dfl_value* x);
/* matrix_impl.h:59-64: the reference's colored element-block scatter into the row-expanded sub-matrix arrays (a14).  The
 * assembly kernels of this library scatter directly into the block array and never call it; it is exported for a host that
 * keeps its own element kernels.  matval = DEVICE array of n_offset^2 device pointers (NULL = sub-matrix absent), offset =
 * DEVICE array [n_offset+1]; one batch must be conflict-free (a color), as in the reference. */
void SetBlockValueToSubmatGPU(dfl_value** matval, dfl_value alpha, dfl_index n_offset, const dfl_index* offset, dfl_index nshl,
                              dfl_index batch_size, const dfl_index* batch_index_ptr, const dfl_index* ien, dfl_index num_row,
                              dfl_index num_col, const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val, int lda,
                              int stride, dfl_value beta, const dfl_index* mask);
/* matrix_impl.h:29-62: the single-matrix forms of the same scatter.  (row_ptr, col_ind) is the NODAL pattern; matval is a
 * scalar CSR array (block 1x1) or a row-expanded block_row x block_col array over it (csr_impl.cu:24-59).  The reference's
 * kernels behind these names (matrix_impl.cu:88-208) derive (a, b) from the element id and mix two layouts -- dead code
 * there; implemented here as their call sites describe them.  One batch must be conflict-free. */
void MatrixCSRAddElemValueBatchedGPU(dfl_value* matval, dfl_value alpha, dfl_index batch_size, const dfl_index* batch_index_ptr,
                                     const dfl_index* ien, dfl_index nshl, dfl_index num_row, dfl_index num_col,
                                     const dfl_index* row_ptr, const dfl_index* col_ind, const dfl_value* val, dfl_value beta,
                                     const dfl_index* mask);
void MatrixCSRAddElemValueBlockedBatchedGPU(dfl_value* matval, dfl_value alpha, dfl_index batch_size, const dfl_index* batch_index_ptr,
                                            const dfl_index* ien, dfl_index nshl, dfl_index num_row, dfl_index num_col,
                                            const dfl_index* row_ptr, const dfl_index* col_ind, dfl_index block_row,
                                            dfl_index block_col, const dfl_value* val, int lda, int stride, dfl_value beta,
                                            const dfl_index* mask);
void MatrixCSRAddElementLHSGPU(dfl_value* matval, dfl_index nshl, dfl_index bs, dfl_index num_row, const dfl_index* row_ptr,
                               dfl_index num_col, const dfl_index* col_ind, dfl_index batch_size, const dfl_index* batch_ptr,
                               const dfl_index* ien, const dfl_value* val, int lda);
void MatrixCSRSetValueBatchedGPU(dfl_value* matval, dfl_value alpha, dfl_index csr_num_row, dfl_index csr_num_col,
                                 const dfl_index* csr_row_ptr, const dfl_index* csr_col_ind, dfl_index batch_size,
                                 const dfl_index* batch_row_ind, const dfl_index* batch_col_ind, const dfl_value* A, dfl_value beta);
void MatrixCSRSetValueBlockedBatchedGPU(dfl_value* matval, dfl_value alpha, dfl_index csr_num_row, dfl_index csr_num_col,
                                        const dfl_index* csr_row_ptr, const dfl_index* csr_col_ind, dfl_index batch_size,
                                        const dfl_index* batch_row_ind, const dfl_index* batch_col_ind, dfl_index block_row,
                                        dfl_index block_col, const dfl_value* A, dfl_value beta, int lda, int stride);
/* the same element-block scatter straight into the 4x4 block array (block-mode MatrixFS): rows / columns 0..3 of every
 * lda-strided (a, b) block, block = alpha * block + beta * element block */
void dfl_bcsr_add_elem_blocked(dfl_value* block_val, dfl_value alpha, dfl_index nshl, dfl_index batch_size,
                               const dfl_index* batch_index_ptr, const dfl_index* ien, const dfl_index* row_ptr,
                               const dfl_index* col_ind, const dfl_value* val, int lda, int stride, dfl_value beta,
                               const dfl_index* mask, void* stream);
/* MatrixZeroRow on the block array: row[i] + shift = scalar row node*3 + comp of the velocity block-row; others skipped */
void dfl_bcsr_zero_scalar_rows(dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, dfl_value* val, dfl_index n,
                               const dfl_index* row, dfl_index shift, dfl_value diag, void* stream);
/* matrix_impl.h:17-26 (scalar CSR value setters; off the hot path, kept for launcher-level completeness) */
void MatrixCSRSetValuesCOOGPU(dfl_value* matval, dfl_value alpha, dfl_index num_row, dfl_index num_col, const dfl_index* row_ptr,
                              const dfl_index* col_ind, dfl_index n, const dfl_index* row, const dfl_index* col,
                              const dfl_value* val, dfl_value beta);
void MatrixCSRSetValuesIndGPU(dfl_value* matval, dfl_value alpha, dfl_index n, const dfl_index* ind, const dfl_value* val,
                              dfl_value beta);
/* all colors at once: stable counting sort of element ids by color.
 * h_batch_offset[num_color+1] (host), batch_ind[T] (device). Synchronises. */
void dfl_color_batches(const dfl_index* color, dfl_index T, dfl_index num_color, dfl_index* h_batch_offset,
                       dfl_index* batch_ind);
/* ien_b[i*4+a] = ien[batch_ind[i]*4+a] : batch-ordered connectivity for streaming reads */
void dfl_gather_ien(dfl_index T, const dfl_index* ien, const dfl_index* batch_ind, dfl_index* ien_b, void* stream);
/* number of adjacent equal-priority pairs (Q1 diagnostic); synchronises */
dfl_index dfl_count_priority_ties(const dfl_index* ien, dfl_index T, const dfl_index* v2e_row, const dfl_index* v2e_col,
                                  const dfl_index* prio);

/* ---- element assembly, one launch per color batch (src/assemble.cu:1559-1738 chain fused).
 *  ien_b / nzmap_b point at the first element of the batch (batch-ordered arrays).
 *  Node data is gathered from a packed per-node record array (one 128-byte line per node:
 *  x[3] u[3] phi T du[3] p dphi dT pad pad) written once per assembly call by dfl_pack_nodes --
 *  replaces the 8 LoadElementValueKernel launches per batch (assemble.cu:1601-1619,1663-1678).
 *  The RHS kernel accumulates into packed 64-byte residual records (F_u[3] F_p F_phi F_T pad pad);
 *  dfl_unpack_rhs adds them to F in the reference layout and clears the packed buffer. */
void dfl_pack_nodes(dfl_index N, const dfl_value* xg, const dfl_value* wgalpha, const dfl_value* dwgalpha /*or NULL*/,
                    dfl_value* nodep /*[N][16]*/, void* stream);
/* the same, and (nodexu != NULL) the compact records of the Jacobian kernel: nodexu[i][8] = x[3] u[3] pad pad, 64 B per node;
 * nodep == NULL writes the compact records only (and reads only xg and the velocity part of wgalpha) */
void dfl_pack_nodes2(dfl_index N, const dfl_value* xg, const dfl_value* wgalpha, const dfl_value* dwgalpha /*or NULL*/,
                     dfl_value* nodep, dfl_value* nodexu /*or NULL*/, void* stream);
void dfl_unpack_rhs(dfl_index N, dfl_value* Fp /*[N][8], zeroed on return*/, dfl_value* F, void* stream);
/* per-element geometry cache (static mesh): egeo[e*16 + ..] = shape gradients[12], |det J|, sum G_ij^2, 1/tr G, pad;
 * `ien_x` is the connectivity in the order the consuming kernel walks (schedule or patch order) */
void dfl_elem_geometry(dfl_index T, const dfl_index* ien_x, const dfl_value* xg, dfl_value* egeo, void* stream);
void dfl_assemble_tet_lhs(dfl_index batch_size, const dfl_index* ien_b, const dfl_index* nzmap_b, const dfl_value* egeo_b,
                          const dfl_value* nodep, dfl_value* val, void* stream);
void dfl_assemble_tet_rhs(dfl_index batch_size, const dfl_index* ien_b, const dfl_value* nodep, dfl_value* Fp, void* stream);
/* patch form of the same residual (host/patch.c: DflBuildRhsPatchSchedule): workgroup p evaluates tets
 * [p_eoff[p], p_eoff[p+1]) whose vertices are the patch nodes pnode[p_noff[p] + lien[tet*4 + a]], and writes one
 * 6-component partial record per patch node, partial[(p_noff[p] + k)*6 ..], summed in the order of the adjacency lists
 * (adj / adj_start).  dfl_rhs_node_sum then adds, for every node, its partial
 * records gidx[goff[n] .. goff[n+1]) in that order into F (reference layout).  No atomics: bitwise reproducible. */
int dfl_rhs_patch_max_nodes(void);
int dfl_rhs_patch_max_tets(void);
void dfl_assemble_tet_rhs_patch(dfl_index npatch, const dfl_index* p_eoff, const dfl_index* p_noff, const dfl_index* pnode,
                                 const unsigned char* lien, const unsigned short* adj, const unsigned short* adj_start,
                                 const dfl_value* nodep, dfl_value* partial, void* stream);
void dfl_rhs_node_sum(dfl_index N, const dfl_index* goff, const dfl_index* gidx, const dfl_value* partial, dfl_value* F,
                      void* stream);
/* patch form of the LHS assembly (assembly schedule 2): one workgroup per spatial patch of tets sums all
 * (a,b) blocks of the patch in an LDS table (ds_add_f64) and read-modify-writes each distinct block once.
 * Launch = the `npatch` patches [patch_base, patch_base+npatch) of one patch color (no shared nodes).
 * p_eoff/p_boff: per-patch element / block-slot offsets; ien_p: connectivity in patch order;
 * lslot[e*16 + a*4+b]: LDS slot of each block; blk_nz: nodal nonzero of each slot; max_slots <= 511. */
void dfl_assemble_tet_lhs_patch(dfl_index npatch, dfl_index patch_base, const dfl_index* p_eoff, const dfl_index* p_boff,
                                const dfl_index* ien_p, const unsigned short* lslot, const dfl_index* blk_nz,
                                const dfl_value* egeo_p, const dfl_value* nodep, dfl_value* val, dfl_index max_slots,
                                void* stream);
/* row-owner patch form (assembly schedule 3, host/rowpatch.c): workgroup p owns the CSR rows of its nodes.
 * item_ea[i] = tet*4 + a for every (tet, owned node a) pair of the patch, item_slot[i*4 + b] = LDS slot of block
 * (a, b), slot_nz = nodal nonzero of every slot; `tet` indexes `ien` / `egeo` (execution-schedule order in the host layer).
 * val = beta * val + assembled rows (beta = 0 overwrites: no prior MatrixZero needed). */
void dfl_assemble_tet_lhs_rowpatch(dfl_index npatch, const dfl_index* p_ioff, const dfl_index* p_soff, const dfl_index* item_ea,
                                   const unsigned short* item_slot, const dfl_index* slot_nz, const dfl_index* ien,
                                   const dfl_value* egeo, const dfl_value* nodep, dfl_value* val, dfl_value beta,
                                   dfl_index max_slots, void* stream);
/* slot-owner form (assembly schedule 4, default; host/slotpatch.c, csrc/k_assemble2.hip): workgroup p owns the CSR rows of
 * its node patch; hdr[p] = {tet_off, num_tet | num_node << 16, pos_off, num_pos, group_off, trips_lo, trips_hi, node_off};
 * pnode[node_off + n] = global id of the patch's n-th distinct node (ascending; every node of every tet touching the patch,
 * <= DFL_SLOT_NODES); ptet_lid[tet_off + k] = the four LOCAL node ids (one byte each) of the k-th tet touching the patch;
 * position q of the patch (lane pair q % 128 in pass q / 128) is nodal nonzero slot_nz[pos_off + q] & 0x3fffffff (bit 30 /
 * 31: first / other part of a slot cut into four adjacent positions) and sums the contributions its two lanes find in the
 * lane-major descriptor groups ldesc (layout: host/slotpatch.c), each (local tet << 4) | (a << 2) | b or 0xFFFF.
 * val = beta * val + assembled rows (beta = 0 overwrites).  max_tets = largest num_tet over the patches (sizes the
 * workgroup's LDS: dfl_lhs_slot_lds_bytes).  No atomics: bitwise reproducible. */
#ifndef DFL_SLOT_BLOCK
#define DFL_SLOT_BLOCK 256 /* threads of a slot-owner workgroup: caps a patch at this many tets and DFL_SLOT_BLOCK - 1 slot
                              positions (host/slotpatch.c builds to these caps) */
#define DFL_SLOT_NODES 64  /* distinct nodes of the tets touching a patch: their (x, u) records are staged in LDS, one lane
                              per node (3 x 16 B each, 3 KB) */
#endif
int dfl_lhs_slot_record_bytes(void);
int64_t dfl_lhs_slot_lds_bytes(dfl_index max_tets);
/* nodexu = the compact node records of dfl_pack_nodes2 ([N][8]: x[3] u[3] pad pad) */
void dfl_assemble_tet_lhs_slot(dfl_index npatch, const int32_t* hdr, const uint32_t* ptet_lid, const dfl_index* pnode,
                               const dfl_index* slot_nz, const uint32_t* ldesc, const dfl_value* nodexu, dfl_value* val,
                               dfl_value beta, dfl_index max_tets, void* stream);
/* wave-per-patch form of the residual (schedule 4): the padded layout of host/patch.c -- patch p holds tet slots
 * [p*pad_tets, ..) of lien / adj, node slots [p*pad_nodes, ..) of pnode / partial and adj_start[p*(pad_nodes+1) ..];
 * cnt[p] = num_tets | num_nodes << 16.  Supported shapes (pad_tets, pad_nodes): (16,32), (32,48), (64,64). */
void dfl_assemble_tet_rhs_wave(dfl_index npatch, dfl_index pad_tets, dfl_index pad_nodes, const dfl_index* cnt,
                               const dfl_index* pnode, const unsigned char* lien, const unsigned short* adj,
                               const unsigned short* adj_start, const dfl_value* nodep, dfl_value* partial, void* stream);
/* lane-per-tet form on the (64,64) shape: the adjacency as sub-lists of exactly 4 result slots (256 = zero slot),
 * sub4[p][128][4], and sub_start[p][65] (first sub-list of each patch node; entry num_nodes.. = number of sub-lists) */
void dfl_assemble_tet_rhs_lane(dfl_index npatch, const dfl_index* cnt, const dfl_index* pnode, const unsigned char* lien,
                               const unsigned short* sub4, const unsigned short* sub_start, const dfl_value* nodep,
                               dfl_value* partial, void* stream);
/* the same kernel gathering the node values from the caller's arrays (xg[N][3] and the reference-layout state vectors
 * wgalpha / dwgalpha, all non-NULL) instead of packed records: a residual-only assembly call needs no pack pass */
void dfl_assemble_tet_rhs_lane_direct(dfl_index npatch, const dfl_index* cnt, const dfl_index* pnode, const unsigned char* lien,
                                      const unsigned short* sub4, const unsigned short* sub_start, const dfl_value* xg,
                                      const dfl_value* wgalpha, const dfl_value* dwgalpha, dfl_index N, dfl_value* partial,
                                      void* stream);
/* developer probe of the patch kernel (bit 0 skip element loop, bit 1 skip flush, bit 2 skip LDS adds) */
void dfl_tune_asm(int flags);
int dfl_tune_asm_flags(void);
/* weak-BC faces of one color (src/assemble.cu:1764-1964): face list entries index f2e/forn of the group */
void dfl_assemble_face(dfl_index n_face, const dfl_index* face_list, const dfl_index* f2e, const dfl_index* forn,
                       const dfl_index* ien, dfl_index N, const dfl_value* xg, const dfl_value* wgalpha,
                       const dfl_value* dwgalpha, dfl_value* F /*or NULL*/, const dfl_index* row_ptr, const dfl_index* col_ind,
                       dfl_value* val /*or NULL*/, void* stream);
/* two-pass form of the same face terms: every face parks its contributions (pF[f][a][4], pJ[f][a*4+b][16]; NULL = part not
 * wanted), then each touched node / nodal nonzero sums its entries ent[off[k] .. off[k+1]) (= f*4+a resp. f*16+a*4+b) in
 * that order.  One launch for all faces instead of one per conflict-free class; summation order fixed by the lists. */
void dfl_assemble_face_park(dfl_index nf, const dfl_index* f2e, const dfl_index* forn, const dfl_index* ien, dfl_index N,
                            const dfl_value* xg, const dfl_value* wg, const dfl_value* dwg, dfl_value* pF, dfl_value* pJ,
                            void* stream);
void dfl_face_sum_F(dfl_index num_node_entries, const dfl_index* fnode, const dfl_index* off, const dfl_index* ent,
                    const dfl_value* pF, dfl_index N, dfl_value* F, void* stream);
void dfl_face_sum_J(dfl_index num_nz_entries, const dfl_index* fnz, const dfl_index* off, const dfl_index* ent,
                    const dfl_value* pJ, dfl_value* val, void* stream);

/* ---- two-level preconditioner (csrc/k_amg.hip, host/pc_twolevel.c; build-defined): piecewise-constant aggregation.
 *  galerkin   : coarse 4x4 blocks val_coarse[cz] = sum of val_fine[idx[off[cz] .. off[cz+1])] (list order)
 *  restrict   : rc[I] = sum of r over the nodes anode[aoff[I] .. aoff[I+1]) of aggregate I, layouts [u: 3N | p: N]
 *  prolong_add: z[i] += xc[agg[i]] */
void dfl_amg_galerkin(dfl_index nnzc, const dfl_index* off, const dfl_index* idx, const dfl_value* val_fine, dfl_value* val_coarse,
                      void* stream);
void dfl_amg_restrict(dfl_index Nc, const dfl_index* aoff, const dfl_index* anode, dfl_index N, const dfl_value* r, dfl_value* rc,
                      void* stream);
/* The matvec with x gathered from an INTERLEAVED copy x4[node][4] = (u0 u1 u2 p): the two lanes of a block row fetch their x
 * entries with one 16-byte load each instead of two 8-byte loads from the u part and the p part of the reference layout --
 * one gather instruction and about one L2 request less per nodal nonzero: 0.50 against 0.57 ms at 10M tets (6.95 TB/s),
 * bitwise the same y.  dfl_interleave4 writes the copy for nodes [node0, node1); y rows [row0, row1) = alpha * A x. */
/* the fused Jacobi-tree application (dfl_pc_jacobi_apply[_scaled]_rows; d_nrm == NULL: unscaled, q_out unused) that ALSO
 * writes y interleaved into y4[node][4] for the owned rows -- the matvec that follows needs no interleave pass; y == NULL:
 * only the interleaved copy (n == 4N then: the phi / T tail has nowhere to go) */
void dfl_pc_jacobi_apply_scaled_rows_x4(dfl_index nrows, dfl_index N, dfl_index n, const dfl_value* dinv33, const dfl_value* dinv1,
                                        const dfl_value* w, const dfl_value* d_nrm, dfl_value* q_out, dfl_value* y, dfl_value* y4,
                                        void* stream);
void dfl_interleave4(dfl_index node0, dfl_index node1, dfl_index N, const dfl_value* x, dfl_value* x4, void* stream);
void dfl_bcsr_spmv_x4(dfl_index row0, dfl_index row1, dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind,
                      const dfl_value* val, dfl_value alpha, const dfl_value* x4, dfl_value* y, void* stream);
/* single-precision copy of block values (n = nnz1 * 16 entries) and the matvec / DILU sweeps reading it: PC_TWOLEVEL's
 * smoother and residual matvec (a preconditioner under FGMRES may be inexact; everything outside it stays double) */
void dfl_bcsr_values_to_f32(int64_t n, const dfl_value* val, float* valf, void* stream);
void dfl_bcsr_spmv_f32(dfl_index nrows, dfl_index N, const dfl_index* row_ptr, const dfl_index* col_ind, const float* valf,
                       const dfl_value* x, dfl_value* y, void* stream); /* y = A x on rows [0, nrows) */
void dfl_dilu_sweep_color_f32(int forward, dfl_index slot0, dfl_index nrows_c, const dfl_index* rows, dfl_index N,
                              const dfl_index* eptr, const dfl_index* enz, const dfl_index* ecol, const float* valf,
                              const dfl_value* Einv, const dfl_value* r, dfl_value* z, void* stream);
/* the same of r - sub (the residual r - A z with sub = A z from a plain matvec) */
void dfl_amg_restrict_diff(dfl_index Nc, const dfl_index* aoff, const dfl_index* anode, dfl_index N, const dfl_value* r,
                           const dfl_value* sub, dfl_value* rc, void* stream);
void dfl_amg_prolong_add(dfl_index N, const dfl_index* agg, dfl_index Nc, const dfl_value* xc, dfl_value* z, void* stream);
/* the same for rows [0, nrows) of N (partitioned runs: the owned nodes come first) */
void dfl_amg_prolong_add_rows(dfl_index nrows, dfl_index N, const dfl_index* agg, dfl_index Nc, const dfl_value* xc, dfl_value* z,
                              void* stream);

/* ---- DEM contact sweep (build-defined; the reference's Particle.c holds storage only)
 *  model: monodisperse spheres, linear spring-dashpot normal contact F = (kn*overlap - gamma_n*vn) n between
 *  particles and against the six walls of the unit box; uniform cell list with cell edge >= 2R:
 *    dfl_dem_build_cells  counting sort of the particles by cell (4 launches, no allocation, no synchronisation):
 *                         cell_start[ncell^3 + 1], order[P] = particle ids by (cell, id), sorted[P][6] = position and
 *                         velocity in that order; count[ncell^3 + 1] and chunk_sum[dfl_dem_num_chunks(ncell^3)] are
 *                         zero-initialised scratch that the call leaves zeroed again
 *    dfl_dem_forces       acc[i] = (sum_j F_ij + F_walls) / mass, neighbours from the 27 surrounding cells
 *    dfl_dem_integrate    v += dt*a ; x += dt*v */
dfl_index dfl_dem_num_chunks(dfl_index ncell3);
void dfl_dem_build_cells(dfl_index P, const dfl_value* coord, const dfl_value* vel, dfl_value cell, dfl_index ncell,
                         dfl_index* cell_of, dfl_index* rank, dfl_index* count, dfl_index* chunk_sum, dfl_index* cell_start,
                         dfl_index* slot, dfl_index* order, dfl_value* sorted, void* stream);
void dfl_dem_integrate(dfl_index P, dfl_value dt, dfl_value* coord, dfl_value* vel, const dfl_value* acc, void* stream);
void dfl_dem_forces(dfl_index P, const dfl_value* sorted, dfl_value radius, dfl_value mass, dfl_value kn, dfl_value gamma_n,
                    dfl_value cell, dfl_index ncell, const dfl_index* order, const dfl_index* cell_start, dfl_value* acc,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEDFLOW_KERNELS_H */
