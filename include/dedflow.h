/* dedflow.h -- host-side object API of the MI355X implementation of DEDFlow's
 * per-timestep hot path.  One consolidated header that keeps the names,
 * argument meaning and struct prefixes of the reference's public headers so a
 * caller written against them (src/main.c) compiles against this library:
 *
 *   common.h:20-111   scalar typedefs, ASSERT/CEIL_DIV, Init/Finalize, GlobalContextGet
 *   alloc.h:19-36     Allocator vtable, CdamMalloc{Host,Device}/CdamFree{Host,Device}
 *   MeshData.h:10-36  Mesh3DData           Mesh.h:14-73   Mesh3D
 *   csr.h:14-36       CSRAttr              matrix.h:27-147 Matrix / MatrixOp / MatrixCSR / MatrixFS
 *   vec.h:7-10        Vec*                 dirichlet.h:8-33 Dirichlet
 *   pc.h:15-88        PC tree              krylov.h:12-30  Krylov
 *   assemble.h:13-14  AssembleSystemTet / AssembleSystemTetFace
 *   Array.h:11-39 / Particle.h:13-35       Array, ParticleContext
 *
 * Differences a caller can observe (all listed in INTEGRATION.md):
 *   - cudaStream_t fields are hipStream_t.
 *   - A MatrixFS holding the reference's (u,p) 2x2 layout stores ONE shared-pattern
 *     array of 4x4 blocks (include/dedflow_kernels.h); sub-matrix `val` arrays in the
 *     reference layout are materialised only by MatrixFSExportSubmatrices().
 *   - structs end with one extra `ext` pointer owned by the library.
 *   - expanded patterns have a correct last row_ptr entry (reference bug Q3).
 * No status returns, guard-and-trap error behaviour, caller-owned buffers, one
 * host thread: as in the reference.
 */
#ifndef DEDFLOW_H
#define DEDFLOW_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalars (common.h:20-66) ------------------------------------------------- */
typedef int8_t i8;
typedef int16_t i16;
typedef int32_t i32;
typedef int64_t i64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef float f32;
typedef double f64;
typedef char byte;
typedef f64 value_type;  /* -DUSE_F64_VALUE */
typedef i32 index_type;  /* -DUSE_I32_INDEX */
typedef int32_t b32;
typedef i32 color_t;
#ifndef TRUE
#define TRUE (1 == 1)
#define FALSE (1 == 0)
#endif
#define SIZE_OF(x) ((index_type)sizeof(x))
#define CEIL_DIV(a, b) (((a) + (b)-1) / (b))
#define UNUSED(args) ((void)(args))
#if defined(NDEBUG)
#define ASSERT(expr) /* empty */
#else
#define ASSERT(expr) \
    while (!(expr)) __builtin_trap()
#endif
typedef hipMemcpyKind MemCopyKind;
#define H2D (hipMemcpyHostToDevice)
#define D2H (hipMemcpyDeviceToHost)
#define D2D (hipMemcpyDeviceToDevice)
#define H2H (hipMemcpyHostToHost)
void DflGuardPrivate(hipError_t code, const char* file, int line);
#define HIPGUARD(err) DflGuardPrivate((err), __FILE__, __LINE__)

#define UNCOLORED (0x0)
#define MAX_COLOR (1 << 8)

/* ---- runtime (common.h:103-111, alloc.h) --------------------------------------- */
void Init(int argc, char** argv);
void Finalize(void);
typedef enum GlobalContextType { GLOBAL_CONTEXT_SPARSE_HANDLE = 0, GLOBAL_CONTEXT_BLAS_HANDLE = 1 } GlobalContextType;
/* the reference returns cuSPARSE/cuBLAS handles here; this library has no vendor
 * handles -- both slots return a pointer to the hipStream_t every launcher uses. */
void* GlobalContextGet(GlobalContextType type);
hipStream_t DflStream(void);
void DflSetStream(hipStream_t s);

typedef enum DeviceType { HOST = 0, DEVICE = 1 } DeviceType;
typedef void* UserCtxPtr;
typedef struct Allocator {
    void* (*malloc)(ptrdiff_t, UserCtxPtr);
    void (*free)(void*, ptrdiff_t, UserCtxPtr);
    UserCtxPtr ctx;
} Allocator;
Allocator* GetDefaultAllocator(int device_id);
#define CdamMallocHost(count) (GetDefaultAllocator(HOST)->malloc((ptrdiff_t)(count), GetDefaultAllocator(HOST)->ctx))
#define CdamFreeHost(ptr, count) (GetDefaultAllocator(HOST)->free(ptr, (ptrdiff_t)(count), GetDefaultAllocator(HOST)->ctx))
#define CdamMallocDevice(count) (GetDefaultAllocator(DEVICE)->malloc((ptrdiff_t)(count), GetDefaultAllocator(DEVICE)->ctx))
#define CdamFreeDevice(ptr, count) (GetDefaultAllocator(DEVICE)->free(ptr, (ptrdiff_t)(count), GetDefaultAllocator(DEVICE)->ctx))
/* the two macros above as functions, for bindings that cannot expand C macros (zero-filled, pooled: host/runtime.c) */
void* DflDeviceMalloc(int64_t bytes);
void DflDeviceFree(void* ptr);

/* ---- mesh (MeshData.h, Mesh.h) --------------------------------------------------- */
typedef struct H5FileInfo H5FileInfo;
typedef struct Mesh3DData {
    b32 is_host;
    index_type num_node, num_tet, num_prism, num_hex;
    f64* xg;         /* xg[3*num_node] */
    index_type* ien; /* ien[4*num_tet + 6*num_prism + 8*num_hex] */
} Mesh3DData;
#define Mesh3DDataNumNode(data) ((data)->num_node)
#define Mesh3DDataNumTet(data) ((data)->num_tet)
#define Mesh3DDataNumPrism(data) ((data)->num_prism)
#define Mesh3DDataNumHex(data) ((data)->num_hex)
#define Mesh3DDataCoord(data) ((data)->xg)
#define Mesh3DDataIEN(data) ((data)->ien)
#define Mesh3DDataTet(data) (Mesh3DDataNumTet(data) ? (data)->ien + 0 : NULL)
Mesh3DData* Mesh3DDataCreateHost(index_type num_node, index_type num_tet, index_type num_prism, index_type num_hex);
Mesh3DData* Mesh3DDataCreateDevice(index_type num_node, index_type num_tet, index_type num_prism, index_type num_hex);
void Mesh3DDataDestroy(Mesh3DData* data);
void Mesh3DDataCopy(Mesh3DData* dst, Mesh3DData* src, MemCopyKind kind);
Mesh3DData* Mesh3DDataCreateH5(H5FileInfo* h5f, const char* group_name); /* MeshData.c:57-109; in libdedflow_h5.so */

typedef struct Mesh3D {
    index_type num_node, num_tet, num_prism, num_hex;
    Mesh3DData* host;
    Mesh3DData* device;
    index_type num_bound;
    index_type* bound_fid;
    index_type* bound_node_offset; /* host */
    index_type* bound_node;        /* device */
    index_type* bound_elem_offset; /* host */
    index_type* bound_ien;
    index_type* bound_f2e;  /* device */
    index_type* bound_forn; /* device */
    index_type num_batch;
    index_type* batch_offset; /* host */
    index_type* batch_ind;    /* device */
    color_t num_color;
    color_t* color; /* device */
    void* ext;      /* library-owned: batch-ordered ien, (elem,a,b)->nz map, face lists */
} Mesh3D;
#define Mesh3DHost(mesh) ((mesh)->host)
#define Mesh3DDevice(mesh) ((mesh)->device)
#define Mesh3DNumNode(mesh) ((mesh)->num_node)
#define Mesh3DNumTet(mesh) ((mesh)->num_tet)
#define Mesh3DNumPrism(mesh) ((mesh)->num_prism)
#define Mesh3DNumHex(mesh) ((mesh)->num_hex)
#define Mesh3DBoundNumNode(mesh, i) ((mesh)->bound_node_offset[(i) + 1] - (mesh)->bound_node_offset[(i)])
#define Mesh3DBoundNode(mesh, i) ((mesh)->bound_node + (mesh)->bound_node_offset[(i)])
#define Mesh3DBoundNumElem(mesh, i) ((mesh)->bound_elem_offset[(i) + 1] - (mesh)->bound_elem_offset[(i)])
#define Mesh3DBoundF2E(mesh, i) ((mesh)->bound_f2e + (mesh)->bound_elem_offset[(i)])
#define Mesh3DBoundFORN(mesh, i) ((mesh)->bound_forn + (mesh)->bound_elem_offset[(i)])
Mesh3D* Mesh3DCreate(index_type num_node, index_type num_tet, index_type num_prism, index_type num_hex);
Mesh3D* Mesh3DCreateH5(H5FileInfo* h5f, const char* group_name);
void Mesh3DDestroy(Mesh3D* mesh);
void Mesh3DUpdateHost(Mesh3D* mesh);
void Mesh3DUpdateDevice(Mesh3D* mesh);
void Mesh3DColor(Mesh3D* mesh);
void Mesh3DGenerateColorBatch(Mesh3D* mesh);
/* boundary groups from host arrays (the reference only fills them from HDF5, Mesh.c:12-59) */
void Mesh3DSetBound(Mesh3D* mesh, index_type num_bound, const index_type* node_offset, const index_type* node,
                    const index_type* elem_offset, const index_type* f2e, const index_type* forn);
void ColorMeshTet(const Mesh3D* mesh, index_type max_color_len, color_t* color);
color_t GetMaxColor(const color_t* color, index_type num_elem);

/* ---- HDF5 formats (h5util.h:24-58; implemented in libdedflow_h5.so, dedflow_amd/h5/h5io.c) ----- */
H5FileInfo* H5OpenFile(const char* filename, const char* mode);
void H5CloseFile(H5FileInfo* h5file);
b32 H5FileIsReadable(H5FileInfo* h5file);
b32 H5FileIsWritable(H5FileInfo* h5file);
b32 H5DatasetExist(H5FileInfo* h5file, const char* dataset_name);
void H5GetDatasetSize(H5FileInfo* h5file, const char* dataset_name, index_type* size);
void H5ReadDatasetf64(H5FileInfo* h5file, const char* dataset_name, f64* data);
void H5ReadDatasetInd(H5FileInfo* h5file, const char* dataset_name, index_type* data);
void H5WriteDatasetf64(H5FileInfo* h5file, const char* dataset_name, index_type len, const f64* data);
void H5WriteDatasetInd(H5FileInfo* h5file, const char* dataset_name, index_type len, const index_type* data);
/* mesh writer in the schema tools/mesh_convert.py:116-126 produces; solution files of main.c:521-532,571-590 */
void DflMeshWriteH5(H5FileInfo* f, const char* group, index_type N, index_type T, const f64* xg, const index_type* ien,
                    index_type nb, const index_type* node_offset, const index_type* node, const index_type* elem_offset,
                    const index_type* bien, const index_type* f2e, const index_type* forn);
void DflSolutionWriteH5(const char* filename, index_type N, const f64* d_wgold, const f64* d_dwgold);
void DflSolutionReadH5(const char* filename, index_type N, f64* d_wgold, f64* d_dwgold);

/* ---- sparsity (csr.h) ------------------------------------------------------------ */
typedef index_type csr_index_type;
typedef struct CSRAttr CSRAttr;
struct CSRAttr {
    index_type num_row, num_col, nnz;
    index_type* row_ptr; /* device */
    index_type* col_ind; /* device */
    const CSRAttr* parent;
};
#define CSRAttrNumRow(attr) ((attr)->num_row)
#define CSRAttrNumCol(attr) ((attr)->num_col)
#define CSRAttrNNZ(attr) ((attr)->nnz)
#define CSRAttrRowPtr(attr) ((attr)->row_ptr)
#define CSRAttrColInd(attr) ((attr)->col_ind)
CSRAttr* CSRAttrCreate(const Mesh3D* mesh);
void CSRAttrDestroy(CSRAttr* attr);
CSRAttr* CSRAttrCreateBlock(const CSRAttr* attr, csr_index_type block_row, csr_index_type block_col);
/* csr.h:32-36.  The reference reads attr->row_ptr (a device pointer) on the host in the first two; here they copy the
 * 4 / 8 bytes they need back.  CSRAttrRow returns a DEVICE pointer into col_ind. */
index_type CSRAttrLength(CSRAttr* attr, csr_index_type row);
csr_index_type* CSRAttrRow(CSRAttr* attr, csr_index_type row);
void CSRAttrGetNonzeroIndBatched(const CSRAttr* attr, csr_index_type batch_size, const index_type* row, const index_type* col,
                                 index_type* ind);
/* csr_impl.h:6-9 */
void ExpandCSRByBlockSize(const CSRAttr* attr, CSRAttr* new_attr, csr_index_type block_size[2]);
void CSRAttrGetNZIndBatchedGPU(const CSRAttr* attr, csr_index_type batch_size, const index_type* row, const index_type* col,
                               csr_index_type* ind);

/* ---- matrices (matrix.h) --------------------------------------------------------- */
typedef enum MatType { MAT_TYPE_NONE = 0, MAT_TYPE_DENSE = 1, MAT_TYPE_CSR = 2, MAT_TYPE_FS = 4, MAT_TYPE_CUSTOM = 8 } MatType;
typedef struct Matrix Matrix;
typedef struct MatrixOp {
    void (*setup)(Matrix* matrix);
    void (*zero)(Matrix* matrix);
    void (*zero_row)(Matrix* matrix, index_type, const index_type* row, index_type shift, value_type diag);
    void (*amvpby)(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y);
    void (*amvpby_mask)(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y, value_type* left_mask,
                        value_type* right_mask);
    void (*matvec)(Matrix* matrix, value_type* x, value_type* y);
    void (*matvec_mask)(Matrix* matrix, value_type* x, value_type* y, value_type* left_mask, value_type* right_mask);
    void (*get_diag)(Matrix* matrix, value_type* diag, index_type bs);
    void (*set_values_coo)(Matrix* matrix, value_type alpha, index_type n, const index_type* row, const index_type* col,
                           const value_type* val, value_type beta);
    void (*set_values_ind)(Matrix* matrix, value_type alpha, index_type n, const index_type* ind, const value_type* val,
                           value_type beta);
    void (*add_elem_value_batched)(Matrix* matrix, index_type nshl, index_type batch_size, const index_type* batch_ptr,
                                   const index_type* ien, const value_type* val, const index_type* mask);
    void (*add_elem_value_blocked_batched)(Matrix* matrix, index_type nshl, index_type batch_size, const index_type* batch_ptr,
                                           const index_type* ien, index_type block_row_size, index_type block_col_size,
                                           const value_type* val, int lda, int stride, const index_type* mask);
    void (*add_value_batched)(Matrix* matrix, index_type batch_size, const index_type* batch_row_ind,
                              const index_type* batch_col_ind, const value_type* A);
    void (*add_value_blocked_batched)(Matrix* matrix, index_type batch_size, const index_type* batch_row_ind,
                                      const index_type* batch_col_ind, index_type block_row, index_type block_col,
                                      const value_type* A, int lda, int stride);
    void (*destroy)(Matrix* matrix);
} MatrixOp;
struct Matrix {
    index_type size[2];
    MatType type;
    void* data;
    hipStream_t stream_ref;
    MatrixOp op[1];
};
#define MatrixNumRow(A) ((A)->size[0])
#define MatrixNumCol(A) ((A)->size[1])
#define MatrixType(A) ((A)->type)
typedef struct MatrixFS MatrixFS;
typedef struct MatrixCSR {
    b32 external_attr;
    const CSRAttr* attr;
    value_type* val; /* reference-layout values; NULL while the matrix is a view into a block-mode MatrixFS */
    void* descr;     /* unused (cusparseSpMatDescr_t in the reference) */
    index_type buffer_size;
    void* buffer;
    /* ext */
    MatrixFS* owner;       /* block-mode parent, or NULL */
    index_type owner_slot; /* i * n_offset + j inside the parent */
} MatrixCSR;
struct MatrixFS {
    index_type n_offset;
    index_type* offset;
    index_type* d_offset;
    hipStream_t* stream;
    const CSRAttr* spy1x1;
    value_type** d_matval;
    Matrix** mat;
    /* ext */
    b32 block_mode;        /* the (u,p) 2x2 layout of src/main.c:374-391 was recognised */
    value_type* block_val; /* [nnz1][16] device, 4x4 blocks over spy1x1 */
    index_type owned_rows; /* node rows this rank owns (== spy1x1->num_row on one GPU): SpMV / PC run on these only */
    b32 reference_layout;  /* MatrixFSUseReferenceLayout: keep the four row-expanded sub-matrix arrays (no block mode) */
    b32 block_val_heap;    /* block_val is a plain hipMalloc block (moved out of the allocator's pool by the Krylov placement
                              calibration, host/solver.c) -- MatrixDestroy frees it accordingly */
    value_type* x4;        /* [num_row][4] interleaved copy of the matvec's input (dfl_bcsr_spmv_x4), allocated on first use */
    b32 x4_pool;           /* x4 came from the allocator's pool (DFL_X4_POOL=1) rather than from hipMalloc */
};
Matrix* MatrixCreateTypeCSR(const CSRAttr* attr, void*);
Matrix* MatrixCreateTypeFS(index_type n_offset, const index_type* offset, void*);
void MatrixDestroy(Matrix* matrix);
void MatrixSetup(Matrix* matrix);
void MatrixZero(Matrix* matrix);
void MatrixZeroRow(Matrix* matrix, index_type n, const index_type* row, index_type shift, value_type diag);
void MatrixAMVPBY(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y);
void MatrixAMVPBYWithMask(Matrix* A, value_type alpha, value_type* x, value_type beta, value_type* y, value_type* left_mask,
                          value_type* right_mask);
void MatrixMatVec(Matrix* matrix, value_type* x, value_type* y);
/* block-mode (u,p) MatrixFS: the interleaved scratch the matvec gathers from (allocated on first call), and y = A x with the
 * scratch already filled by the caller for all columns the rows [row0, row1) read (dfl_interleave4, or a producer that
 * writes both layouts, e.g. dfl_pc_jacobi_apply_scaled_rows_x4) */
value_type* DflMatrixFSInterleavedScratch(Matrix* matrix);
void DflMatrixFSMatVecX4Range(Matrix* matrix, const value_type* x4, value_type* y, index_type row0, index_type row1);
void MatrixMatVecWithMask(Matrix* matrix, value_type* x, value_type* y, value_type* left_mask, value_type* right_mask);
void MatrixGetDiag(Matrix* matrix, value_type* diag, index_type bs);
void MatrixSetValuesCOO(Matrix* matrix, value_type alpha, index_type n, const index_type* row, const index_type* col,
                        const value_type* val, value_type beta);
void MatrixSetValuesInd(Matrix* matrix, value_type alpha, index_type n, const index_type* ind, const value_type* val,
                        value_type beta);
void MatrixAddElemValueBatched(Matrix* matrix, index_type nshl, index_type num_batch, const index_type* batch_ptr,
                               const index_type* ien, const value_type* val, const index_type* mask);
/* the reference's live LHS scatter entry point (assemble.cu:253-271): one batch = one color; val holds one lda-strided
 * block per (batch slot, a, b), `stride` values apart.  On the (u,p) MatrixFS the rows / columns 0..3 of every block go
 * into the 4x4 block array (block mode) or through SetBlockValueToSubmatGPU into the four sub-matrices. */
void MatrixAddElemValueBlockedBatched(Matrix* matrix, index_type nshl, index_type num_batch, const index_type* batch_ptr,
                                      const index_type* ien, index_type block_row_size, index_type block_col_size,
                                      const value_type* val, int lda, int stride, const index_type* mask);
void MatrixAddValueBatched(Matrix* matrix, index_type batch_size, const index_type* batch_row_ind, const index_type* batch_col_ind,
                           const value_type* A);
void MatrixAddValueBlockedBatched(Matrix* matrix, index_type batch_size, const index_type* batch_row_ind,
                                  const index_type* batch_col_ind, index_type block_row_size, index_type block_col_size,
                                  const value_type* A, int lda, int stride);
/* matrix.h:141-147 */
MatrixCSR* MatrixCSRCreate(const CSRAttr* attr, void*);
void MatrixCSRDestroy(Matrix* matrix);
MatrixFS* MatrixFSCreate(index_type n_offset, const index_type* offset, void*);
void MatrixFSDestroy(Matrix* matrix);
/* call before MatrixSetup: keep the reference's storage (four row-expanded CSR value arrays, per-sub-matrix loops of
 * matrix.c:449-551) instead of the 4x4 block array -- for hosts that write into MatrixCSR.val themselves */
void MatrixFSUseReferenceLayout(Matrix* matrix, b32 on);
/* block-mode helpers (not in the reference) */
value_type* MatrixFSBlockValues(Matrix* matrix); /* NULL unless block mode */
/* fills the four sub-matrices' `val` arrays (reference layout) from the block storage, allocating them on first use */
void MatrixFSExportSubmatrices(Matrix* matrix);
void MatrixFSImportSubmatrices(Matrix* matrix);
/* element-partitioned runs: local nodes are numbered owned-first; rows >= n are ghost rows */
void MatrixFSSetOwnedRows(Matrix* matrix, index_type n);
/* y = A x on node rows [row0, row1) of the block-mode (u,p) system only */
void MatrixFSMatVecRange(Matrix* matrix, value_type* x, value_type* y, index_type row0, index_type row1);
index_type MatrixFSOwnedRows(Matrix* matrix);
/* recursive coordinate bisection of tet centroids into num_part parts (the reference's METIS
 * wrapper, src/partition.c:16-77, is dead code and METIS is unavailable): epart[T] on the host */
void DflPartitionRCB(index_type num_tet, const index_type* ien, const f64* xg, index_type num_part, index_type* epart);

/* ---- vectors (vec.h) --------------------------------------------------------------- */
void VecAXPY(value_type a, const value_type* x, value_type* y, index_type n);
void VecPointwiseMult(const value_type* x, const value_type* y, value_type* z, index_type n);
void VecPointwiseDiv(const value_type* x, const value_type* y, value_type* z, index_type n);
void VecPointwiseInv(value_type* x, index_type n);

/* ---- Dirichlet (dirichlet.h) --------------------------------------------------------- */
typedef enum BCType { BC_NONE = 0, BC_STRONG = 1, BC_WEAK = 2, BC_OUTFLOW = 4 } BCType;
typedef struct Dirichlet {
    const Mesh3D* mesh;
    index_type face_ind;
    index_type shape;
    size_t buffer_size;
    void* buffer;
    BCType bctype[];
} Dirichlet;
Dirichlet* DirichletCreate(const Mesh3D* mesh, index_type face_ind, index_type shape);
void DirichletDestroy(Dirichlet* dirichlet);
void DirichletApplyVec(Dirichlet* dirichlet, value_type* b);
void DirichletApplyMat(Dirichlet* dirichlet, Matrix* A);

/* ---- preconditioners (pc.h) ---------------------------------------------------------- */
typedef enum PCType { PC_NONE = 0x0, PC_JACOBI = 0x1, PC_DECOMPOSITION = 0x2, PC_AMGX = 0x3, PC_CUSTOM = 0x4, PC_ILU0 = 0x5,
                      PC_TWOLEVEL = 0x6 } PCType;
typedef struct PC PC;
typedef struct PCOps {
    void (*setup)(PC*);
    void (*destroy)(PC*);
    void (*apply)(PC*, value_type*, value_type*);
} PCOps;
struct PC {
    PCType type;
    void* mat;
    PCOps op[1];
    void* data;
    void* cublas_handle; /* kept for layout compatibility; unused */
};
typedef struct PCNone { index_type n; } PCNone;
typedef struct PCJacobi { index_type n; index_type bs; void* diag; } PCJacobi;
typedef struct PCDecomposition { index_type n_sec; index_type* offset; PC** pc; void* ext; } PCDecomposition;
PC* PCCreateNone(Matrix* mat, index_type n);
PC* PCCreateJacobi(Matrix* mat, index_type bs, void* handle);
PC* PCCreateDecomposition(Matrix* mat, index_type n, const index_type* offset, void* handle);
PC* PCCreateAMGX(Matrix* mat, void* options); /* returns NULL: NVIDIA-only external library (pc.c:300-304) */
/* PC_ILU0: multicolor block-DILU on the block-mode (u,p) matrix (host/pc_dilu.c, csrc/k_dilu.hip); build-defined:
 * the reference's PCType ends at PC_CUSTOM = 0x4 (pc.h:15-21); 0x5 and 0x6 are this build's additions.  KrylovSetPCType(ksp, PC_ILU0) makes KrylovSolve build it instead
 * of the reference's Jacobi tree (PC_DECOMPOSITION = the reference default). */
PC* PCCreateDILU(Matrix* mat);
void PCDILUSetActiveLength(PC* pc, index_type n_active);
/* sweeps read the off-diagonal blocks from a caller-kept single-precision copy (NULL = the matrix's own values); host/pc_dilu.c */
void PCDILUSetF32Values(PC* pc, const float* valf);
index_type PCDILUGetColors(PC* pc, u8* color_out);
const f64* PCDILUGetInverseBlocks(PC* pc);
/* PC_TWOLEVEL (host/pc_twolevel.c, csrc/k_amg.hip; build-defined, in the spirit of the AMGX aggregation configuration the
 * reference sketches at krylov.c:409-437): block-DILU smoothing plus a coarse-grid correction on node aggregates --
 * z = S r;  z += P Ac^-1 P^T (r - A z), P = piecewise constant over spatial aggregates of ~agg_size nodes (recursive
 * coordinate bisection of the mesh nodes), Ac = P^T A P (Galerkin, 4x4 blocks over the aggregate graph, rebuilt at every
 * PCSetup), Ac^-1 by a few inner Jacobi-GMRES iterations (rtol 0.1) -- hence the outer solver runs as FGMRES.  Iteration
 * counts become (nearly) independent of the mesh size: 50M tets converge in tens of iterations instead of ~600. */
PC* PCCreateTwoLevel(Matrix* mat, const Mesh3D* mesh, index_type agg_size);
/* the same on an element-partitioned matrix (owned rows first, MatrixFSSetOwnedRows): every rank aggregates the nodes it
 * owns, the Galerkin coarse matrix is assembled from the owned rows and REPLICATED (its values all-reduced at PCSetup, the
 * restricted residual all-reduced per application: one halo exchange + one all-reduce of 4 Nc doubles per apply), every rank
 * solves the same small coarse problem, the smoother is the rank-local DILU.  `comm` (copied) must carry rank / world;
 * NULL = PCCreateTwoLevel.  Collective: every rank of the partition calls it, and PCSetup / PCApply, together. */
struct DflComm;
PC* PCCreateTwoLevelDist(Matrix* mat, const Mesh3D* mesh, index_type agg_size, const struct DflComm* comm);
void PCTwoLevelSetActiveLength(PC* pc, index_type n_active);
void PCTwoLevelSetInner(PC* pc, index_type max_iter, f64 rtol); /* inner coarse solve: default 40 iterations, rtol 0.1 */
void PCTwoLevelInfo(PC* pc, index_type* num_aggregate, index_type* coarse_nnz, int64_t* inner_iterations);
const index_type* PCTwoLevelAggregates(PC* pc); /* device [N]: aggregate of every node (tests) */
Matrix* PCTwoLevelCoarseMatrix(PC* pc);          /* the Galerkin coarse matrix, a block-mode MatrixFS (tests) */
void PCSetup(PC* pc);
void PCDestroy(PC* pc);
void PCApply(PC* pc, f64* x, f64* y);

/* ---- Krylov (krylov.h) ---------------------------------------------------------------- */
typedef void (*KSPSolveFunc)(Matrix*, value_type*, value_type*, void*);
typedef struct Krylov {
    index_type max_iter;
    f64 atol, rtol;
    void* handle;
    KSPSolveFunc ksp_solve;
    size_t ksp_ctx_size;
    void* ksp_ctx;
    void* pc;
    void* ext; /* library-owned workspace + statistics (KrylovStats) */
} Krylov;
Krylov* KrylovCreateCG(index_type, f64, f64, void*);    /* stub in the reference (krylov.c:42-51); build-defined here */
Krylov* KrylovCreateGMRES(index_type, f64, f64, void*);
void KrylovDestroy(Krylov* krylov);
/* argument order of the DEFINITION and the call site (krylov.c:386, main.c:217): (ksp, A, x, b) -- Q10 */
void KrylovSolve(Krylov* krylov, Matrix* A, f64* x, f64* b);
typedef struct KrylovStats {
    index_type iterations;
    f64 rnrm_init;
    f64 res_hist[512]; /* |beta[k+1]| after iteration k */
    b32 converged;
    b32 fused_norm_cancelled; /* KrylovSetFusedNorm: an iteration kept < 1e-6 of w.w -- history unreliable */
    /* running totals since the solver was created (a Newton loop makes several solves per step) */
    index_type total_solves, total_converged;
    int64_t total_iterations;
} KrylovStats;
const KrylovStats* KrylovGetStats(const Krylov* krylov);
/* 0: GMRES tests convergence every 20 iterations like the reference (krylov.c:281-290); k>0: every k */
void KrylovSetCheckInterval(Krylov* krylov, index_type k);
void KrylovSetVerbose(Krylov* krylov, b32 verbose);
void KrylovSetPCType(Krylov* krylov, PCType type); /* PC_DECOMPOSITION (default, reference tree), PC_ILU0 or PC_TWOLEVEL */
PC* KrylovGetPC(const Krylov* krylov);
/* off by default: the norm of the orthogonalised vector from w.w - sum h_j^2, so that a partitioned Arnoldi step needs ONE
 * all-reduce (h and w.w together) instead of two and, with the Jacobi tree on <= 500k rows, ONE launch for update + Givens
 * step + next preconditioner application (7 -> 4 launches per step; since round 3 also without a communicator -- the
 * last-level solver of PC_TWOLEVEL uses it); rounding differs from the explicit norm and heavy cancellation raises
 * KrylovStats.fused_norm_cancelled */
void KrylovSetFusedNorm(Krylov* krylov, b32 on);
/* p(1)-pipelined GMRES (host/solver.c, gmres_pipelined; build-defined, off by default): ONE reduction per Arnoldi step -- the CGS
 * coefficients and w.w together, the norm from the Pythagorean identity -- overlapped with the matvec of the NEXT step through
 * the auxiliary basis z_{j+1} = A M^-1 v_j (Ghysels et al. 2013).  Hides the all-reduce latency of partitioned runs behind the
 * matvec (own stream with the RCCL communicator); costs a second basis, a third basis pass per step and some accuracy of the
 * residual history (~1e-8 r0 instead of 1e-10; heavy cancellation raises KrylovStats.fused_norm_cancelled).  Ignored with
 * FGMRES / PC_TWOLEVEL and with restarts. */
void KrylovSetPipelined(Krylov* krylov, b32 on);
/* A non-blocking stream (default priority) that was PROBED to run concurrently with `main_stream`: HIP maps streams onto a few
 * hardware queues round-robin, and a side stream on the main stream's own queue runs behind it instead of beside it; a
 * highest-priority stream avoids that but was seen to delay every kernel of the normal-priority stream (host/comm_rccl.c).
 * What the RCCL communicator uses for its halo exchange; a host that fills DflComm.halo_stream itself should create its stream
 * with this.  The caller destroys it with hipStreamDestroy. */
hipStream_t DflPickConcurrentStream(hipStream_t main_stream);
/* GMRES(m): restart after m basis columns (x updated, true residual recomputed); m <= 0 or m >= max_iter (default) = the
 * reference's full GMRES.  Keeps the basis at m+1 vectors for long solves (config 5: 50M tets, PC_ILU0). */
void KrylovSetRestart(Krylov* krylov, index_type m);
/* FGMRES: keep Z[:,k] = M_k^-1 Q[:,k] (doubles the basis memory) so that the preconditioner may change between iterations;
 * switched on automatically by PC_TWOLEVEL */
void KrylovSetFlexible(Krylov* krylov, b32 on);
/* node coordinates for preconditioners that aggregate nodes (PC_TWOLEVEL); SolveFlowSystem passes its mesh itself */
void KrylovSetMesh(Krylov* krylov, const Mesh3D* mesh);
void KrylovSetAggregateSize(Krylov* krylov, index_type nodes_per_aggregate); /* PC_TWOLEVEL, default 64 */
/* optional communicator for element-partitioned runs (one process per GPU); NULL = single GPU */
typedef struct DflComm {
    void (*allreduce_sum)(void* ctx, f64* d_buf, index_type n); /* in place, device buffer */
    void (*halo_exchange)(void* ctx, f64* d_x);                /* fill ghost entries of a [u|p|..] vector */
    void* ctx;
    index_type num_owned_node; /* dots / norms run over owned nodes only */
    /* optional split exchange: halo_begin starts filling the ghost entries of d_x asynchronously (the caller has
     * finished writing d_x on the library stream), halo_end makes the library stream wait for it.  With
     * num_interior_node > 0 (owned nodes [0, num_interior_node) have no ghost neighbour) the Krylov matvec runs the
     * interior rows between the two.  NULL / 0: halo_exchange before the whole matvec. */
    void (*halo_begin)(void* ctx, f64* d_x);
    void (*halo_end)(void* ctx, f64* d_x);
    index_type num_interior_node;
    /* this process's place in the partition (0 / 0 when the implementer does not say: preconditioners that need a global
     * numbering -- PC_TWOLEVEL on a partitioned matrix -- then refuse and KrylovSolve falls back to PC_ILU0) */
    int rank, world;
    /* optional: the stream the exchange started by halo_begin runs on (NULL / returns NULL: synchronous).  The Krylov matvec
     * enqueues the boundary rows there, behind the unpack, so that they overlap the interior rows on the library stream and
     * the library stream waits once (in halo_end) for both */
    hipStream_t (*halo_stream)(void* ctx);
} DflComm;
void KrylovSetComm(Krylov* krylov, const DflComm* comm);
const DflComm* KrylovGetComm(const Krylov* krylov); /* NULL on a single GPU */

/* RCCL implementation of DflComm (host/comm_rccl.c): collectives enqueued from C on the library stream.
 * Bootstrap: every rank DflRcclLoad(path to librccl.so, NULL/"" = "librccl.so.1"); rank 0 DflRcclGetUniqueId and
 * broadcasts the DflRcclUniqueIdBytes() bytes (MPI, torch.distributed, a file ...); all ranks DflRcclCommCreate
 * (collective), optionally DflRcclCommCreateHaloComm with a second broadcast id, DflRcclCommSetHalo with their halo plan, KrylovSetComm(ksp, DflRcclCommVtable(c)). */
typedef struct DflRcclComm DflRcclComm;
int DflRcclLoad(const char* path);
int DflRcclUniqueIdBytes(void);
int DflRcclGetUniqueId(char* out_bytes);
DflRcclComm* DflRcclCommCreate(const char* id_bytes, int rank, int world);
/* optional own communicator for the halo stream; 0 = created.  If it fails on ANY rank, every rank must call
 * DflRcclCommDropHaloComm (agree on the result first): mixed use pairs sends and receives of different communicators */
int DflRcclCommCreateHaloComm(DflRcclComm* c, const char* id_bytes);
void DflRcclCommDropHaloComm(DflRcclComm* c);
void DflRcclCommSetInterior(DflRcclComm* c, index_type n_interior); /* owned nodes [0, n_interior) touch no ghost */
void DflRcclCommSetHalo(DflRcclComm* c, index_type n_local, index_type n_owned, const index_type* send_count,
                        const index_type* send_idx, const index_type* recv_count, const index_type* recv_idx);
const DflComm* DflRcclCommVtable(const DflRcclComm* c);
void DflRcclCommCounters(const DflRcclComm* c, int64_t* n_allreduce, int64_t* n_halo);
void DflRcclCommDestroy(DflRcclComm* c);

/* ---- assembly (assemble.h) ------------------------------------------------------------ */
void AssembleSystemTet(Mesh3D* mesh, f64* wgalpha_dptr, f64* dwgalpha_dptr, f64* F, Matrix* J);
void AssembleSystemTetFace(Mesh3D* mesh, f64* wgalpha_dptr, f64* dwgalpha_dptr, f64* F, Matrix* J);
/* the caller of the hot path, src/main.c:31-75 (static there) */
void AssembleSystem(Mesh3D* mesh, f64* wgalpha, f64* dwgalpha, f64* F, Matrix* J, Dirichlet** bcs, index_type nbc);
/* generalized-alpha Newton solve of one time level (src/main.c:77-283, static there; `maxit` <= 0 -> 4).
 * Returns the Newton iteration count; rnorm_out / rnorm_init_out (4 each: u, p, phi, T) may be NULL. */
index_type SolveFlowSystem(Mesh3D* mesh, f64* wgold, f64* dwgold, f64* dwg, Matrix* J, f64* F, f64* dx, Krylov* ksp,
                           Dirichlet** bcs, index_type nbc, index_type maxit, f64* rnorm_out, f64* rnorm_init_out);
struct ParticleContext;
/* one pass of the time loop body (src/main.c:537-565): predictor, SolveFlowSystem, corrector; with a
 * particle context also `dem_substeps` contact sweeps + particle updates (the calls the reference has
 * commented out at main.c:547-569) */
index_type DflTimeStep(Mesh3D* mesh, f64* wgold, f64* dwgold, f64* dwg, Matrix* J, f64* F, f64* dx, Krylov* ksp, Dirichlet** bcs,
                       index_type nbc, index_type newton_maxit, struct ParticleContext* pctx, index_type dem_substeps,
                       f64* rnorm_out, f64* rnorm_init_out);
/* the assembly caches J^-1-derived element geometry per mesh (the reference recomputes it every call); after writing new
 * node coordinates into Mesh3DDevice(mesh)->xg call this once so the next assembly rebuilds the cache */
void DflMeshGeometryChanged(Mesh3D* mesh);
/* device memory pool of the default DEVICE allocator (host/runtime.c; DFL_DEVICE_POOL_GB=0 disables it) */
void DflDevicePoolStats(int64_t* reserved_bytes, int64_t* in_use_bytes);
/* The driver wipes freed device memory in the background (~36 GB/s) and streaming kernels run up to 8 % slower meanwhile;
 * hipMemGetInfo counts that memory as free at once.  DflDeviceMemoryInUse: the driver's own "VRAM in use" figure for the
 * current device in bytes (it includes memory still to be wiped), -1 when rocm_smi is unavailable.  DflWaitDeviceMemoryQuiet:
 * blocks while a wipe is in progress (that figure falling, or the SOC clock at its high level), at most max_seconds; returns
 * the seconds waited, 0 when there was nothing to wait for, -1 when unknown.  Init() and the Krylov work-space calibration
 * call it; a host program that frees tens of GB right before a timed region may want to as well (host/runtime.c). */
int64_t DflDeviceMemoryInUse(void);
/* text log of this process's most recent Krylov work-space calibration (host/solver.c: what every candidate placement
 * measured, what was chosen, how long the waits were); "" when none has run.  DFL_WS_VERBOSE=1 prints the same to stderr. */
const char* DflKrylovCalibrationLog(void);
/* Placement of the Krylov work space (host/ws_placement.c).  The first KrylovSolve that allocates a GMRES basis for a large
 * block-mode system times at most four candidate blocks and keeps the fastest: bounded (<= 16 GB or a quarter of the free
 * memory extra, < 1 s), nothing the caller can see moves (DFL_WS_CANDIDATES=1 switches even that off).  This call is the
 * explicit, heavy form: builds the solver's preconditioner and work space for A now and draws many more placements -- basis
 * blocks and heap copies of the block value array behind spacers of up to 5/8 of the free memory, waiting for the driver's
 * memory wipe around the timings (5-15 s) -- holding at most max_extra_bytes of transient device memory (<= 0: no cap).  It
 * MAY move the block value array: ask MatrixFSBlockValues(A) again afterwards (DFL_VAL_RELOCATE=0 forbids the move). */
void DflKrylovCalibratePlacement(Krylov* krylov, Matrix* A, int64_t max_extra_bytes);
double DflWaitDeviceMemoryQuiet(double max_seconds);
/* boundary group whose faces get the weak-BC terms of AssembleSystemTetFace (default 4 = the reference's hard-coded group,
 * assemble.cu:1826-1828); lists are rebuilt when the group changes */
void DflSetWeakBCGroup(index_type group);
/* DflSetWeakBCGroup, DflSetAssemblySchedule and the Dfl*Parameters setters below change the PROCESS DEFAULTS; every mesh
 * copies them at Mesh3DCreate and keeps its own configuration from then on (two meshes with different schedules or face
 * groups coexist).  The two per-mesh forms (the schedule before Mesh3DGenerateColorBatch): */
void DflMeshSetAssemblySchedule(Mesh3D* mesh, int mode);
void DflMeshSetWeakBCGroup(Mesh3D* mesh, index_type group);
void DflSetQuiet(b32 quiet); /* suppress the reference's stdout chatter ("Assemble: F J", timers) */
/* which conflict-free launches the assembly kernels execute (set BEFORE Mesh3DGenerateColorBatch):
 *   0  the reference's JPL color batches, one launch per color (reference summation order)
 *   1  compact balanced re-coloring, ~4x fewer / larger launches; mesh->color,
 *      batch_offset and batch_ind are the reference's JPL result in both modes */
void DflSetAssemblySchedule(int mode);
/*   2  as 1 for the RHS; the Jacobian is assembled patch-wise: one workgroup sums all blocks of a spatial
 *      patch of tets in LDS and read-modify-writes each distinct block once per patch (host/patch.c) --
 *      ~3x less HBM traffic than one RMW per tet; LDS atomics => values reproducible to rounding, not bitwise */
void DflSetPatchParameters(index_type leaf, index_type slot_cap);
/*   3  (default) as 1 for the RHS; the Jacobian is assembled by row-owner node patches: one launch, every workgroup sums the
 *      block rows of its nodes in LDS and writes them once (host/rowpatch.c); AssembleSystem skips the zero pass.
 *      Reproducible to rounding, not bitwise (LDS atomics). */
/* schedule 3 (row-owner node patches): nodes per patch and cap on their summed nodal row lengths
 * (one 128-byte LDS line per nodal nonzero; 255 -> 32 KB per workgroup) */
void DflSetRowPatchParameters(index_type leaf_nodes, index_type slot_cap);
/*   4  slot-owner node patches for J (host/slotpatch.c): as 3, but every nodal nonzero is summed in registers by its owner
 *      lanes and written once -- no LDS atomics, fixed summation order (bitwise reproducible), no geometry cache */
void DflSetSlotPatchParameters(index_type leaf_nodes, index_type slot_cap, index_type tet_cap);
/* schedule 4 assembles the residual with one wave per patch of <= tets tets / <= nodes nodes: (16,32), (32,48) or (64,64) */
void DflSetRhsWaveParameters(index_type tets, index_type nodes);
/* schedules 2 and 3 assemble the residual by spatial tet patches (<= 64 tets, <= node_cap <= 96 distinct nodes each) */
void DflSetRhsPatchParameters(index_type leaf_tets, index_type node_cap);

/* ---- arrays / particles (Array.h, Particle.h) ------------------------------------------ */
typedef struct Array {
    b32 is_host;
    index_type len;
    f64* data;
} Array;
#define ArrayLen(a) ((a)->len)
#define ArrayData(a) ((a)->data)
Array* ArrayCreateHost(index_type len);
Array* ArrayCreateDevice(index_type len);
void ArrayDestroy(Array* a);
void ArrayCopy(Array* dst, const Array* src, MemCopyKind kind);
void ArrayLoad(Array* a, H5FileInfo* h5f, const char* dataset_name);       /* Array.c:242-253; in libdedflow_h5.so */
void ArraySave(const Array* a, H5FileInfo* h5f, const char* dataset_name); /* Array.c:255-261 */
/* BLAS-1 wrappers of Array.h:24-36 (Array.c:83-238).  A device array runs the library's own kernels on the library stream and
 * returns when the result is on the host / the array is updated, like the reference's cuBLAS calls with a host result pointer;
 * a host array is walked on the host in index order, like the reference.  Both operands must live on the same side.
 * ArrayZero is ArrayScale(a, 0.0) as in the reference (Array.c:103-105: a NaN stays a NaN).  SetAt with a repeated index keeps
 * the LAST value, as the reference's sequential loop does.  The reference declares ArrayGetAt and defines ArrayAt: both names. */
void ArraySet(Array* a, f64 val);
void ArrayZero(Array* a);
void ArraySetAt(Array* a, index_type n, const index_type* idx, const f64* val);
void ArrayGetAt(const Array* a, index_type n, const index_type* idx, f64* val);
void ArrayAt(const Array* a, index_type n, const index_type* idx, f64* val);
void ArrayScale(Array* a, f64 val);
void ArrayDot(f64* result, const Array* a, const Array* b);
void ArrayNorm2(f64* result, const Array* a);
void ArrayAXPY(Array* y, f64 a, const Array* x);
void ArrayAXPBY(Array* y, f64 a, const Array* x, f64 b); /* y = a x + b y */

/* ---- nodal fields (Field.h:13-33, Field.c:15-77): a host and a device Array of num_node * num_nodal_dof values ------- */
typedef struct Field {
    index_type shape[2];
    Array* host;
    Array* device;
} Field;
#define FieldHost(f) ((f)->host)
#define FieldDevice(f) ((f)->device)
Field* FieldCreate3D(const Mesh3D* mesh, i32 num_nodal_dof);
void FieldDestroy(Field* f);
void FieldInit(Field* f, void (*func)(f64*, void* ctx), void* ctx); /* func fills the host array; then H2D */
void FieldLoad(Field* f, H5FileInfo* h5f, const char* group_name);       /* Field.c:47-51; in libdedflow_h5.so */
void FieldSave(const Field* f, H5FileInfo* h5f, const char* group_name); /* Field.c:53-57 (saves the HOST copy) */
void FieldCopy(Field* dst, const Field* src);
void FieldUpdateHost(Field* f);
void FieldUpdateDevice(Field* f);
typedef struct ParticleContext {
    index_type num_particle;
    i32 num_pointwise_dof;
    Array* h_arr[3];
    Array* d_arr[3];
    f64 buff[2];
    void* ext; /* cell list workspace of the DEM sweep */
} ParticleContext;
#define ParticleCTXNumParticle(pctx) ((pctx)->num_particle)
#define ParticleCTXHostCoord(pctx) ((pctx)->h_arr[0])
#define ParticleCTXHostVel(pctx) ((pctx)->h_arr[1])
#define ParticleCTXHostAcc(pctx) ((pctx)->h_arr[2])
#define ParticleCTXDeviceCoord(pctx) ((pctx)->d_arr[0])
#define ParticleCTXDeviceVel(pctx) ((pctx)->d_arr[1])
#define ParticleCTXDeviceAcc(pctx) ((pctx)->d_arr[2])
#define ParticleMass(pctx) (((pctx)->buff)[0])
#define ParticleRadius(pctx) (((pctx)->buff)[1])
ParticleContext* ParticleContextCreate(index_type num_particle);
void ParticleContextDestroy(ParticleContext* ctx);
void ParticleContextCopy(ParticleContext* dst, const ParticleContext* src);
/* <group_name>/{coord,vel,acc} (Particle.c:66-103); in libdedflow_h5.so */
void ParticleContextLoad(ParticleContext* ctx, H5FileInfo* h5f, const char* group_name);
void ParticleContextSave(const ParticleContext* ctx, H5FileInfo* h5f, const char* group_name);
void ParticleContextUpdateHost(ParticleContext* ctx);
void ParticleContextUpdateDevice(ParticleContext* ctx);
void ParticleContextAdd(ParticleContext* ctx);
/* empty in the reference (Particle.c:120-130); here: contact-force sweep + explicit update (build-defined) */
void ParticleContextUpdate(ParticleContext* ctx);
void ParticleContextRemove(ParticleContext* ctx);
void ParticleContextSetContactModel(ParticleContext* ctx, f64 kn, f64 gamma_n, f64 dt);
void ParticleContextComputeForces(ParticleContext* ctx); /* acc <- contact forces / mass */

#ifdef __cplusplus
}
#endif
#endif /* DEDFLOW_H */
