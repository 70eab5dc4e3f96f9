/* Library-private declarations shared by the host C files. */
#ifndef DFL_HOST_PRIVATE_H
#define DFL_HOST_PRIVATE_H
#include "dedflow.h"

/* Patch schedule of the LHS assembly (host/patch.c): spatial patches of <= leaf tets whose (row,col)
 * blocks fit an LDS table; patches of one color share no node. */
typedef struct PatchSched {
    const CSRAttr* attr;       /* pattern the slot maps were built for */
    index_type num_patch, num_color, max_slots;
    index_type* color_offset;  /* host [num_color+1], patches sorted by color */
    index_type* d_eoff;        /* device [num_patch+1] element offsets (patch order) */
    index_type* d_boff;        /* device [num_patch+1] block-slot offsets */
    index_type* d_ien;         /* device [T][4] connectivity in patch order */
    uint16_t* d_lslot;         /* device [T][16] LDS slot of each (elem,a,b) block */
    index_type* d_blk_nz;      /* device [sum slots] nodal nonzero of each slot */
    f64* d_egeo;               /* device [T][16] element geometry cache in patch order */
    int64_t total_slots;
} PatchSched;
PatchSched* DflBuildPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap);
void DflFreePatchSchedule(PatchSched* ps);

/* RHS patch schedule (host/patch.c): spatial patches of <= 64 tets / <= node_cap nodes; partial residual
 * records per (patch, node) + a node -> partials list for the ordered second pass. */
typedef struct RhsPatchSched {
    index_type num_patch, total_nodes;
    index_type* d_eoff;      /* device [P+1] tet offsets (patch order) */
    index_type* d_noff;      /* device [P+1] patch-node offsets = partial record offsets */
    index_type* d_pnode;     /* device [total_nodes] global node of each patch node */
    u8* d_lien;              /* device [T][4] local (patch) node index of each tet vertex */
    uint16_t* d_adj;         /* device [4T] per patch: (local tet)*4 + a grouped by patch node, ascending tet */
    uint16_t* d_adj_start;   /* device [total_nodes + P] per patch nn+1 group starts */
    index_type* d_goff;      /* device [N+1] node -> range of gidx */
    index_type* d_gidx;      /* device [total_nodes] partial record ids of each node, ascending patch */
    f64* d_partial;          /* device [total_nodes][6] */
    index_type pad_tets, pad_nodes; /* > 0: fixed-stride layout (patch p at tet slot p*pad_tets, node slot p*pad_nodes) */
    index_type* d_cnt;       /* device [P] num_tets | num_nodes << 16 */
    /* 64-tet padded layout only (lane-per-tet kernel): the adjacency of a patch node cut into sub-lists of <= 4 entries */
    uint16_t* d_sub4;        /* device [P][128][4] result slots (local tet * 4 + a) of each sub-list, 256 = the zero slot */
    uint16_t* d_sub_start;   /* device [P][pad_nodes + 1] first sub-list of each patch node; entry nn = number of sub-lists */
} RhsPatchSched;
RhsPatchSched* DflBuildRhsPatchSchedule(Mesh3D* mesh, index_type leaf, index_type node_cap, index_type pad_tets, index_type pad_nodes);
void DflFreeRhsPatchSchedule(RhsPatchSched* ps);

/* Row-owner patch schedule (host/rowpatch.c): spatial patches of nodes, each owning its CSR rows. */
typedef struct RowPatchSched {
    const CSRAttr* attr;
    index_type num_patch, max_slots;
    index_type* d_ioff;      /* device [num_patch+1] item offsets */
    index_type* d_soff;      /* device [num_patch+1] slot offsets */
    index_type* d_item_ea;   /* device [4T] (schedule position of the tet)*4 + a */
    uint16_t* d_item_slot;   /* device [4T][4] LDS slot of block (a, b) */
    index_type* d_slot_nz;   /* device [nnz1] nodal nonzero of each slot */
} RowPatchSched;
RowPatchSched* DflBuildRowPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap);
void DflFreeRowPatchSchedule(RowPatchSched* ps);

/* Slot-owner patch schedule (host/slotpatch.c): node patches; every nodal nonzero is summed by one lane pair. */
typedef struct SlotPatchSched {
    const CSRAttr* attr;
    index_type num_patch, max_tets, max_slots, max_contrib;
    int64_t total_tets;
    int32_t* d_hdr;          /* device [num_patch][8]: tet_off, num_tet | num_node << 16, pos_off, num_pos, group_off, trips_lo, trips_hi, node_off */
    uint32_t* d_ptet_lid;    /* device [total_tets]: the four patch-local node ids (a byte each) of every (patch, tet) pair */
    index_type* d_pnode;     /* device [total_nodes]: global ids of every patch's distinct nodes, ascending inside a patch */
    index_type max_nodes;
    int64_t total_nodes;
    index_type* d_slot_nz;   /* device [positions] nodal nonzero of each slot position (+ split flags) */
    uint32_t* d_ldesc;       /* device lane-major descriptor groups: [group][64 lanes] x 2 x ((local tet << 4) | (a << 2) | b) */
} SlotPatchSched;
SlotPatchSched* DflBuildSlotPatchSchedule(Mesh3D* mesh, const CSRAttr* spy, index_type leaf, index_type slot_cap, index_type tet_cap);
void DflFreeSlotPatchSchedule(SlotPatchSched* ps);

/* Assembly configuration of ONE mesh: a copy of the process defaults (the DflSet* setters of include/dedflow.h) taken
 * at Mesh3DCreate, so that two meshes with different schedules / face groups / patch shapes coexist in one process. */
typedef struct AsmConfig {
    int sched_mode;              /* 0 reference colors, 1 compact colors, 2 tet patches, 3 row-owner patches, 4 slot-owner (default) */
    index_type face_group;       /* boundary group of the weak-BC faces (4 in the reference, assemble.cu:1826-1828) */
    index_type patch_leaf, patch_cap;           /* schedule 2: tets per patch, LDS block slots */
    index_type rowpatch_leaf, rowpatch_cap;     /* schedule 3: nodes per patch, LDS block slots */
    index_type slot_leaf, slot_cap, slot_tets;  /* schedule 4: nodes per patch, nodal nonzeros, tets touching the patch */
    index_type rhspatch_leaf, rhspatch_nodes;   /* schedules 2, 3: residual patches */
    index_type rhswave_tets, rhswave_nodes;     /* schedule 4: one wave per residual patch */
} AsmConfig;
const AsmConfig* DflAsmDefaults(void);
struct FlowWork;
void DflFreeFlowWork(struct FlowWork* fw);

typedef struct MeshExt {
    AsmConfig cfg;                 /* assembly configuration of this mesh */
    struct FlowWork* flow;         /* alpha-level state vectors + norm scratch of SolveFlowSystem (host/driver.c) */
    b32 nodep_current;             /* the packed node records already hold the states the next AssembleSystem is given */
    index_type* ien_b;             /* device [T][4], elements in execution-schedule order */
    index_type sched_num;          /* number of conflict-free launches of the execution schedule */
    index_type* sched_offset;      /* host [sched_num+1] */
    index_type* nzmap_b;           /* device [T][16], (elem,a,b) -> nodal nonzero, batch order */
    const CSRAttr* nzmap_attr;     /* pattern the map was built for */
    index_type face_group;         /* boundary group the face lists below belong to (-1: none) */
    index_type face_nf;            /* faces of that group */
    index_type face_nn;            /* nodes touched by their parent tets */
    index_type *face_node, *face_node_off, *face_node_ent; /* device: node, CSR offsets, entries f*4+a (ascending f) */
    const CSRAttr* face_attr;      /* pattern the nonzero lists below were built for */
    index_type face_nnz;           /* nodal nonzeros touched by the faces' 4x4 node blocks */
    index_type *face_nz, *face_nz_off, *face_nz_ent;       /* device: nonzero, CSR offsets, entries f*16+a*4+b */
    f64 *face_pF, *face_pJ;        /* device parking buffers [nf][16] and [nf][256] */
    index_type* h_f2e;             /* host copy of bound_f2e */
    index_type* h_sched_elem;      /* host [T]: element id at each position of the execution schedule */
    f64* egeo_b;                   /* device [T][16] element geometry cache in schedule order (LHS kernel) */
    f64* nodep;                    /* device [N][16] packed gather records (x,u,phi,T,du,p,dphi,dT) */
    f64* nodexu;                   /* device [N][8] compact (x,u) records: all the slot-owner Jacobian kernel reads of a node */
    f64* Fp;                       /* device [N][8] packed residual accumulator, zero between calls */
    RhsPatchSched* rhspatch;       /* RHS patch schedule (modes 2, 3), built on first use */
    RowPatchSched* rowpatch;       /* LHS row-owner patch schedule (mode 3), built on first use */
    PatchSched* patch;             /* LHS patch schedule (assembly schedule mode 2), built on first use */
    SlotPatchSched* slotpatch;     /* LHS slot-owner schedule (mode 4, default), built on first use */
} MeshExt;

void DflMeshPrepareFaces(Mesh3D* mesh, index_type group);
void DflMeshPrepareFaceNonzeros(Mesh3D* mesh, index_type group, const CSRAttr* spy);
void DflMeshFreeFaceLists(struct MeshExt* x);
/* AssembleSystemTet with J = beta_J * J + contributions (beta_J = 0 only takes effect in schedules 3 and 4) */
void DflAssembleSystemTetBeta(Mesh3D* mesh, f64* wgalpha, f64* dwgalpha, f64* F, Matrix* J, f64 beta_J);
b32 DflQuiet(void);

/* profiling tags (runtime.c) */
enum { DFL_TAG_SPMV = 0, DFL_TAG_CGS_DOTS = 1, DFL_TAG_CGS_UPDATE = 2, DFL_TAG_PC = 3, DFL_TAG_ASM_LHS = 4,
       DFL_TAG_ASM_RHS = 5, DFL_TAG_FACE = 6, DFL_TAG_DIRICHLET = 7, DFL_TAG_SMALL = 8 };
void DflProfileEnable(int on);
int DflProfileBegin(int tag);
void DflProfileEnd(int slot);
int DflProfileCollect(int tag, double* total_ms, double* min_ms);
int DflProfileDurations(int tag, double* out_ms, int max_out);
#define DFL_TIMED(tag, call) do { int _s = DflProfileBegin(tag); call; DflProfileEnd(_s); } while (0)

/* AssembleSystem for the Newton driver: `prepacked` = the packed node records were just written from these very states
 * by the fused alpha-state kernel (host/driver.c), so the pack launch is skipped */
void DflAssembleSystemPrepacked(Mesh3D* mesh, f64* wgalpha, f64* dwgalpha, f64* F, Matrix* J, Dirichlet** bcs, index_type nbc,
                                b32 prepacked);
f64* DflMeshNodeRecords(Mesh3D* mesh); /* [N][16] packed gather records, allocated on first use */
void DflKrylovWorkspaceInPool(int on);
void* DflVectorArenaAlloc(size_t bytes);
int DflVectorArenaFree(void* p);
/* driver.c */
index_type SolveFlowSystem(Mesh3D* mesh, f64* wgold, f64* dwgold, f64* dwg, Matrix* J, f64* F, f64* dx, Krylov* ksp,
                           Dirichlet** bcs, index_type nbc, index_type maxit, f64* rnorm_out, f64* rnorm_init_out);
void DflKrylovSolvePrepared(Krylov* ksp, Matrix* A, f64* x, f64* b); /* KrylovSolve without PC (re)build and PCSetup */

/* named ranges for rocprofv3 --marker-trace (DFL_ROCTX=1); no-ops otherwise */
void DflRangePush(const char* name);
void DflRangePop(void);

int DflDevicePoolEnabled(void); /* the default DEVICE allocator carves large requests out of its pool */
void* DflDevicePoolAllocNoGrow(size_t bytes); /* from the chunks already reserved, or NULL */

void DflMatrixFSRelocateBlockValues(Matrix* m, value_type* new_val); /* host/matrix.c */
value_type* DflMatrixFSScratchBlockBegin(Matrix* m); /* reference-layout (u,p) FS matrix -> scratch block array (host/matrix.c) */
void DflMatrixFSScratchBlockEnd(Matrix* m);
/* host/slotpatch.c */
int DflSlotPatchLimitCheck(int64_t num_positions, int64_t num_tets, int64_t num_nodes, int64_t max_contributions_of_a_position, char* why,
                           size_t why_len);
void DflSlotPatchSetTestLimits(int positions, int tets);

#endif
