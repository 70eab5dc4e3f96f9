/* AssembleSystemTet / AssembleSystemTetFace (src/assemble.h:13-14) and their caller
 * AssembleSystem (src/main.c:31-75): the color-batch loop of src/assemble.cu:1559-1738
 * with one fused launch per color, no per-call device allocation (the reference
 * mallocs+memsets+frees nine work buffers per call, :1533-1553,1746-1760). */
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

#define BS (6)
static b32 g_quiet = FALSE;
/* process defaults of the per-mesh assembly configuration (copied into every mesh at Mesh3DCreate):
 *   schedule 2  tets per patch before the LDS-slot cap applies, and that cap (16 * (cap|1) * 8 B of dynamic LDS per
 *               workgroup; 448 -> 57.5 KB -> two workgroups per CU)
 *   schedule 3  nodes per patch and the cap on their summed row lengths (LDS slots)
 *   schedule 4  nodes per patch, nodal nonzeros and tets per patch (one slot offset / one tet per lane of a 256-thread
 *               workgroup: <= 255 / <= 256); residual: one wave per patch of (tets, nodes) = (16,32), (32,48) or (64,64)
 *   face group  boundary group whose faces carry the weak-BC terms (the reference hard-codes group 4) */
static AsmConfig g_asm = {4, 4, 96, 448, 16, 255, 7, 200, 128, 64, 64, 64, 64};
const AsmConfig* DflAsmDefaults(void) { return &g_asm; }
void DflSetAssemblySchedule(int mode) { g_asm.sched_mode = mode; }
void DflSetPatchParameters(index_type leaf, index_type slot_cap) {
    if (leaf > 0) g_asm.patch_leaf = leaf;
    if (slot_cap > 0 && slot_cap <= 511) g_asm.patch_cap = slot_cap;
}
void DflSetRowPatchParameters(index_type leaf_nodes, index_type slot_cap) {
    if (leaf_nodes > 0) g_asm.rowpatch_leaf = leaf_nodes;
    if (slot_cap > 0 && slot_cap <= 1023) g_asm.rowpatch_cap = slot_cap;
}
void DflSetSlotPatchParameters(index_type leaf_nodes, index_type slot_cap, index_type tet_cap) {
    if (leaf_nodes > 0) g_asm.slot_leaf = leaf_nodes;
    if (slot_cap > 0) g_asm.slot_cap = slot_cap > DFL_SLOT_BLOCK - 1 ? DFL_SLOT_BLOCK - 1 : slot_cap;
    if (tet_cap > 0) g_asm.slot_tets = tet_cap > DFL_SLOT_BLOCK ? DFL_SLOT_BLOCK : tet_cap;
}
void DflSetRhsPatchParameters(index_type leaf_tets, index_type node_cap) {
    if (leaf_tets > 0 && leaf_tets <= dfl_rhs_patch_max_tets()) g_asm.rhspatch_leaf = leaf_tets;
    if (node_cap >= 4 && node_cap <= dfl_rhs_patch_max_nodes()) g_asm.rhspatch_nodes = node_cap;
}
void DflSetRhsWaveParameters(index_type tets, index_type nodes) {
    if ((tets == 16 && nodes == 32) || (tets == 32 && nodes == 48) || (tets == 64 && nodes == 64)) {
        g_asm.rhswave_tets = tets;
        g_asm.rhswave_nodes = nodes;
    }
}
void DflSetWeakBCGroup(index_type group) { g_asm.face_group = group; }
/* the same two switches for one existing mesh (the schedule before Mesh3DGenerateColorBatch) */
void DflMeshSetAssemblySchedule(Mesh3D* mesh, int mode) { ((MeshExt*)mesh->ext)->cfg.sched_mode = mode; }
void DflMeshSetWeakBCGroup(Mesh3D* mesh, index_type group) { ((MeshExt*)mesh->ext)->cfg.face_group = group; }
void DflSetQuiet(b32 quiet) { g_quiet = quiet; }
/* node coordinates were modified (moving mesh): drop the per-element geometry cache, rebuilt at the next assembly */
void DflMeshGeometryChanged(Mesh3D* mesh) {
    MeshExt* x = (MeshExt*)mesh->ext;
    if (!x) return;
    CdamFreeDevice(x->egeo_b, 0);
    x->egeo_b = NULL;
    if (x->patch) { /* schedule 2 keeps its own copy in patch order */
        CdamFreeDevice(x->patch->d_egeo, 0);
        x->patch->d_egeo = NULL;
    }
    /* schedule 4 and the residual kernels recompute the geometry from the node records; the face lists hold no geometry */
}
b32 DflQuiet(void) { return g_quiet; }

/* the 4x4-block array the kernels write and its nodal pattern.  Block mode: the matrix's own storage.  Reference layout
 * (MatrixFSUseReferenceLayout): a scratch block array filled from the four sub-matrix arrays, written back by block_done --
 * *scratch says which.  Anything else is not the (u,p) matrix of main.c:374-391: message + trap, as the reference's guards. */
static const CSRAttr* block_pattern(Matrix* J, value_type** val, b32* scratch) {
    *val = MatrixFSBlockValues(J);
    *scratch = FALSE;
    if (!*val) {
        *val = DflMatrixFSScratchBlockBegin(J);
        *scratch = *val != NULL;
    }
    if (!*val) {
        fprintf(stderr, "AssembleSystemTet: J must be the (u,p) field-split matrix of src/main.c:374-391 after MatrixSetup\n");
        ASSERT(FALSE);
        return NULL;
    }
    return ((MatrixFS*)J->data)->spy1x1;
}
static void block_done(Matrix* J, b32 scratch) {
    if (scratch) DflMatrixFSScratchBlockEnd(J);
}

/* (elem,a,b) -> nonzero map for this pattern, batch order; built once per (mesh, pattern) */
static void ensure_nzmap(Mesh3D* mesh, const CSRAttr* spy) {
    MeshExt* x = (MeshExt*)mesh->ext;
    if (x->nzmap_b && x->nzmap_attr == spy) return;
    CdamFreeDevice(x->nzmap_b, 0);
    x->nzmap_b = (index_type*)CdamMallocDevice((ptrdiff_t)mesh->num_tet * 16 * SIZE_OF(index_type));
    dfl_elem_nzmap(mesh->num_tet, x->ien_b, spy->row_ptr, spy->col_ind, x->nzmap_b, DflStream());
    x->nzmap_attr = spy;
}

f64* DflMeshNodeRecords(Mesh3D* mesh) {
    MeshExt* x = (MeshExt*)mesh->ext;
    if (!x->nodep) {
        const index_type N = Mesh3DNumNode(mesh);
        x->nodep = (f64*)CdamMallocDevice((ptrdiff_t)N * 16 * SIZE_OF(f64));
        x->nodexu = (f64*)CdamMallocDevice((ptrdiff_t)N * 8 * SIZE_OF(f64));
        x->Fp = (f64*)CdamMallocDevice((ptrdiff_t)N * 8 * SIZE_OF(f64));
    }
    return x->nodep;
}

/* The Jacobian schedule of this mesh for this pattern (modes 2-4), built on first use.  A builder that refuses the mesh --
 * it has printed why: a node patch beyond the kernel's slot / tet limits, too many patch colors -- makes the mesh FALL BACK
 * to schedule 1 (compact colors, the reference-shaped scatter) for good instead of trapping.  Returns the mode in force. */
static int ensure_lhs_schedule(Mesh3D* mesh, const CSRAttr* spy) {
    MeshExt* x = (MeshExt*)mesh->ext;
    const int mode = x->cfg.sched_mode;
    b32 failed = FALSE;
    if (mode == 2) {
        if (x->patch && x->patch->attr != spy) { DflFreePatchSchedule(x->patch); x->patch = NULL; }
        if (!x->patch) x->patch = DflBuildPatchSchedule(mesh, spy, x->cfg.patch_leaf, x->cfg.patch_cap);
        failed = x->patch == NULL;
    } else if (mode == 3) {
        if (x->rowpatch && x->rowpatch->attr != spy) { DflFreeRowPatchSchedule(x->rowpatch); x->rowpatch = NULL; }
        if (!x->rowpatch) x->rowpatch = DflBuildRowPatchSchedule(mesh, spy, x->cfg.rowpatch_leaf, x->cfg.rowpatch_cap);
        failed = x->rowpatch == NULL;
    } else if (mode == 4) {
        if (x->slotpatch && x->slotpatch->attr != spy) { DflFreeSlotPatchSchedule(x->slotpatch); x->slotpatch = NULL; }
        if (!x->slotpatch) x->slotpatch = DflBuildSlotPatchSchedule(mesh, spy, x->cfg.slot_leaf, x->cfg.slot_cap, x->cfg.slot_tets);
        failed = x->slotpatch == NULL;
    }
    if (failed) {
        fprintf(stderr, "dedflow: assembly schedule %d is not available for this mesh; falling back to schedule 1 (compact colors)\n", mode);
        x->cfg.sched_mode = 1;
    }
    return x->cfg.sched_mode;
}

void AssembleSystemTet(Mesh3D* mesh, f64* wgalpha_dptr, f64* dwgalpha_dptr, f64* F, Matrix* J) {
    DflAssembleSystemTetBeta(mesh, wgalpha_dptr, dwgalpha_dptr, F, J, 1.0);
}

void DflAssembleSystemTetBeta(Mesh3D* mesh, f64* wgalpha_dptr, f64* dwgalpha_dptr, f64* F, Matrix* J, f64 beta_J) {
    MeshExt* x = (MeshExt*)mesh->ext;
    const Mesh3DData* dev = Mesh3DDevice(mesh);
    const index_type N = Mesh3DNumNode(mesh);
    hipStream_t s = DflStream();
    ASSERT((F || J) && "Either F or J should be provided");
    ASSERT(x && x->ien_b && "Mesh3DGenerateColorBatch must run before assembly");
    if (!g_quiet) printf("Assemble: %s %s\n", F ? "F" : "", J ? "J" : "");
    value_type* val = NULL;
    const CSRAttr* spy = NULL;
    b32 scratch = FALSE;
    if (J) {
        spy = block_pattern(J, &val, &scratch);
        const int before = x->cfg.sched_mode;
        if (ensure_lhs_schedule(mesh, spy) != before && beta_J == 0.0) {
            /* the caller counted on an overwriting schedule; the colored scatter adds */
            HIPGUARD(hipMemsetAsync(val, 0, (size_t)spy->nnz * 16 * sizeof(value_type), s));
            beta_J = 1.0;
        }
        if (x->cfg.sched_mode < 2) ensure_nzmap(mesh, spy);
    }
    /* packed gather records (one line per node) and packed residual accumulator */
    DflMeshNodeRecords(mesh);
    /* (the compact (x, u) copy only when a Jacobian follows: the residual kernels read the full records) */
    const b32 patch_lhs = J && x->cfg.sched_mode == 2;
    const b32 rowpatch_lhs = J && x->cfg.sched_mode == 3;
    const b32 slot_lhs = J && x->cfg.sched_mode == 4;
    const b32 patch_rhs = F && x->cfg.sched_mode >= 2;
    /* Who reads which records: the slot-owner J kernel only the compact (x, u) ones, every other element kernel the full
       ones; a Jacobian-only call on the default schedule therefore packs only the compact records.  (DFL_RHS_DIRECT=1, a
       measured alternative: the lane-per-tet residual kernel gathers from the caller's arrays itself and a residual-only
       call packs nothing -- 14 8-byte gathers per node cost the kernel more, 0.38 M of 1.28 M cycles per wave, than the
       0.10 ms pack pass it saves: 1.01 against 0.96 ms per call.) */
    static int rhs_direct_ok = -1;
    if (rhs_direct_ok < 0) rhs_direct_ok = getenv("DFL_RHS_DIRECT") && atoi(getenv("DFL_RHS_DIRECT")) == 1;
    const b32 lane_rhs = patch_rhs && x->cfg.sched_mode >= 4 && x->cfg.rhswave_tets == 64 && x->cfg.rhswave_nodes == 64 &&
                         !dfl_tune_asm_flags();
    const b32 rhs_direct = lane_rhs && rhs_direct_ok && dwgalpha_dptr && wgalpha_dptr && !x->nodep_current &&
                           !(getenv("DFL_RHS_WPB") && atoi(getenv("DFL_RHS_WPB")) == 4);
    const b32 need_full = (F && !rhs_direct) || (J && !slot_lhs);
    if (!x->nodep_current)
        dfl_pack_nodes2(N, dev->xg, wgalpha_dptr, dwgalpha_dptr, need_full ? x->nodep : NULL, slot_lhs ? x->nodexu : NULL, s);
    if (J && !slot_lhs && !x->egeo_b) { /* geometry cache in schedule order (static mesh), built once; the LHS kernels read it */
        x->egeo_b = (f64*)CdamMallocDevice((ptrdiff_t)mesh->num_tet * 16 * SIZE_OF(f64));
        dfl_elem_geometry(mesh->num_tet, x->ien_b, dev->xg, x->egeo_b, s);
    }
    /* one launch per class of the execution schedule (mesh.c: the reference's color batches in
       mode 0, the compact re-coloring otherwise) */
    for (index_type b = 0; b < x->sched_num; ++b) {
        const index_type off = x->sched_offset[b];
        const index_type bsz = x->sched_offset[b + 1] - off;
        if (bsz == 0) continue;
        const index_type* ien_b = x->ien_b + (size_t)off * 4;
        if (F && !patch_rhs) DFL_TIMED(DFL_TAG_ASM_RHS, dfl_assemble_tet_rhs(bsz, ien_b, x->nodep, x->Fp, s));
        if (J && !patch_lhs && !rowpatch_lhs && !slot_lhs)
            DFL_TIMED(DFL_TAG_ASM_LHS, dfl_assemble_tet_lhs(bsz, ien_b, x->nzmap_b + (size_t)off * 16, x->egeo_b + (size_t)off * 16,
                                                            x->nodep, val, s));
    }
    if (patch_lhs) { /* schedule 2: one launch per PATCH color, each block RMW'd once per patch (host/patch.c) */
        if (!x->patch->d_egeo) {
            x->patch->d_egeo = (f64*)CdamMallocDevice((ptrdiff_t)mesh->num_tet * 16 * SIZE_OF(f64));
            dfl_elem_geometry(mesh->num_tet, x->patch->d_ien, dev->xg, x->patch->d_egeo, s);
        }
        const PatchSched* ps = x->patch;
        for (index_type c = 0; c < ps->num_color; ++c) {
            const index_type p0 = ps->color_offset[c], np = ps->color_offset[c + 1] - p0;
            if (!np) continue;
            DFL_TIMED(DFL_TAG_ASM_LHS, dfl_assemble_tet_lhs_patch(np, p0, ps->d_eoff, ps->d_boff, ps->d_ien, ps->d_lslot,
                                                                  ps->d_blk_nz, ps->d_egeo, x->nodep, val, ps->max_slots, s));
        }
    }
    if (rowpatch_lhs) { /* schedule 3: ONE launch, every workgroup owns the rows of its node patch (host/rowpatch.c) */
        const RowPatchSched* rs = x->rowpatch;
        DFL_TIMED(DFL_TAG_ASM_LHS, dfl_assemble_tet_lhs_rowpatch(rs->num_patch, rs->d_ioff, rs->d_soff, rs->d_item_ea, rs->d_item_slot,
                                                                 rs->d_slot_nz, x->ien_b, x->egeo_b, x->nodep, val, beta_J,
                                                                 rs->max_slots, s));
    }
    if (slot_lhs) { /* schedule 4: ONE launch, every nodal nonzero is summed in registers by its owner lanes (host/slotpatch.c) */
        const SlotPatchSched* ss = x->slotpatch;
        DFL_TIMED(DFL_TAG_ASM_LHS, dfl_assemble_tet_lhs_slot(ss->num_patch, ss->d_hdr, ss->d_ptet_lid, ss->d_pnode, ss->d_slot_nz, ss->d_ldesc,
                                                             x->nodexu, val, beta_J, ss->max_tets, s));
    }
    if (patch_rhs) { /* schedules 2, 3, 4: patch-staged residual, two launches, fixed summation order (host/patch.c) */
        const b32 wave = x->cfg.sched_mode >= 4; /* schedule 4: one wave per patch, padded layout */
        if (x->rhspatch && (x->rhspatch->pad_tets > 0) != wave) {
            DflFreeRhsPatchSchedule(x->rhspatch);
            x->rhspatch = NULL;
        }
        if (!x->rhspatch)
            x->rhspatch = wave ? DflBuildRhsPatchSchedule(mesh, x->cfg.rhswave_tets, x->cfg.rhswave_nodes, x->cfg.rhswave_tets, x->cfg.rhswave_nodes)
                               : DflBuildRhsPatchSchedule(mesh, x->cfg.rhspatch_leaf, x->cfg.rhspatch_nodes, 0, 0);
        const RhsPatchSched* rp = x->rhspatch;
        int slot = DflProfileBegin(DFL_TAG_ASM_RHS);
        if (wave && rp->d_sub4 && rhs_direct)
            dfl_assemble_tet_rhs_lane_direct(rp->num_patch, rp->d_cnt, rp->d_pnode, rp->d_lien, rp->d_sub4, rp->d_sub_start, dev->xg,
                                             wgalpha_dptr, dwgalpha_dptr, N, rp->d_partial, s);
        else if (wave && rp->d_sub4 && !(dfl_tune_asm_flags() & 32))
            dfl_assemble_tet_rhs_lane(rp->num_patch, rp->d_cnt, rp->d_pnode, rp->d_lien, rp->d_sub4, rp->d_sub_start, x->nodep,
                                      rp->d_partial, s);
        else if (wave)
            dfl_assemble_tet_rhs_wave(rp->num_patch, rp->pad_tets, rp->pad_nodes, rp->d_cnt, rp->d_pnode, rp->d_lien, rp->d_adj,
                                      rp->d_adj_start, x->nodep, rp->d_partial, s);
        else
            dfl_assemble_tet_rhs_patch(rp->num_patch, rp->d_eoff, rp->d_noff, rp->d_pnode, rp->d_lien, rp->d_adj, rp->d_adj_start,
                                       x->nodep, rp->d_partial, s);
        dfl_rhs_node_sum(N, rp->d_goff, rp->d_gidx, rp->d_partial, F, s);
        DflProfileEnd(slot);
    } else if (F) {
        dfl_unpack_rhs(N, x->Fp, F, s);
    }
    if (J) block_done(J, scratch);
}

void AssembleSystemTetFace(Mesh3D* mesh, f64* wgalpha_dptr, f64* dwgalpha_dptr, f64* F, Matrix* J) {
    const index_type group = ((MeshExt*)mesh->ext)->cfg.face_group; /* 4 in the reference, hard-coded at assemble.cu:1826-1828 */
    const Mesh3DData* dev = Mesh3DDevice(mesh);
    const index_type N = Mesh3DNumNode(mesh);
    hipStream_t s = DflStream();
    if (mesh->num_bound <= group) return;
    DflMeshPrepareFaces(mesh, group);
    MeshExt* x = (MeshExt*)mesh->ext;
    value_type* val = NULL;
    const CSRAttr* spy = NULL;
    b32 scratch = FALSE;
    if (J) spy = block_pattern(J, &val, &scratch);
    const index_type* f2e = Mesh3DBoundF2E(mesh, group);
    const index_type* forn = Mesh3DBoundFORN(mesh, group);
    int slot = DflProfileBegin(DFL_TAG_FACE);
    if (spy) DflMeshPrepareFaceNonzeros(mesh, group, spy);
    /* pass 1: every face parks its terms; pass 2: ordered sums into F and the block values */
    dfl_assemble_face_park(x->face_nf, f2e, forn, dev->ien, N, dev->xg, wgalpha_dptr, dwgalpha_dptr, F ? x->face_pF : NULL,
                           spy ? x->face_pJ : NULL, s);
    if (F) dfl_face_sum_F(x->face_nn, x->face_node, x->face_node_off, x->face_node_ent, x->face_pF, N, F, s);
    if (spy) dfl_face_sum_J(x->face_nnz, x->face_nz, x->face_nz_off, x->face_nz_ent, x->face_pJ, val, s);
    DflProfileEnd(slot);
    if (J) block_done(J, scratch);
}

void AssembleSystem(Mesh3D* mesh, f64* wgalpha, f64* dwgalpha, f64* F, Matrix* J, Dirichlet** bcs, index_type nbc) {
    DflAssembleSystemPrepacked(mesh, wgalpha, dwgalpha, F, J, bcs, nbc, FALSE);
}

void DflAssembleSystemPrepacked(Mesh3D* mesh, f64* wgalpha, f64* dwgalpha, f64* F, Matrix* J, Dirichlet** bcs, index_type nbc,
                                b32 prepacked) {
    index_type num_node = Mesh3DNumNode(mesh);
    MeshExt* x = (MeshExt*)mesh->ext;
    x->nodep_current = prepacked && x->nodep != NULL;
    hipStream_t s = DflStream();
    DflRangePush(F && J ? "AssembleSystem(F,J)" : F ? "AssembleSystem(F)" : "AssembleSystem(J)");
    if (F) HIPGUARD(hipMemsetAsync(F, 0, (size_t)num_node * sizeof(f64) * BS, s));
    /* schedule 3 writes every row of J exactly once: the zero pass folds into that write */
    if (J && Mesh3DNumTet(mesh) && MatrixFSBlockValues(J) && x->cfg.sched_mode >= 2)
        (void)ensure_lhs_schedule(mesh, ((MatrixFS*)J->data)->spy1x1); /* may fall back to schedule 1: decides `overwrite` */
    const b32 overwrite = J && x->cfg.sched_mode >= 3 && Mesh3DNumTet(mesh) && MatrixFSBlockValues(J);
    if (J && !overwrite) MatrixZero(J);
    if (Mesh3DNumTet(mesh)) {
        DflAssembleSystemTetBeta(mesh, wgalpha, dwgalpha, F, J, overwrite ? 0.0 : 1.0);
        AssembleSystemTetFace(mesh, wgalpha, dwgalpha, F, J);
    }
    x->nodep_current = FALSE;
    if (F) HIPGUARD(hipMemsetAsync(F + 4 * (size_t)num_node, 0, (size_t)num_node * sizeof(f64) * 2, s)); /* main.c:63-66 */
    for (index_type ibc = 0; ibc < nbc; ++ibc) {
        if (F) DirichletApplyVec(bcs[ibc], F);
        if (J) DirichletApplyMat(bcs[ibc], J);
    }
    DflRangePop();
}
