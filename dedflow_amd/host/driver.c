/* The immediate caller of the hot path: generalized-alpha predictor / multi-corrector of
 * src/main.c:77-283 (SolveFlowSystem) and the per-step prediction/update of :535-565.
 * Same state algebra (fact1/fact2, :95-97; fac_pred/fac_corr, :535-536), same Newton control
 * (<= 4 iterations, 4-way relative test at 5e-4, :157,271-276), same printed lines.  The reference
 * spells the state algebra as ~10 cuBLAS BLAS-1 passes over 6N-vectors and four Dnrm2 host syncs per
 * Newton iteration; here every group is ONE pass (dfl_alpha_states, which also writes the packed node
 * records of the assembly kernels; dfl_alpha_predict / dfl_alpha_correct; dfl_norms4 + one 32-byte copy).
 * The work vectors live in the mesh (no process-global state). */
#include <math.h>
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

#define kRHOC (0.5)
#define kDT (5e-2)
#define kALPHAM ((3.0 - kRHOC) / (1.0 + kRHOC))
#define kALPHAF (1.0 / (1.0 + kRHOC))
#define kGAMMA (0.5 + kALPHAM - kALPHAF)
#define BS (6)

typedef struct FlowWork {
    index_type num_node;
    f64 *wgalpha, *dwgalpha, *nrm, *work;
} FlowWork;

/* work vectors of one mesh (kept in its MeshExt: two meshes / two solvers in one process do not collide) */
static FlowWork* fw_get(Mesh3D* mesh) {
    MeshExt* x = (MeshExt*)mesh->ext;
    const index_type N = Mesh3DNumNode(mesh);
    if (x->flow && x->flow->num_node == N) return x->flow;
    DflFreeFlowWork(x->flow);
    FlowWork* fw = (FlowWork*)CdamMallocHost(SIZE_OF(FlowWork));
    fw->wgalpha = (f64*)CdamMallocDevice((ptrdiff_t)N * BS * SIZE_OF(f64));
    fw->dwgalpha = (f64*)CdamMallocDevice((ptrdiff_t)N * BS * SIZE_OF(f64));
    fw->nrm = (f64*)CdamMallocDevice(8 * SIZE_OF(f64));
    fw->work = (f64*)CdamMallocDevice((ptrdiff_t)(dfl_reduce_work_size() + 16) * SIZE_OF(f64));
    fw->num_node = N;
    x->flow = fw;
    return fw;
}
void DflFreeFlowWork(FlowWork* fw) {
    if (!fw) return;
    CdamFreeDevice(fw->wgalpha, 0); CdamFreeDevice(fw->dwgalpha, 0); CdamFreeDevice(fw->nrm, 0); CdamFreeDevice(fw->work, 0);
    CdamFreeHost(fw, SIZE_OF(FlowWork));
}

/* alpha-level states, main.c:107-118 and :242-253: one pass (8 full-vector passes in the reference), which also writes
 * the packed node records the assembly kernels gather from -- the two AssembleSystem calls that follow skip their pack */
static void alpha_states(Mesh3D* mesh, const f64* wgold, const f64* dwgold, const f64* dwg, f64* wgalpha, f64* dwgalpha) {
    const f64 fact1[] = {1.0 - kALPHAM, kALPHAM};
    const f64 fact2[] = {kDT * kALPHAF * (1.0 - kGAMMA), kDT * kALPHAF * kGAMMA};
    f64* const nodep = DflMeshNodeRecords(mesh); /* (allocates the compact (x, u) records of the Jacobian kernel as well) */
    dfl_alpha_states2(Mesh3DNumNode(mesh), wgold, dwgold, dwg, fact1[0], fact1[1], fact2[0], fact2[1], Mesh3DDevice(mesh)->xg, wgalpha,
                      dwgalpha, nodep, ((MeshExt*)mesh->ext)->nodexu, DflStream());
}

static void four_norms(FlowWork* fw, index_type N, const f64* F, f64* out, const DflComm* comm) {
    hipStream_t s = DflStream();
    /* element-partitioned run: ghost entries of F are zero, sums of squares are all-reduced */
    dfl_norms4(N, F, fw->nrm, comm ? 0 : 1, fw->work, s);
    if (comm) comm->allreduce_sum(comm->ctx, fw->nrm, 4);
    HIPGUARD(hipMemcpyAsync(out, fw->nrm, 4 * sizeof(f64), D2H, s));
    HIPGUARD(hipStreamSynchronize(s));
    if (comm) for (int k = 0; k < 4; ++k) out[k] = sqrt(out[k]);
}

/* partitioned run: residual entries of ghost nodes are partial sums that belong to another rank */
static void zero_ghost_residual(index_type N, f64* F, const DflComm* comm) {
    if (!comm) return;
    const index_type no = comm->num_owned_node;
    hipStream_t s = DflStream();
    if (no < N) {
        HIPGUARD(hipMemsetAsync(F + (size_t)no * 3, 0, (size_t)(N - no) * 3 * sizeof(f64), s));
        HIPGUARD(hipMemsetAsync(F + (size_t)N * 3 + no, 0, (size_t)(N - no) * sizeof(f64), s));
    }
}

/* SolveFlowSystem, main.c:77-283.  Returns the number of Newton iterations; rnorm_out[0..3] / rnorm_init_out[0..3]
 * receive the last and the initial residual norms (u, p, phi, T). */
index_type SolveFlowSystem(Mesh3D* mesh, f64* wgold, f64* dwgold, f64* dwg, Matrix* J, f64* F, f64* dx, Krylov* ksp,
                           Dirichlet** bcs, index_type nbc, index_type maxit, f64* rnorm_out, f64* rnorm_init_out) {
    const index_type N = Mesh3DNumNode(mesh);
    const f64 tol = 0.5e-3;
    hipStream_t s = DflStream();
    f64 rnorm[4] = {0, 0, 0, 0}, rnorm_init[4];
    index_type iter = 0;
    b32 converged = FALSE;
    if (maxit <= 0) maxit = 4;
    FlowWork* fw = fw_get(mesh);
    const DflComm* comm = KrylovGetComm(ksp);
    KrylovSetMesh(ksp, mesh); /* node coordinates for aggregation-based preconditioners */
    f64 *wgalpha = fw->wgalpha, *dwgalpha = fw->dwgalpha;
    alpha_states(mesh, wgold, dwgold, dwg, wgalpha, dwgalpha);
    DflAssembleSystemPrepacked(mesh, wgalpha, dwgalpha, F, NULL, bcs, nbc, TRUE);
    zero_ghost_residual(N, F, comm);
    four_norms(fw, N, F, rnorm_init, comm);
    if (!DflQuiet())
        for (int k = 0; k < 4; ++k)
            fprintf(stdout, "Newton %d) abs = %.17e rel = %6.4e (tol = %6.4e)\n", 0, rnorm_init[k], 1.0, tol);
    if (rnorm_init_out) memcpy(rnorm_init_out, rnorm_init, sizeof rnorm_init);
    for (int k = 0; k < 4; ++k) rnorm_init[k] += 1e-16;
    while (!converged && iter < maxit) {
        DflAssembleSystemPrepacked(mesh, wgalpha, dwgalpha, NULL, J, bcs, nbc, TRUE); /* same states as the residual before */
        HIPGUARD(hipMemsetAsync(dx, 0, (size_t)N * BS * sizeof(f64), s));
        KrylovSolve(ksp, J, dx, F);
        if (comm) comm->halo_exchange(comm->ctx, dx); /* ghost copies of the increment from their owners */
        dfl_daxpy(N * BS, -1.0, dx, dwg, s); /* dwg -= dx, main.c:226 */
        alpha_states(mesh, wgold, dwgold, dwg, wgalpha, dwgalpha);
        DflAssembleSystemPrepacked(mesh, wgalpha, dwgalpha, F, NULL, bcs, nbc, TRUE);
        zero_ghost_residual(N, F, comm);
        four_norms(fw, N, F, rnorm, comm);
        if (!DflQuiet())
            for (int k = 0; k < 4; ++k)
                fprintf(stdout, "Newton %d) abs = %.17e rel = %6.4e (tol = %6.4e)\n", iter + 1, rnorm[k], rnorm[k] / rnorm_init[k], tol);
        if (rnorm[0] < tol * rnorm_init[0] && rnorm[1] < tol * rnorm_init[1] && rnorm[2] < tol * rnorm_init[2] &&
            rnorm[3] < tol * rnorm_init[3])
            converged = TRUE;
        iter++;
    }
    if (rnorm_out) memcpy(rnorm_out, rnorm, sizeof rnorm);
    return iter;
}

/* one time step of main.c:537-565: predictor, Newton solve, corrector; optional DEM sub-steps */
index_type DflTimeStep(Mesh3D* mesh, f64* wgold, f64* dwgold, f64* dwg, Matrix* J, f64* F, f64* dx, Krylov* ksp, Dirichlet** bcs,
                       index_type nbc, index_type newton_maxit, ParticleContext* pctx, index_type dem_substeps, f64* rnorm_out,
                       f64* rnorm_init_out) {
    const index_type N = Mesh3DNumNode(mesh);
    hipStream_t s = DflStream();
    const f64 fac_pred = (kGAMMA - 1.0) / kGAMMA;
    const f64 fac_corr[] = {kDT * (1.0 - kGAMMA), kDT * kGAMMA};
    DflRangePush("DflTimeStep");
    dfl_alpha_predict(N, fac_pred, dwg, s);
    index_type it = SolveFlowSystem(mesh, wgold, dwgold, dwg, J, F, dx, ksp, bcs, nbc, newton_maxit, rnorm_out, rnorm_init_out);
    if (pctx)
        for (index_type k = 0; k < dem_substeps; ++k) ParticleContextUpdate(pctx); /* coupled step: contact sweep (config 4) */
    dfl_alpha_correct(N, fac_corr[0], fac_corr[1], wgold, dwgold, dwg, s);
    DflRangePop();
    return it;
}
