/* The immediate caller of the hot path: generalized-alpha predictor / multi-corrector of
 * src/main.c:77-283 (SolveFlowSystem) and the per-step prediction/update of :535-565.
 * Same state algebra (fact1/fact2, :95-97; fac_pred/fac_corr, :535-536), same Newton control
 * (<= 4 iterations, 4-way relative test at 5e-4, :157,271-276), same printed lines; the
 * cuBLAS BLAS-1 calls are dfl_* launchers and the four Dnrm2 host syncs per Newton iteration
 * are one 32-byte copy. */
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
static FlowWork g_fw = {0, NULL, NULL, NULL, NULL};

static void fw_ensure(index_type N) {
    if (g_fw.num_node == N) return;
    CdamFreeDevice(g_fw.wgalpha, 0); CdamFreeDevice(g_fw.dwgalpha, 0); CdamFreeDevice(g_fw.nrm, 0); CdamFreeDevice(g_fw.work, 0);
    g_fw.wgalpha = (f64*)CdamMallocDevice((ptrdiff_t)N * BS * SIZE_OF(f64));
    g_fw.dwgalpha = (f64*)CdamMallocDevice((ptrdiff_t)N * BS * SIZE_OF(f64));
    g_fw.nrm = (f64*)CdamMallocDevice(8 * SIZE_OF(f64));
    g_fw.work = (f64*)CdamMallocDevice((ptrdiff_t)(dfl_reduce_work_size() + 16) * SIZE_OF(f64));
    g_fw.num_node = N;
}

/* alpha-level states, main.c:107-118 and :242-253 */
static void alpha_states(index_type N, const f64* wgold, const f64* dwgold, const f64* dwg, f64* wgalpha, f64* dwgalpha) {
    hipStream_t s = DflStream();
    const f64 fact1[] = {1.0 - kALPHAM, kALPHAM};
    const f64 fact2[] = {kDT * kALPHAF * (1.0 - kGAMMA), kDT * kALPHAF * kGAMMA};
    HIPGUARD(hipMemsetAsync(dwgalpha, 0, (size_t)N * BS * sizeof(f64), s));
    dfl_daxpy(N * BS, fact1[0], dwgold, dwgalpha, s);
    dfl_daxpy(N * BS, fact1[1], dwg, dwgalpha, s);
    dfl_dcopy(N, dwg + (size_t)N * 3, dwgalpha + (size_t)N * 3, s); /* pressure is not alpha-interpolated */
    dfl_dcopy(N * BS, wgold, wgalpha, s);
    dfl_daxpy(N * BS, fact2[0], dwgold, wgalpha, s);
    dfl_daxpy(N * BS, fact2[1], dwg, wgalpha, s);
    HIPGUARD(hipMemsetAsync(wgalpha + (size_t)N * 3, 0, (size_t)N * sizeof(f64), s));
}

static void four_norms(index_type N, const f64* F, f64* out, const DflComm* comm) {
    hipStream_t s = DflStream();
    if (comm) { /* element-partitioned run: ghost entries of F are zero, sums of squares are all-reduced */
        dfl_ddot(N * 3, F, F, g_fw.nrm + 0, g_fw.work, s);
        dfl_ddot(N, F + (size_t)N * 3, F + (size_t)N * 3, g_fw.nrm + 1, g_fw.work, s);
        dfl_ddot(N, F + (size_t)N * 4, F + (size_t)N * 4, g_fw.nrm + 2, g_fw.work, s);
        dfl_ddot(N, F + (size_t)N * 5, F + (size_t)N * 5, g_fw.nrm + 3, g_fw.work, s);
        comm->allreduce_sum(comm->ctx, g_fw.nrm, 4);
    } else {
        dfl_dnrm2(N * 3, F, g_fw.nrm + 0, g_fw.work, s);
        dfl_dnrm2(N, F + (size_t)N * 3, g_fw.nrm + 1, g_fw.work, s);
        dfl_dnrm2(N, F + (size_t)N * 4, g_fw.nrm + 2, g_fw.work, s);
        dfl_dnrm2(N, F + (size_t)N * 5, g_fw.nrm + 3, g_fw.work, s);
    }
    HIPGUARD(hipMemcpyAsync(out, g_fw.nrm, 4 * sizeof(f64), D2H, s));
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
    fw_ensure(N);
    const DflComm* comm = KrylovGetComm(ksp);
    f64 *wgalpha = g_fw.wgalpha, *dwgalpha = g_fw.dwgalpha;
    alpha_states(N, wgold, dwgold, dwg, wgalpha, dwgalpha);
    AssembleSystem(mesh, wgalpha, dwgalpha, F, NULL, bcs, nbc);
    zero_ghost_residual(N, F, comm);
    four_norms(N, F, rnorm_init, comm);
    if (!DflQuiet())
        for (int k = 0; k < 4; ++k)
            fprintf(stdout, "Newton %d) abs = %.17e rel = %6.4e (tol = %6.4e)\n", 0, rnorm_init[k], 1.0, tol);
    if (rnorm_init_out) memcpy(rnorm_init_out, rnorm_init, sizeof rnorm_init);
    for (int k = 0; k < 4; ++k) rnorm_init[k] += 1e-16;
    while (!converged && iter < maxit) {
        AssembleSystem(mesh, wgalpha, dwgalpha, NULL, J, bcs, nbc);
        HIPGUARD(hipMemsetAsync(dx, 0, (size_t)N * BS * sizeof(f64), s));
        KrylovSolve(ksp, J, dx, F);
        if (comm) comm->halo_exchange(comm->ctx, dx); /* ghost copies of the increment from their owners */
        dfl_daxpy(N * BS, -1.0, dx, dwg, s); /* dwg -= dx, main.c:226 */
        alpha_states(N, wgold, dwgold, dwg, wgalpha, dwgalpha);
        AssembleSystem(mesh, wgalpha, dwgalpha, F, NULL, bcs, nbc);
        zero_ghost_residual(N, F, comm);
        four_norms(N, F, rnorm, comm);
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
    dfl_dscal(N * 3, fac_pred, dwg, s);
    dfl_dscal(N * 2, fac_pred, dwg + (size_t)N * 4, s);
    index_type it = SolveFlowSystem(mesh, wgold, dwgold, dwg, J, F, dx, ksp, bcs, nbc, newton_maxit, rnorm_out, rnorm_init_out);
    if (pctx)
        for (index_type k = 0; k < dem_substeps; ++k) ParticleContextUpdate(pctx); /* coupled step: contact sweep (config 4) */
    dfl_daxpy(N * 3, fac_corr[0], dwgold, wgold, s);
    dfl_daxpy(N * 2, fac_corr[0], dwgold + (size_t)N * 4, wgold + (size_t)N * 4, s);
    dfl_daxpy(N * 3, fac_corr[1], dwg, wgold, s);
    dfl_daxpy(N * 2, fac_corr[1], dwg + (size_t)N * 4, wgold + (size_t)N * 4, s);
    dfl_dcopy(N * 6, dwg, dwgold, s);
    return it;
}
