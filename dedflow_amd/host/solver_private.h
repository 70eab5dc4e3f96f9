/* Krylov solver internals shared by host/solver.c and host/ws_placement.c. */
#ifndef DFL_SOLVER_PRIVATE_H
#define DFL_SOLVER_PRIVATE_H
#include "dedflow.h"

typedef struct KrylovExt {
    KrylovStats stats;
    index_type check_interval;
    b32 check_interval_set; /* KrylovSetCheckInterval was called: PC_TWOLEVEL leaves the interval alone */
    b32 verbose;
    DflComm comm;
    b32 has_comm;
    PCType pc_type; /* tree KrylovSolve builds: PC_DECOMPOSITION (reference) or PC_ILU0 */
    index_type restart; /* GMRES(m): basis columns per cycle; <= 0 or >= max_iter = full GMRES (the reference, krylov.c:56-334) */
    b32 flexible_user; /* KrylovSetFlexible(on): FGMRES whatever the preconditioner; otherwise PC_TWOLEVEL alone switches it on */
    b32 flexible;   /* FGMRES: keep Z[:,k] = M_k^-1 Q[:,k] (a second basis) so that the preconditioner may vary from step to step */
    const Mesh3D* mesh; /* optional: node coordinates for preconditioners that aggregate nodes (PC_TWOLEVEL) */
    index_type agg_size; /* PC_TWOLEVEL: nodes per aggregate */
    b32 fused_norm; /* partitioned runs: ||w - Qh|| from w.w - sum h^2, one all-reduce per Arnoldi step (off by default) */
    int q_pooled;   /* the basis Q came from the device pool (placement calibration may pick either kind) */
    int* d_flag;    /* device int raised by the fused-norm kernel on heavy cancellation */
    f64* hraw;      /* [ldh] raw CGS coefficients + w.w of the current column (fused update + PC + Givens kernel) */
    /* cached GMRES work space */
    index_type ws_n, ws_maxit, ws_hist;
    int ws_pooled; /* where Q and tmp of the cached work space came from */
    b32 ws_fresh;  /* the basis was (re)allocated and its placement has not been calibrated yet */
    f64 *Q, *Z, *H, *tmp, *gv, *beta, *res_hist, *nrm, *work;
    int64_t work_len;
    /* host reads of a solve without idling the GPU (GMRESSolvePrivate): probes + ||r0|| travel with the first convergence
       check, the check itself is read one iteration late */
    f64* nrm_base;      /* allocation behind nrm: [tail probe, x probe, -, -, nrm[0], ...] */
    f64* h_stat;        /* pinned host staging [16] */
    hipEvent_t ev_stat;
    b32 assume_valid, assume_tail_zero, assume_x_zero; /* what the previous solve of this solver found (verified per solve) */
    /* p(1)-pipelined GMRES (KrylovSetPipelined): auxiliary basis z_{j+1} = A M^-1 v_j, the reduction's own stream */
    b32 pipelined;
    f64* Zp;
    index_type zp_n, zp_m;
    int zp_pooled;
    hipStream_t red_stream;
    hipEvent_t ev_w, ev_h;
    b32 no_calibration; /* inner / coarse solvers of PC_TWOLEVEL: never time basis placements (DflKrylovMarkInner) */
} KrylovExt;


/* host/solver.c */
int DflWsInPool(void);
void DflWsVecFreeAs(f64* p, int pooled);
void DflPcApplyFused(PC* pc, index_type na, f64* w, const f64* d_nrm, f64* z);
b32 DflPcApplyFusedX4(PC* pc, index_type na, f64* w, const f64* d_nrm, f64* z, f64* z4); /* TRUE: z4 written too */
PC* DflKrylovBuildPC(Krylov* ksp, Matrix* A);            /* the (re)build step of KrylovSolve */
b32 DflKrylovEnsureWorkspace(Krylov* ksp, Matrix* A, index_type* n, index_type* m, index_type* ldh);
void DflKrylovMarkInner(Krylov* ksp);

/* host/ws_placement.c: where the Krylov basis (and, in the explicit heavy form, the value array) is placed */
f64* DflWsPickBasis(KrylovExt* ex, Matrix* A, PC* pc, f64* first, ptrdiff_t count, index_type na, index_type m, index_type ldh);

#endif
