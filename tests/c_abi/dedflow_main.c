/* A plain C99 host program written against include/dedflow.h only -- the shape of the reference's src/main.c
 * (read the HDF5 mesh, build the (u,p) field-split matrix, color the mesh, set the four Dirichlet groups, run the
 * generalized-alpha time loop, write sol.<k>.h5).  It is the drop-in claim in executable form: no Python, no HIP call,
 * no kernel launcher -- only the object API the reference's own host code uses.
 *   usage: dedflow_main <mesh.h5> <sol.0.h5> <out prefix> <steps> <newton iterations> [jacobi|ilu0|twolevel]
 * Test infrastructure (tests/test_gpu_c_driver.py builds and runs it); not part of the library. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dedflow.h"

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s mesh.h5 sol.0.h5 out_prefix steps newton_its [jacobi|ilu0|twolevel]\n", argv[0]);
        return 2;
    }
    const int nstep = atoi(argv[4]), newton = atoi(argv[5]);
    Init(argc, argv);
    DflSetQuiet(TRUE);

    /* mesh (main.c:362-372) */
    H5FileInfo* h5 = H5OpenFile(argv[1], "r");
    Mesh3D* mesh = Mesh3DCreateH5(h5, "mesh");
    H5CloseFile(h5);
    const index_type N = Mesh3DNumNode(mesh);

    /* matrix (main.c:374-391): spy1x1 + expanded patterns, 4x4 field-split grid with the (u,p) corner populated */
    CSRAttr* spy1x1 = CSRAttrCreate(mesh);
    CSRAttr* spy1x3 = CSRAttrCreateBlock(spy1x1, 1, 3);
    CSRAttr* spy3x1 = CSRAttrCreateBlock(spy1x1, 3, 1);
    CSRAttr* spy3x3 = CSRAttrCreateBlock(spy1x1, 3, 3);
    index_type offset[5] = {0, 3, 4, 5, 6};
    Matrix* J = MatrixCreateTypeFS(4, offset, NULL);
    MatrixFS* fs = (MatrixFS*)J->data;
    fs->spy1x1 = spy1x1;
    fs->mat[0 * 4 + 0] = MatrixCreateTypeCSR(spy3x3, NULL);
    fs->mat[0 * 4 + 1] = MatrixCreateTypeCSR(spy3x1, NULL);
    fs->mat[1 * 4 + 0] = MatrixCreateTypeCSR(spy1x3, NULL);
    fs->mat[1 * 4 + 1] = MatrixCreateTypeCSR(spy1x1, NULL);
    MatrixSetup(J);

    Krylov* ksp = KrylovCreateGMRES(120, 1e-12, 1e-4, NULL);
    /* the reference's tree (krylov.c:439-453) unless asked otherwise; the two build-defined preconditioners need one call
       each (SolveFlowSystem hands PC_TWOLEVEL the mesh itself) */
    if (argc > 6 && strcmp(argv[6], "ilu0") == 0) KrylovSetPCType(ksp, PC_ILU0);
    if (argc > 6 && strcmp(argv[6], "twolevel") == 0) KrylovSetPCType(ksp, PC_TWOLEVEL);
    Mesh3DGenerateColorBatch(mesh);

    /* boundary conditions (main.c:454-476) */
    const index_type bc_group[4] = {0, 2, 3, 4};
    const BCType bc_type[4][3] = {{BC_STRONG, BC_STRONG, BC_STRONG}, {BC_NONE, BC_STRONG, BC_NONE},
                                  {BC_NONE, BC_NONE, BC_STRONG}, {BC_NONE, BC_NONE, BC_NONE}};
    Dirichlet* bcs[4];
    for (int i = 0; i < 4; ++i) {
        bcs[i] = DirichletCreate(mesh, bc_group[i], 3);
        for (int c = 0; c < 3; ++c) bcs[i]->bctype[c] = bc_type[i][c];
    }

    /* state (main.c:479-532): wgold / dwgold from the solution file, dwg starts as dwgold */
    const ptrdiff_t nb = (ptrdiff_t)N * 6 * SIZE_OF(f64);
    f64* wgold = (f64*)CdamMallocDevice(nb);
    f64* dwgold = (f64*)CdamMallocDevice(nb);
    f64* dwg = (f64*)CdamMallocDevice(nb);
    f64* F = (f64*)CdamMallocDevice(nb);
    f64* dx = (f64*)CdamMallocDevice(nb);
    DflSolutionReadH5(argv[2], N, wgold, dwgold);
    VecAXPY(1.0, dwgold, dwg, N * 6); /* dwg = 0 + dwgold (device allocations are zero-filled) */

    /* time loop (main.c:535-592) */
    for (int step = 1; step <= nstep; ++step) {
        f64 rn[4], r0[4];
        index_type it = DflTimeStep(mesh, wgold, dwgold, dwg, J, F, dx, ksp, bcs, 4, newton, NULL, 0, rn, r0);
        printf("step %d: %d newton iterations, |R| = %.6e %.6e %.6e %.6e\n", step, (int)it, rn[0], rn[1], rn[2], rn[3]);
        char name[1024];
        snprintf(name, sizeof name, "%s.%d.h5", argv[3], step);
        DflSolutionWriteH5(name, N, wgold, dwgold);
    }

    CdamFreeDevice(dx, nb); CdamFreeDevice(F, nb); CdamFreeDevice(dwg, nb); CdamFreeDevice(dwgold, nb); CdamFreeDevice(wgold, nb);
    for (int i = 0; i < 4; ++i) DirichletDestroy(bcs[i]);
    KrylovDestroy(ksp);
    MatrixDestroy(J);
    CSRAttrDestroy(spy3x3); CSRAttrDestroy(spy3x1); CSRAttrDestroy(spy1x3); CSRAttrDestroy(spy1x1);
    Mesh3DDestroy(mesh);
    Finalize();
    return 0;
}
