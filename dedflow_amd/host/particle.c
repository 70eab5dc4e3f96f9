/* ParticleContext behind src/Particle.h:13-35.  The reference provides storage only
 * (coord / vel / acc, host + device, mass 1.0, radius 0.1 hard-coded, Particle.c:8-26) and
 * empty Add/Update/Remove hooks (:120-130).  The contact sweep is build-defined
 * (dedflow_amd/csrc/k_dem.hip): ParticleContextComputeForces = cell list + force kernel,
 * ParticleContextUpdate = forces + semi-implicit Euler step. */
#include <math.h>
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

typedef struct ParticleExt {
    f64 kn, gamma_n, dt;
    f64 cell;
    index_type ncell;
    /* persistent workspace of the sweep (device): nothing is allocated, freed or synchronised per sweep */
    index_type *cell_of, *rank, *slot, *order;  /* [P] */
    index_type *count, *cell_start, *chunk_sum; /* [ncell^3 + 1], [ncell^3 + 1], [chunks] */
    f64* sorted;                                /* [P][6] position + velocity in (cell, id) order */
    index_type cap_particle, cap_cell;
} ParticleExt;

ParticleContext* ParticleContextCreate(index_type num_particle) {
    ParticleContext* ctx = (ParticleContext*)CdamMallocHost(SIZE_OF(ParticleContext));
    memset(ctx, 0, sizeof *ctx);
    ctx->num_particle = num_particle;
    ctx->num_pointwise_dof = 9;
    for (int k = 0; k < 3; ++k) {
        ctx->h_arr[k] = ArrayCreateHost(num_particle * 3);
        ctx->d_arr[k] = ArrayCreateDevice(num_particle * 3);
    }
    ParticleMass(ctx) = 1.0;   /* Particle.c:23-24 */
    ParticleRadius(ctx) = 0.1;
    ParticleExt* x = (ParticleExt*)CdamMallocHost(SIZE_OF(ParticleExt));
    memset(x, 0, sizeof *x);
    x->kn = 1.0e4;
    x->gamma_n = 1.0;
    x->dt = 1.0e-4;
    ctx->ext = x;
    return ctx;
}

void ParticleContextDestroy(ParticleContext* ctx) {
    if (!ctx) return;
    ParticleExt* x = (ParticleExt*)ctx->ext;
    for (int k = 0; k < 3; ++k) {
        ArrayDestroy(ctx->h_arr[k]);
        ArrayDestroy(ctx->d_arr[k]);
    }
    if (x) {
        CdamFreeDevice(x->cell_of, 0); CdamFreeDevice(x->rank, 0); CdamFreeDevice(x->slot, 0); CdamFreeDevice(x->order, 0); CdamFreeDevice(x->sorted, 0);
        CdamFreeDevice(x->count, 0); CdamFreeDevice(x->cell_start, 0); CdamFreeDevice(x->chunk_sum, 0);
        CdamFreeHost(x, SIZE_OF(ParticleExt));
    }
    CdamFreeHost(ctx, SIZE_OF(ParticleContext));
}

void ParticleContextCopy(ParticleContext* dst, const ParticleContext* src) {
    ASSERT(dst && src && dst->num_particle == src->num_particle);
    for (int k = 0; k < 3; ++k) {
        ArrayCopy(dst->h_arr[k], src->h_arr[k], H2H);
        ArrayCopy(dst->d_arr[k], src->d_arr[k], D2D);
    }
}
void ParticleContextUpdateHost(ParticleContext* ctx) {
    for (int k = 0; k < 3; ++k) ArrayCopy(ctx->h_arr[k], ctx->d_arr[k], D2H);
}
void ParticleContextUpdateDevice(ParticleContext* ctx) {
    for (int k = 0; k < 3; ++k) ArrayCopy(ctx->d_arr[k], ctx->h_arr[k], H2D);
}
void ParticleContextAdd(ParticleContext* ctx) { UNUSED(ctx); }
void ParticleContextRemove(ParticleContext* ctx) { UNUSED(ctx); }

void ParticleContextSetContactModel(ParticleContext* ctx, f64 kn, f64 gamma_n, f64 dt) {
    ParticleExt* x = (ParticleExt*)ctx->ext;
    x->kn = kn;
    x->gamma_n = gamma_n;
    x->dt = dt;
}

void ParticleContextComputeForces(ParticleContext* ctx) {
    ParticleExt* x = (ParticleExt*)ctx->ext;
    const index_type P = ctx->num_particle;
    const f64 R = ParticleRadius(ctx);
    hipStream_t s = DflStream();
    DflRangePush("ParticleContextComputeForces");
    /* cell edge >= 4R: the interaction range of a particle covers at most two cells per axis; and not (much) finer than a few
       particles per cell -- the scan over the cells is what an over-fine grid pays for (125^3 cells for 100k particles cost
       14 us of scan; 58^3 cells 4 us, and the force kernel still tests only ~1.5 neighbours per particle) */
    index_type ncell = (index_type)floor(1.0 / (4.0 * R));
    const index_type by_count = (index_type)floor(cbrt(2.0 * (f64)P) + 0.5); /* about half a particle per cell */
    if (ncell > by_count) ncell = by_count;
    if (ncell < 1) ncell = 1;
    if (ncell > 256) ncell = 256; /* 2^24 cells at most (the dense cell arrays) */
    const f64 cell = 1.0 / (f64)ncell; /* >= 4R */
    const index_type ncell3 = ncell * ncell * ncell;
    if (x->cap_particle < P) {
        CdamFreeDevice(x->cell_of, 0); CdamFreeDevice(x->rank, 0); CdamFreeDevice(x->slot, 0); CdamFreeDevice(x->order, 0); CdamFreeDevice(x->sorted, 0);
        x->cell_of = (index_type*)CdamMallocDevice((ptrdiff_t)P * SIZE_OF(index_type));
        x->rank = (index_type*)CdamMallocDevice((ptrdiff_t)P * SIZE_OF(index_type));
        x->slot = (index_type*)CdamMallocDevice((ptrdiff_t)P * SIZE_OF(index_type));
        x->order = (index_type*)CdamMallocDevice((ptrdiff_t)P * SIZE_OF(index_type));
        x->sorted = (f64*)CdamMallocDevice((ptrdiff_t)P * 6 * SIZE_OF(f64));
        x->cap_particle = P;
    }
    if (x->cap_cell < ncell3 + 1) { /* zero-filled by the allocator; every sweep leaves count / chunk_sum zeroed again */
        CdamFreeDevice(x->count, 0); CdamFreeDevice(x->cell_start, 0); CdamFreeDevice(x->chunk_sum, 0);
        x->count = (index_type*)CdamMallocDevice(((ptrdiff_t)ncell3 + 1) * SIZE_OF(index_type));
        x->cell_start = (index_type*)CdamMallocDevice(((ptrdiff_t)ncell3 + 1) * SIZE_OF(index_type));
        x->chunk_sum = (index_type*)CdamMallocDevice((ptrdiff_t)dfl_dem_num_chunks(ncell3) * SIZE_OF(index_type));
        x->cap_cell = ncell3 + 1;
    }
    x->cell = cell;
    x->ncell = ncell;
    const f64* coord = ArrayData(ParticleCTXDeviceCoord(ctx));
    const f64* vel = ArrayData(ParticleCTXDeviceVel(ctx));
    f64* acc = ArrayData(ParticleCTXDeviceAcc(ctx));
    dfl_dem_build_cells(P, coord, vel, cell, ncell, x->cell_of, x->rank, x->count, x->chunk_sum, x->cell_start, x->slot, x->order, x->sorted, s);
    int slot = DflProfileBegin(DFL_TAG_SMALL + 1);
    dfl_dem_forces(P, x->sorted, R, ParticleMass(ctx), x->kn, x->gamma_n, cell, ncell, x->order, x->cell_start, acc, s);
    DflProfileEnd(slot);
    DflRangePop();
}

void ParticleContextUpdate(ParticleContext* ctx) {
    ParticleExt* x = (ParticleExt*)ctx->ext;
    ParticleContextComputeForces(ctx);
    dfl_dem_integrate(ctx->num_particle, x->dt, ArrayData(ParticleCTXDeviceCoord(ctx)), ArrayData(ParticleCTXDeviceVel(ctx)),
                      ArrayData(ParticleCTXDeviceAcc(ctx)), DflStream());
}
