/* Mesh containers, coloring/batching and the nodal sparsity pattern.
 * API of src/MeshData.c, src/Mesh.c, src/color.c, src/csr.c; the algorithms run on
 * the device through include/dedflow_kernels.h (the reference builds the pattern on
 * one host thread, src/csr.c:81-190). */
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

static Mesh3DData* data_create(b32 is_host, index_type nn, index_type nt, index_type np, index_type nh) {
    Mesh3DData* d = (Mesh3DData*)CdamMallocHost(SIZE_OF(Mesh3DData));
    ptrdiff_t elem = (ptrdiff_t)nt * 4 + (ptrdiff_t)np * 6 + (ptrdiff_t)nh * 8;
    ASSERT(nn >= 4 && "Invalid number of nodes");
    ASSERT(nt + np + nh && "Invalid number of elements");
    memset(d, 0, sizeof *d);
    d->is_host = is_host;
    d->num_node = nn; d->num_tet = nt; d->num_prism = np; d->num_hex = nh;
    if (is_host) {
        d->xg = (f64*)CdamMallocHost((ptrdiff_t)nn * 3 * SIZE_OF(f64));
        d->ien = (index_type*)CdamMallocHost(elem * SIZE_OF(index_type));
        memset(d->xg, 0, (size_t)nn * 3 * sizeof(f64));
        memset(d->ien, 0, (size_t)elem * sizeof(index_type));
    } else {
        d->xg = (f64*)CdamMallocDevice((ptrdiff_t)nn * 3 * SIZE_OF(f64));
        d->ien = (index_type*)CdamMallocDevice(elem * SIZE_OF(index_type));
    }
    return d;
}
Mesh3DData* Mesh3DDataCreateHost(index_type nn, index_type nt, index_type np, index_type nh) { return data_create(TRUE, nn, nt, np, nh); }
Mesh3DData* Mesh3DDataCreateDevice(index_type nn, index_type nt, index_type np, index_type nh) { return data_create(FALSE, nn, nt, np, nh); }

void Mesh3DDataDestroy(Mesh3DData* d) {
    if (!d) return;
    if (d->is_host) { CdamFreeHost(d->xg, 0); CdamFreeHost(d->ien, 0); }
    else { CdamFreeDevice(d->xg, 0); CdamFreeDevice(d->ien, 0); }
    CdamFreeHost(d, SIZE_OF(Mesh3DData));
}

void Mesh3DDataCopy(Mesh3DData* dst, Mesh3DData* src, MemCopyKind kind) {
    ASSERT(dst && src && "Invalid data");
    if (src == dst) return;
    ASSERT(dst->num_node == src->num_node && dst->num_tet == src->num_tet && dst->num_prism == src->num_prism &&
           dst->num_hex == src->num_hex);
    size_t elem = (size_t)src->num_tet * 4 + (size_t)src->num_prism * 6 + (size_t)src->num_hex * 8;
    HIPGUARD(hipMemcpy(dst->xg, src->xg, (size_t)src->num_node * 3 * sizeof(f64), kind));
    HIPGUARD(hipMemcpy(dst->ien, src->ien, elem * sizeof(index_type), kind));
}

Mesh3D* Mesh3DCreate(index_type nn, index_type nt, index_type np, index_type nh) {
    Mesh3D* m = (Mesh3D*)CdamMallocHost(SIZE_OF(Mesh3D));
    memset(m, 0, sizeof *m);
    m->num_node = nn; m->num_tet = nt; m->num_prism = np; m->num_hex = nh;
    m->host = Mesh3DDataCreateHost(nn, nt, np, nh);
    m->device = Mesh3DDataCreateDevice(nn, nt, np, nh);
    MeshExt* x = (MeshExt*)CdamMallocHost(SIZE_OF(MeshExt));
    memset(x, 0, sizeof *x);
    x->face_group = -1;
    x->cfg = *DflAsmDefaults();
    m->ext = x;
    return m;
}

void Mesh3DSetBound(Mesh3D* m, index_type nb, const index_type* node_offset, const index_type* node,
                    const index_type* elem_offset, const index_type* f2e, const index_type* forn) {
    m->num_bound = nb;
    m->bound_node_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nb + 1) * 2);
    m->bound_elem_offset = m->bound_node_offset + nb + 1;
    memcpy(m->bound_node_offset, node_offset, sizeof(index_type) * (size_t)(nb + 1));
    memcpy(m->bound_elem_offset, elem_offset, sizeof(index_type) * (size_t)(nb + 1));
    index_type nbn = node_offset[nb], nf = elem_offset[nb];
    m->bound_node = (index_type*)CdamMallocDevice(SIZE_OF(index_type) * ((ptrdiff_t)nbn + (ptrdiff_t)nf * 2));
    m->bound_f2e = m->bound_node + nbn;
    m->bound_forn = m->bound_f2e + nf;
    HIPGUARD(hipMemcpy(m->bound_node, node, sizeof(index_type) * (size_t)nbn, H2D));
    HIPGUARD(hipMemcpy(m->bound_f2e, f2e, sizeof(index_type) * (size_t)nf, H2D));
    HIPGUARD(hipMemcpy(m->bound_forn, forn, sizeof(index_type) * (size_t)nf, H2D));
    MeshExt* x = (MeshExt*)m->ext;
    x->h_f2e = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (ptrdiff_t)(nf > 0 ? nf : 1));
    memcpy(x->h_f2e, f2e, sizeof(index_type) * (size_t)nf);
}

void Mesh3DDestroy(Mesh3D* m) {
    if (!m) return;
    MeshExt* x = (MeshExt*)m->ext;
    Mesh3DDataDestroy(m->host);
    Mesh3DDataDestroy(m->device);
    if (m->bound_node_offset) CdamFreeHost(m->bound_node_offset, 0);
    CdamFreeDevice(m->bound_node, 0);
    if (m->batch_offset) CdamFreeHost(m->batch_offset, 0);
    CdamFreeDevice(m->batch_ind, 0);
    CdamFreeDevice(m->color, 0);
    if (x) {
        CdamFreeDevice(x->ien_b, 0);
        CdamFreeDevice(x->nzmap_b, 0);
        DflFreePatchSchedule(x->patch);
        DflFreeRowPatchSchedule(x->rowpatch);
        DflFreeSlotPatchSchedule(x->slotpatch);
        DflFreeRhsPatchSchedule(x->rhspatch);
        DflFreeFlowWork(x->flow);
        CdamFreeDevice(x->egeo_b, 0);
        if (x->h_sched_elem) CdamFreeHost(x->h_sched_elem, 0);
        CdamFreeDevice(x->nodep, 0);
        CdamFreeDevice(x->nodexu, 0);
        CdamFreeDevice(x->Fp, 0);
        DflMeshFreeFaceLists(x);
        if (x->sched_offset) CdamFreeHost(x->sched_offset, 0);
        if (x->h_f2e) CdamFreeHost(x->h_f2e, 0);
        CdamFreeHost(x, SIZE_OF(MeshExt));
    }
    CdamFreeHost(m, SIZE_OF(Mesh3D));
}

void Mesh3DUpdateHost(Mesh3D* m) { Mesh3DDataCopy(m->host, m->device, D2H); }
void Mesh3DUpdateDevice(Mesh3D* m) { Mesh3DDataCopy(m->device, m->host, H2D); }

/* ColorMeshTet, src/color.c:14-57: V2E map, XORWOW priorities, JPL rounds */
void ColorMeshTet(const Mesh3D* mesh, index_type max_color_len, color_t* color) {
    const Mesh3DData* dev = Mesh3DDevice(mesh);
    const index_type* ien = dev->ien;
    index_type T = dev->num_tet, N = dev->num_node;
    UNUSED(max_color_len);
    index_type* row = (index_type*)CdamMallocDevice(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    GenerateV2EMapRowTetGPU(ien, T, N, row);
    index_type nnz = 0;
    HIPGUARD(hipMemcpy(&nnz, row + N, sizeof nnz, D2H));
    index_type* col = (index_type*)CdamMallocDevice((ptrdiff_t)nnz * SIZE_OF(index_type));
    GenerateV2EMapColTetGPU(ien, T, N, row, col);
    ColorElementJPLTetGPU(ien, row, col, MAX_COLOR, color, T);
    CdamFreeDevice(row, 0);
    CdamFreeDevice(col, 0);
}

color_t GetMaxColor(const color_t* color, index_type n) {
    color_t mc = 0;
    GetMaxColorGPU(color, n, &mc);
    return mc;
}

/* connectivity sanity before any kernel indexes node arrays with it (an out-of-range vertex id would be a device fault,
 * a degenerate tet a division by zero in the geometry): host pass over the host copy, once per coloring */
static void mesh_validate(const Mesh3D* mesh) {
    const Mesh3DData* h = mesh->host;
    if (!h || !h->ien) return;
    const index_type N = mesh->num_node, T = mesh->num_tet;
    for (index_type e = 0; e < T; ++e) {
        const index_type* nd = h->ien + (size_t)e * 4;
        for (int a = 0; a < 4; ++a)
            if (nd[a] < 0 || nd[a] >= N) {
                fprintf(stderr, "dedflow: tet %d has vertex id %d outside [0, %d)\n", (int)e, (int)nd[a], (int)N);
                ASSERT(FALSE && "mesh connectivity out of range");
                abort();
            }
        if (nd[0] == nd[1] || nd[0] == nd[2] || nd[0] == nd[3] || nd[1] == nd[2] || nd[1] == nd[3] || nd[2] == nd[3]) {
            fprintf(stderr, "dedflow: tet %d repeats a vertex (%d %d %d %d)\n", (int)e, (int)nd[0], (int)nd[1], (int)nd[2], (int)nd[3]);
            ASSERT(FALSE && "degenerate tet");
            abort();
        }
    }
}

void Mesh3DColor(Mesh3D* mesh) {
    index_type T = mesh->num_tet;
    mesh_validate(mesh);
    if (!mesh->color) mesh->color = (color_t*)CdamMallocDevice((ptrdiff_t)T * SIZE_OF(color_t));
    ColorMeshTet(mesh, MAX_COLOR, mesh->color);
    mesh->num_color = GetMaxColor(mesh->color, T) + 1;
}

/* Execution schedule of the assembly kernels (MeshExt.cfg.sched_mode, include/dedflow.h: DflSetAssemblySchedule):
 * 0: launches follow the reference's JPL color batches one to one (same summation order as
 *    the reference inside every matrix / RHS entry);
 * 1: compact schedule below.  Results differ from mode 0 by summation order only;
 * 2-4: patch schedules (patch.c, rowpatch.c, slotpatch.c), built on first use by host/assemble.c. */

/* JPL colors one independent set of local maxima per round, i.e. ~140 colors of ~T/140 tets
 * for a tet mesh whose conflict graph needs ~30.  A 77k-tet launch is 4.4 waves per SIMD for
 * the 4-lane RHS kernel: latency-bound.  The coloring stays exactly the reference's
 * (mesh->color / batch_offset / batch_ind are bit-identical and remain the public result);
 * what the assembly kernels EXECUTE is a second, balanced greedy coloring with ~4x fewer and
 * ~4x larger conflict-free launches.  First-fit over node bit masks, least-loaded free class. */
/* FALSE (nothing allocated) when the mesh needs more than 64 classes -- a vertex shared by more than 64 tets, which the
 * reference's own limits allow (a row of <= 64 nonzeros, csr.c:10, bounds a closed fan at 2 * 63 - 4 = 122 tets): the caller
 * then executes the reference's JPL color batches instead (they go up to 256 colors). */
static b32 build_compact_schedule(Mesh3D* mesh, MeshExt* x) {
    const index_type T = mesh->num_tet, N = mesh->num_node;
    const index_type* ien = mesh->host->ien;
    u64* node_mask = (u64*)CdamMallocHost((ptrdiff_t)N * (ptrdiff_t)sizeof(u64));
    memset(node_mask, 0, (size_t)N * sizeof(u64));
    u8* cls = (u8*)CdamMallocHost((ptrdiff_t)T);
    index_type count[64];
    memset(count, 0, sizeof count);
    int nopen = 1;
    for (index_type e = 0; e < T; ++e) {
        const index_type* nd = ien + (size_t)e * 4;
        const u64 used = node_mask[nd[0]] | node_mask[nd[1]] | node_mask[nd[2]] | node_mask[nd[3]];
        int best = -1;
        for (int c = 0; c < nopen; ++c)
            if (!((used >> c) & 1ULL) && (best < 0 || count[c] < count[best])) best = c;
        if (best < 0) {
            if (nopen == 64) {
                fprintf(stderr, "dedflow: the compact assembly schedule needs more than 64 conflict-free classes (a vertex of tet %d "
                                "is shared by more than 64 tets); executing the reference's color batches instead\n", e);
                CdamFreeHost(cls, 0);
                CdamFreeHost(node_mask, 0);
                return FALSE;
            }
            best = nopen++;
        }
        cls[e] = (u8)best;
        count[best]++;
        const u64 bit = 1ULL << best;
        node_mask[nd[0]] |= bit; node_mask[nd[1]] |= bit; node_mask[nd[2]] |= bit; node_mask[nd[3]] |= bit;
    }
    x->sched_num = nopen;
    x->sched_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nopen + 1));
    x->sched_offset[0] = 0;
    for (int c = 0; c < nopen; ++c) x->sched_offset[c + 1] = x->sched_offset[c] + count[c];
    index_type* ind = (index_type*)CdamMallocHost((ptrdiff_t)T * SIZE_OF(index_type));
    index_type cur[64];
    memcpy(cur, x->sched_offset, sizeof(index_type) * (size_t)nopen);
    for (index_type e = 0; e < T; ++e) ind[cur[cls[e]]++] = e; /* ascending element id inside a class */
    index_type* d_ind = (index_type*)CdamMallocDevice((ptrdiff_t)T * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(d_ind, ind, sizeof(index_type) * (size_t)T, H2D));
    dfl_gather_ien(T, mesh->device->ien, d_ind, x->ien_b, DflStream());
    HIPGUARD(hipStreamSynchronize(DflStream()));
    CdamFreeDevice(d_ind, 0);
    x->h_sched_elem = ind; /* schedule position -> element id, kept for the row-owner schedule builder */
    CdamFreeHost(cls, 0);
    CdamFreeHost(node_mask, 0);
    return TRUE;
}

/* Mesh3DGenerateColorBatch, src/Mesh.c:165-206.  The per-color count + copy_if passes
 * are one stable sort by color; additionally the connectivity is re-laid out in batch
 * order so that each color's launch streams its ien (and nz map) contiguously. */
void Mesh3DGenerateColorBatch(Mesh3D* mesh) {
    index_type T = mesh->num_tet;
    MeshExt* x = (MeshExt*)mesh->ext;
    if (!T) return;
    Mesh3DColor(mesh);
    index_type nc = mesh->num_color;
    mesh->num_batch = nc;
    mesh->batch_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nc + 1));
    mesh->batch_ind = (index_type*)CdamMallocDevice((ptrdiff_t)T * SIZE_OF(index_type));
    dfl_color_batches(mesh->color, T, nc, mesh->batch_offset, mesh->batch_ind);
    x->ien_b = (index_type*)CdamMallocDevice((ptrdiff_t)T * 4 * SIZE_OF(index_type));
    if (x->sched_offset) CdamFreeHost(x->sched_offset, 0);
    if (x->h_sched_elem) CdamFreeHost(x->h_sched_elem, 0);
    x->h_sched_elem = NULL;
    if (x->cfg.sched_mode == 0 || !build_compact_schedule(mesh, x)) {
        /* execution schedule == the reference's JPL color batches (asked for, or the compact schedule refused the mesh: the
           colored kernels of schedule 1 then run one launch per reference color; schedules 2-4 are unaffected) */
        x->sched_num = nc;
        x->sched_offset = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (nc + 1));
        memcpy(x->sched_offset, mesh->batch_offset, sizeof(index_type) * (size_t)(nc + 1));
        dfl_gather_ien(T, mesh->device->ien, mesh->batch_ind, x->ien_b, DflStream());
        x->h_sched_elem = (index_type*)CdamMallocHost((ptrdiff_t)T * SIZE_OF(index_type));
        HIPGUARD(hipMemcpy(x->h_sched_elem, mesh->batch_ind, sizeof(index_type) * (size_t)T, D2H));
    }
    HIPGUARD(hipStreamSynchronize(DflStream()));
}

/* faces of one boundary group ordered by the color of their parent tet
 * (replaces the per-color SetupMaskKernel passes, src/assemble.cu:1916-1945) */
static int cmp_pair(const void* a, const void* b) {
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}
/* key-sorted (key, entry) pairs -> distinct keys, CSR offsets, entries; uploaded */
static index_type upload_groups(int64_t* pairs, index_type n, index_type** d_key, index_type** d_off, index_type** d_ent) {
    qsort(pairs, (size_t)n, sizeof(int64_t), cmp_pair); /* key in the high word, entry in the low word: entries ascend */
    index_type* key = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (ptrdiff_t)(n > 0 ? n : 1));
    index_type* off = (index_type*)CdamMallocHost(SIZE_OF(index_type) * ((ptrdiff_t)n + 1));
    index_type* ent = (index_type*)CdamMallocHost(SIZE_OF(index_type) * (ptrdiff_t)(n > 0 ? n : 1));
    index_type nk = 0;
    for (index_type i = 0; i < n; ++i) {
        const index_type k = (index_type)(pairs[i] >> 32);
        if (i == 0 || k != key[nk - 1]) { key[nk] = k; off[nk] = i; ++nk; }
        ent[i] = (index_type)(pairs[i] & 0xffffffffLL);
    }
    off[nk] = n;
    *d_key = (index_type*)CdamMallocDevice(SIZE_OF(index_type) * (ptrdiff_t)(nk > 0 ? nk : 1));
    *d_off = (index_type*)CdamMallocDevice(SIZE_OF(index_type) * ((ptrdiff_t)nk + 1));
    *d_ent = (index_type*)CdamMallocDevice(SIZE_OF(index_type) * (ptrdiff_t)(n > 0 ? n : 1));
    HIPGUARD(hipMemcpy(*d_key, key, sizeof(index_type) * (size_t)nk, H2D));
    HIPGUARD(hipMemcpy(*d_off, off, sizeof(index_type) * ((size_t)nk + 1), H2D));
    HIPGUARD(hipMemcpy(*d_ent, ent, sizeof(index_type) * (size_t)n, H2D));
    CdamFreeHost(ent, 0); CdamFreeHost(off, 0); CdamFreeHost(key, 0);
    return nk;
}

static void face_free_nz_lists(MeshExt* x) {
    CdamFreeDevice(x->face_nz, 0); CdamFreeDevice(x->face_nz_off, 0); CdamFreeDevice(x->face_nz_ent, 0);
    CdamFreeDevice(x->face_pJ, 0);
    x->face_nz = x->face_nz_off = x->face_nz_ent = NULL;
    x->face_pJ = NULL;
    x->face_attr = NULL;
    x->face_nnz = 0;
}
void DflMeshFreeFaceLists(MeshExt* x) {
    face_free_nz_lists(x);
    CdamFreeDevice(x->face_node, 0); CdamFreeDevice(x->face_node_off, 0); CdamFreeDevice(x->face_node_ent, 0);
    CdamFreeDevice(x->face_pF, 0);
    x->face_node = x->face_node_off = x->face_node_ent = NULL;
    x->face_pF = NULL;
    x->face_group = -1;
}

/* Weak-BC faces of one boundary group (the reference scatters them once per ELEMENT color: 141 masked passes at 10M
 * tets for 28k faces).  Here ONE launch parks every face's contributions and a second sums them per node / per nodal
 * nonzero in ascending face order -- deterministic, no conflict classes.  This builds the node lists. */
void DflMeshPrepareFaces(Mesh3D* mesh, index_type group) {
    MeshExt* x = (MeshExt*)mesh->ext;
    if (x->face_group == group) return;
    DflMeshFreeFaceLists(x);
    const index_type nf = Mesh3DBoundNumElem(mesh, group);
    const index_type lo = mesh->bound_elem_offset[group];
    const index_type* h_ien = mesh->host->ien;
    int64_t* pairs = (int64_t*)CdamMallocHost((ptrdiff_t)(nf > 0 ? nf : 1) * 4 * (ptrdiff_t)sizeof(int64_t));
    for (index_type f = 0; f < nf; ++f) {
        const index_type* nd = h_ien + (size_t)x->h_f2e[lo + f] * 4;
        for (int a = 0; a < 4; ++a) pairs[(size_t)f * 4 + a] = ((int64_t)nd[a] << 32) | (int64_t)(f * 4 + a);
    }
    x->face_nn = upload_groups(pairs, nf * 4, &x->face_node, &x->face_node_off, &x->face_node_ent);
    CdamFreeHost(pairs, 0);
    x->face_pF = (f64*)CdamMallocDevice((ptrdiff_t)(nf > 0 ? nf : 1) * 16 * SIZE_OF(f64));
    x->face_nf = nf;
    x->face_group = group;
}

/* nonzero lists of the same faces for a given nodal pattern (built when a Jacobian is first assembled with it) */
void DflMeshPrepareFaceNonzeros(Mesh3D* mesh, index_type group, const CSRAttr* spy) {
    MeshExt* x = (MeshExt*)mesh->ext;
    DflMeshPrepareFaces(mesh, group);
    if (x->face_attr == spy && x->face_nz) return;
    face_free_nz_lists(x);
    const index_type nf = x->face_nf, N = mesh->num_node;
    const index_type lo = mesh->bound_elem_offset[group];
    const index_type* h_ien = mesh->host->ien;
    index_type* rp = (index_type*)CdamMallocHost(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    index_type* ci = (index_type*)CdamMallocHost((ptrdiff_t)spy->nnz * SIZE_OF(index_type));
    HIPGUARD(hipMemcpy(rp, spy->row_ptr, sizeof(index_type) * ((size_t)N + 1), D2H));
    HIPGUARD(hipMemcpy(ci, spy->col_ind, sizeof(index_type) * (size_t)spy->nnz, D2H));
    int64_t* pairs = (int64_t*)CdamMallocHost((ptrdiff_t)(nf > 0 ? nf : 1) * 16 * (ptrdiff_t)sizeof(int64_t));
    for (index_type f = 0; f < nf; ++f) {
        const index_type* nd = h_ien + (size_t)x->h_f2e[lo + f] * 4;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                index_type l = rp[nd[a]], h = rp[nd[a] + 1] - 1; /* ascending col_ind: binary search */
                while (l < h) {
                    index_type mid = (l + h) >> 1;
                    if (ci[mid] < nd[b]) l = mid + 1; else h = mid;
                }
                pairs[(size_t)f * 16 + a * 4 + b] = ((int64_t)l << 32) | (int64_t)(f * 16 + a * 4 + b);
            }
    }
    x->face_nnz = upload_groups(pairs, nf * 16, &x->face_nz, &x->face_nz_off, &x->face_nz_ent);
    CdamFreeHost(pairs, 0); CdamFreeHost(ci, 0); CdamFreeHost(rp, 0);
    x->face_pJ = (f64*)CdamMallocDevice((ptrdiff_t)(nf > 0 ? nf : 1) * 256 * SIZE_OF(f64));
    x->face_attr = spy;
}

/* ---- csr.h ------------------------------------------------------------------------ */
/* csr_impl.h:7-9 */
void CSRAttrGetNZIndBatchedGPU(const CSRAttr* attr, csr_index_type batch_size, const index_type* row, const index_type* col,
                               csr_index_type* ind) {
    dfl_csr_find_nz(batch_size, attr->row_ptr, attr->col_ind, row, col, ind, DflStream());
}

CSRAttr* CSRAttrCreate(const Mesh3D* mesh) {
    CSRAttr* attr = (CSRAttr*)CdamMallocHost(SIZE_OF(CSRAttr));
    memset(attr, 0, sizeof *attr);
    const Mesh3DData* dev = Mesh3DDevice(mesh);
    index_type N = dev->num_node, T = dev->num_tet;
    hipStream_t s = DflStream();
    attr->num_row = N;
    attr->num_col = N;
    /* vertex->element map, then sorted unique neighbour lists per node */
    index_type* vrow = (index_type*)CdamMallocDevice(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    GenerateV2EMapRowTetGPU(dev->ien, T, N, vrow);
    index_type vnnz = 0;
    HIPGUARD(hipMemcpy(&vnnz, vrow + N, sizeof vnnz, D2H));
    index_type* vcol = (index_type*)CdamMallocDevice((ptrdiff_t)vnnz * SIZE_OF(index_type));
    GenerateV2EMapColTetGPU(dev->ien, T, N, vrow, vcol);
    index_type* len = (index_type*)CdamMallocDevice(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    index_type* d_over = (index_type*)CdamMallocDevice(SIZE_OF(index_type));
    dfl_pattern_count(N, dev->ien, vrow, vcol, len, d_over, s);
    attr->row_ptr = (index_type*)CdamMallocDevice(((ptrdiff_t)N + 1) * SIZE_OF(index_type));
    int64_t tb = dfl_scan_temp_bytes(N);
    void* tmp = CdamMallocDevice((ptrdiff_t)tb);
    dfl_exclusive_scan_i32(N, len, attr->row_ptr, tmp, tb, s);
    index_type over = 0, nnz = 0;
    HIPGUARD(hipMemcpyAsync(&over, d_over, sizeof over, D2H, s));
    HIPGUARD(hipMemcpyAsync(&nnz, attr->row_ptr + N, sizeof nnz, D2H, s));
    HIPGUARD(hipStreamSynchronize(s));
    ASSERT(!over && "CSRHashMapPush: row overflow"); /* csr.c:63 */
    attr->nnz = nnz;
    attr->col_ind = (index_type*)CdamMallocDevice((ptrdiff_t)nnz * SIZE_OF(index_type));
    dfl_pattern_fill(N, dev->ien, vrow, vcol, attr->row_ptr, attr->col_ind, s);
    HIPGUARD(hipStreamSynchronize(s));
    CdamFreeDevice(tmp, 0);
    CdamFreeDevice(len, 0);
    CdamFreeDevice(d_over, 0);
    CdamFreeDevice(vrow, 0);
    CdamFreeDevice(vcol, 0);
    return attr;
}

CSRAttr* CSRAttrCreateBlock(const CSRAttr* attr, csr_index_type br, csr_index_type bc) {
    CSRAttr* na = (CSRAttr*)CdamMallocHost(SIZE_OF(CSRAttr));
    memset(na, 0, sizeof *na);
    na->num_row = attr->num_row * br;
    na->num_col = attr->num_col * bc;
    na->nnz = attr->nnz * br * bc;
    na->parent = attr;
    na->row_ptr = (index_type*)CdamMallocDevice(((ptrdiff_t)na->num_row + 1) * SIZE_OF(index_type));
    na->col_ind = (index_type*)CdamMallocDevice((ptrdiff_t)na->nnz * SIZE_OF(index_type));
    if (br == 1 && bc == 1) {
        HIPGUARD(hipMemcpy(na->row_ptr, attr->row_ptr, sizeof(index_type) * (size_t)(attr->num_row + 1), D2D));
        HIPGUARD(hipMemcpy(na->col_ind, attr->col_ind, sizeof(index_type) * (size_t)attr->nnz, D2D));
    } else {
        dfl_csr_expand(attr->num_row, attr->row_ptr, attr->col_ind, br, bc, na->row_ptr, na->col_ind, DflStream());
        HIPGUARD(hipStreamSynchronize(DflStream()));
    }
    return na;
}

/* ExpandCSRByBlockSize, csr_impl.cu:126-156: fills the (already allocated) expanded pattern; last row_ptr entry written (Q3) */
void ExpandCSRByBlockSize(const CSRAttr* attr, CSRAttr* new_attr, csr_index_type block_size[2]) {
    dfl_csr_expand(attr->num_row, attr->row_ptr, attr->col_ind, block_size[0], block_size[1], new_attr->row_ptr, new_attr->col_ind,
                   DflStream());
}

/* csr.c:226-238 */
index_type CSRAttrLength(CSRAttr* attr, csr_index_type row) {
    index_type rp[2];
    HIPGUARD(hipMemcpy(rp, attr->row_ptr + row, 2 * sizeof(index_type), D2H));
    return rp[1] - rp[0];
}
csr_index_type* CSRAttrRow(CSRAttr* attr, csr_index_type row) {
    index_type start = 0;
    HIPGUARD(hipMemcpy(&start, attr->row_ptr + row, sizeof(index_type), D2H));
    return attr->col_ind + start;
}
void CSRAttrGetNonzeroIndBatched(const CSRAttr* attr, csr_index_type batch_size, const index_type* row, const index_type* col,
                                 index_type* ind) {
    CSRAttrGetNZIndBatchedGPU(attr, batch_size, row, col, ind);
}

void CSRAttrDestroy(CSRAttr* attr) {
    if (!attr) return;
    CdamFreeDevice(attr->row_ptr, 0);
    CdamFreeDevice(attr->col_ind, 0);
    CdamFreeHost(attr, SIZE_OF(CSRAttr));
}
