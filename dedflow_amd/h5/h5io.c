/* HDF5 mesh / solution formats of the reference, as a separate
 * library (libdedflow_h5.so) so that the core library carries no HDF5 dependency.
 *
 * API of src/h5util.h:24-58 (open/close, dataset size, 1-D read/write with automatic group
 * creation) and Mesh3DCreateH5 (src/Mesh.c:78-107 + ReadBoundFromH5Private :12-59 +
 * Mesh3DDataCreateH5, src/MeshData.c:57-109).  Schema (written by tools/mesh_convert.py:116-126):
 *   <grp>/xg f64[3N]; <grp>/ien/tet i32[4T] (prism/hex optional, absent => size 0);
 *   <grp>/bound/{node_offset[nb+1], node, elem_offset[nb+1], ien(3/face), f2e, forn}.
 * Solution files sol.<k>.h5 (src/main.c:521-532, 571-590): u[3N], p, phi, T, du[3N], dphi, dT. */
#include <hdf5.h>
#include <string.h>
#include "dedflow.h"

struct H5FileInfo {
    char filename[256];
    hid_t file_id;
};

H5FileInfo* H5OpenFile(const char* filename, const char* mode) {
    H5FileInfo* f = (H5FileInfo*)CdamMallocHost(sizeof(H5FileInfo));
    memset(f, 0, sizeof *f);
    if (strcmp(mode, "r") == 0) f->file_id = H5Fopen(filename, H5F_ACC_RDONLY, H5P_DEFAULT);
    else if (strcmp(mode, "w") == 0) f->file_id = H5Fcreate(filename, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    else if (strcmp(mode, "a") == 0) f->file_id = H5Fopen(filename, H5F_ACC_RDWR, H5P_DEFAULT);
    else ASSERT(0 && "H5OpenFile: Invalid mode!");
    ASSERT(f->file_id >= 0 && "H5OpenFile: Failed to open file!");
    strncpy(f->filename, filename, sizeof(f->filename) - 1);
    return f;
}

void H5CloseFile(H5FileInfo* f) {
    H5Fclose(f->file_id);
    CdamFreeHost(f, sizeof(H5FileInfo));
}

/* file access intent, h5util.c:43-57 */
b32 H5FileIsWritable(H5FileInfo* f) {
    unsigned intent = 0;
    H5Fget_intent(f->file_id, &intent);
    return (intent & (H5F_ACC_RDWR | H5F_ACC_TRUNC)) != 0;
}
b32 H5FileIsReadable(H5FileInfo* f) {
    unsigned intent = 0;
    H5Fget_intent(f->file_id, &intent);
    return intent == H5F_ACC_RDONLY || (intent & H5F_ACC_RDWR) != 0;
}

static int link_exists(hid_t file, const char* path) {
    /* H5Lexists needs every intermediate group to exist */
    char buf[512];
    size_t n = strlen(path);
    if (n >= sizeof buf) return 0;
    for (size_t i = 1; i <= n; ++i) {
        if (path[i] == '/' || path[i] == '\0') {
            memcpy(buf, path, i);
            buf[i] = '\0';
            if (H5Lexists(file, buf, H5P_DEFAULT) <= 0) return 0;
        }
    }
    return 1;
}

b32 H5DatasetExist(H5FileInfo* f, const char* name) {
    if (!link_exists(f->file_id, name)) return FALSE;
    H5O_info_t info;
    return H5Oget_info_by_name(f->file_id, name, &info, H5P_DEFAULT) >= 0 && info.type == H5O_TYPE_DATASET;
}

void H5GetDatasetSize(H5FileInfo* f, const char* name, index_type* size) {
    if (!H5DatasetExist(f, name)) { *size = 0; return; } /* h5util.c: missing dataset => 0 */
    hid_t d = H5Dopen2(f->file_id, name, H5P_DEFAULT), s = H5Dget_space(d);
    hsize_t n = 0;
    ASSERT(H5Sget_simple_extent_ndims(s) == 1 && "All arrays are flattened into 1D.");
    H5Sget_simple_extent_dims(s, &n, NULL);
    *size = (index_type)n;
    H5Sclose(s);
    H5Dclose(d);
}

static void read_ds(H5FileInfo* f, const char* name, hid_t mem_type, void* data) {
    hid_t d = H5Dopen2(f->file_id, name, H5P_DEFAULT);
    ASSERT(d >= 0 && "H5ReadDataset: no such dataset");
    H5Dread(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    H5Dclose(d);
}
void H5ReadDatasetf64(H5FileInfo* f, const char* name, f64* data) { read_ds(f, name, H5T_NATIVE_DOUBLE, data); }
void H5ReadDatasetInd(H5FileInfo* f, const char* name, index_type* data) { read_ds(f, name, H5T_NATIVE_INT32, data); }

static void write_ds(H5FileInfo* f, const char* name, hid_t file_type, hid_t mem_type, index_type len, const void* data) {
    /* create missing groups along the path (h5util.c does the same) */
    char buf[512];
    size_t n = strlen(name);
    ASSERT(n < sizeof buf);
    for (size_t i = 1; i < n; ++i) {
        if (name[i] != '/') continue;
        memcpy(buf, name, i);
        buf[i] = '\0';
        if (H5Lexists(f->file_id, buf, H5P_DEFAULT) <= 0) {
            hid_t g = H5Gcreate2(f->file_id, buf, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            H5Gclose(g);
        }
    }
    hsize_t dims = (hsize_t)len;
    hid_t s = H5Screate_simple(1, &dims, NULL);
    hid_t d = H5Dcreate2(f->file_id, name, file_type, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    ASSERT(d >= 0 && "H5WriteDataset: cannot create dataset");
    if (len > 0) H5Dwrite(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    H5Dclose(d);
    H5Sclose(s);
}
void H5WriteDatasetf64(H5FileInfo* f, const char* name, index_type len, const f64* data) {
    write_ds(f, name, H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, len, data);
}
void H5WriteDatasetInd(H5FileInfo* f, const char* name, index_type len, const index_type* data) {
    write_ds(f, name, H5T_STD_I32LE, H5T_NATIVE_INT32, len, data);
}

/* Mesh3DDataCreateH5, MeshData.c:57-109: host-side coordinates + connectivity of <group> (prism / hex blocks are read when
 * present, although nothing on the path uses them) */
Mesh3DData* Mesh3DDataCreateH5(H5FileInfo* f, const char* group) {
    char p[320];
    index_type n3 = 0, t4 = 0, np6 = 0, nh8 = 0;
    ASSERT(f && H5FileIsReadable(f) && "Mesh3DDataCreateH5: Invalid file");
    ASSERT(group && strlen(group) < 192 && "Mesh3DDateCreateH5: Invalid group name.");
    snprintf(p, sizeof p, "%s/xg", group); H5GetDatasetSize(f, p, &n3);
    snprintf(p, sizeof p, "%s/ien/tet", group); H5GetDatasetSize(f, p, &t4);
    snprintf(p, sizeof p, "%s/ien/prism", group); H5GetDatasetSize(f, p, &np6);
    snprintf(p, sizeof p, "%s/ien/hex", group); H5GetDatasetSize(f, p, &nh8);
    ASSERT(n3 % 3 == 0 && t4 % 4 == 0 && np6 % 6 == 0 && nh8 % 8 == 0);
    Mesh3DData* d = Mesh3DDataCreateHost(n3 / 3, t4 / 4, np6 / 6, nh8 / 8);
    if (n3) { snprintf(p, sizeof p, "%s/xg", group); H5ReadDatasetf64(f, p, d->xg); }
    if (t4) { snprintf(p, sizeof p, "%s/ien/tet", group); H5ReadDatasetInd(f, p, d->ien); }
    if (np6) { snprintf(p, sizeof p, "%s/ien/prism", group); H5ReadDatasetInd(f, p, d->ien + t4); }
    if (nh8) { snprintf(p, sizeof p, "%s/ien/hex", group); H5ReadDatasetInd(f, p, d->ien + t4 + np6); }
    return d;
}

/* ArrayLoad / ArraySave (Array.c:242-261) and ParticleContextLoad / Save (Particle.c:66-103): <group>/{coord,vel,acc} */
void ArrayLoad(Array* a, H5FileInfo* f, const char* name) {
    ASSERT(a && f && name && "ArrayLoad: NULL pointer");
    ASSERT(a->is_host && "ArrayLoad: Array must be host type");
    ASSERT(H5FileIsReadable(f) && "ArrayLoad: File is not readable");
    ASSERT(H5DatasetExist(f, name) && "ArrayLoad: Dataset does not exist");
    index_type len = 0;
    H5GetDatasetSize(f, name, &len);
    ASSERT(len == ArrayLen(a) && "ArrayLoad: Array length mismatch");
    H5ReadDatasetf64(f, name, ArrayData(a));
}
void ArraySave(const Array* a, H5FileInfo* f, const char* name) {
    ASSERT(a && f && name && "ArraySave: NULL pointer");
    ASSERT(a->is_host && "ArraySave: Array must be host type");
    ASSERT(H5FileIsWritable(f) && "ArraySave: File is not writable");
    H5WriteDatasetf64(f, name, ArrayLen(a), ArrayData(a));
}
/* FieldLoad / FieldSave (Field.c:47-57): the dataset holds the host copy; a load refreshes the device copy */
void FieldLoad(Field* f, H5FileInfo* h5f, const char* name) {
    ASSERT(f && h5f && name && "FieldLoad: NULL pointer.");
    ArrayLoad(FieldHost(f), h5f, name);
    ArrayCopy(FieldDevice(f), FieldHost(f), H2D);
}
void FieldSave(const Field* f, H5FileInfo* h5f, const char* name) {
    ASSERT(f && h5f && name && "FieldSave: NULL pointer.");
    ArraySave(FieldHost(f), h5f, name);
}
void ParticleContextLoad(ParticleContext* ctx, H5FileInfo* f, const char* group) {
    char path[256];
    static const char* part[3] = {"coord", "vel", "acc"};
    ASSERT(ctx && group && strlen(group) < 192 && "ParticleContextLoad: bad arguments");
    for (int k = 0; k < 3; ++k) {
        snprintf(path, sizeof path, "%s/%s", group, part[k]);
        ArrayLoad(ctx->h_arr[k], f, path);
        ArrayCopy(ctx->d_arr[k], ctx->h_arr[k], H2D);
    }
}
void ParticleContextSave(const ParticleContext* ctx, H5FileInfo* f, const char* group) {
    char path[256];
    static const char* part[3] = {"coord", "vel", "acc"};
    ASSERT(ctx && group && strlen(group) < 192 && "ParticleContextSave: bad arguments");
    for (int k = 0; k < 3; ++k) {
        snprintf(path, sizeof path, "%s/%s", group, part[k]);
        ArraySave(ctx->h_arr[k], f, path);
    }
}

Mesh3D* Mesh3DCreateH5(H5FileInfo* f, const char* group) {
    char p[320];
    index_type n3 = 0, t4 = 0, np6 = 0, nh8 = 0, nb1 = 0;
    ASSERT(f && group && strlen(group) < 192);
    snprintf(p, sizeof p, "%s/xg", group); H5GetDatasetSize(f, p, &n3);
    snprintf(p, sizeof p, "%s/ien/tet", group); H5GetDatasetSize(f, p, &t4);
    snprintf(p, sizeof p, "%s/ien/prism", group); H5GetDatasetSize(f, p, &np6);
    snprintf(p, sizeof p, "%s/ien/hex", group); H5GetDatasetSize(f, p, &nh8);
    ASSERT(n3 % 3 == 0 && t4 % 4 == 0 && np6 % 6 == 0 && nh8 % 8 == 0);
    ASSERT(np6 == 0 && nh8 == 0 && "prism / hex elements are empty stubs in the reference (main.c:57-61)");
    Mesh3D* mesh = Mesh3DCreate(n3 / 3, t4 / 4, 0, 0);
    snprintf(p, sizeof p, "%s/xg", group); H5ReadDatasetf64(f, p, mesh->host->xg);
    snprintf(p, sizeof p, "%s/ien/tet", group); H5ReadDatasetInd(f, p, mesh->host->ien);
    Mesh3DUpdateDevice(mesh);
    snprintf(p, sizeof p, "%s/bound/node_offset", group); H5GetDatasetSize(f, p, &nb1);
    if (nb1 > 0) {
        index_type nb = nb1 - 1, nbn = 0, nf = 0;
        index_type* noff = (index_type*)malloc(sizeof(index_type) * (size_t)nb1);
        index_type* eoff = (index_type*)malloc(sizeof(index_type) * (size_t)nb1);
        H5ReadDatasetInd(f, p, noff);
        snprintf(p, sizeof p, "%s/bound/elem_offset", group); H5ReadDatasetInd(f, p, eoff);
        nbn = noff[nb]; nf = eoff[nb];
        index_type* node = (index_type*)malloc(sizeof(index_type) * (size_t)(nbn > 0 ? nbn : 1));
        index_type* f2e = (index_type*)malloc(sizeof(index_type) * (size_t)(nf > 0 ? nf : 1));
        index_type* forn = (index_type*)malloc(sizeof(index_type) * (size_t)(nf > 0 ? nf : 1));
        snprintf(p, sizeof p, "%s/bound/node", group); H5ReadDatasetInd(f, p, node);
        snprintf(p, sizeof p, "%s/bound/f2e", group); H5ReadDatasetInd(f, p, f2e);
        snprintf(p, sizeof p, "%s/bound/forn", group); H5ReadDatasetInd(f, p, forn);
        Mesh3DSetBound(mesh, nb, noff, node, eoff, f2e, forn);
        free(noff); free(eoff); free(node); free(f2e); free(forn);
    }
    return mesh;
}

/* writer for synthetic meshes in the same schema (the reference relies on tools/mesh_convert.py) */
void DflMeshWriteH5(H5FileInfo* f, const char* group, index_type N, index_type T, const f64* xg, const index_type* ien,
                    index_type nb, const index_type* node_offset, const index_type* node, const index_type* elem_offset,
                    const index_type* bien, const index_type* f2e, const index_type* forn) {
    char p[320];
    snprintf(p, sizeof p, "%s/xg", group); H5WriteDatasetf64(f, p, 3 * N, xg);
    snprintf(p, sizeof p, "%s/ien/tet", group); H5WriteDatasetInd(f, p, 4 * T, ien);
    snprintf(p, sizeof p, "%s/bound/node_offset", group); H5WriteDatasetInd(f, p, nb + 1, node_offset);
    snprintf(p, sizeof p, "%s/bound/node", group); H5WriteDatasetInd(f, p, node_offset[nb], node);
    snprintf(p, sizeof p, "%s/bound/elem_offset", group); H5WriteDatasetInd(f, p, nb + 1, elem_offset);
    snprintf(p, sizeof p, "%s/bound/ien", group); H5WriteDatasetInd(f, p, 3 * elem_offset[nb], bien);
    snprintf(p, sizeof p, "%s/bound/f2e", group); H5WriteDatasetInd(f, p, elem_offset[nb], f2e);
    snprintf(p, sizeof p, "%s/bound/forn", group); H5WriteDatasetInd(f, p, elem_offset[nb], forn);
}

/* sol.<k>.h5 writer of main.c:571-590: u,phi,T from wgold; p and the rates from dwgold (device vectors) */
void DflSolutionWriteH5(const char* filename, index_type N, const f64* d_wgold, const f64* d_dwgold) {
    f64* buf = (f64*)malloc(sizeof(f64) * (size_t)N * 6);
    H5FileInfo* f = H5OpenFile(filename, "w");
    HIPGUARD(hipMemcpy(buf, d_wgold, sizeof(f64) * (size_t)N * 6, D2H));
    H5WriteDatasetf64(f, "u", N * 3, buf);
    H5WriteDatasetf64(f, "phi", N, buf + (size_t)N * 4);
    H5WriteDatasetf64(f, "T", N, buf + (size_t)N * 5);
    HIPGUARD(hipMemcpy(buf, d_dwgold, sizeof(f64) * (size_t)N * 6, D2H));
    H5WriteDatasetf64(f, "du", N * 3, buf);
    H5WriteDatasetf64(f, "p", N, buf + (size_t)N * 3);
    H5WriteDatasetf64(f, "dphi", N, buf + (size_t)N * 4);
    H5WriteDatasetf64(f, "dT", N, buf + (size_t)N * 5);
    H5CloseFile(f);
    free(buf);
}

/* restart reader: the inverse of the writer (the reference's restart branch, main.c:480-503, reads phi/T into
 * the RATE slots by mistake -- not reproduced) */
void DflSolutionReadH5(const char* filename, index_type N, f64* d_wgold, f64* d_dwgold) {
    f64* buf = (f64*)calloc((size_t)N * 6, sizeof(f64));
    H5FileInfo* f = H5OpenFile(filename, "r");
    H5ReadDatasetf64(f, "u", buf);
    H5ReadDatasetf64(f, "phi", buf + (size_t)N * 4);
    H5ReadDatasetf64(f, "T", buf + (size_t)N * 5);
    HIPGUARD(hipMemcpy(d_wgold, buf, sizeof(f64) * (size_t)N * 6, H2D));
    memset(buf, 0, sizeof(f64) * (size_t)N * 6);
    H5ReadDatasetf64(f, "du", buf);
    H5ReadDatasetf64(f, "p", buf + (size_t)N * 3);
    H5ReadDatasetf64(f, "dphi", buf + (size_t)N * 4);
    H5ReadDatasetf64(f, "dT", buf + (size_t)N * 5);
    HIPGUARD(hipMemcpy(d_dwgold, buf, sizeof(f64) * (size_t)N * 6, H2D));
    H5CloseFile(f);
    free(buf);
}
