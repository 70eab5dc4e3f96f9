/* Array BLAS-1 wrappers and nodal fields: Array.h:24-36 / Array.c:83-238, Field.h:22-33 / Field.c:15-77.
 * Device arrays run the library's kernels (k_blas.hip) on the library stream; nothing here computes device data on the host. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

static void same_side(const Array* a, const Array* b, const char* who) {
    if (a->len != b->len || a->is_host != b->is_host) {
        fprintf(stderr, "%s: array length or memory-side mismatch\n", who);
        abort();
    }
}

void ArraySet(Array* a, f64 val) {
    ASSERT(a && "ArraySet: NULL pointer");
    if (a->is_host) {
        for (index_type i = 0; i < a->len; ++i) a->data[i] = val;
    } else {
        dfl_dset(a->len, val, a->data, DflStream());
        HIPGUARD(hipStreamSynchronize(DflStream()));
    }
}

void ArrayZero(Array* a) { ArrayScale(a, 0.0); }

void ArrayScale(Array* a, f64 val) {
    ASSERT(a && "ArrayScale: NULL pointer");
    if (a->is_host) {
        for (index_type i = 0; i < a->len; ++i) a->data[i] *= val;
    } else {
        dfl_dscal(a->len, val, a->data, DflStream());
        HIPGUARD(hipStreamSynchronize(DflStream()));
    }
}

/* (idx, val) pairs on the device for the scatter / gather kernels; SetAt keeps the last value of a repeated index */
typedef struct { index_type idx; index_type pos; } IdxPos;
static int by_idx_then_pos(const void* a, const void* b) {
    const IdxPos *x = (const IdxPos*)a, *y = (const IdxPos*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

void ArraySetAt(Array* a, index_type n, const index_type* idx, const f64* val) {
    ASSERT(a && idx && val && "ArraySetAt: NULL pointer");
    if (n <= 0) return;
    for (index_type i = 0; i < n; ++i) ASSERT(idx[i] >= 0 && idx[i] < a->len && "ArraySetAt: index out of range");
    if (a->is_host) {
        for (index_type i = 0; i < n; ++i) a->data[idx[i]] = val[i];
        return;
    }
    IdxPos* ip = (IdxPos*)malloc((size_t)n * sizeof *ip);
    index_type* uidx = (index_type*)malloc((size_t)n * sizeof *uidx);
    f64* uval = (f64*)malloc((size_t)n * sizeof *uval);
    ASSERT(ip && uidx && uval);
    for (index_type i = 0; i < n; ++i) { ip[i].idx = idx[i]; ip[i].pos = i; }
    qsort(ip, (size_t)n, sizeof *ip, by_idx_then_pos);
    index_type m = 0;
    for (index_type i = 0; i < n; ++i)
        if (i + 1 == n || ip[i + 1].idx != ip[i].idx) { uidx[m] = ip[i].idx; uval[m] = val[ip[i].pos]; ++m; }
    index_type* d_idx = (index_type*)CdamMallocDevice((ptrdiff_t)m * SIZE_OF(index_type));
    f64* d_val = (f64*)CdamMallocDevice((ptrdiff_t)m * SIZE_OF(f64));
    HIPGUARD(hipMemcpyAsync(d_idx, uidx, (size_t)m * sizeof *uidx, hipMemcpyHostToDevice, DflStream()));
    HIPGUARD(hipMemcpyAsync(d_val, uval, (size_t)m * sizeof *uval, hipMemcpyHostToDevice, DflStream()));
    dfl_scatter_idx(m, d_idx, d_val, a->data, DflStream());
    HIPGUARD(hipStreamSynchronize(DflStream()));
    CdamFreeDevice(d_idx, (ptrdiff_t)m * SIZE_OF(index_type));
    CdamFreeDevice(d_val, (ptrdiff_t)m * SIZE_OF(f64));
    free(ip); free(uidx); free(uval);
}

void ArrayGetAt(const Array* a, index_type n, const index_type* idx, f64* val) {
    ASSERT(a && idx && val && "ArrayAt: NULL pointer");
    if (n <= 0) return;
    for (index_type i = 0; i < n; ++i) ASSERT(idx[i] >= 0 && idx[i] < a->len && "ArrayAt: index out of range");
    if (a->is_host) {
        for (index_type i = 0; i < n; ++i) val[i] = a->data[idx[i]];
        return;
    }
    index_type* d_idx = (index_type*)CdamMallocDevice((ptrdiff_t)n * SIZE_OF(index_type));
    f64* d_val = (f64*)CdamMallocDevice((ptrdiff_t)n * SIZE_OF(f64));
    HIPGUARD(hipMemcpyAsync(d_idx, idx, (size_t)n * sizeof *idx, hipMemcpyHostToDevice, DflStream()));
    dfl_gather_idx(n, d_idx, a->data, d_val, DflStream());
    HIPGUARD(hipMemcpyAsync(val, d_val, (size_t)n * sizeof *val, hipMemcpyDeviceToHost, DflStream()));
    HIPGUARD(hipStreamSynchronize(DflStream()));
    CdamFreeDevice(d_idx, (ptrdiff_t)n * SIZE_OF(index_type));
    CdamFreeDevice(d_val, (ptrdiff_t)n * SIZE_OF(f64));
}
void ArrayAt(const Array* a, index_type n, const index_type* idx, f64* val) { ArrayGetAt(a, n, idx, val); }

/* one device reduction: result and two-stage workspace from the pool, read back before returning */
static f64 device_reduce(const Array* a, const Array* b) {
    const ptrdiff_t bytes = ((ptrdiff_t)dfl_reduce_work_size() + 1) * SIZE_OF(f64);
    f64* w = (f64*)CdamMallocDevice(bytes);
    f64 r = 0.0;
    if (b) dfl_ddot(a->len, a->data, b->data, w, w + 1, DflStream());
    else dfl_dnrm2(a->len, a->data, w, w + 1, DflStream());
    HIPGUARD(hipMemcpyAsync(&r, w, sizeof r, hipMemcpyDeviceToHost, DflStream()));
    HIPGUARD(hipStreamSynchronize(DflStream()));
    CdamFreeDevice(w, bytes);
    return r;
}

void ArrayDot(f64* result, const Array* a, const Array* b) {
    ASSERT(result && a && b && "ArrayDot: NULL pointer");
    same_side(a, b, "ArrayDot");
    if (a->is_host) {
        f64 s = 0.0;
        for (index_type i = 0; i < a->len; ++i) s += a->data[i] * b->data[i];
        *result = s;
    } else {
        *result = a->len > 0 ? device_reduce(a, b) : 0.0;
    }
}

void ArrayNorm2(f64* result, const Array* a) {
    ASSERT(result && a && "ArrayNorm2: NULL pointer");
    if (a->is_host) {
        f64 s = 0.0;
        for (index_type i = 0; i < a->len; ++i) s += a->data[i] * a->data[i];
        *result = sqrt(s);
    } else {
        *result = a->len > 0 ? device_reduce(a, NULL) : 0.0;
    }
}

void ArrayAXPY(Array* y, f64 a, const Array* x) {
    ASSERT(y && x && "ArrayAXPY: NULL pointer");
    same_side(y, x, "ArrayAXPY");
    if (y->is_host) {
        for (index_type i = 0; i < y->len; ++i) y->data[i] += a * x->data[i];
    } else {
        dfl_daxpy(y->len, a, x->data, y->data, DflStream());
        HIPGUARD(hipStreamSynchronize(DflStream()));
    }
}

void ArrayAXPBY(Array* y, f64 a, const Array* x, f64 b) {
    ASSERT(y && x && "ArrayAXPBY: NULL pointer");
    same_side(y, x, "ArrayAXPBY");
    if (y->is_host) {
        for (index_type i = 0; i < y->len; ++i) y->data[i] = a * x->data[i] + b * y->data[i];
    } else { /* scale, then axpy: the reference's two cuBLAS calls (Array.c:232-235) and their two roundings */
        dfl_dscal(y->len, b, y->data, DflStream());
        dfl_daxpy(y->len, a, x->data, y->data, DflStream());
        HIPGUARD(hipStreamSynchronize(DflStream()));
    }
}

/* ---- Field ---------------------------------------------------------------------------- */
Field* FieldCreate3D(const Mesh3D* mesh, i32 num_nodal_dof) {
    ASSERT(mesh && num_nodal_dof > 0 && "Number of nodal degrees of freedom must be positive.");
    Field* f = (Field*)CdamMallocHost(SIZE_OF(Field));
    memset(f, 0, sizeof *f);
    f->shape[0] = Mesh3DNumNode(mesh);
    f->shape[1] = num_nodal_dof;
    f->host = ArrayCreateHost(f->shape[0] * num_nodal_dof);
    f->device = ArrayCreateDevice(f->shape[0] * num_nodal_dof);
    return f;
}

void FieldDestroy(Field* f) {
    if (!f) return;
    ArrayDestroy(f->host);
    ArrayDestroy(f->device);
    CdamFreeHost(f, SIZE_OF(Field));
}

void FieldInit(Field* f, void (*func)(f64*, void*), void* ctx) {
    ASSERT(f && func && "FieldInitCond: NULL pointer.");
    (*func)(ArrayData(f->host), ctx);
    ArrayCopy(f->device, f->host, hipMemcpyHostToDevice);
}

void FieldCopy(Field* dst, const Field* src) {
    ASSERT(dst && src && "FieldCopy: NULL pointer.");
    ArrayCopy(dst->host, src->host, hipMemcpyHostToHost);
    ArrayCopy(dst->device, src->device, hipMemcpyDeviceToDevice);
}

void FieldUpdateHost(Field* f) {
    ASSERT(f && "FieldUpdateHost: NULL pointer.");
    ArrayCopy(f->host, f->device, hipMemcpyDeviceToHost);
}

void FieldUpdateDevice(Field* f) {
    ASSERT(f && "FieldUpdateDevice: NULL pointer.");
    ArrayCopy(f->device, f->host, hipMemcpyHostToDevice);
}
