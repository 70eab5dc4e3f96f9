/* Runtime: error guard, global context, allocators.
 * Mirrors src/common.c:21-51 (Init/Finalize/GlobalContextGet) and src/alloc.c:8-50
 * (host malloc / device malloc + zero fill).  There are no vendor-library handles
 * on this path; the "handle" slots hand out the library stream. */
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"

static hipStream_t g_stream = 0;
static int g_initialised = 0;

void DflGuardPrivate(hipError_t code, const char* file, int line) {
    if (code != hipSuccess) {
        printf("GPUAssert: %s %s %d\n", hipGetErrorString(code), file, line);
        fflush(stdout);
        ASSERT(FALSE);
    }
}

void Init(int argc, char** argv) {
    UNUSED(argc);
    UNUSED(argv);
    if (g_initialised) return;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fprintf(stderr, "dedflow: no HIP device available (%s) -- the product path has no CPU fallback\n",
                hipGetErrorString(e));
        abort();
    }
    g_initialised = 1;
}

void Finalize(void) {
    if (!g_initialised) return;
    HIPGUARD(hipDeviceSynchronize());
    g_initialised = 0;
}

void* GlobalContextGet(GlobalContextType type) {
    UNUSED(type);
    return (void*)&g_stream;
}

hipStream_t DflStream(void) { return g_stream; }
void DflSetStream(hipStream_t s) { g_stream = s; }

static void* host_malloc(ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(ctx);
    if (size <= 0) return NULL;
    return malloc((size_t)size);
}
static void host_free(void* ptr, ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(size);
    UNUSED(ctx);
    free(ptr);
}
/* zero-filled device allocation, as alloc.c:23-30 */
static void* device_malloc(ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(ctx);
    void* p = NULL;
    if (size <= 0) return NULL;
    HIPGUARD(hipMalloc(&p, (size_t)size));
    HIPGUARD(hipMemsetAsync(p, 0, (size_t)size, g_stream));
    return p;
}
static void device_free(void* ptr, ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(size);
    UNUSED(ctx);
    if (ptr) HIPGUARD(hipFree(ptr));
}

static Allocator g_alloc[2] = {{host_malloc, host_free, NULL}, {device_malloc, device_free, NULL}};

Allocator* GetDefaultAllocator(int device_id) { return &g_alloc[device_id == DEVICE ? 1 : 0]; }

/* ---- vec.h ------------------------------------------------------------------ */
void VecAXPY(value_type a, const value_type* x, value_type* y, index_type n) { dfl_daxpy(n, a, x, y, g_stream); }
void VecPointwiseMult(const value_type* x, const value_type* y, value_type* z, index_type n) {
    dfl_pointwise_mult(n, x, y, z, g_stream);
}
void VecPointwiseDiv(const value_type* x, const value_type* y, value_type* z, index_type n) {
    dfl_pointwise_div(n, x, y, z, g_stream);
}
void VecPointwiseInv(value_type* x, index_type n) { dfl_pointwise_inv(n, x, g_stream); }

/* ---- Array.h (storage only; the reference's BLAS wrappers are dead on the path) ---- */
Array* ArrayCreateHost(index_type len) {
    Array* a = (Array*)CdamMallocHost(SIZE_OF(Array));
    a->is_host = TRUE;
    a->len = len;
    a->data = (f64*)CdamMallocHost((ptrdiff_t)len * SIZE_OF(f64));
    if (a->data) memset(a->data, 0, (size_t)len * sizeof(f64));
    return a;
}
Array* ArrayCreateDevice(index_type len) {
    Array* a = (Array*)CdamMallocHost(SIZE_OF(Array));
    a->is_host = FALSE;
    a->len = len;
    a->data = (f64*)CdamMallocDevice((ptrdiff_t)len * SIZE_OF(f64));
    return a;
}
void ArrayDestroy(Array* a) {
    if (!a) return;
    if (a->is_host) CdamFreeHost(a->data, (ptrdiff_t)a->len * SIZE_OF(f64));
    else CdamFreeDevice(a->data, (ptrdiff_t)a->len * SIZE_OF(f64));
    CdamFreeHost(a, SIZE_OF(Array));
}
void ArrayCopy(Array* dst, const Array* src, MemCopyKind kind) {
    ASSERT(dst && src && dst->len == src->len);
    HIPGUARD(hipMemcpy(dst->data, src->data, (size_t)src->len * sizeof(f64), kind));
}

/* ---- in-library kernel timing (hipEvent pairs on the library stream) --------------------
 * bench.py brackets the timed region with these so that per-kernel launch durations are
 * measured live, on the stream the kernels run on. */
#define DFL_PROF_TAGS 16
#define DFL_PROF_MAX 8192
static int g_prof_on = 0;
static struct { hipEvent_t a, b; int tag; } g_ev[DFL_PROF_MAX];
static int g_ev_n = 0, g_ev_alloc = 0;

void DflProfileEnable(int on) {
    g_prof_on = on;
    g_ev_n = 0;
}
int DflProfileBegin(int tag) {
    if (!g_prof_on || g_ev_n >= DFL_PROF_MAX) return -1;
    int i = g_ev_n++;
    if (i >= g_ev_alloc) {
        HIPGUARD(hipEventCreate(&g_ev[i].a));
        HIPGUARD(hipEventCreate(&g_ev[i].b));
        g_ev_alloc = i + 1;
    }
    g_ev[i].tag = tag;
    HIPGUARD(hipEventRecord(g_ev[i].a, g_stream));
    return i;
}
void DflProfileEnd(int slot) {
    if (slot >= 0) HIPGUARD(hipEventRecord(g_ev[slot].b, g_stream));
}
/* sums the elapsed time of every recorded interval with this tag; synchronises */
int DflProfileCollect(int tag, double* total_ms, double* min_ms) {
    int count = 0;
    double tot = 0.0, mn = 1e300;
    HIPGUARD(hipStreamSynchronize(g_stream));
    for (int i = 0; i < g_ev_n; ++i) {
        if (g_ev[i].tag != tag) continue;
        float ms = 0.f;
        HIPGUARD(hipEventElapsedTime(&ms, g_ev[i].a, g_ev[i].b));
        tot += ms;
        if (ms < mn) mn = ms;
        ++count;
    }
    if (total_ms) *total_ms = tot;
    if (min_ms) *min_ms = count ? mn : 0.0;
    return count;
}
