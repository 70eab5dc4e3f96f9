/* Runtime: error guard, global context, allocators.
 * Mirrors src/common.c:21-51 (Init/Finalize/GlobalContextGet) and src/alloc.c:8-50
 * (host malloc / device malloc + zero fill).  There are no vendor-library handles
 * on this path; the "handle" slots hand out the library stream. */
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>
#include <omp.h>
#include "dedflow.h"
#include "dedflow_kernels.h"

static hipStream_t g_stream = 0;
static int g_initialised = 0;
static void pool_configure(void);
static void pool_release_all(void);
struct PoolChunk;
static struct PoolChunk* pool_new_chunk(size_t bytes);
static int g_pool_state;
static char* g_varena = NULL;
static size_t g_varena_size = 0, g_varena_off = 0;
static int g_varena_live = 0;
static int g_nchunk;
static size_t g_chunk_bytes;

void DflGuardPrivate(hipError_t code, const char* file, int line) {
    if (code != hipSuccess) {
        printf("GPUAssert: %s %s %d\n", hipGetErrorString(code), file, line);
        fflush(stdout);
        ASSERT(FALSE);
    }
}

/* ---- "is the driver still wiping freed device memory?" ---------------------------------------------------------------
 * Device memory that a process frees (or leaves behind when it exits) is wiped by the driver in the background at about
 * 36 GB/s before it is handed out again; meanwhile streaming kernels run 0-8 % slower in episodes, the SOC clock sits at
 * 1200 MHz, and allocations made in that state land where the wipe has already been (tools/probe_exit_wipe.py,
 * tools/probe_clocks.py).  hipMemGetInfo does not see it (it counts the memory as free at once); the
 * driver's own "VRAM used" figure does, and rocm_smi reads it (read-only, no privileges).  rocm_smi is bound at run time
 * like RCCL and ROCTx; when it is missing every function below reports "unknown" and nobody waits. */
static struct {
    int state; /* 0 = not tried, 1 = library bound, -1 = unavailable */
    int (*mem_usage)(uint32_t, int, uint64_t*);
    int (*clk_freq)(uint32_t, int, void*);
    int (*num)(uint32_t*);
    int (*pci)(uint32_t, uint64_t*);
    int dv_of[16]; /* rocm_smi index of HIP device d: 0 = not looked up yet, -1 = no match, i + 1 = index i */
} g_smi;
/* rsmi_frequencies_t of rocm_smi.h (ROCm 6-7: bool has_deep_sleep; uint32_t num_supported, current; uint64_t frequency[33]),
 * mirrored by hand because the header is not a build dependency; rocm_smi writes at most this many bytes into it */
typedef struct { uint8_t has_deep_sleep; uint32_t num_supported, current; uint64_t frequency[33]; } SmiFrequencies;

static void smi_bind(void) {
    if (g_smi.state) return;
    g_smi.state = -1;
    if (getenv("DFL_NO_SMI")) return;
    void* h = dlopen("librocm_smi64.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librocm_smi64.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librocm_smi64.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    int (*init)(uint64_t) = (int (*)(uint64_t))dlsym(h, "rsmi_init");
    *(void**)(&g_smi.num) = dlsym(h, "rsmi_num_monitor_devices");
    *(void**)(&g_smi.pci) = dlsym(h, "rsmi_dev_pci_id_get");
    *(void**)(&g_smi.mem_usage) = dlsym(h, "rsmi_dev_memory_usage_get");
    *(void**)(&g_smi.clk_freq) = dlsym(h, "rsmi_dev_gpu_clk_freq_get");
    if (!init || !g_smi.num || !g_smi.pci || !g_smi.mem_usage || init(0) != 0) return;
    g_smi.state = 1;
}

/* rocm_smi index of the CURRENT HIP device (a rank may select its device after Init()): matched by PCI domain : bus :
 * device . function, looked up once per HIP device; -1 when unknown */
static int smi_device(void) {
    smi_bind();
    if (g_smi.state != 1) return -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (dev < 0 || dev >= 16) return -1;
    if (g_smi.dv_of[dev]) return g_smi.dv_of[dev] - 1 >= 0 ? g_smi.dv_of[dev] - 1 : -1;
    g_smi.dv_of[dev] = -1;
    char bus[64] = {0};
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return -1;
    uint32_t n = 0;
    if (g_smi.num(&n) != 0) return -1;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t id = 0;
        if (g_smi.pci(i, &id) != 0) continue;
        if ((unsigned)(id >> 32) == dom && (unsigned)((id >> 8) & 0xff) == b && (unsigned)((id >> 3) & 0x1f) == d && (unsigned)(id & 7) == f) {
            g_smi.dv_of[dev] = (int)i + 1;
            return (int)i;
        }
    }
    return -1;
}

/* bytes of VRAM the driver counts as in use on the current device, INCLUDING memory that was freed but is not wiped yet;
 * -1 when unknown */
int64_t DflDeviceMemoryInUse(void) {
    const int dv = smi_device();
    uint64_t used = 0;
    if (dv < 0 || g_smi.mem_usage((uint32_t)dv, 0 /* RSMI_MEM_TYPE_VRAM */, &used) != 0) return -1;
    return (int64_t)used;
}

/* 1 while the SOC clock is at its high level (>= 600 MHz; it idles and computes below 150 MHz): the second sign of a wipe,
 * and the only one for memory freed inside the running process (that leaves the "in use" figure at once); 0 otherwise or
 * when unknown */
static int smi_soc_clock_high(void) {
    union { SmiFrequencies f; char pad[sizeof(SmiFrequencies) + 256]; } u; /* slack should the library's struct be larger */
    memset(&u, 0, sizeof u);
    const int dv = smi_device();
    if (dv < 0 || !g_smi.clk_freq || g_smi.clk_freq((uint32_t)dv, 3 /* RSMI_CLK_TYPE_SOC */, &u.f) != 0 || u.f.current >= 33) return 0;
    return u.f.frequency[u.f.current] >= 600000000ull;
}

/* Blocks while a wipe is in progress -- the "in use" figure FALLING (nothing of ours is being freed meanwhile) or the SOC
 * clock high -- at most max_seconds.  Returns the seconds waited, 0 when there was no sign of one, -1 when rocm_smi is
 * unavailable.  A first look of 60 ms decides whether to wait at all; the wait ends after 0.4 s without either sign. */
double DflWaitDeviceMemoryQuiet(double max_seconds) {
    int64_t prev = DflDeviceMemoryInUse();
    if (prev < 0) return -1.0;
    const double t0 = omp_get_wtime();
    const int64_t step = (int64_t)64 << 20;
    usleep(60000);
    int64_t cur = DflDeviceMemoryInUse();
    if (cur < 0) return -1.0;
    if (prev - cur < step && !smi_soc_clock_high()) return 0.0;
    int calm = 0;
    while (calm < 8 && omp_get_wtime() - t0 < max_seconds) {
        prev = cur;
        usleep(50000);
        cur = DflDeviceMemoryInUse();
        if (cur < 0) return -1.0;
        calm = (prev - cur < step && !smi_soc_clock_high()) ? calm + 1 : 0;
    }
    return omp_get_wtime() - t0;
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
    /* a predecessor's memory may still be under the driver's wipe: wait for that before anything is placed -- but briefly by
       default (5 s; 129 GB take 3.8 s to wipe): on a shared GPU the signs may be another tenant's doing.  DFL_INIT_QUIET_S=0
       never blocks; INTEGRATION.md documents the wait */
    {
        const char* eq = getenv("DFL_INIT_QUIET_S");
        const double cap = eq ? atof(eq) : 5.0;
        if (cap > 0.0) {
            const double w = DflWaitDeviceMemoryQuiet(cap);
            if (w > 0.0 && getenv("DFL_WS_VERBOSE")) fprintf(stderr, "[init] waited %.2f s for the driver to finish wiping freed device memory\n", w);
        }
    }
    /* reserve the first pool chunk while VRAM is pristine (see "device memory pool" below) */
    pool_configure();
    if (g_pool_state == 1 && g_nchunk == 0) (void)pool_new_chunk(g_chunk_bytes);
    /* optional second early reservation for solver vectors (DFL_VECTOR_ARENA_GB, developer A/B; see host/solver.c) */
    {
        const char* e = getenv("DFL_VECTOR_ARENA_GB");
        double gb = e ? atof(e) : 0.0;
        if (gb > 0.0 && !g_varena) {
            g_varena_size = (size_t)(gb * 1073741824.0);
            if (hipMalloc((void**)&g_varena, g_varena_size) != hipSuccess) { (void)hipGetLastError(); g_varena = NULL; g_varena_size = 0; }
        }
    }
}

/* bump allocator over the vector arena: returns NULL when the arena is absent or full; the arena is reused from the start
 * once everything taken from it has been returned */
void* DflVectorArenaAlloc(size_t bytes) {
    bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (!g_varena || g_varena_off + bytes > g_varena_size) return NULL;
    void* p = g_varena + g_varena_off;
    g_varena_off += bytes;
    g_varena_live++;
    return p;
}
int DflVectorArenaFree(void* p) {
    if (!g_varena || (char*)p < g_varena || (char*)p >= g_varena + g_varena_size) return 0;
    if (--g_varena_live == 0) g_varena_off = 0;
    return 1;
}

void Finalize(void) {
    if (!g_initialised) return;
    HIPGUARD(hipDeviceSynchronize());
    pool_release_all();
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
/* ---- device memory pool --------------------------------------------------------------------
 * Measured on MI355X (tools/probe_spmv_placement*.py): the block-CSR SpMV runs at 0.555-0.57 ms when its
 * 3.3 GB value array sits in a large allocation made while VRAM is still pristine, and at 0.64-0.71 ms when the
 * same array is hipMalloc'ed after the setup temporaries have come and gone (placement of that one array decides;
 * a plain 4 TB/s streaming read does not see the difference).  So large buffers are sub-allocated from big
 * chunks reserved early: the first chunk at Init(), further chunks on demand.  DFL_DEVICE_POOL_GB sets the chunk
 * size (0 disables the pool; default min(32 GiB, a quarter of the free VRAM)).  First fit with coalescing; the
 * library has one host thread and stream-ordered use, so a freed block may be handed out again immediately. */
#define POOL_MIN_REQUEST ((size_t)16 << 20)
#define POOL_ALIGN ((size_t)2 << 20)
typedef struct PoolBlock { size_t off, size; int used; } PoolBlock;
typedef struct PoolChunk { char* base; size_t size; PoolBlock* blk; int nblk, cap; } PoolChunk;
static PoolChunk g_chunk[32];
/* g_nchunk, g_chunk_bytes, g_pool_state (0 = not configured, 1 = on, -1 = off) are declared at the top of the file */

static void pool_configure(void) {
    if (g_pool_state) return;
    const char* e = getenv("DFL_DEVICE_POOL_GB");
    double gb = -1.0;
    if (e) gb = atof(e);
    if (gb == 0.0) { g_pool_state = -1; return; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { g_pool_state = -1; return; }
    size_t want = gb > 0.0 ? (size_t)(gb * 1073741824.0) : (size_t)32 << 30;
    if (gb < 0.0 && want > free_b / 4) want = free_b / 4;
    want = want / POOL_ALIGN * POOL_ALIGN;
    if (want < ((size_t)256 << 20)) { g_pool_state = -1; return; }
    g_chunk_bytes = want;
    g_pool_state = 1;
}

static PoolChunk* pool_new_chunk(size_t bytes) {
    if (g_nchunk == (int)(sizeof g_chunk / sizeof g_chunk[0])) return NULL;
    void* p = NULL;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError(); /* out of memory for a whole chunk: the caller falls back to a plain hipMalloc */
        return NULL;
    }
    PoolChunk* c = &g_chunk[g_nchunk++];
    c->base = (char*)p;
    c->size = bytes;
    c->cap = 64;
    c->blk = (PoolBlock*)malloc(sizeof(PoolBlock) * (size_t)c->cap);
    c->nblk = 1;
    c->blk[0].off = 0;
    c->blk[0].size = bytes;
    c->blk[0].used = 0;
    return c;
}

static void* pool_take(PoolChunk* c, size_t bytes) {
    for (int i = 0; i < c->nblk; ++i) {
        PoolBlock* b = &c->blk[i];
        if (b->used || b->size < bytes) continue;
        if (b->size > bytes) { /* split: [used bytes][free rest] */
            if (c->nblk == c->cap) {
                c->cap *= 2;
                c->blk = (PoolBlock*)realloc(c->blk, sizeof(PoolBlock) * (size_t)c->cap);
                b = &c->blk[i];
            }
            memmove(b + 2, b + 1, sizeof(PoolBlock) * (size_t)(c->nblk - i - 1));
            b[1].off = b->off + bytes;
            b[1].size = b->size - bytes;
            b[1].used = 0;
            b->size = bytes;
            c->nblk++;
        }
        b->used = 1;
        return c->base + b->off;
    }
    return NULL;
}

int DflDevicePoolEnabled(void) {
    pool_configure();
    return g_pool_state == 1;
}

static void* pool_alloc(size_t bytes) {
    pool_configure();
    if (g_pool_state != 1 || bytes < POOL_MIN_REQUEST) return NULL;
    bytes = (bytes + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN;
    for (int k = 0; k < g_nchunk; ++k) {
        void* p = pool_take(&g_chunk[k], bytes);
        if (p) return p;
    }
    PoolChunk* c = pool_new_chunk(bytes > g_chunk_bytes ? bytes : g_chunk_bytes);
    return c ? pool_take(c, bytes) : NULL;
}

/* a zero-filled block from the chunks the pool ALREADY holds, NULL when none has room (never reserves a new chunk: the
 * bounded work-space calibration must not grow the pool for a candidate it may throw away); free with CdamFreeDevice */
void* DflDevicePoolAllocNoGrow(size_t bytes) {
    pool_configure();
    if (g_pool_state != 1 || bytes < POOL_MIN_REQUEST) return NULL;
    bytes = (bytes + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN;
    for (int k = 0; k < g_nchunk; ++k) {
        void* p = pool_take(&g_chunk[k], bytes);
        if (p) {
            HIPGUARD(hipMemsetAsync(p, 0, bytes, g_stream));
            return p;
        }
    }
    return NULL;
}

static int pool_free(void* ptr) {
    for (int k = 0; k < g_nchunk; ++k) {
        PoolChunk* c = &g_chunk[k];
        if ((char*)ptr < c->base || (char*)ptr >= c->base + c->size) continue;
        const size_t off = (size_t)((char*)ptr - c->base);
        for (int i = 0; i < c->nblk; ++i) {
            if (c->blk[i].off != off) continue;
            ASSERT(c->blk[i].used && "device pool: double free");
            c->blk[i].used = 0;
            if (i + 1 < c->nblk && !c->blk[i + 1].used) { /* coalesce with the right neighbour */
                c->blk[i].size += c->blk[i + 1].size;
                memmove(&c->blk[i + 1], &c->blk[i + 2], sizeof(PoolBlock) * (size_t)(c->nblk - i - 2));
                c->nblk--;
            }
            if (i > 0 && !c->blk[i - 1].used) { /* and the left one */
                c->blk[i - 1].size += c->blk[i].size;
                memmove(&c->blk[i], &c->blk[i + 1], sizeof(PoolBlock) * (size_t)(c->nblk - i - 1));
                c->nblk--;
            }
            return 1;
        }
        ASSERT(FALSE && "device pool: pointer is not the start of a block");
    }
    return 0;
}

static void pool_release_all(void) {
    for (int k = 0; k < g_nchunk; ++k) {
        (void)hipFree(g_chunk[k].base);
        free(g_chunk[k].blk);
    }
    g_nchunk = 0;
}

/* bytes reserved / in use, for diagnostics and tests */
void DflDevicePoolStats(int64_t* reserved, int64_t* in_use) {
    int64_t r = 0, u = 0;
    for (int k = 0; k < g_nchunk; ++k) {
        r += (int64_t)g_chunk[k].size;
        for (int i = 0; i < g_chunk[k].nblk; ++i)
            if (g_chunk[k].blk[i].used) u += (int64_t)g_chunk[k].blk[i].size;
    }
    if (reserved) *reserved = r;
    if (in_use) *in_use = u;
}

/* zero-filled device allocation, as alloc.c:23-30 */
static void* device_malloc(ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(ctx);
    void* p = NULL;
    if (size <= 0) return NULL;
    p = pool_alloc((size_t)size);
    if (!p) HIPGUARD(hipMalloc(&p, (size_t)size));
    HIPGUARD(hipMemsetAsync(p, 0, (size_t)size, g_stream));
    return p;
}
static void device_free(void* ptr, ptrdiff_t size, UserCtxPtr ctx) {
    UNUSED(size);
    UNUSED(ctx);
    if (!ptr) return;
    if (!pool_free(ptr)) HIPGUARD(hipFree(ptr));
}

static Allocator g_alloc[2] = {{host_malloc, host_free, NULL}, {device_malloc, device_free, NULL}};

Allocator* GetDefaultAllocator(int device_id) { return &g_alloc[device_id == DEVICE ? 1 : 0]; }
/* the CdamMallocDevice / CdamFreeDevice macros as functions (bindings that cannot expand C macros) */
void* DflDeviceMalloc(int64_t bytes) { return CdamMallocDevice((ptrdiff_t)bytes); }
void DflDeviceFree(void* ptr) { CdamFreeDevice(ptr, 0); }

/* ---- vec.h ------------------------------------------------------------------ */
void VecAXPY(value_type a, const value_type* x, value_type* y, index_type n) { dfl_daxpy(n, a, x, y, g_stream); }
void VecPointwiseMult(const value_type* x, const value_type* y, value_type* z, index_type n) {
    dfl_pointwise_mult(n, x, y, z, g_stream);
}
void VecPointwiseDiv(const value_type* x, const value_type* y, value_type* z, index_type n) {
    dfl_pointwise_div(n, x, y, z, g_stream);
}
void VecPointwiseInv(value_type* x, index_type n) { dfl_pointwise_inv(n, x, g_stream); }

/* ---- Array.h storage (the BLAS-1 wrappers and Field are in field.c) ---- */
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
/* the individual durations (ms) of the recorded intervals with this tag, in launch order; returns how many were written */
int DflProfileDurations(int tag, double* out_ms, int max_out) {
    int count = 0;
    HIPGUARD(hipStreamSynchronize(g_stream));
    for (int i = 0; i < g_ev_n && count < max_out; ++i) {
        if (g_ev[i].tag != tag) continue;
        float ms = 0.f;
        HIPGUARD(hipEventElapsedTime(&ms, g_ev[i].a, g_ev[i].b));
        out_ms[count++] = ms;
    }
    return count;
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

/* ---- optional ROCTX ranges (tracing).  DFL_ROCTX=1 binds roctxRangePushA / roctxRangePop from
 * librocprofiler-sdk-roctx.so (or libroctx64.so) at the first call; `rocprofv3 --kernel-trace --marker-trace` then shows AssembleSystem(F) / AssembleSystem(J) /
 * KrylovSolve / DEM sweep / DflTimeStep as named ranges around their kernels.  Off (two loads of a static) otherwise. */
static int g_roctx_state = 0; /* 0 = not looked at, 1 = bound, -1 = off */
static int (*g_roctx_push)(const char*) = NULL;
static int (*g_roctx_pop)(void) = NULL;
static void roctx_bind(void) {
    g_roctx_state = -1;
    const char* e = getenv("DFL_ROCTX");
    if (!e || !atoi(e)) return;
    /* rocprofv3 (rocprofiler-sdk) listens to its own ROCTx library; the roctracer-era libroctx64 is the fallback */
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        fprintf(stderr, "DFL_ROCTX: %s\n", dlerror());
        return;
    }
    *(void**)(&g_roctx_push) = dlsym(h, "roctxRangePushA");
    *(void**)(&g_roctx_pop) = dlsym(h, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) g_roctx_state = 1;
}
void DflRangePush(const char* name) {
    if (g_roctx_state == 0) roctx_bind();
    if (g_roctx_state == 1) g_roctx_push(name);
}
void DflRangePop(void) {
    if (g_roctx_state == 1) g_roctx_pop();
}
