/* RCCL implementation of the DflComm callbacks (include/dedflow.h): the collectives of an
 * element-partitioned solve are enqueued from C on the library's HIP stream, so a GMRES iteration
 * involves no host round trip (the torch.distributed callbacks of dedflow_amd/dist.py cross into
 * Python three times per iteration).
 *   allreduce_sum : ncclAllReduce in place (CGS coefficients, norms, Newton norms)
 *   halo_exchange : pack kernel -> one ncclGroup of ncclSend/ncclRecv to/from every neighbour (xGMI
 *                   is a full mesh: all neighbours at once, no ring) -> unpack kernel
 * RCCL is bound at run time (dlopen of the path the caller passes, e.g. the librccl.so that
 * PyTorch-ROCm already loaded), so single-GPU users of libdedflow.so carry no RCCL dependency.
 * The reference has no multi-GPU path; this file is build-defined. */
#include <dlfcn.h>
#include <string.h>
#include "dedflow.h"
#include "dedflow_kernels.h"
#include "host_private.h"

typedef struct { char internal[128]; } rccl_unique_id; /* ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128 */
typedef void* rccl_comm;
enum { RCCL_SUCCESS = 0, RCCL_FLOAT64 = 8, RCCL_SUM = 0 }; /* ncclSuccess, ncclFloat64 (= ncclDouble), ncclSum */

static struct {
    void* handle;
    int (*GetUniqueId)(rccl_unique_id*);
    int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int);
    int (*CommDestroy)(rccl_comm);
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
    int (*Send)(const void*, size_t, int, int, rccl_comm, hipStream_t);
    int (*Recv)(void*, size_t, int, int, rccl_comm, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char* (*GetErrorString)(int);
} R;

#define RCCLGUARD(call)                                                                             \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_ != RCCL_SUCCESS) {                                                                  \
            fprintf(stderr, "RCCL error %d (%s) at %s:%d\n", rc_, R.GetErrorString ? R.GetErrorString(rc_) : "?", \
                    __FILE__, __LINE__);                                                            \
            ASSERT(FALSE);                                                                          \
        }                                                                                           \
    } while (0)

struct DflRcclComm {
    rccl_comm comm;      /* all-reduces, on the library stream */
    rccl_comm halo_comm; /* point-to-point halo traffic; a communicator of its own because it runs on the side stream
                            concurrently with the library stream (NULL: shares `comm`) */
    int rank, world;
    index_type n_local, n_owned;
    index_type nsend, nrecv;
    index_type *send_count, *recv_count; /* host [world], in f64 entries */
    index_type *d_send_idx, *d_recv_idx; /* device: flat dof indices, concatenated in rank order */
    f64 *d_send, *d_recv;
    DflComm vt;
    int64_t n_allreduce, n_halo;
    hipStream_t side;            /* non-blocking stream the split halo exchange runs on */
    hipEvent_t ev_ready, ev_done;
};

/* 0 on success; the library stays loaded for the life of the process */
int DflRcclLoad(const char* path) {
    if (R.handle) return 0;
    void* h = dlopen(path && path[0] ? path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "DflRcclLoad: %s\n", dlerror());
        return 1;
    }
#define BIND(field, name)                                                   \
    *(void**)(&R.field) = dlsym(h, name);                                   \
    if (!R.field) { fprintf(stderr, "DflRcclLoad: %s missing\n", name); return 2; }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllReduce, "ncclAllReduce")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    R.handle = h;
    return 0;
}

int DflRcclUniqueIdBytes(void) { return (int)sizeof(rccl_unique_id); }

/* rank 0 creates the id; the caller broadcasts the bytes to the other ranks by any means */
int DflRcclGetUniqueId(char* out128) {
    if (!R.handle) return 1;
    rccl_unique_id id;
    memset(&id, 0, sizeof id);
    int rc = R.GetUniqueId(&id);
    memcpy(out128, &id, sizeof id);
    return rc;
}

static void rccl_allreduce(void* ctx, f64* d_buf, index_type n) {
    DflRcclComm* c = (DflRcclComm*)ctx;
    if (n <= 0) return;
    RCCLGUARD(R.AllReduce(d_buf, d_buf, (size_t)n, RCCL_FLOAT64, RCCL_SUM, c->comm, DflStream()));
    c->n_allreduce++;
}

static void rccl_halo_on(DflRcclComm* c, f64* d_x, hipStream_t s) {
    if (c->nsend == 0 && c->nrecv == 0) return;
    rccl_comm hc = c->halo_comm ? c->halo_comm : c->comm;
    dfl_gather_idx(c->nsend, c->d_send_idx, d_x, c->d_send, s);
    RCCLGUARD(R.GroupStart());
    index_type so = 0, ro = 0;
    for (int q = 0; q < c->world; ++q) {
        if (c->send_count[q]) RCCLGUARD(R.Send(c->d_send + so, (size_t)c->send_count[q], RCCL_FLOAT64, q, hc, s));
        if (c->recv_count[q]) RCCLGUARD(R.Recv(c->d_recv + ro, (size_t)c->recv_count[q], RCCL_FLOAT64, q, hc, s));
        so += c->send_count[q];
        ro += c->recv_count[q];
    }
    RCCLGUARD(R.GroupEnd());
    dfl_scatter_idx(c->nrecv, c->d_recv_idx, c->d_recv, d_x, s);
}

static void rccl_halo(void* ctx, f64* d_x) {
    DflRcclComm* c = (DflRcclComm*)ctx;
    c->n_halo++;
    rccl_halo_on(c, d_x, DflStream());
}

/* split form: the exchange runs on a side stream behind an event recorded on the library stream (d_x is complete),
 * the library stream goes on with the interior rows and waits for the side stream in halo_end.  Between the two calls the
 * caller may enqueue work of its own on the side stream (halo_stream): it runs behind the unpack, and halo_end makes the
 * library stream wait for it as well -- the Krylov matvec puts the boundary rows there, so they overlap the interior rows
 * and the library stream pays ONE wait per matvec and no launch of its own for them. */
static void rccl_halo_begin(void* ctx, f64* d_x) {
    DflRcclComm* c = (DflRcclComm*)ctx;
    c->n_halo++;
    HIPGUARD(hipEventRecord(c->ev_ready, DflStream()));
    HIPGUARD(hipStreamWaitEvent(c->side, c->ev_ready, 0));
    rccl_halo_on(c, d_x, c->side);
}
static void rccl_halo_end(void* ctx, f64* d_x) {
    DflRcclComm* c = (DflRcclComm*)ctx;
    UNUSED(d_x);
    HIPGUARD(hipEventRecord(c->ev_done, c->side));
    HIPGUARD(hipStreamWaitEvent(DflStream(), c->ev_done, 0));
}
static hipStream_t rccl_halo_stream(void* ctx) { return ((DflRcclComm*)ctx)->side; }

/* collective over all ranks: every rank passes the same 128-byte id */
/* A stream that really runs BESIDE `main_stream`.  HIP maps streams onto a handful of hardware queues round-robin; a side
 * stream that lands on the main stream's own queue runs its kernels AFTER the main stream's instead of beside them (rank 0 of
 * the 8-way partition: boundary rows behind the interior rows, both event hops 10 us instead of 6), and which queue a new
 * stream gets depends on how many streams the process made before (torch, RCCL).  A stream of the HIGHEST priority always gets
 * a queue of its own -- and was worse: in some processes every kernel of the normal-priority library stream then ran with random
 * delays of 0-100 us for as long as that stream lived (first problem of a process: 59 instead of 46 ms per 10M-tet step, 17.6
 * instead of 7.2 ms for the 8-way share; tools/probe_rank_local.py with DFL_HALO_STREAM_PRIORITY=high).  So: default priority,
 * and the candidate is PROBED -- a 300-us resident wave on the main stream, an empty launch on the candidate: if the
 * candidate's launch ends first the two overlap.  Up to 6 candidates; the rejected ones are destroyed afterwards (destroying
 * one at once would hand its queue to the next).  DFL_HALO_STREAM_PRIORITY=high|default skips the probe. */
hipStream_t DflPickConcurrentStream(hipStream_t main_stream) {
    hipStream_t pick = NULL;
    const char* force = getenv("DFL_HALO_STREAM_PRIORITY");
    if (force && !strcmp(force, "high")) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        HIPGUARD(hipStreamCreateWithPriority(&pick, hipStreamNonBlocking, hi));
        return pick;
    }
    if (force && !strcmp(force, "default")) {
        HIPGUARD(hipStreamCreateWithFlags(&pick, hipStreamNonBlocking));
        return pick;
    }
    enum { MAXC = 6 };
    hipStream_t cand[MAXC];
    int nc = 0;
    hipEvent_t em, es;
    HIPGUARD(hipEventCreate(&em));
    HIPGUARD(hipEventCreate(&es));
    HIPGUARD(hipStreamSynchronize(main_stream));
    for (; nc < MAXC && !pick; ++nc) {
        HIPGUARD(hipStreamCreateWithFlags(&cand[nc], hipStreamNonBlocking));
        for (int rep = 0; rep < 2 && !pick; ++rep) { /* (a first launch on a new stream may pay set-up time: two tries) */
            dfl_spin_us(300, main_stream);
            HIPGUARD(hipEventRecord(em, main_stream));
            dfl_spin_us(1, cand[nc]);
            HIPGUARD(hipEventRecord(es, cand[nc]));
            HIPGUARD(hipEventSynchronize(em));
            HIPGUARD(hipEventSynchronize(es));
            float ms = 0.f; /* from the end of the candidate's launch to the end of the main stream's wave */
            HIPGUARD(hipEventElapsedTime(&ms, es, em));
            if (ms > 0.1f) pick = cand[nc]; /* the candidate was done >= 100 us before the 300-us wave: they overlapped */
        }
    }
    if (!pick) pick = cand[0]; /* every candidate shares the main stream's queue (GPU_MAX_HW_QUEUES=1?): correct, only slower */
    for (int k = 0; k < nc; ++k)
        if (cand[k] != pick) HIPGUARD(hipStreamDestroy(cand[k]));
    HIPGUARD(hipEventDestroy(em));
    HIPGUARD(hipEventDestroy(es));
    if (getenv("DFL_WS_VERBOSE")) fprintf(stderr, "[comm] halo stream: candidate %d of %d runs beside the library stream\n", nc, MAXC);
    return pick;
}

DflRcclComm* DflRcclCommCreate(const char* id128, int rank, int world) {
    if (!R.handle) {
        fprintf(stderr, "DflRcclCommCreate: call DflRcclLoad first\n");
        return NULL;
    }
    DflRcclComm* c = (DflRcclComm*)CdamMallocHost(SIZE_OF(DflRcclComm));
    memset(c, 0, sizeof *c);
    rccl_unique_id id;
    memcpy(&id, id128, sizeof id);
    c->rank = rank;
    c->world = world;
    {   /* a communicator that cannot be created is not fatal: the caller falls back to its torch.distributed callbacks */
        const int rc = R.CommInitRank(&c->comm, world, id, rank);
        if (rc != RCCL_SUCCESS) {
            fprintf(stderr, "DflRcclCommCreate: ncclCommInitRank failed (%d: %s)\n", rc, R.GetErrorString ? R.GetErrorString(rc) : "?");
            CdamFreeHost(c, SIZE_OF(DflRcclComm));
            return NULL;
        }
    }
    c->send_count = (index_type*)CdamMallocHost(SIZE_OF(index_type) * world);
    c->recv_count = (index_type*)CdamMallocHost(SIZE_OF(index_type) * world);
    memset(c->send_count, 0, sizeof(index_type) * (size_t)world);
    memset(c->recv_count, 0, sizeof(index_type) * (size_t)world);
    c->side = DflPickConcurrentStream(DflStream());
    /* both events order work of ONE device across two of its streams (x complete -> pack; unpack -> boundary rows): no
       system-scope fence needed -- the peers' data arrives through RCCL's own kernels -- and without it the record / wait
       pair costs less idle time on the library stream (DFL_EVENT_SYSTEM_FENCE=1 restores the default) */
    unsigned evflags = hipEventDisableTiming;
    if (!(getenv("DFL_EVENT_SYSTEM_FENCE") && atoi(getenv("DFL_EVENT_SYSTEM_FENCE")))) evflags |= hipEventDisableSystemFence;
    HIPGUARD(hipEventCreateWithFlags(&c->ev_ready, evflags));
    HIPGUARD(hipEventCreateWithFlags(&c->ev_done, evflags));
    c->vt.allreduce_sum = rccl_allreduce;
    c->vt.halo_exchange = rccl_halo;
    c->vt.halo_begin = rccl_halo_begin;
    c->vt.halo_end = rccl_halo_end;
    c->vt.halo_stream = rccl_halo_stream;
    c->vt.rank = rank;
    c->vt.world = world;
    c->vt.ctx = c;
    return c;
}

/* optional second communicator for the halo traffic (collective; every rank passes the same second id).  Returns 0 on
 * success.  On failure this rank's halo traffic would share the main communicator -- which only works if EVERY rank does
 * the same, so the caller must agree on the outcome across ranks and call DflRcclCommDropHaloComm on all of them if any
 * failed (dedflow_amd/dist.py does). */
int DflRcclCommCreateHaloComm(DflRcclComm* c, const char* id128) {
    rccl_unique_id id;
    memcpy(&id, id128, sizeof id);
    const int rc = R.CommInitRank(&c->halo_comm, c->world, id, c->rank);
    if (rc != RCCL_SUCCESS) {
        fprintf(stderr, "DflRcclCommCreateHaloComm: ncclCommInitRank failed (%d)\n", rc);
        c->halo_comm = NULL;
        return 1;
    }
    return 0;
}
void DflRcclCommDropHaloComm(DflRcclComm* c) {
    if (c->halo_comm) R.CommDestroy(c->halo_comm);
    c->halo_comm = NULL; /* halo traffic shares the all-reduce communicator (never concurrent with it on one stream) */
}

/* halo plan: send_idx / recv_idx are HOST arrays of flat dof indices into the local [u|p|..] vector,
 * concatenated in rank order; send_count / recv_count [world] give each rank's share */
void DflRcclCommSetHalo(DflRcclComm* c, index_type n_local, index_type n_owned, const index_type* send_count,
                        const index_type* send_idx, const index_type* recv_count, const index_type* recv_idx) {
    c->n_local = n_local;
    c->n_owned = n_owned;
    c->vt.num_owned_node = n_owned;
    CdamFreeDevice(c->d_send_idx, 0); CdamFreeDevice(c->d_recv_idx, 0);
    CdamFreeDevice(c->d_send, 0); CdamFreeDevice(c->d_recv, 0);
    c->d_send_idx = c->d_recv_idx = NULL;
    c->d_send = c->d_recv = NULL;
    c->nsend = c->nrecv = 0;
    for (int q = 0; q < c->world; ++q) {
        ASSERT(q != c->rank || (send_count[q] == 0 && recv_count[q] == 0));
        c->send_count[q] = send_count[q];
        c->recv_count[q] = recv_count[q];
        c->nsend += send_count[q];
        c->nrecv += recv_count[q];
    }
    if (c->nsend) {
        c->d_send_idx = (index_type*)CdamMallocDevice((ptrdiff_t)c->nsend * SIZE_OF(index_type));
        c->d_send = (f64*)CdamMallocDevice((ptrdiff_t)c->nsend * SIZE_OF(f64));
        HIPGUARD(hipMemcpy(c->d_send_idx, send_idx, sizeof(index_type) * (size_t)c->nsend, H2D));
    }
    if (c->nrecv) {
        c->d_recv_idx = (index_type*)CdamMallocDevice((ptrdiff_t)c->nrecv * SIZE_OF(index_type));
        c->d_recv = (f64*)CdamMallocDevice((ptrdiff_t)c->nrecv * SIZE_OF(f64));
        HIPGUARD(hipMemcpy(c->d_recv_idx, recv_idx, sizeof(index_type) * (size_t)c->nrecv, H2D));
    }
}

void DflRcclCommSetInterior(DflRcclComm* c, index_type n_interior) { c->vt.num_interior_node = n_interior; }
const DflComm* DflRcclCommVtable(const DflRcclComm* c) { return &c->vt; }
void DflRcclCommCounters(const DflRcclComm* c, int64_t* n_allreduce, int64_t* n_halo) {
    if (n_allreduce) *n_allreduce = c->n_allreduce;
    if (n_halo) *n_halo = c->n_halo;
}

void DflRcclCommDestroy(DflRcclComm* c) {
    if (!c) return;
    HIPGUARD(hipStreamSynchronize(DflStream()));
    HIPGUARD(hipStreamSynchronize(c->side));
    if (c->halo_comm) R.CommDestroy(c->halo_comm);
    if (c->comm) R.CommDestroy(c->comm);
    HIPGUARD(hipEventDestroy(c->ev_ready));
    HIPGUARD(hipEventDestroy(c->ev_done));
    HIPGUARD(hipStreamDestroy(c->side));
    CdamFreeDevice(c->d_send_idx, 0); CdamFreeDevice(c->d_recv_idx, 0);
    CdamFreeDevice(c->d_send, 0); CdamFreeDevice(c->d_recv, 0);
    CdamFreeHost(c->send_count, 0); CdamFreeHost(c->recv_count, 0);
    CdamFreeHost(c, SIZE_OF(DflRcclComm));
}
