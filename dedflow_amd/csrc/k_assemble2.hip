// Owner-computes assembly kernels for gfx950 (assembly schedule 4, the default):
//
//   tet_lhs_slot_kernel   Jacobian: one workgroup per spatial patch of nodes owns their CSR rows.  Phase 1 evaluates,
//                         once per tet touching the patch, everything the sixteen (a,b) blocks of that tet share
//                         (shape gradients, |det J|, convective shape derivatives and stabilisation parameters at the four
//                         quadrature points; src/assemble.cu:528-603) into an LDS record.  Phase 2 gives every nodal
//                         nonzero ("slot") of the owned rows to one lane PAIR: each lane walks every other entry of the
//                         slot's (tet, a, b) contribution list (host/slotpatch.c), evaluates those blocks
//                         (assemble.cu:618-661) from the LDS records and adds them up in registers; one DPP exchange
//                         inside the pair, then the 128-byte block line is written ONCE (val = beta * val + sum).
//                         No atomics anywhere, fixed summation order => bitwise reproducible like the reference's colored
//                         scatter (matrix_impl.cu:370-453), at one launch instead of one per color.
//
// HBM view: every block line written once (128 B x nnz1) + slot map (4 B) + contribution descriptors (2 B x 16 T)
// + patch-tet connectivity (16 B per patch-tet); node records are gathered through L2.  Compute: 16 block evaluations
// per tet (the same count as the reference's kernel), the per-tet part once per (patch, tet) pair instead of 16 times.
#include "asm_device.hpp"

namespace {

constexpr int SBLK = DFL_SLOT_BLOCK;
constexpr int SP_RV = 36;  // doubles of an LDS tet record in use: shg[12] conv[a][q] (16) tauM[4] | cT, w | sum tauM, cK (asm_device.hpp)
// Record stride = record size.  (A stride of 38 doubles -- all 16 start banks of a 16-byte access reachable instead of 8 --
// left the bank-conflict count where it was, 2.5e8 of 5.0e8 LDS cycles, profiles/r03_pmc_lhs.txt: the lanes of a wave gather the
// records of unrelated tets, a birthday problem whatever the stride.  120 tets x 288 B + the 4.5 KB of node records keep four
// workgroups per CU.)
constexpr int SP_RS = 36;

typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));

// Dynamic LDS of one workgroup: [max_tets] tet records, nothing else.  The workgroups are PERSISTENT (grid = resident
// workgroups; each walks its share of the patches) and software-pipelined: the slot map and the lane-major contribution
// descriptors of the NEXT patch (host/slotpatch.c) are requested into registers before phase 1 of this one and have arrived
// before its row stores are issued, so neither phase waits on HBM and the row stores of patch p drain behind the work on
// patch p+1.  (The first form staged offset / slot / descriptor lists in a double LDS buffer: three dependent LDS round trips
// per slot before the first block evaluation and 45 KB of LDS per workgroup = 3 workgroups per CU; now 34 KB = 4.)
__host__ __device__ inline size_t slot_lds_bytes(int max_tets) { return (size_t)max_tets * SP_RS * 8 + (size_t)3 * DFL_SLOT_NODES * 16 + 16; }

// Lane-major contribution descriptors (host/slotpatch.c).  Per (pass, wave) with T trips: floor(T / 2) groups of [64 lanes] x
// one 32-bit word (trips 2g and 2g + 1) and, for odd T, a tail of [64 lanes] x u16 (32 words); offsets count in UNITS of one
// trip = 32 words, so the first unit of (pass, wave w) is the sum of the trip bytes before it (one byte per (pass, wave) in lo
// (pass 0) / hi (pass 1), at most 254 each).
__device__ __forceinline__ int slot_byte_sum(unsigned x) { return (int)((x & 255u) + ((x >> 8) & 255u) + ((x >> 16) & 255u) + (x >> 24)); }
__device__ __forceinline__ int slot_first_group(unsigned lo, unsigned hi, int pass, int w) {
    const unsigned below = (1u << (8 * w)) - 1u;  // w = 0..3
    return pass ? slot_byte_sum(lo) + slot_byte_sum(hi & below) : slot_byte_sum(lo & below);
}
// this lane's descriptor word of group g (trips 2g | 2g + 1 << 16; the tail: trip T - 1 in the low half, the high half is never
// consumed) of a (pass, wave) with T trips whose first unit is u0; unconditional: past the last group it reads and drops
__device__ __forceinline__ unsigned slot_desc_word(const unsigned* __restrict__ patch_desc, int u0, int g, int T, int lane) {
    const bool tail = 2 * g + 1 == T;
    const unsigned v = patch_desc[(u0 + 2 * g) * 32 + (tail ? (lane >> 1) : lane)];
    return tail ? v >> (16 * (lane & 1)) : v;
}

// phase 1 of one tet: everything its sixteen (a,b) blocks share -> one LDS record
__device__ __forceinline__ void slot_tet_record(const double2* r, double* rec) {
    double x[12], u[12];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        x[b * 3] = r[3 * b].x; x[b * 3 + 1] = r[3 * b].y; x[b * 3 + 2] = r[3 * b + 1].x;
        u[b * 3] = r[3 * b + 1].y; u[b * 3 + 1] = r[3 * b + 2].x; u[b * 3 + 2] = r[3 * b + 2].y;
    }
    double invJ[9], shg[12], G[9], detJ;
    tet_geometry_t<true>(x, invJ, detJ, shg);
    tet_metric(shg, G);
    double gg = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) gg += G[k] * G[k];
    const double itr = rcp_nr2(G[0] + G[4] + G[8]);
#pragma unroll
    for (int k = 0; k < 12; ++k) rec[k] = shg[k];
    double su[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) su[d] = ((u[d] + u[3 + d]) + u[6 + d]) + u[9 + d];
    const double knu = kMU / kRHO;
    double s_t1 = 0.0, t0v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // u at quadrature point q (qr_wgalpha, :1648-1655): shl(b,q) = SHB + (SHA-SHB)[b == q]
        double uq[3], cv[4];
#pragma unroll
        for (int d = 0; d < 3; ++d) uq[d] = SHB * su[d] + (SHA - SHB) * u[q * 3 + d];
#pragma unroll
        for (int a = 0; a < 4; ++a) {  // shconv (:574-583)
            cv[a] = shg[a * 3] * uq[0] + shg[a * 3 + 1] * uq[1] + shg[a * 3 + 2] * uq[2];
            rec[12 + a * 4 + q] = cv[a];
        }
        // |J^-1 u|^2 (rows of J^-1 = shape gradients of nodes 1..3) and the stabilisation parameters (:587-603)
        const double y = cv[1] * cv[1] + cv[2] * cv[2] + cv[3] * cv[3] + (3.0 * knu * knu) * gg;
        t0v[q] = rsqrt_nr1(4.0 / (kDT * kDT) + y) * (1.0 / kRHO);
        rec[28 + q] = t0v[q];
        s_t1 += y * rsqrt_nr1(y) * itr;  // tauC enters the block only through its sum over the quadrature points
    }
    // the four numbers every block of the tet starts from (lhs_block_accumulate), formed once here instead of 16 times there
    const double fact2 = kDT * kALPHAF * kGAMMA;
    const double w = detJ * GW;
    rec[32] = (fact2 * kRHO) * s_t1 * w;
    rec[33] = w;
    rec[34] = (t0v[0] + t0v[1]) + (t0v[2] + t0v[3]);
    rec[35] = (4.0 * fact2 * kMU) * w;
}

// Developer timeline (probe build, dfl_tune_asm bit 2048): lane 0 of every wave of the first TRACE_WG workgroups stamps the
// shader clock at nine points of each of its first TRACE_IT patches; tools/slot_timeline.py turns them into a per-phase budget.
constexpr int TRACE_WG = 32, TRACE_IT = 48, TRACE_PT = 10;
__device__ unsigned long long g_slot_trace[TRACE_WG * TRACE_IT * 4 * TRACE_PT];
__device__ unsigned long long g_slot_wgtime[2048 * 2];  // per workgroup: cycles from entry to exit, patches walked
#define SLOT_TR(k)                                                                                                    \
    do {                                                                                                              \
        if (PROBE && (dbg & 2048) && blockIdx.x < TRACE_WG && tr_it < TRACE_IT && (threadIdx.x & 63) == 0)             \
            g_slot_trace[((blockIdx.x * TRACE_IT + tr_it) * 4 + (threadIdx.x >> 6)) * TRACE_PT + (k)] = __builtin_readcyclecounter(); \
    } while (0)

// Node records of a patch in LDS.  The tets touching a 7-node patch (~90) have ~55 distinct nodes; round 2 gathered four
// records per (patch, tet) lane straight from HBM / L2 at the start of phase 1 (1080 16-byte requests per patch, their
// latency -- 3000+ cycles of the 13 000 a patch takes, tools/slot_timeline.py -- exposed in every patch because keeping them
// in flight across phase 2 costs 48 registers).  Now ONE lane per distinct node moves its (x, u) record -- 3 x 16 B -- by
// LDS-DMA (global_load_lds_dwordx4: no destination registers) into s_nrec[piece][node], ONE PATCH AHEAD: issued right after
// the barrier that ends phase 1 (nobody reads s_nrec any more), in flight during phase 2, retired by the issuing wave's own
// s_waitcnt vmcnt(0) placed BEFORE its first row store of the patch (the counter is in order: a wait placed after the stores
// would wait for their acknowledgement too), visible to everybody behind the end-of-patch barrier.  Phase 1 then reads its
// four nodes by patch-local id (a byte each, host/slotpatch.c) from LDS.  165 requests per patch instead of 1080.
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_byte_address(const void* p) { return (unsigned)(size_t)(lds_char_t*)p; }
// one wave-instruction: lane l copies 16 B from its own global address to lds_dst + 16 l (lds_dst wave-uniform); hipcc does not
// count this load: its completion is the caller's s_waitcnt vmcnt (cdna_hip_programming.md, inline-asm LDS-DMA recipe)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int SP_NC = DFL_SLOT_NODES;  // node capacity of s_nrec
constexpr int SP_CHUNK = 8;             // consecutive patches a workgroup claims at a time

// the (x, u) records of the nodes nid (lane t: node t of the patch, t < nn) -> s_nrec; whole waves without a node skip
__device__ __forceinline__ void slot_stage_nodes(const T* __restrict__ nodep, int nid, int nn, int tphys, unsigned nrec_base) {
    const int w = __builtin_amdgcn_readfirstlane(tphys >> 6);
    if (w * 64 >= nn) return;  // wave-uniform
    if (tphys < nn) {
        const T* src = nodep + (long long)nid * 8;  // compact record: x[3] u[3] pad pad
#pragma unroll
        for (int k = 0; k < 3; ++k) glds16(src + 2 * k, nrec_base + (unsigned)((k * SP_NC + w * 64) * 16));
    }
}

// PROBE != 0 only under dfl_tune_asm (developer phase split: bit 0 skip phase 2, bit 1 skip phase 1, bit 2 skip the
// block evaluation, bit 3 skip the store, 32 every lane reads tet record 0 = no LDS bank conflicts, 2048 timeline); the
// shipped instantiation carries no probe branches
template <bool BETA0, int PROBE>
__global__ __launch_bounds__(SBLK, 4) void tet_lhs_slot_kernel(I P, const int4* __restrict__ hdr, const unsigned* __restrict__ ptet_lid,
                                                               const I* __restrict__ pnode, const I* __restrict__ slot_nz,
                                                               const unsigned* __restrict__ ldesc, const T* __restrict__ nodep,
                                                               T* __restrict__ val, T beta, int max_tets, int dbg_in,
                                                               int* __restrict__ claim) {
    static_assert(SBLK == 256, "trip bytes: four waves x two passes");
    const int dbg = PROBE ? (dbg_in & ~(1 << 30)) : 0;
    extern __shared__ __attribute__((aligned(16))) double s_tet[];
    const double2* const s_nrec = reinterpret_cast<const double2*>(s_tet + (size_t)max_tets * SP_RS);  // [3][SP_NC]
    volatile int* const s_claim = reinterpret_cast<volatile int*>(s_tet + (size_t)max_tets * SP_RS + 6 * SP_NC);  // [4], behind s_nrec
    const unsigned nrec_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte_address(s_nrec));
    // XCD-aware order: workgroup w runs on XCD w % 8; every XCD gets one contiguous range of the spatially ordered
    // patches (neighbouring patches share tets and node records -> one L2).  Inside that range the patches are CLAIMED, not
    // dealt: the hardware issues from the oldest wave first, so of the four workgroups that share a CU the one launched first
    // runs fastest -- with an equal static share each it was done after 2.75 M cycles, the last after 4.31 M, and for the last
    // third of the kernel the CUs ran half empty (tools/slot_timeline.py, gpurun_out/r3j).  A workgroup's first CHUNK of
    // SP_CHUNK consecutive patches is its own index; every further chunk comes from the XCD's counter, claimed two chunks
    // ahead of its use, so neither the atomic nor the header loads behind it are ever waited for.  (One atomic per PATCH was
    // tried first: device-scope atomics on one address retire every ~95 ns -- 36 000 of them per counter made the kernel
    // 3.4 ms whatever else it did.)  Which workgroup takes which patch has no influence on the values: a patch's rows are
    // summed by that patch alone, in a fixed order.
    const int per = (P + 7) >> 3;
    const int g8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int pbeg = xcd * per;
    const int pend = min((int)P, pbeg + per);
    const int base0 = pbeg + SP_CHUNK * (int)(blockIdx.x >> 3);  // chunk 0 of this workgroup
    int p = base0;
    if (p >= pend) return;
    const int cbase = pbeg + SP_CHUNK * g8;  // claim c of this XCD = patches cbase + SP_CHUNK c ...
    // the j-th patch of this workgroup (j / SP_CHUNK = chunk number: 0 static, the others claimed; s_claim is a ring of 4)
#define SLOT_SEQ(j) (((j) < SP_CHUNK ? base0 : cbase + SP_CHUNK * __builtin_amdgcn_readfirstlane(s_claim[((j) / SP_CHUNK) & 3])) + ((j) % SP_CHUNK))
    // Role rotation (developer A/B, DFL_SLOT_ROTATION=1; off by default): co-resident workgroups rotate which physical wave
    // plays logical wave 0 (phase 1 + the heaviest positions).  Measured slower (2.32 against 2.17 ms, gpurun_out/r3e).
    const int rot = (dbg_in & (1 << 30)) ? (int)((blockIdx.x >> 8) & 3u) : 0;
    const int t = (int)((threadIdx.x + 64u * (unsigned)rot) & (unsigned)(SBLK - 1));
    const int lane = t & 63, pr = t >> 1;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int4 zero4 = make_int4(0, 0, 0, 0);

    // ---- prologue: lists of the first patch, its node records staged and waited for ------------------------------------
    int4 h = hdr[2 * p], h2 = hdr[2 * p + 1];
    int nz0, nz1;
    unsigned lid;
    unsigned d0a, d0b, d1a, d1b;  // descriptor groups 0, 1 of pass 0 and of pass 1 (two trips each)
    {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(h2.y), hi = (unsigned)__builtin_amdgcn_readfirstlane(h2.z);
        const int g0 = slot_first_group(lo, hi, 0, w), g1 = slot_first_group(lo, hi, 1, w);
        const int T0 = (int)((lo >> (8 * w)) & 255u), T1 = (int)((hi >> (8 * w)) & 255u);
        const unsigned* lp = ldesc + (long long)__builtin_amdgcn_readfirstlane(h2.x) * 32;
        const int nt0 = h.y & 0xffff, nn0 = h.y >> 16;
        nz0 = slot_nz[h.z + max(0, min(pr, h.w - 1))];
        nz1 = slot_nz[h.z + max(0, min(pr + SBLK / 2, h.w - 1))];
        d0a = slot_desc_word(lp, g0, 0, T0, lane); d0b = slot_desc_word(lp, g0, 1, T0, lane);
        d1a = slot_desc_word(lp, g1, 0, T1, lane); d1b = slot_desc_word(lp, g1, 1, T1, lane);
        lid = ptet_lid[h.x + max(0, min(t, nt0 - 1))];
        const int nid0 = pnode[h2.w + max(0, min(t, nn0 - 1))];
        slot_stage_nodes(nodep, nid0, nn0, t, nrec_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) s_claim[1] = atomicAdd(claim + xcd, 1);  // the second chunk of this workgroup
    // the first patch's lists have arrived before the loop is entered: inside the loop the same registers carry the next
    // patch's (already waited for), and the compiler merges the two states
    asm volatile("" ::"v"(lid), "v"(nz0), "v"(nz1), "v"(d0a), "v"(d0b), "v"(d1a), "v"(d1b));
    __syncthreads();  // s_nrec of the first patch is complete (every wave waited for its own LDS-DMA above); claims visible
    int seq = 0;  // position of patch p in this workgroup's sequence
    int pn = SLOT_SEQ(1);
    int pnn = SLOT_SEQ(2);
    int4 hn = zero4, hn2 = zero4;
    if (pn < pend) { hn = hdr[2 * pn]; hn2 = hdr[2 * pn + 1]; }

    int tr_it = 0;
    const unsigned long long tr_begin = (PROBE && (dbg & 2048)) ? __builtin_readcyclecounter() : 0ull;
    for (;;) {
        SLOT_TR(0);
        const bool has_next = pn < pend;
        const int nt = h.y & 0xffff;
        const int np = __builtin_amdgcn_readfirstlane(h.w);
        const unsigned tlo = (unsigned)__builtin_amdgcn_readfirstlane(h2.y), thi = (unsigned)__builtin_amdgcn_readfirstlane(h2.z);
        // (a) the lists of the NEXT patch into registers: local node ids of this lane's tet, this lane's node of the node
        // table, slot map and descriptors.  Every load is unconditional (indices clamped into the patch; past the last patch the
        // header is all zero and patch 0's first entries are read and dropped) and nothing is computed from the results here
        int4 hnn = zero4, hnn2 = zero4;
        if (has_next && pnn < pend) { hnn = hdr[2 * pnn]; hnn2 = hdr[2 * pnn + 1]; }
        const bool claims = (seq % SP_CHUNK) == 0;  // at the start of chunk c: claim chunk c + 2
        int my_claim = 0;
        if (claims && threadIdx.x == 0) my_claim = atomicAdd(claim + xcd, 1);
        const unsigned nlo = (unsigned)__builtin_amdgcn_readfirstlane(hn2.y), nhi = (unsigned)__builtin_amdgcn_readfirstlane(hn2.z);
        const int gn0 = slot_first_group(nlo, nhi, 0, w), gn1 = slot_first_group(nlo, nhi, 1, w);
        const int Tn0 = (int)((nlo >> (8 * w)) & 255u), Tn1 = (int)((nhi >> (8 * w)) & 255u);
        const unsigned* lpn = ldesc + (long long)__builtin_amdgcn_readfirstlane(hn2.x) * 32;
        const int ntn = hn.y & 0xffff, nnn = hn.y >> 16;
        const unsigned lidn = ptet_lid[hn.x + max(0, min(t, ntn - 1))];
        const int nidn = pnode[hn2.w + max(0, min(t, nnn - 1))];
        const int nzn0 = slot_nz[hn.z + max(0, min(pr, hn.w - 1))];
        const int nzn1 = slot_nz[hn.z + max(0, min(pr + SBLK / 2, hn.w - 1))];
        const unsigned dn0a = slot_desc_word(lpn, gn0, 0, Tn0, lane), dn0b = slot_desc_word(lpn, gn0, 1, Tn0, lane);
        const unsigned dn1a = slot_desc_word(lpn, gn1, 0, Tn1, lane), dn1b = slot_desc_word(lpn, gn1, 1, Tn1, lane);
        // (b) phase 1: one lane per (patch, tet), node records from LDS by patch-local id
        SLOT_TR(1);
        if (t < nt && !((PROBE & 2) && (dbg & 2))) {
            double2 r[12];
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int n = (int)((lid >> (8 * b4)) & 255u);
                r[3 * b4] = s_nrec[n];
                r[3 * b4 + 1] = s_nrec[SP_NC + n];
                r[3 * b4 + 2] = s_nrec[2 * SP_NC + n];
            }
            slot_tet_record(r, s_tet + t * SP_RS);
        }
        SLOT_TR(2);
        __syncthreads();  // tet records complete; nobody reads s_nrec any more
        SLOT_TR(3);
        // (c) everything requested for the next patch has arrived
        asm volatile("" ::"v"(lidn), "v"(nidn), "v"(nzn0), "v"(nzn1), "v"(dn0a), "v"(dn0b), "v"(dn1a), "v"(dn1b));
        SLOT_TR(4);
        if (claims && threadIdx.x == 0) s_claim[(seq / SP_CHUNK + 2) & 3] = my_claim;  // (returned with the lists above); first read chunks later
        // (d) the next patch's node records start their way into s_nrec; they land during phase 2
        if (has_next) slot_stage_nodes(nodep, nidn, nnn, t, nrec_base);
        bool dma_retired = !has_next;

        // ---- phase 2: one lane pair per slot position, two passes over the positions ---------------------------------------
        if (!((PROBE & 1) && (dbg & 1))) {
            const int j = t & 1;
            const bool hi1 = j != 0;
            const int g0 = slot_first_group(tlo, thi, 0, w), g1 = slot_first_group(tlo, thi, 1, w);
            const unsigned* lp = ldesc + (long long)__builtin_amdgcn_readfirstlane(h2.x) * 32;
#pragma nounroll
            for (int pass = 0; pass < 2; ++pass) {
                const int trips = (int)(((pass ? thi : tlo) >> (8 * w)) & 255u);  // the same for the 32 pairs of this wave
                if (trips == 0) break;  // positions are dealt to the waves pass by pass: none in this pass, none in the next
                const int pos = pass * (SBLK / 2) + pr;
                int nzr = pass ? nz1 : nz0;  // bit 30: first pair of a split slot, bit 31: one of its other three pairs
                if (pos >= np) nzr = (int)0x80000000;  // past the patch: nothing to store (its descriptors are all 0xFFFF)
                const long long nz = nzr & 0x3fffffff;
                unsigned dlo = pass ? d1a : d0a, dhi = pass ? d1b : d0b;
                const int gb = pass ? g1 : g0;
                double acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0;
                // one contribution: descriptor d = (local tet << 4) | (a << 2) | b, its block added to acc
                auto contribute = [&](unsigned d) {
                    const int aa = (d >> 2) & 3, bb = d & 3;
                    const double* rec = s_tet + (((PROBE & 4) && (dbg & 32)) ? 0 : (d >> 4)) * SP_RS;  // probe 32: no bank conflicts
                    double ga[3], gb3[3], t0q[4], ca[4], cb[4];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        ga[i] = rec[aa * 3 + i];
                        gb3[i] = rec[bb * 3 + i];
                    }
                    const d2a* r2 = reinterpret_cast<const d2a*>(rec);
                    const d2a ca01 = r2[6 + aa * 2], ca23 = r2[7 + aa * 2], cb01 = r2[6 + bb * 2], cb23 = r2[7 + bb * 2];
                    const d2a ta01 = r2[14], ta23 = r2[15], sc = r2[16], sk = r2[17];
                    const double cb_a = rec[12 + bb * 4 + aa], ca_b = rec[12 + aa * 4 + bb], t0_b = rec[28 + bb];
                    ca[0] = ca01.x; ca[1] = ca01.y; ca[2] = ca23.x; ca[3] = ca23.y;
                    cb[0] = cb01.x; cb[1] = cb01.y; cb[2] = cb23.x; cb[3] = cb23.y;
                    t0q[0] = ta01.x; t0q[1] = ta01.y; t0q[2] = ta23.x; t0q[3] = ta23.y;
                    if ((PROBE & 4) && (dbg & 4)) {
                        acc[0] += ga[0] + gb3[1] + ca[2] + cb[3] + t0q[0] + sc.x + sc.y + sk.x + sk.y + cb_a + ca_b + t0_b;
                    } else {
                        lhs_block_accumulate(aa == bb, ga, gb3, sc.y, sk.y, sc.x, sk.x, t0q, ca, cb, cb_a, ca_b, t0_b, acc);
                    }
                };
                // Up to four trips come out of the two prefetched descriptor words: NO memory instruction in that loop, so the
                // compiler places no s_waitcnt vmcnt in it -- with a load inside (the refill below) it waits vmcnt(0) at the top
                // of EVERY trip, i.e. the first trip of a patch waits for the LDS-DMA issued a moment ago, and the first trip of a
                // second pass for the first pass's row stores.  More than four trips (a part of a slot with > 8 contributions:
                // rare) take the general loop.
                if (trips <= 4) {
#pragma nounroll
                    for (int k = 0; k < trips; ++k) {
                        const unsigned d = dlo & 0xffffu;
                        dlo = (dlo >> 16) | (dhi << 16);
                        dhi >>= 16;
                        if (d != 0xffffu) contribute(d);
                    }
                } else {
#pragma nounroll
                    for (int k = 0; k < trips; ++k) {
                        if (k >= 4 && !(k & 1)) dlo = slot_desc_word(lp, gb, k >> 1, trips, lane);
                        const unsigned d = dlo & 0xffffu;
                        dlo = (dlo >> 16) | (dhi << 16);
                        dhi >>= 16;
                        if (d != 0xffffu) contribute(d);
                    }
                }
                if (pass == 0) SLOT_TR(5);
                // reduce-scatter inside the pair: lane j ends up with the 16-byte pieces {j, j + 2, j + 4, j + 6} of the
                // 128-byte line (entries 4k + 2j, 4k + 2j + 1) summed over both lanes, so that every store instruction
                // of a pair covers one contiguous 32-byte sector
                double2 e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k].x = (hi1 ? acc[4 * k + 2] : acc[4 * k]) + dpp_quad<0xB1>(hi1 ? acc[4 * k] : acc[4 * k + 2]);
                    e[k].y = (hi1 ? acc[4 * k + 3] : acc[4 * k + 1]) + dpp_quad<0xB1>(hi1 ? acc[4 * k + 1] : acc[4 * k + 3]);
                }
                // split slot (host/slotpatch.c): its four parts sit in four adjacent pairs (half a 16-lane DPP row, aligned);
                // lane j of the first pair collects lane j of the others: (p0 + p1) + (p2 + p3), two row shifts.  The branch
                // is on a WAVE-uniform condition: the split positions are ranked first, so in a 7-node patch they all sit in
                // one wave and the other three skip the 32 DPP moves + 16 adds (left to the compiler the per-lane `if` became
                // predicated code that every position paid for: 65 of the 116 instructions of this epilogue)
                if (__builtin_amdgcn_ballot_w64((nzr & 0xC0000000) != 0) != 0ull) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double sx = dpp_quad<0x102>(e[k].x), sy = dpp_quad<0x102>(e[k].y);
                        if (nzr & 0xC0000000) { e[k].x += sx; e[k].y += sy; }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double sx = dpp_quad<0x104>(e[k].x), sy = dpp_quad<0x104>(e[k].y);
                        if (nzr & 0xC0000000) { e[k].x += sx; e[k].y += sy; }
                    }
                }
                // this wave's LDS-DMA of the next patch's node records must be retired BEFORE the first row store: the
                // counter is in order, a wait behind the stores would wait for their acknowledgement as well
                if (!dma_retired) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    dma_retired = true;
                }
                if (nzr < 0) continue;  // followers of a split slot (and positions past the patch): only the first pair stores
                if ((PROBE & 8) && (dbg & 8)) {
                    if (e[0].x == 1.2345e300) val[nz] = e[0].x + e[0].y + e[1].x + e[1].y + e[2].x + e[2].y + e[3].x + e[3].y;
                    continue;
                }
                double2* dst = reinterpret_cast<double2*>(val + nz * 16) + j;
                if (!BETA0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2 o = dst[2 * k];
                        e[k].x += beta * o.x; e[k].y += beta * o.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[2 * k] = e[k];
            }
        }
        if (!dma_retired) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a wave without positions in this patch
        SLOT_TR(6);
        if (!has_next) {
            if (PROBE && (dbg & 2048) && threadIdx.x == 0 && blockIdx.x < 2048) {
                g_slot_wgtime[2 * blockIdx.x] = __builtin_readcyclecounter() - tr_begin;
                g_slot_wgtime[2 * blockIdx.x + 1] = (unsigned long long)(tr_it + 1);
            }
            break;
        }
        lid = lidn;
        nz0 = nzn0; nz1 = nzn1;
        d0a = dn0a; d0b = dn0b; d1a = dn1a; d1b = dn1b;
        __syncthreads();  // every wave is done with the tet records of this patch; s_nrec holds the next patch's nodes
        SLOT_TR(7);
        ++tr_it;
        h = hn; h2 = hn2; hn = hnn; hn2 = hnn2;
        p = pn; pn = pnn;
        ++seq;
        // the patch two ahead: the next one of its chunk, or (every SP_CHUNK-th time) the first one of a claimed chunk
        if (((seq + 2) % SP_CHUNK) != 0) pnn = pnn + 1;
        else pnn = SLOT_SEQ(seq + 2);
    }
#undef SLOT_SEQ
}

// ====================================================================================
//  Residual, wave-per-patch form (schedule 4).  Same arithmetic as tet_rhs_patch_kernel (k_assemble.hip): 4 lanes per
//  tet (lane = vertex for the result, quadrature point for the weak form; assemble.cu:761-924), node records staged in
//  LDS once per patch, per-(tet, vertex) results summed per patch node in adjacency order, one partial record per patch
//  node.  What changes is the schedule: ONE WAVE owns a patch of <= TETS tets / <= NODES nodes, everything it shares is
//  wave-local (no workgroup barrier), and the padded layout of host/patch.c makes every index list of a patch
//  addressable from the patch id, so the loads of a patch are a two-hop chain (lists -> node records).  Many independent
//  waves per CU then hide those two hops.
// ====================================================================================
template <int TETS, int NODES>
__global__ __launch_bounds__(256) void tet_rhs_wave_kernel(I P, const I* __restrict__ cnt, const I* __restrict__ pnode,
                                                          const unsigned char* __restrict__ lien,
                                                          const unsigned short* __restrict__ adj,
                                                          const unsigned short* __restrict__ adj_start,
                                                          const T* __restrict__ nodep, T* __restrict__ partial) {
    constexpr int RS = NV + 1;  // padded record
    __shared__ double s_rec[4][NODES][RS];
    __shared__ double s_out[4][6][TETS * 4];
    __shared__ unsigned short s_adj[4][TETS * 4];
    __shared__ unsigned short s_st[4][NODES + 2];
    __shared__ int s_node[4][NODES];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // XCD-aware order over waves: workgroup b runs on XCD b % 8; every XCD gets one contiguous range of patches
    const int per = ((P + 31) >> 5) << 2;  // patches per XCD, a multiple of 4
    const int pid = (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + w;
    if (pid >= P) return;  // whole waves leave; nothing below synchronises across waves
    const int c = cnt[pid];
    const int ne = c & 0xffff, nn = c >> 16;
    const long long e0 = (long long)pid * TETS, n0 = (long long)pid * NODES;
    // hop 1: every list of the patch, addressed from the patch id
    if (lane < nn) s_node[w][lane] = pnode[n0 + lane];
    uchar4 lnv[TETS / 16];
#pragma unroll
    for (int ps = 0; ps < TETS / 16; ++ps) lnv[ps] = *reinterpret_cast<const uchar4*>(lien + (e0 + ps * 16 + (lane >> 2)) * 4);
    for (int k = lane; k < ne * 4; k += 64) s_adj[w][k] = adj[e0 * 4 + k];
    for (int k = lane; k <= nn; k += 64) s_st[w][k] = adj_start[n0 + pid + k];
    WAVE_SYNC();
    // hop 2: node records, 7 x 16 B per node
    for (int k = lane; k < nn * 7; k += 64) {
        const int ln = k / 7, part = k - ln * 7;
        const long long node = s_node[w][ln];
        const double2 v = reinterpret_cast<const double2*>(nodep + node * NREC)[part];
        s_rec[w][ln][2 * part] = v.x;
        s_rec[w][ln][2 * part + 1] = v.y;
    }
    WAVE_SYNC();
    const int a = lane & 3;
#pragma unroll
    for (int ps = 0; ps < TETS / 16; ++ps) {
        const int le = ps * 16 + (lane >> 2);
        if (le < ne) {  // whole quads
            const double* r[4] = {s_rec[w][lnv[ps].x], s_rec[w][lnv[ps].y], s_rec[w][lnv[ps].z], s_rec[w][lnv[ps].w]};
            double mine[6];
            rhs_quad(r, a, mine);
#pragma unroll
            for (int j = 0; j < 6; ++j) s_out[w][j][le * 4 + a] = mine[j];
        }
    }
    WAVE_SYNC();
    // ordered sum per patch node: contributions in ascending local tet order
    for (int k = lane; k < nn * 6; k += 64) {
        const int ln = k / 6, j = k - ln * 6;
        double sum = 0.0;
        for (int q = s_st[w][ln]; q < s_st[w][ln + 1]; ++q) sum += s_out[w][j][s_adj[w][q]];
        partial[n0 * 6 + k] = sum;
    }
}


// ====================================================================================
//  Residual, lane-per-tet form (schedule 4 with 64-tet patches).  One WAVE owns a padded patch of <= 64 tets / <= NODES
//  nodes and ONE LANE a tet: for a linear tet only the values at the quadrature point change from point to point --
//  geometry, gradients and the vertex sums are computed once per tet instead of once per (tet, point) lane as in the
//  4-lanes-per-tet kernels -- and the sum over the four points of shl(a,q) X(q) + shg[a].Y(q) (assemble.cu:761-924)
//  collapses to  SHB sum_q X(q) + (SHA-SHB) X(a) + shg[a].sum_q Y(q).  Same terms as rhs_quad, different association.
//  The waves are PERSISTENT and software-pipelined: while patch p is computed from LDS, the node records of patch p+1
//  are in flight into registers (its node ids were loaded one patch earlier) and the index lists of patch p+2 are
//  requested, so neither hop of the dependent chain (lists -> node ids -> records) is waited for.  The LDS slice of a wave
//  holds the node records during the element pass and is reused for the per-(tet, vertex) results of the ordered sum.
// ====================================================================================
struct LaneLists {
    int c, nid;   // requested two patches ahead (the node ids feed the record gather of the next patch)
    unsigned ln;  // 4 local node ids of this lane's tet, one byte each; this and the rest: one patch ahead
    uint4 sub;    // two sub-lists of the ordered sum (4 result slots each, u16)
    unsigned st0, st1;
};

// every load unconditional (padded layout: all addresses valid) so that the compiler can count the loads in flight exactly
template <int NODES, bool HEAD_ONLY>
__device__ __forceinline__ LaneLists lane_load_lists(int pp, int lane, const I* __restrict__ cnt, const I* __restrict__ pnode,
                                                     const unsigned char* __restrict__ lien, const unsigned short* __restrict__ sub4,
                                                     const unsigned short* __restrict__ sub_start) {
    static_assert(NODES == 64, "one node id per lane");
    LaneLists L;
    L.c = cnt[pp];
    L.nid = pnode[(long long)pp * NODES + lane];
    if (HEAD_ONLY) return L;
    L.ln = reinterpret_cast<const unsigned*>(lien)[(long long)pp * 64 + lane];
    L.sub = reinterpret_cast<const uint4*>(sub4 + (long long)pp * 512)[lane];
    const unsigned short* st = sub_start + (long long)pp * (NODES + 1);
    L.st0 = st[lane];
    L.st1 = st[NODES];  // the last entry (= number of sub-lists), same address for every lane
    return L;
}

// PROBE (developer phase split under dfl_tune_asm): bit 128 skip the element pass, 256 skip the ordered sum, 512 skip the
// record gather, 1024 skip the stores; the shipped instantiation carries none of these branches
// DIRECT: the node values are gathered from the caller's arrays (xg, and the reference-layout state vectors wg, dwg) as 14
// 8-byte pieces per node instead of 7 16-byte pieces of a packed record: an F-only assembly call then needs no pack pass
// SUM6: the ordered sum with one lane per sub-list / per node carrying all six components (16-byte LDS accesses on slot-major
// results) instead of one lane per (sub-list, component) / (node, component): the same additions in the same order, 32 instead
// of 72 LDS instructions per patch at level 1 and one trip instead of six at level 2
template <int NODES, int WPS, bool PROBE, int WPB = 4, bool DIRECT = false, bool SUM6 = false>  // WPB = waves per workgroup; WPS = waves per SIMD the registers are budgeted for (1: no spills; 2: 256 VGPRs)
__global__ __launch_bounds__(64 * WPB, WPS) void tet_rhs_lane_kernel(I P, const I* __restrict__ cnt, const I* __restrict__ pnode,
                                                             const unsigned char* __restrict__ lien,
                                                             const unsigned short* __restrict__ sub4,
                                                             const unsigned short* __restrict__ sub_start,
                                                             const T* __restrict__ nodep, T* __restrict__ partial, int dbg_in,
                                                             unsigned long long* __restrict__ wtime, const T* __restrict__ xg = nullptr,
                                                             const T* __restrict__ wg = nullptr, const T* __restrict__ dwg = nullptr,
                                                             I Nn = 0) {
    const int dbg = PROBE ? dbg_in : 0;
    // developer probe (DFL_RHS_WTIME=1, tools/rhs_wavetime.py): cycles and patches per persistent wave.  Finding (round 3): the
    // workgroup launched first on a CU finishes its equal share after 1.35 M cycles, the second after 1.65 M (the hardware issues
    // from the oldest wave first); claiming chunks of 8 patches from per-XCD counters levels that (1.64 / 1.72 M) but costs a
    // memory operation and registers per iteration (18 spilled): 1.09 ms against 1.05 ms per F assembly -- not adopted; the
    // WPB == 8 build claims from an LDS counter of its workgroup instead (below), which costs neither.
    const unsigned long long w_begin = wtime ? __builtin_readcyclecounter() : 0ull;
    int w_patches = 0;
    constexpr int RS = NV + 1;                 // padded node record in LDS
    constexpr int NJ = (NODES * 7 + 63) / 64;  // 16-byte pieces of the node records per lane
    constexpr int OS = 260;                    // stride of one component of the parked results; slot 256 holds 0.0
    constexpr int BUF = NODES * RS > 6 * OS ? NODES * RS : 6 * OS;
    __shared__ __attribute__((aligned(16))) double s_buf[WPB][BUF];
    __shared__ __attribute__((aligned(16))) double s_subv[WPB][(128 + 4) * 6];  // sub-list sums, [sub-list][component]
    __shared__ __attribute__((aligned(16))) unsigned short s_sub4[WPB][512];
    __shared__ unsigned short s_st[WPB][NODES + 2];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // w in an SGPR: patch ids,
                                                                                              // LDS bases scalar
    // XCD-aware persistent ranges: workgroup b runs on XCD b % 8; every XCD walks one contiguous range of patches
    const int per = (P + 7) >> 3;
    const int pbeg = (blockIdx.x & 7) * per;
    const int pend = min((int)P, pbeg + per);
    const int wx = (gridDim.x >> 3) * WPB;  // waves per XCD
    // WPB == 8 (one workgroup per CU): the first two patches of a wave are its static ones, every later one is claimed from
    // the workgroup's LDS counter -- item v of the workgroup is slot v % WPB of its round v / WPB, the same set of patches as
    // the static walk.  The hardware issues from the OLDEST wave of a SIMD first, so of the two waves sharing a SIMD the one
    // launched first finished an equal share after 1.35 M cycles and the other after 1.65 M; with the claim both run to the end.
    constexpr bool CLAIM = WPB == 8;
    __shared__ int s_next;
    if (CLAIM) {
        if (threadIdx.x == 0) s_next = 2 * WPB;
        __syncthreads();
    }
    const int pwg = pbeg + (blockIdx.x >> 3) * WPB;
    int p = pwg + w;
    if (p >= pend) return;  // whole waves leave; nothing below synchronises across waves
    double* const sb = s_buf[w];

    // the pieces of the node records this lane fetches: piece k = j * 64 + lane of node k / 7 (DIRECT: value k % 14 of node
    // k / 14, in record order x(3) u(3) phi T du(3) p dphi dT -- pack_nodes_kernel's)
    constexpr int NJR = DIRECT ? 2 * NJ : NJ;
    double2 rv[DIRECT ? 1 : NJ];
    double rd[DIRECT ? NJR : 1];
#define G_LN(j) ((j * 64 + lv) / 7)
#define G_PART(j) ((j * 64 + lv) - 7 * G_LN(j))
#define D_LN(j) ((j * 64 + lv) / 14)
#define D_Q(j) ((j * 64 + lv) - 14 * D_LN(j))
    const long long bxg = (long long)xg, bwg = (long long)wg, bdw = (long long)dwg;
    auto gather = [&](int nid) {
        if (DIRECT) {
            int lv = lane;
            asm volatile("" : "+v"(lv));  // the per-(lane, j) constants are recomputed here, not kept in 60 registers across the loop
#pragma unroll
            for (int j = 0; j < NJR; ++j) {
                int node = __shfl(nid, D_LN(j) & 63, WAVE);
                node = node < 0 ? 0 : node;
                const int q = D_Q(j);
                // x: xg[3n+q]; u: wg[3n+q-3]; phi, T: wg[(q-2)N+n]; du: dwg[3n+q-8]; p, dphi, dT: dwg[(q-8)N+n] -- as integer
                // arithmetic on the three base addresses (a select among pointers becomes a lookup table in scratch)
                const long long m3 = -(long long)(q >= 3), m8 = -(long long)(q >= 8);
                const long long base = bxg + (m3 & (bwg - bxg)) + (m8 & (bdw - bwg));
                const int sub = q - (int)(3 & m3) - (int)(5 & m8) + (q >= 6 && q < 8 ? 1 : 0);  // 0 1 2 | 0 1 2 | 4 5 | 0 1 2 | 3 4 5
                const bool aos = sub < 3 && !(q >= 6 && q < 8);
                const long long off = aos ? 3LL * node + sub : (long long)sub * Nn + node;
                rd[j] = *reinterpret_cast<const T*>(base + 8 * off);
            }
            return;
        }
        int lv = lane;
        if (WPB == 8) asm volatile("" : "+v"(lv));  // (as above; the 4-wave builds are kept as they were measured)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int node = __shfl(nid, G_LN(j) & 63, WAVE);  // executed by every lane
            node = node < 0 ? 0 : node;                  // padding entries fetch record 0 (never staged): every load is issued
            if (PROBE && (dbg & 512)) rv[j] = make_double2(1.0 + node, 2.0);
            else rv[j] = reinterpret_cast<const double2*>(nodep + (long long)node * NREC)[G_PART(j)];
        }
    };
    auto stage = [&](const LaneLists& L) {  // registers -> the wave's LDS slice
        const int nn = L.c >> 16;
        if (DIRECT) {
            int lv = lane;
            asm volatile("" : "+v"(lv));
#pragma unroll
            for (int j = 0; j < NJR; ++j)
                if (D_LN(j) < nn) sb[D_LN(j) * RS + D_Q(j)] = rd[j];
        } else {
            int lv = lane;
            if (WPB == 8) asm volatile("" : "+v"(lv));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (G_LN(j) < nn) {
                    sb[G_LN(j) * RS + 2 * G_PART(j)] = rv[j].x;
                    sb[G_LN(j) * RS + 2 * G_PART(j) + 1] = rv[j].y;
                }
        }
        reinterpret_cast<uint4*>(s_sub4[w])[lane] = L.sub;
        s_st[w][lane] = (unsigned short)L.st0;
        if (lane == 0) s_st[w][NODES] = (unsigned short)L.st1;
    };

    LaneLists L1 = lane_load_lists<NODES, false>(p, lane, cnt, pnode, lien, sub4, sub_start);
    gather(L1.nid);
    int pn = p + wx;
    bool has_n = pn < pend;
    LaneLists L2 = lane_load_lists<NODES, true>(min(pn, pend - 1), lane, cnt, pnode, lien, sub4, sub_start);
    stage(L1);
    // the head of the second patch's lists is complete before the loop is entered: inside the loop the same registers
    // are waited for with the stores of the previous patch still in flight, and the compiler merges the two states
    asm volatile("" ::"v"(L2.c), "v"(L2.nid));

    for (;;) {
        const int ne = L1.c & 0xffff, nn = L1.c >> 16;
        const long long n0 = (long long)p * NODES;
        // hop 2 of the next patch and hop 1 of the one after it: in flight during the element pass
        WAVE_SYNC();

        double out[24];  // out[a * 6 + j]
        if (PROBE && (dbg & 128)) {
#pragma unroll
            for (int k = 0; k < 24; ++k) out[k] = sb[(lane + k) & 63];
        } else if (lane < ne) {
            const double* r[4] = {sb + (L1.ln & 255u) * RS, sb + ((L1.ln >> 8) & 255u) * RS, sb + ((L1.ln >> 16) & 255u) * RS,
                                  sb + (L1.ln >> 24) * RS};
            double shg[12], detJ, gg, itr;
            {
                double x[12], invJ[9], G[9];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int d = 0; d < 3; ++d) x[b * 3 + d] = r[b][d];
                tet_geometry_t<true>(x, invJ, detJ, shg);  // reciprocal / rsqrt: hardware seed + Newton (asm_device.hpp), as the J kernel
                tet_metric(shg, G);
                gg = 0.0;
#pragma unroll
                for (int k = 0; k < 9; ++k) gg += G[k] * G[k];
                itr = rcp_nr2(G[0] + G[4] + G[8]);
            }
            const double DS = SHA - SHB;
            const double fb[3] = {FB0, FB1, FB2};
            const double mu = kMU / kRHO, kappa = kKAPPA / (kRHO * kCP);
            const double t0 = 4.0 / (kDT * kDT);
            const double wq = GW * detJ;
            double Su[3] = {0.0, 0.0, 0.0};  // vertex sums of u: both passes
            // ---- pass 1: momentum + continuity rows.  out[a*6+j] first collects (SHA-SHB) X_j(a) (the point q = a), then
            // takes SHB sum_q X_j + shg[a].sum_q Y_j
            {
                double grad[12], Sp = 0.0, Sd[3] = {0.0, 0.0, 0.0};  // gradients of u0 u1 u2 p: the same at every point
#pragma unroll
                for (int k = 0; k < 12; ++k) grad[k] = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double v[4] = {r[b][3], r[b][4], r[b][5], r[b][11]};
#pragma unroll
                    for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                        for (int d = 0; d < 3; ++d) grad[d + 3 * comp] += shg[d + 3 * b] * v[comp];
                    Su[0] += v[0]; Su[1] += v[1]; Su[2] += v[2]; Sp += v[3];
                    Sd[0] += r[b][8]; Sd[1] += r[b][9]; Sd[2] += r[b][10];
                }
                const double divu = grad[0] + grad[4] + grad[8];
                double SY[12], SX[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 12; ++k) SY[k] = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double uadv[3], qd[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        uadv[i] = SHB * Su[i] + DS * r[q][3 + i];
                        qd[i] = SHB * Sd[i] + DS * r[q][8 + i];
                    }
                    const double pq = SHB * Sp + DS * r[q][11];
                    double rLi[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        rLi[i] = kRHO * (qd[i] - fb[i]) + kRHO * uadv[0] * grad[3 * i] + kRHO * uadv[1] * grad[3 * i + 1] +
                                 kRHO * uadv[2] * grad[3 * i + 2] + grad[9 + i];
                    double t1 = 0.0;
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        const double v = shg[3 + rr] * uadv[0] + shg[6 + rr] * uadv[1] + shg[9 + rr] * uadv[2];
                        t1 += v * v;
                    }
                    const double y = t1 + 3.0 * mu * mu * gg;
                    const double tau0 = rsqrt_nr1(t0 + y) * (1.0 / kRHO);
                    const double tau1 = y * rsqrt_nr1(y) * itr;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double xi = kRHO * (qd[i] - fb[i]) + kRHO * (uadv[0] - tau0 * rLi[0]) * grad[3 * i] +
                                          kRHO * (uadv[1] - tau0 * rLi[1]) * grad[3 * i + 1] +
                                          kRHO * (uadv[2] - tau0 * rLi[2]) * grad[3 * i + 2];
                        SX[i] += xi;
                        out[q * 6 + i] = DS * xi;
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            SY[3 * i + j] += kMU * (grad[3 * i + j] + grad[3 * j + i]) + kRHO * tau0 * rLi[i] * uadv[j] -
                                             kRHO * tau0 * tau0 * rLi[i] * rLi[j];
                        SY[3 * i + i] += -pq + kRHO * tau1 * divu;
                        SY[9 + i] += tau0 * rLi[i];
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double xs = j == 3 ? divu : SHB * SX[j] + out[a * 6 + j];  // X_3 = div u at every point
                        out[a * 6 + j] = wq * (xs + shg[a * 3] * SY[3 * j] + shg[a * 3 + 1] * SY[3 * j + 1] + shg[a * 3 + 2] * SY[3 * j + 2]);
                    }
            }
            // ---- pass 2: phi and T rows (u at the points and |J^-1 u|^2 recomputed: cheaper than keeping them)
            {
                double grad[6], Sd[2] = {0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 6; ++k) grad[k] = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double v[2] = {r[b][6], r[b][7]};
#pragma unroll
                    for (int comp = 0; comp < 2; ++comp)
#pragma unroll
                        for (int d = 0; d < 3; ++d) grad[d + 3 * comp] += shg[d + 3 * b] * v[comp];
                    Sd[0] += r[b][12]; Sd[1] += r[b][13];
                }
                double SY[6], SX[2] = {0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 6; ++k) SY[k] = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double uadv[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) uadv[i] = SHB * Su[i] + DS * r[q][3 + i];
                    const double dphi = SHB * Sd[0] + DS * r[q][12], dT = SHB * Sd[1] + DS * r[q][13];
                    double t1 = 0.0;
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        const double v = shg[3 + rr] * uadv[0] + shg[6 + rr] * uadv[1] + shg[9 + rr] * uadv[2];
                        t1 += v * v;
                    }
                    const double tau2 = rsqrt_nr1(t0 + t1);
                    const double tau3 = rsqrt_nr1(t0 + t1 + 3.0 * kappa * kappa * gg) * (1.0 / (kRHO * kCP));
                    const double bp = dphi + uadv[0] * grad[0] + uadv[1] * grad[1] + uadv[2] * grad[2];
                    const double btc = kRHO * kCP * (dT + uadv[0] * grad[3] + uadv[1] * grad[4] + uadv[2] * grad[5]);
                    SX[0] += bp;
                    SX[1] += btc;
                    out[q * 6 + 4] = DS * bp;
                    out[q * 6 + 5] = DS * btc;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        SY[i] += bp * tau2 * uadv[i];
                        SY[3 + i] += btc * (kRHO * kCP * tau3) * uadv[i] + kKAPPA * grad[3 + i];
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        out[a * 6 + 4 + j] = wq * (SHB * SX[j] + out[a * 6 + 4 + j] + shg[a * 3] * SY[3 * j] + shg[a * 3 + 1] * SY[3 * j + 1] +
                                                   shg[a * 3 + 2] * SY[3 * j + 2]);
            }
        }
        // hop 2 of the next patch and hop 1 of the one after it, requested AFTER the element pass (their 40 registers are
        // free again) and in flight during the ordered sum; past the end of the range the loads are repeated on the last
        // patch and dropped: no branch around a load
        int pnn = pn + wx;
        if (CLAIM) {
            int v = 0;
            if (lane == 0) v = atomicAdd(&s_next, 1);
            v = __builtin_amdgcn_readfirstlane(v);
            pnn = pwg + (v / WPB) * wx + (v % WPB);
        }
        gather(L2.nid);
        const bool has_nn = pnn < pend;
        {
            const LaneLists T2 = lane_load_lists<NODES, false>(min(pn, pend - 1), lane, cnt, pnode, lien, sub4, sub_start);
            L2.ln = T2.ln; L2.sub = T2.sub; L2.st0 = T2.st0; L2.st1 = T2.st1;  // (c and nid are re-read: same values)
        }
        const LaneLists L3 = lane_load_lists<NODES, true>(min(pnn, pend - 1), lane, cnt, pnode, lien, sub4, sub_start);
        WAVE_SYNC();  // every lane is done with the node records: the slice now takes the per-(tet, vertex) results
        if (SUM6) {
            static_assert(!SUM6 || BUF >= 257 * 6, "slot-major results + the zero slot");
            if (lane < ne) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {  // result slot lane * 4 + a: six components, 48 B
                    double2* dst = reinterpret_cast<double2*>(sb + (lane * 4 + a) * 6);
                    dst[0] = make_double2(out[a * 6], out[a * 6 + 1]);
                    dst[1] = make_double2(out[a * 6 + 2], out[a * 6 + 3]);
                    dst[2] = make_double2(out[a * 6 + 4], out[a * 6 + 5]);
                }
            }
            if (lane < 6) sb[256 * 6 + lane] = 0.0;  // the slot the padding of a sub-list points at
            WAVE_SYNC();
            const int ns = (int)s_st[w][NODES];  // number of sub-lists (<= 128)
            for (int sl = lane; sl < ns; sl += 64) {
                const uint2 id = reinterpret_cast<const uint2*>(s_sub4[w])[sl];
                const double2* p0 = reinterpret_cast<const double2*>(sb + (id.x & 0xffffu) * 6);
                const double2* p1 = reinterpret_cast<const double2*>(sb + (id.x >> 16) * 6);
                const double2* p2 = reinterpret_cast<const double2*>(sb + (id.y & 0xffffu) * 6);
                const double2* p3 = reinterpret_cast<const double2*>(sb + (id.y >> 16) * 6);
                double2* d = reinterpret_cast<double2*>(s_subv[w] + sl * 6);
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const double2 v0 = p0[h], v1 = p1[h], v2 = p2[h], v3 = p3[h];
                    d[h] = make_double2(((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y);
                }
            }
            WAVE_SYNC();
            double2 f[3] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
            if (lane < nn) {
                const int s1 = s_st[w][lane + 1];
                for (int q = s_st[w][lane]; q < s1; ++q) {  // the node's 1-6 sub-list sums, in order
                    const double2* v = reinterpret_cast<const double2*>(s_subv[w] + q * 6);
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const double2 t = v[h];
                        f[h].x += t.x;
                        f[h].y += t.y;
                    }
                }
            }
            // every lane stores its node's 48 bytes (zeros past the last node): a fixed number of unconditional stores
            double2* dst = reinterpret_cast<double2*>(partial + (n0 + lane) * 6);
            dst[0] = f[0];
            dst[1] = f[1];
            dst[2] = f[2];
        } else {
        if (lane < ne) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double2* dst = reinterpret_cast<double2*>(sb + j * OS + lane * 4);
                dst[0] = make_double2(out[j], out[6 + j]);
                dst[1] = make_double2(out[12 + j], out[18 + j]);
            }
        }
        if (lane < 6) sb[lane * OS + 256] = 0.0;  // the slot the padding of a sub-list points at
        WAVE_SYNC();
        // Ordered sum per patch node, two fixed-shape levels (host/patch.c): level 1 -- every (sub-list, component) pair sums
        // its 4 result slots, the same work for every lane whatever the valence of the node (a per-node walk made the
        // whole wave wait for its 24-tet nodes: 0.37 of the kernel's 1.1 ms); level 2 -- every (node, component) pair adds
        // its 1-6 sub-list sums in order.  Fixed association ((a0+a1)+a2)+a3 per sub-list, then sub-lists in order.
        {
            const int ns6 = (int)s_st[w][NODES] * 6;  // entry NODES = number of sub-lists (entries >= nn repeat it)
            if (!(PROBE && (dbg & 256))) {
#pragma unroll 2
                for (int tk = lane; tk < ns6; tk += 64) {
                    const int s6 = tk / 6, j = tk - 6 * s6;
                    const uint2 id = reinterpret_cast<const uint2*>(s_sub4[w])[s6];
                    const double* o = sb + j * OS;
                    const double v0 = o[id.x & 0xffffu], v1 = o[id.x >> 16], v2 = o[id.y & 0xffffu], v3 = o[id.y >> 16];
                    s_subv[w][tk] = ((v0 + v1) + v2) + v3;
                }
            }
        }
        WAVE_SYNC();
        // (all NODES * 6 entries of the padded partial block are written, the unused ones as zeros: a fixed number of
        // unconditional stores per lane, so that waiting for the loads above never waits for these stores)
#pragma unroll
        for (int i = 0; i < NODES * 6 / 64; ++i) {
            const int k = lane + 64 * i;
            const int ln = k / 6, j = k - ln * 6;
            double sum = 0.0;
            if (PROBE && (dbg & 256)) sum = sb[k];
            else if (ln < nn) {
                const int s1 = s_st[w][ln + 1];
                for (int q = s_st[w][ln]; q < s1; q += 4) {  // 1-6 sub-lists per node: one trip, two for a 24-tet node
                    const double v0 = s_subv[w][q * 6 + j], v1 = s_subv[w][(q + 1) * 6 + j], v2 = s_subv[w][(q + 2) * 6 + j],
                                 v3 = s_subv[w][(q + 3) * 6 + j];
                    sum += v0;
                    if (q + 1 < s1) sum += v1;
                    if (q + 2 < s1) sum += v2;
                    if (q + 3 < s1) sum += v3;
                }
            }
            if (!(PROBE && (dbg & 1024))) partial[n0 * 6 + k] = sum;
            else if (sum == 1.2345e300) partial[0] = sum;
        }
        }  // !SUM6
        ++w_patches;
        if (!has_n) break;
        WAVE_SYNC();  // the sums are read: the slice and the lists take the next patch
        L1 = L2;
        stage(L1);
        p = pn;
        pn = pnn;
        L2 = L3;
        has_n = has_nn;
    }
    if (wtime && lane == 0) {
        const int gw = blockIdx.x * WPB + (threadIdx.x >> 6);
        if (gw < 4096) { wtime[2 * gw] = __builtin_readcyclecounter() - w_begin; wtime[2 * gw + 1] = (unsigned long long)w_patches; }
    }
#undef G_LN
#undef G_PART
#undef D_LN
#undef D_Q
}

}  // namespace

extern "C" {

int dfl_lhs_slot_record_bytes(void) { return SP_RS * (int)sizeof(double); }
// developer timeline of the slot kernel (probe build, dfl_tune_asm bit 2048): [TRACE_WG][TRACE_IT][4 waves][TRACE_PT] clock stamps
int dfl_slot_wgtime_fetch(unsigned long long* out, int max_entries) {
    if (max_entries < 4096) return -4096;
    DFL_GUARD(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_wgtime), sizeof(unsigned long long) * 4096, 0, hipMemcpyDeviceToHost));
    return 4096;
}
int dfl_slot_trace_fetch(unsigned long long* out, int max_entries) {
    const int n = TRACE_WG * TRACE_IT * 4 * TRACE_PT;
    if (max_entries < n) return -n;
    DFL_GUARD(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_trace), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost));
    return n;
}
int64_t dfl_lhs_slot_lds_bytes(I max_tets) { return (int64_t)slot_lds_bytes(max_tets); }

extern int g_patch_dbg;
static unsigned long long* g_rhs_wtime = nullptr;
int dfl_rhs_wtime_fetch(unsigned long long* out, int max_entries) {
    if (!g_rhs_wtime || max_entries < 8192) return 0;
    DFL_GUARD(hipMemcpy(out, g_rhs_wtime, 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 8192;
}
int g_rhs_lane_grid_cap = 0;  // developer / test knob (dfl_tune(2, n)): few workgroups make a small mesh walk the pipelined loop
void dfl_assemble_tet_lhs_slot(I npatch, const int32_t* hdr, const uint32_t* ptet_lid, const I* pnode, const I* slot_nz,
                               const uint32_t* ldesc, const T* nodep, T* val, T beta, I max_tets, void* stream) {
    if (npatch <= 0) return;
    if (max_tets > SBLK) {
        fprintf(stderr, "dfl_assemble_tet_lhs_slot: patch limit exceeded (tets %d <= %d)\n", (int)max_tets, SBLK);
        abort();
    }
    const size_t lds = slot_lds_bytes(max_tets);
    // persistent grid: as many workgroups as the device keeps resident at this LDS size (re-derived when the size changes)
    static size_t lds_set = 0;
    static int resident = 0;
    if (lds != lds_set) {
        const void* kernels[3] = {(const void*)tet_lhs_slot_kernel<true, 0>, (const void*)tet_lhs_slot_kernel<false, 0>,
                                  (const void*)tet_lhs_slot_kernel<true, 15>};
        for (int k = 0; k < 3; ++k) DFL_GUARD(hipFuncSetAttribute(kernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, cus = 0, occ = 0;
        DFL_GUARD(hipGetDevice(&dev));
        DFL_GUARD(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 8) cus = 8;
        DFL_GUARD(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernels[0], SBLK, lds));
        resident = cus * (occ < 1 ? 1 : occ);
        lds_set = lds;
        if (getenv("DFL_PATCH_VERBOSE")) fprintf(stderr, "[slot kernel] %zu B LDS per workgroup, %d resident workgroups on %d CUs\n", lds, resident, cus);
    }
    const int probe = g_patch_dbg;  // any bit: the probe build (always overwrites: beta = 0)
    int grid = resident / 8 * 8;
    const int need = 8 * ((((npatch + 7) / 8) + SP_CHUNK - 1) / SP_CHUNK);  // every workgroup starts on a chunk of its own
    if (grid > need) grid = need;
    const int4* h4 = reinterpret_cast<const int4*>(hdr);
    static int rot_on = -1;  // DFL_SLOT_ROTATION=1: co-resident workgroups rotate the wave roles (A/B; slower, see the kernel)
    if (rot_on < 0) rot_on = getenv("DFL_SLOT_ROTATION") ? 1 : 0;
    const int kflags = probe | (rot_on ? (1 << 30) : 0);
    static int* d_claim = nullptr;  // one patch counter per XCD, zeroed before every launch (stream-ordered)
    if (!d_claim) DFL_GUARD(hipMalloc((void**)&d_claim, 8 * sizeof(int)));
    DFL_GUARD(hipMemsetAsync(d_claim, 0, 8 * sizeof(int), S(stream)));
#define SLOT_LAUNCH(B0, PR) tet_lhs_slot_kernel<B0, PR><<<grid, SBLK, lds, S(stream)>>>(npatch, h4, ptet_lid, pnode, slot_nz, ldesc, nodep, val, beta, max_tets, kflags, d_claim)
    if (probe) SLOT_LAUNCH(true, 15);
    else if (beta == 0.0) SLOT_LAUNCH(true, 0);
    else SLOT_LAUNCH(false, 0);
#undef SLOT_LAUNCH
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_tet_rhs_wave(I npatch, I pad_tets, I pad_nodes, const I* cnt, const I* pnode, const unsigned char* lien,
                                const unsigned short* adj, const unsigned short* adj_start, const T* nodep, T* partial,
                                void* stream) {
    if (npatch <= 0) return;
    const int per = ((npatch + 31) >> 5) << 2;
    const int grid = 8 * (per / 4);
    if (pad_tets == 32 && pad_nodes == 48)
        tet_rhs_wave_kernel<32, 48><<<grid, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, adj, adj_start, nodep, partial);
    else if (pad_tets == 64 && pad_nodes == 64)
        tet_rhs_wave_kernel<64, 64><<<grid, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, adj, adj_start, nodep, partial);
    else if (pad_tets == 16 && pad_nodes == 32)
        tet_rhs_wave_kernel<16, 32><<<grid, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, adj, adj_start, nodep, partial);
    else {
        fprintf(stderr, "dfl_assemble_tet_rhs_wave: unsupported patch shape %d tets / %d nodes\n", (int)pad_tets, (int)pad_nodes);
        abort();
    }
    DFL_LAUNCH_CHECK();
}

// lane-per-tet kernel, persistent waves: 2 workgroups per CU (built for 2 waves per SIMD; measured 0.85 ms at 10M tets against
// 1.07 ms for the 1-wave build and 1.29 ms for the 4-lanes-per-tet wave kernel)
static void rhs_lane_launch(I npatch, const I* cnt, const I* pnode, const unsigned char* lien, const unsigned short* sub4,
                            const unsigned short* sub_start, const T* nodep, T* partial, const T* xg, const T* wg, const T* dwg,
                            I Nn, void* stream) {
    if (npatch <= 0) return;
    const bool direct = nodep == nullptr;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        DFL_GUARD(hipGetDevice(&dev));
        DFL_GUARD(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus < 8) cus = 8;
    }
    const int wps = (g_patch_dbg & 64) ? 1 : 2;  // developer A/B: bit 64 = the 1-wave-per-SIMD build (no spills, 512 registers)
    int g = (wps * cus) / 8 * 8;
    const int need = 8 * ((((npatch + 7) / 8) + 3) / 4);  // one wave per patch of an XCD's share
    if (g > need) g = need;
    if (g_rhs_lane_grid_cap > 0 && g > g_rhs_lane_grid_cap) g = (g_rhs_lane_grid_cap + 7) / 8 * 8;
    static const bool wpb8 = !(getenv("DFL_RHS_WPB") && atoi(getenv("DFL_RHS_WPB")) == 4);  // A/B: 4 = the static walk
    static unsigned long long* d_wtime = nullptr;  // DFL_RHS_WTIME=1: cycles and patches per wave (dfl_rhs_wtime_fetch)
    if (!d_wtime && getenv("DFL_RHS_WTIME")) {
        DFL_GUARD(hipMalloc((void**)&d_wtime, 8192 * sizeof(unsigned long long)));
        DFL_GUARD(hipMemset(d_wtime, 0, 8192 * sizeof(unsigned long long)));
        g_rhs_wtime = d_wtime;
    }
    if ((g_patch_dbg & ~(64 | 32)) && !direct)
        tet_rhs_lane_kernel<64, 1, true><<<g, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, g_patch_dbg, d_wtime);
    else if (wps == 2 && wpb8) {
        // one workgroup of 8 waves per CU claiming patches from an LDS counter (see the kernel): 1.60 M cycles for the
        // slowest wave against 1.72 M of two static 4-wave workgroups per CU (tools/ab_rhs_wpb.sh)
        int g8 = cus / 8 * 8;
        const int need8 = 8 * ((((npatch + 7) / 8) + 7) / 8);
        if (g8 > need8) g8 = need8;
        static const bool sum6 = !(getenv("DFL_RHS_SUM6") && atoi(getenv("DFL_RHS_SUM6")) == 0);  // developer A/B
        if (direct)
            tet_rhs_lane_kernel<64, 2, false, 8, true, true><<<g8, 512, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nullptr, partial, 0, d_wtime, xg, wg, dwg, Nn);
        else if (sum6)
            tet_rhs_lane_kernel<64, 2, false, 8, false, true><<<g8, 512, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, 0, d_wtime);
        else
            tet_rhs_lane_kernel<64, 2, false, 8><<<g8, 512, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, 0, d_wtime);
    } else if (direct) {
        fprintf(stderr, "dfl_assemble_tet_rhs_lane_direct: only the default 8-wave build gathers from the caller's arrays\n");
        abort();
    } else if (wps == 2)
        tet_rhs_lane_kernel<64, 2, false><<<g, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, 0, d_wtime);
    else
        tet_rhs_lane_kernel<64, 1, false><<<g, 256, 0, S(stream)>>>(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, 0, d_wtime);
    DFL_LAUNCH_CHECK();
}
void dfl_assemble_tet_rhs_lane(I npatch, const I* cnt, const I* pnode, const unsigned char* lien, const unsigned short* sub4,
                               const unsigned short* sub_start, const T* nodep, T* partial, void* stream) {
    if (!nodep) { fprintf(stderr, "dfl_assemble_tet_rhs_lane: NULL node records\n"); abort(); }
    rhs_lane_launch(npatch, cnt, pnode, lien, sub4, sub_start, nodep, partial, nullptr, nullptr, nullptr, 0, stream);
}
void dfl_assemble_tet_rhs_lane_direct(I npatch, const I* cnt, const I* pnode, const unsigned char* lien, const unsigned short* sub4,
                                      const unsigned short* sub_start, const T* xg, const T* wg, const T* dwg, I N, T* partial,
                                      void* stream) {
    if (!xg || !wg || !dwg) { fprintf(stderr, "dfl_assemble_tet_rhs_lane_direct: NULL input array\n"); abort(); }
    rhs_lane_launch(npatch, cnt, pnode, lien, sub4, sub_start, nullptr, partial, xg, wg, dwg, N, stream);
}

}  // extern "C"
