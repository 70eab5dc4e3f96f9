// Colored element assembly for gfx950: one launch per color batch does the whole
// chain of src/assemble.cu:1559-1738 (Jacobian -> inverse/det -> shape gradients
// -> metric -> nodal gathers -> quadrature interpolation -> element tensor ->
// race-free scatter) without the reference's 22-24 launches and ~1 KB/elem of
// intermediate global traffic per batch.
//
// HBM view (what bounds these kernels):
//   LHS : per tet 16 (a,b) blocks x one 128-byte line RMW in the block-CSR value
//         array (4096 B/tet algorithmic) + 80 B of batch-ordered ien/nzmap stream
//         + L2/MALL-resident node gathers.  The element blocks are transposed
//         through LDS so that every wave instruction of the scatter touches 8
//         whole lines (8 lanes x 16 B per line) instead of 64 partial ones.
//   RHS : per tet 4 nodes x 14 gathered f64 and 24 f64 RMW into F.
// Summation order inside a matrix entry / F entry is color order, as in the
// reference (non-atomic colored scatter, assemble.cu:188-208, matrix_impl.cu:447).
#include "asm_device.hpp"

namespace {

// ====================================================================================
//  LHS: AssembleWeakFormLHSKernel (assemble.cu:495-759) + SetBlockValueToSubmatKernel
//  (matrix_impl.cu:370-453), 16 lanes per element, 16 elements per 256-thread block.
// ====================================================================================
constexpr int EPB = 16;
constexpr int LBLK = 256;
constexpr int TRS = 257;  // padded row length of the transpose buffer


// per-element staging shared by the 16 lanes of an element (all inside one wave)
struct LhsStage {
    double u[EPB][12];
    double shg[EPB][12];
    double conv[EPB][4][4];  // [a][q]
    double tau[EPB][4][2];   // [q][tauM, tauC]
    double scal[EPB][4];     // detJ, gg, 1/tr
};

// The 4x4 (u,p) block of node pair (a,b) = (p>>2, p&3) of one tet: gather, geometry, convection at the 4
// quadrature points, stabilisation parameters, block.  `nodes` = the element's 4 node ids.
// `egeo` = the element's cached geometry record (elem_geometry_kernel: shg[12], detJ, gg, 1/tr, pad), read as
// one coalesced 128-byte line by the element's 16 lanes -- the mesh does not move inside the time loop, so the
// Jacobian inverse / shape gradients / metric are computed once per mesh instead of once per assembly.
__device__ __forceinline__ void lhs_element_block(LhsStage& S, int te, int p, bool valid, const I* __restrict__ nodes,
                                                  const T* __restrict__ egeo, const T* __restrict__ nodep, double* Bk) {
    double (*s_u)[12] = S.u;
    double (*s_shg)[12] = S.shg;
    double (*s_conv)[4][4] = S.conv;
    double (*s_tau)[4][2] = S.tau;
    double (*s_scal)[4] = S.scal;
    if (valid) {
        const double g = egeo[p];
        if (p < 12) {
            const int a = p / 3, d = p - a * 3;
            const long long node = nodes[a];
            s_shg[te][p] = g;
            s_u[te][p] = nodep[node * 16 + 3 + d];  // packed node record: u at [3..5]
        } else {
            s_scal[te][p - 12] = g;  // detJ, gg, 1/tr, pad
        }
    }
    WAVE_SYNC();

    if (valid) {  // p = a*4 + q : shconv[a] at quadrature point q (:574-583)
        const int a = p >> 2, q = p & 3;
        double uq[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s += shl(b, q) * s_u[te][b * 3 + d];  // qr_wgalpha, :1648-1655
            uq[d] = s;
        }
        double c = 0.0;
        c += s_shg[te][a * 3 + 0] * uq[0];
        c += s_shg[te][a * 3 + 1] * uq[1];
        c += s_shg[te][a * 3 + 2] * uq[2];
        s_conv[te][a][q] = c;
    }
    WAVE_SYNC();

    if (valid && p < 4) {  // stabilisation parameters at quadrature point p (:587-603)
        const double knu = kMU / kRHO;
        double tmp = 0.0;
        tmp += s_conv[te][1][p] * s_conv[te][1][p];
        tmp += s_conv[te][2][p] * s_conv[te][2][p];
        tmp += s_conv[te][3][p] * s_conv[te][3][p];
        const double gg = s_scal[te][1];
        s_tau[te][p][0] = (1.0 / sqrt(4.0 / (kDT * kDT) + tmp + 3.0 * knu * knu * gg)) / kRHO;
        s_tau[te][p][1] = sqrt(tmp + 3.0 * knu * knu * gg) * s_scal[te][2];
    }
    WAVE_SYNC();

#pragma unroll
    for (int i = 0; i < 16; ++i) Bk[i] = 0.0;
    if (valid) {
        const int aa = p >> 2, bb = p & 3;
        double ga[3], gb[3], t0[4], t1[4], ca[4], cb[4];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ga[d] = s_shg[te][aa * 3 + d];
            gb[d] = s_shg[te][bb * 3 + d];
        }
#pragma unroll
        for (int iq = 0; iq < 4; ++iq) {
            t0[iq] = s_tau[te][iq][0];
            t1[iq] = s_tau[te][iq][1];
            ca[iq] = s_conv[te][aa][iq];
            cb[iq] = s_conv[te][bb][iq];
        }
        lhs_block_eval(aa, bb, ga, gb, s_scal[te][0], t0, t1, ca, cb, Bk);
    }
}

__global__ __launch_bounds__(LBLK) void tet_lhs_kernel(I B, const I* __restrict__ ien_b, const I* __restrict__ nzmap_b,
                                                      const T* __restrict__ egeo, const T* __restrict__ nodep,
                                                      T* __restrict__ val) {
    __shared__ LhsStage S;
    __shared__ double s_blk[16 * TRS];
    __shared__ int s_nz[LBLK];

    const int t = threadIdx.x;
    const int te = t >> 4, p = t & 15;
    const long long e = (long long)blockIdx.x * EPB + te;
    const bool valid = e < B;

    s_nz[t] = valid ? nzmap_b[(long long)blockIdx.x * LBLK + t] : -1;
    double Bk[16];
    lhs_element_block(S, te, p, valid, ien_b + (valid ? e : 0) * 4, egeo + (valid ? e : 0) * 16, nodep, Bk);
    // transpose through LDS: s_blk[i][t] = entry i of the block owned by thread t
#pragma unroll
    for (int i = 0; i < 16; ++i) s_blk[i * TRS + t] = Bk[i];
    WAVE_SYNC();

    // scatter: 8 lanes per 128-byte block line, 32 blocks per pass, 8 passes
    const int l8 = t & 7;
    double2 oldv[8];
    long long addr[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int bi = (t & ~63) + ((t & 63) >> 3) + 8 * pass;
        const int nz = s_nz[bi];
        addr[pass] = (nz >= 0) ? ((long long)nz * 16 + 2 * l8) : -1;
        if (nz >= 0) oldv[pass] = *reinterpret_cast<const double2*>(val + addr[pass]);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int bi = (t & ~63) + ((t & 63) >> 3) + 8 * pass;
        if (addr[pass] >= 0) {
            double2 nv;
            nv.x = oldv[pass].x + s_blk[(2 * l8) * TRS + bi];
            nv.y = oldv[pass].y + s_blk[(2 * l8 + 1) * TRS + bi];
            *reinterpret_cast<double2*>(val + addr[pass]) = nv;
        }
    }
}

// ====================================================================================
//  LHS, patch form (assembly schedule 2; host/patch.c).  One workgroup owns one spatial patch
//  of tets: every (a,b) block contribution is summed into an LDS table indexed by the patch's
//  local block slots (LDS f64 atomics: ds_add_f64), and each distinct block of the patch is
//  read-modify-written in HBM once per patch instead of once per tet.  Patches of one launch
//  share no node (patch coloring), so the global RMW needs no atomics.
//  LDS table layout: entry-major, tab[i * NS + slot] (NS odd), so the 8 lanes that flush one
//  block line read 8 different banks.
// ====================================================================================
__global__ __launch_bounds__(LBLK) void tet_lhs_patch_kernel(const I* __restrict__ p_eoff, const I* __restrict__ p_boff,
                                                            I patch_base, const I* __restrict__ ien_p,
                                                            const unsigned short* __restrict__ lslot,
                                                            const I* __restrict__ blk_nz, const T* __restrict__ egeo,
                                                            const T* __restrict__ nodep, T* __restrict__ val, int NS, int dbg) {
    extern __shared__ double dyn_lds[];
    __shared__ LhsStage S;
    double* tab = dyn_lds;  // [16][NS]
    const int t = threadIdx.x;
    const int te = t >> 4, p = t & 15;
    const int patch = patch_base + blockIdx.x;
    const int e0 = p_eoff[patch], ne = p_eoff[patch + 1] - e0;
    const int b0 = p_boff[patch], nb = p_boff[patch + 1] - b0;
    for (int i = t; i < 16 * NS; i += LBLK) tab[i] = 0.0;
    __syncthreads();
    for (int base = 0; base < ((dbg & 1) ? 0 : ne); base += EPB) {
        const int le = base + te;
        const bool valid = le < ne;
        const long long e = e0 + (valid ? le : 0);
        double Bk[16];
        WAVE_SYNC();  // the wave's staging slots are about to be overwritten
        lhs_element_block(S, te, p, valid, ien_p + e * 4, egeo + e * 16, nodep, Bk);
        if (valid) {
            const int slot = lslot[e * 16 + p];
#pragma unroll
            for (int i = 0; i < 16; ++i) if (!(dbg & 4)) atomicAdd(&tab[i * NS + slot], Bk[i]); else if (Bk[i] == 1.2345e300) tab[i] = 1.0;
        }
    }
    __syncthreads();
    // flush: 8 lanes per block line, 32 lines per pass, 4 passes in flight
    const int l8 = t & 7;
    if (dbg & 2) return;
    for (int s0 = t >> 3; s0 < nb; s0 += 32 * 4) {
        double2 oldv[4];
        long long addr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = s0 + 32 * k;
            addr[k] = -1;
            if (s < nb) {
                addr[k] = (long long)blk_nz[b0 + s] * 16 + 2 * l8;
                oldv[k] = *reinterpret_cast<const double2*>(val + addr[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = s0 + 32 * k;
            if (addr[k] >= 0) {
                double2 nv;
                nv.x = oldv[k].x + tab[(2 * l8) * NS + s];
                nv.y = oldv[k].y + tab[(2 * l8 + 1) * NS + s];
                *reinterpret_cast<double2*>(val + addr[k]) = nv;
            }
        }
    }
}

// ====================================================================================
//  RHS: AssembleWeakFormKernel<..,1> (assemble.cu:761-924) + ElemRHSLocal2Global x4
//  (:188-208, 1709-1724).  4 lanes per element: lane = node a for the gather and the
//  scatter, lane = quadrature point q for the weak form; the 4 quadrature
//  contributions are summed across the lanes by two xor-shuffles.
// ====================================================================================


__global__ __launch_bounds__(RBLK) void tet_rhs_kernel(I B, const I* __restrict__ ien_b, const T* __restrict__ nodep,
                                                      T* __restrict__ Fp) {
    __shared__ double s_n[REPB][4][NV + 1];
    const int t = threadIdx.x;
    const int te = t >> 2, a = t & 3;
    const long long e = (long long)blockIdx.x * REPB + te;
    const bool valid = e < B;
    long long node = 0;
    if (valid) {
        node = ien_b[e * 4 + a];
        double* s = s_n[te][a];
        const double2* rec = reinterpret_cast<const double2*>(nodep + node * NREC);
#pragma unroll
        for (int k = 0; k < 7; ++k) {  // 7 x 16 B out of one line instead of 14 scattered 8-byte gathers
            const double2 v = rec[k];
            s[2 * k] = v.x;
            s[2 * k + 1] = v.y;
        }
    }
    WAVE_SYNC();
    if (!valid) return;  // whole 4-lane groups leave together; shuffles below stay inside a group
    const double* r[4] = {s_n[te][0], s_n[te][1], s_n[te][2], s_n[te][3]};
    double mine[6];
    rhs_quad(r, a, mine);
    // ElemRHSLocal2Global: non-atomic, race-free inside a class; one 64-byte record per node
    double2* dst = reinterpret_cast<double2*>(Fp + node * FREC);
    double2 f0 = dst[0], f1 = dst[1], f2 = dst[2];
    f0.x += mine[0]; f0.y += mine[1];
    f1.x += mine[2]; f1.y += mine[3];
    f2.x += mine[4]; f2.y += mine[5];
    dst[0] = f0; dst[1] = f1; dst[2] = f2;
}

// ---- RHS, patch form (host/patch.c: DflBuildRhsPatchSchedule).  One workgroup = one spatial patch of <= 64 tets:
// the patch's node records are staged in LDS once (each record is fetched once per patch instead of once per tet),
// every tet is evaluated by 4 lanes exactly as above, the per-(tet, vertex) results are parked in LDS and summed per
// patch node in a fixed order (adjacency lists), and ONE partial record per patch node is written.
constexpr int RP_MAXN = 96;  // node records a patch may stage (the host caps patches at this many distinct nodes)
constexpr int RP_MAXT = 64;   // tets per patch (the tet loop below takes 64 per trip)

__global__ __launch_bounds__(RBLK) void tet_rhs_patch_kernel(const I* __restrict__ p_eoff, const I* __restrict__ p_noff,
                                                            const I* __restrict__ pnode, const unsigned char* __restrict__ lien,
                                                            const unsigned short* __restrict__ adj,
                                                            const unsigned short* __restrict__ adj_start,
                                                            const T* __restrict__ nodep, T* __restrict__ partial, I P, int xcd) {
    __shared__ double s_rec[RP_MAXN][NV + 1];
    __shared__ double s_out[RP_MAXT * 4][6 + 1];
    __shared__ unsigned short s_adj[RP_MAXT * 4];
    __shared__ unsigned short s_st[RP_MAXN + 1];
    const int t = threadIdx.x;
    // XCD-aware order (see tet_lhs_rowpatch_kernel): neighbouring patches share node records -> same L2
    const int per = (P + 7) >> 3;
    const int pid = (xcd & 1) ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (pid >= P) return;
    const int e0 = p_eoff[pid], ne = p_eoff[pid + 1] - e0;
    const int n0 = p_noff[pid], nn = p_noff[pid + 1] - n0;
    const int a = t & 3;
    uchar4 lnv = make_uchar4(0, 0, 0, 0);
    if ((t >> 2) < ne) lnv = *reinterpret_cast<const uchar4*>(lien + ((long long)e0 + (t >> 2)) * 4);
    for (int k = t; k < ne * 4; k += RBLK) s_adj[k] = adj[(long long)e0 * 4 + k];
    for (int k = t; k <= nn; k += RBLK) s_st[k] = adj_start[n0 + pid + k];
    for (int k = t; k < nn * 7; k += RBLK) {
        const int ln = k / 7, part = k - ln * 7;
        const double2 v = reinterpret_cast<const double2*>(nodep + (long long)pnode[n0 + ln] * NREC)[part];
        s_rec[ln][2 * part] = v.x;
        s_rec[ln][2 * part + 1] = v.y;
    }
    __syncthreads();
    if (xcd & 64) return;
    for (int le = t >> 2; le < ne; le += REPB) {  // whole quads
        if (le >= REPB) lnv = *reinterpret_cast<const uchar4*>(lien + ((long long)e0 + le) * 4);
        const double* r[4] = {s_rec[lnv.x], s_rec[lnv.y], s_rec[lnv.z], s_rec[lnv.w]};
        double mine[6];
        rhs_quad(r, a, mine);
#pragma unroll
        for (int j = 0; j < 6; ++j) s_out[le * 4 + a][j] = mine[j];
    }
    __syncthreads();
    // ordered sum per patch node: contributions in ascending local tet order (adjacency staged in LDS above)
    for (int k = t; k < nn * 6; k += RBLK) {
        const int ln = k / 6, j = k - ln * 6;
        double sum = 0.0;
        for (int q = s_st[ln]; q < s_st[ln + 1]; ++q) sum += s_out[s_adj[q]][j];
        partial[(long long)n0 * 6 + k] = sum;
    }
}

// F (reference layout) += sum of the node's partial records, ascending patch order (fixed => reproducible)
__global__ __launch_bounds__(256) void rhs_node_sum_kernel(I N, const I* __restrict__ goff, const I* __restrict__ gidx,
                                                          const T* __restrict__ partial, T* __restrict__ F) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double f[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const I q1 = goff[i + 1];
    I q = goff[i];
    // four records in flight (indices first, then the 12 loads): a node has 2-8 partial records and the one-at-a-time walk was a
    // chain of dependent loads per record; the additions stay in record order
    for (; q + 4 <= q1; q += 4) {
        const I g0 = gidx[q], g1 = gidx[q + 1], g2 = gidx[q + 2], g3 = gidx[q + 3];
        const double2* s0 = reinterpret_cast<const double2*>(partial + (long long)g0 * 6);
        const double2* s1 = reinterpret_cast<const double2*>(partial + (long long)g1 * 6);
        const double2* s2 = reinterpret_cast<const double2*>(partial + (long long)g2 * 6);
        const double2* s3 = reinterpret_cast<const double2*>(partial + (long long)g3 * 6);
        const double2 a0 = s0[0], a1 = s0[1], a2 = s0[2], b0 = s1[0], b1 = s1[1], b2 = s1[2];
        const double2 c0 = s2[0], c1 = s2[1], c2 = s2[2], d0 = s3[0], d1 = s3[1], d2 = s3[2];
        f[0] += a0.x; f[1] += a0.y; f[2] += a1.x; f[3] += a1.y; f[4] += a2.x; f[5] += a2.y;
        f[0] += b0.x; f[1] += b0.y; f[2] += b1.x; f[3] += b1.y; f[4] += b2.x; f[5] += b2.y;
        f[0] += c0.x; f[1] += c0.y; f[2] += c1.x; f[3] += c1.y; f[4] += c2.x; f[5] += c2.y;
        f[0] += d0.x; f[1] += d0.y; f[2] += d1.x; f[3] += d1.y; f[4] += d2.x; f[5] += d2.y;
    }
    for (; q < q1; ++q) {
        const double2* src = reinterpret_cast<const double2*>(partial + (long long)gidx[q] * 6);
        const double2 a0 = src[0], a1 = src[1], a2 = src[2];
        f[0] += a0.x; f[1] += a0.y; f[2] += a1.x; f[3] += a1.y; f[4] += a2.x; f[5] += a2.y;
    }
    F[3 * i + 0] += f[0];
    F[3 * i + 1] += f[1];
    F[3 * i + 2] += f[2];
    F[3LL * N + i] += f[3];
    F[4LL * N + i] += f[4];
    F[5LL * N + i] += f[5];
}

// gather layout: one line per node, written once per assembly call from the reference-layout vectors
// nodexu != NULL: also the compact (x, u) records of the Jacobian kernel, 64 B per node (it reads nothing else of a node:
// half the bytes per record fetched from HBM)
__global__ __launch_bounds__(256) void pack_nodes_kernel(I N, const T* __restrict__ xg, const T* __restrict__ wg,
                                                        const T* __restrict__ dwg, T* __restrict__ nodep, T* __restrict__ nodexu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double r[NREC];
    r[0] = xg[3 * i]; r[1] = xg[3 * i + 1]; r[2] = xg[3 * i + 2];
    r[3] = wg[3 * i]; r[4] = wg[3 * i + 1]; r[5] = wg[3 * i + 2];
    if (!nodep) {  // (uniform) a Jacobian-only call on the slot-owner schedule: the compact records are all that is read
        double2* c = reinterpret_cast<double2*>(nodexu + i * 8);
        c[0] = make_double2(r[0], r[1]);
        c[1] = make_double2(r[2], r[3]);
        c[2] = make_double2(r[4], r[5]);
        c[3] = make_double2(0.0, 0.0);
        return;
    }
    r[6] = wg[4LL * N + i];
    r[7] = wg[5LL * N + i];
    if (dwg) {
        r[8] = dwg[3 * i]; r[9] = dwg[3 * i + 1]; r[10] = dwg[3 * i + 2];
        r[11] = dwg[3LL * N + i];  // pressure always from the rate vector (Q9, assemble.cu:1606-1609)
        r[12] = dwg[4LL * N + i];
        r[13] = dwg[5LL * N + i];
    } else {
        r[8] = r[9] = r[10] = r[11] = r[12] = r[13] = 0.0;
    }
    r[14] = r[15] = 0.0;
    double2* o = reinterpret_cast<double2*>(nodep + i * NREC);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = make_double2(r[2 * k], r[2 * k + 1]);
    if (nodexu) {
        double2* c = reinterpret_cast<double2*>(nodexu + i * 8);
        c[0] = make_double2(r[0], r[1]);
        c[1] = make_double2(r[2], r[3]);
        c[2] = make_double2(r[4], r[5]);
        c[3] = make_double2(0.0, 0.0);
    }
}

// The same records written through LDS: a thread still reads its own node (coalesced reads of the reference-layout arrays) but
// the wave writes its 64 records as 8 (full) / 4 (compact) store instructions of 1 KB of consecutive addresses each, instead
// of 64 different lines per instruction.  LDS slice of a wave: piece k (16 B) of node n at [k * 65 + n].
__global__ __launch_bounds__(256) void pack_nodes_lds_kernel(I N, const T* __restrict__ xg, const T* __restrict__ wg,
                                                            const T* __restrict__ dwg, T* __restrict__ nodep, T* __restrict__ nodexu) {
    __shared__ double2 s_rec[4][8 * 65];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long nb = (long long)blockIdx.x * 256 + w * 64;  // first node of the wave
    const long long i = nb + lane;
    double2* const sw = s_rec[w];
    if (i < N) {
        const double x0 = xg[3 * i], x1 = xg[3 * i + 1], x2 = xg[3 * i + 2];
        const double u0 = wg[3 * i], u1 = wg[3 * i + 1], u2 = wg[3 * i + 2];
        sw[lane] = make_double2(x0, x1);
        sw[65 + lane] = make_double2(x2, u0);
        sw[130 + lane] = make_double2(u1, u2);
        if (nodep) {
            double r[8];
            r[0] = wg[4LL * N + i];
            r[1] = wg[5LL * N + i];
            if (dwg) {
                r[2] = dwg[3 * i]; r[3] = dwg[3 * i + 1]; r[4] = dwg[3 * i + 2];
                r[5] = dwg[3LL * N + i];  // pressure always from the rate vector (Q9, assemble.cu:1606-1609)
                r[6] = dwg[4LL * N + i];
                r[7] = dwg[5LL * N + i];
            } else {
                r[2] = r[3] = r[4] = r[5] = r[6] = r[7] = 0.0;
            }
            sw[195 + lane] = make_double2(r[0], r[1]);
            sw[260 + lane] = make_double2(r[2], r[3]);
            sw[325 + lane] = make_double2(r[4], r[5]);
            sw[390 + lane] = make_double2(r[6], r[7]);
            sw[455 + lane] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    const long long left = N - nb;  // nodes of this wave that exist (may be <= 0 in the last block)
    if (nodep) {
        double2* o = reinterpret_cast<double2*>(nodep + nb * NREC);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int p = it * 64 + lane, n = p >> 3, k = p & 7;
            if (n < left) o[p] = sw[k * 65 + n];
        }
    }
    if (nodexu) {
        double2* o = reinterpret_cast<double2*>(nodexu + nb * 8);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p = it * 64 + lane, n = p >> 2, k = p & 3;
            if (n < left) o[p] = k < 3 ? sw[k * 65 + n] : make_double2(0.0, 0.0);
        }
    }
}

// F (reference layout) += packed residual; the packed buffer is cleared for the next call
__global__ __launch_bounds__(256) void unpack_rhs_kernel(I N, T* __restrict__ Fp, T* __restrict__ F) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double2* src = reinterpret_cast<double2*>(Fp + i * FREC);
    const double2 f0 = src[0], f1 = src[1], f2 = src[2];
    F[3 * i + 0] += f0.x;
    F[3 * i + 1] += f0.y;
    F[3 * i + 2] += f1.x;
    F[3LL * N + i] += f1.y;
    F[4LL * N + i] += f2.x;
    F[5LL * N + i] += f2.y;
    const double2 z = make_double2(0.0, 0.0);
    src[0] = z; src[1] = z; src[2] = z;
}

// ====================================================================================
//  Faces: GetElemFaceNVKernel (:279-319) + FaceAssemblyKernel (:1038-1214), one
//  thread per face of one parent color.
// ====================================================================================
__device__ __forceinline__ int find_nz(const I* __restrict__ rp, const I* __restrict__ ci, int row, int col) {
    int lo = rp[row], hi = rp[row + 1] - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (ci[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(64) void face_kernel(I nf, const I* __restrict__ face_list, const I* __restrict__ f2e,
                                                 const I* __restrict__ forn, const I* __restrict__ ien, I N,
                                                 const T* __restrict__ xg, const T* __restrict__ wg, const T* __restrict__ dwg,
                                                 T* __restrict__ F, const I* __restrict__ rp, const I* __restrict__ ci,
                                                 T* __restrict__ val, T* __restrict__ pF, T* __restrict__ pJ) {
    // pF / pJ != NULL: two-pass form -- the face's contributions are parked ([i][a][4] and [i][a*4+b][16]) and summed per
    // node / per nonzero in a fixed order by face_sum_*_kernel: one launch for all faces, no conflict classes
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nf) return;
    const int f = face_list ? face_list[i] : i;
    const long long el = f2e[f];
    const int iorn = forn[f];
    int nodes[4];
    double x[12], buf[4][4];  // buf[comp][a], comps u0 u1 u2 p
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        nodes[a] = ien[el * 4 + a];
        const long long n = nodes[a];
        x[a * 3 + 0] = xg[3 * n]; x[a * 3 + 1] = xg[3 * n + 1]; x[a * 3 + 2] = xg[3 * n + 2];
        buf[0][a] = wg[3 * n]; buf[1][a] = wg[3 * n + 1]; buf[2][a] = wg[3 * n + 2];
        buf[3][a] = dwg[3LL * N + n];
    }
    double invJ[9], shg[12], detJ;
    tet_geometry(x, invJ, detJ, shg);
    double nv[3];
    {
        double b[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int n = 0; n < 3; ++n) b[n] += invJ[n * 3 + k] * c_nv2[iorn * 3 + k];
        nv[0] = b[0] * detJ; nv[1] = b[1] * detJ; nv[2] = b[2] * detJ;
    }
    double grad[12];  // comps 0..3 (phi/T rows of the reference buffer are zero)
#pragma unroll
    for (int comp = 0; comp < 4; ++comp)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) s += shg[d + 3 * a] * buf[comp][a];
            grad[d + 3 * comp] = s;
        }
    const double* sb = c_shlub + 12 * iorn;
    double qb[4][3];
#pragma unroll
    for (int comp = 0; comp < 4; ++comp)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) s += sb[a + 4 * q] * buf[comp][a];
            qb[comp][q] = s;
        }
    double hinv = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double w = invJ[k + 0] * nv[0] + invJ[k + 3] * nv[1] + invJ[k + 6] * nv[2];
        hinv += w * w;
    }
    hinv = sqrt(hinv);
    const double tau_b = 4.0 * kMU * hinv;

    if (F) {
        double eF[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) eF[a][j] = 0.0;
#pragma unroll
        for (int iq = 0; iq < 3; ++iq) {
            const double uadv[3] = {qb[0][iq], qb[1][iq], qb[2][iq]};
            const double unor = uadv[0] * nv[0] + uadv[1] * nv[1] + uadv[2] * nv[2];
            const double uneg = (unor - fabs(unor)) * 0.5;
            double tmp0[3], tmp1[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double v = 0.0;
                v += nv[k] * qb[3][iq];
                v -= kMU * (nv[0] * grad[3 * k + 0] + nv[1] * grad[3 * k + 1] + nv[2] * grad[3 * k + 2]);
                v -= kMU * (nv[0] * grad[3 * 0 + k] + nv[1] * grad[3 * 1 + k] + nv[2] * grad[3 * 2 + k]);
                v -= kRHO * uneg * uadv[k];
                v += tau_b * uadv[k];
                tmp0[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) tmp1[k * 3 + j] = -kMU * (nv[k] * uadv[j] + nv[j] * uadv[k]);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    double bm = 0.0;
                    bm += sb[iq * 4 + aa] * tmp0[ii];
                    bm += shg[aa * 3 + 0] * tmp1[ii * 3 + 0];
                    bm += shg[aa * 3 + 1] * tmp1[ii * 3 + 1];
                    bm += shg[aa * 3 + 2] * tmp1[ii * 3 + 2];
                    eF[aa][ii] += bm * GWB;
                }
                eF[aa][3] -= sb[iq * 4 + aa] * unor * GWB;
            }
        }
        if (pF) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) pF[(long long)i * 16 + a * 4 + j] = eF[a][j];
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const long long n = nodes[a];
                F[3 * n + 0] += eF[a][0];
                F[3 * n + 1] += eF[a][1];
                F[3 * n + 2] += eF[a][2];
                F[3LL * N + n] += eF[a][3];
            }
        }
    }
    if (val || pJ) {
        const double fact2 = kDT * kALPHAF * kGAMMA;
        double shnorm[4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            double s = 0.0;
            s += shg[aa * 3 + 0] * nv[0];
            s += shg[aa * 3 + 1] * nv[1];
            s += shg[aa * 3 + 2] * nv[2];
            shnorm[aa] = s;
        }
        for (int aa = 0; aa < 4; ++aa)
            for (int bb = 0; bb < 4; ++bb) {
                double Bk[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) Bk[k] = 0.0;
#pragma unroll
                for (int iq = 0; iq < 3; ++iq) {
                    const double uadv[3] = {qb[0][iq], qb[1][iq], qb[2][iq]};
                    const double unor = uadv[0] * nv[0] + uadv[1] * nv[1] + uadv[2] * nv[2];
                    const double uneg = (unor - fabs(unor)) * 0.5;
                    const double sa = sb[iq * 4 + aa], sbb = sb[iq * 4 + bb];
                    double tmp0 = 0.0;
                    tmp0 -= kMU * (shnorm[bb] * sa + shnorm[aa] * sbb);
                    tmp0 -= kRHO * sa * sbb * uneg;
                    tmp0 += tau_b * sa * sbb;
                    Bk[0] += fact2 * tmp0 * GWB;
                    Bk[5] += fact2 * tmp0 * GWB;
                    Bk[10] += fact2 * tmp0 * GWB;
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 3; ++jj) {
                            double v = 0.0;
                            v -= kMU * sa * shg[bb * 3 + ii] * nv[jj];
                            v -= kMU * sbb * shg[aa * 3 + jj] * nv[ii];
                            Bk[ii * 4 + jj] += fact2 * v * GWB;
                        }
                    const double ss = sa * sbb;
#pragma unroll
                    for (int ii = 0; ii < 3; ++ii) {
                        Bk[12 + ii] -= fact2 * ss * nv[ii] * GWB;  // dRC/dU
                        Bk[ii * 4 + 3] += ss * nv[ii] * GWB;       // dRM/dP
                    }
                }
                if (pJ) {
                    T* dst = pJ + ((long long)i * 16 + aa * 4 + bb) * 16;
#pragma unroll
                    for (int k = 0; k < 16; ++k) dst[k] = Bk[k];
                } else {
                    const int nz = find_nz(rp, ci, nodes[aa], nodes[bb]);
                    T* dst = val + (long long)nz * 16;
#pragma unroll
                    for (int k = 0; k < 16; ++k) dst[k] += Bk[k];
                }
            }
    }
}

// second pass of the face assembly: F[node] += parked contributions of the faces around it, ascending face order
__global__ __launch_bounds__(256) void face_sum_F_kernel(I nn, const I* __restrict__ fnode, const I* __restrict__ off,
                                                        const I* __restrict__ ent, const T* __restrict__ pF, I N,
                                                        T* __restrict__ F) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long k = t >> 2;
    const int j = (int)(t & 3);
    if (k >= nn) return;
    double s = 0.0;
    for (I q = off[k]; q < off[k + 1]; ++q) s += pF[(long long)ent[q] * 4 + j];
    const long long n = fnode[k];
    F[j < 3 ? 3 * n + j : 3LL * N + n] += s;
}
// ... and val[nz] += parked blocks, 16 lanes per nonzero
__global__ __launch_bounds__(256) void face_sum_J_kernel(I nz_count, const I* __restrict__ fnz, const I* __restrict__ off,
                                                        const I* __restrict__ ent, const T* __restrict__ pJ, T* __restrict__ val) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long k = t >> 4;
    const int c = (int)(t & 15);
    if (k >= nz_count) return;
    double s = 0.0;
    for (I q = off[k]; q < off[k + 1]; ++q) s += pJ[(long long)ent[q] * 16 + c];
    val[(long long)fnz[k] * 16 + c] += s;
}

// per-element geometry cache for the LHS kernels: shg[12], detJ, gg = sum G_ij^2, 1/trace(G), pad
// (GetElemInvJ3D + GetShapeGradKernel + metric gemm + the gg/tr prologue of assemble.cu:528-535)
__global__ __launch_bounds__(256) void elem_geometry_kernel(I T_, const I* __restrict__ ien_x, const T* __restrict__ xg,
                                                           T* __restrict__ egeo) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= T_) return;
    double x[12];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long long n = ien_x[e * 4 + a];
        x[a * 3] = xg[3 * n]; x[a * 3 + 1] = xg[3 * n + 1]; x[a * 3 + 2] = xg[3 * n + 2];
    }
    double invJ[9], shg[12], G[9], detJ;
    tet_geometry(x, invJ, detJ, shg);
    tet_metric(shg, G);
    double gg = 0.0, tr = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        gg += G[i] * G[i];
        if (!(i & 3)) tr += G[i];
    }
    double2* o = reinterpret_cast<double2*>(egeo + e * 16);
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = make_double2(shg[2 * k], shg[2 * k + 1]);
    o[6] = make_double2(detJ, gg);
    o[7] = make_double2(1.0 / tr, 0.0);
}

__global__ __launch_bounds__(256) void nzmap_kernel(I T_, const I* __restrict__ ien_b, const I* __restrict__ rp,
                                                   const I* __restrict__ ci, I* __restrict__ nzmap) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T_ * 16) return;
    const long long e = i >> 4;
    const int p = (int)(i & 15), aa = p >> 2, bb = p & 3;
    nzmap[i] = find_nz(rp, ci, ien_b[e * 4 + aa], ien_b[e * 4 + bb]);
}

__global__ __launch_bounds__(256) void gather_ien_kernel(I T_, const I* __restrict__ ien, const I* __restrict__ batch_ind,
                                                        I* __restrict__ ien_b) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T_ * 4) return;
    ien_b[i] = ien[(long long)batch_ind[i >> 2] * 4 + (i & 3)];
}

// ====================================================================================
//  LHS, row-owner patch form (assembly schedule 3; host/rowpatch.c).  One workgroup owns the
//  matrix rows of one spatial patch of NODES.  Work item = (tet e, local node a) with node a
//  owned by the patch: its 4 lanes (lane = b = quadrature point q) evaluate the block row
//  (a, 0..3) of the tet and add it into an LDS image of the owned CSR rows; afterwards the rows
//  stream out once (val = beta * val + table; beta = 0 folds the MatrixZero pass into the write).
//  All staging between the 4 lanes of an item goes through DPP quad broadcasts: no LDS staging,
//  no barriers inside the item loop.
// ====================================================================================

// PROBE = true only under dfl_tune_asm (developer phase split): the shipped instantiation carries no probe branches
template <bool PROBE>
__global__ __launch_bounds__(LBLK, 5) void tet_lhs_rowpatch_kernel(I P, const I* __restrict__ p_ioff, const I* __restrict__ p_soff,
                                                               const I* __restrict__ item_ea,
                                                               const unsigned short* __restrict__ item_slot,
                                                               const I* __restrict__ slot_nz, const I* __restrict__ ien,
                                                               const T* __restrict__ egeo, const T* __restrict__ nodep,
                                                               T* __restrict__ val, T beta, int dbg_in) {
    const int dbg = PROBE ? dbg_in : 0;
    extern __shared__ double dyn_lds[];
    double* tab = dyn_lds;  // [16][nsp], entry-major
    // XCD-aware order: workgroup w runs on XCD w % 8; give every XCD one contiguous range of the spatially
    // ordered patches so that neighbouring patches (which share halo tets and node records) share an L2
    const int per = (P + 7) >> 3;
    const int pid = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (pid >= P) return;
    const int t = threadIdx.x;
    const int b = t & 3;
    const int i0 = p_ioff[pid], ni = p_ioff[pid + 1] - i0;
    const int s0 = p_soff[pid], ns = p_soff[pid + 1] - s0;
    const int nsp = ns | 1;
    for (int i = t; i < 16 * nsp; i += LBLK) tab[i] = 0.0;
    __syncthreads();
    // the index chain item -> tet -> node of iteration k+1 is fetched while iteration k computes
    const int nloop = (dbg & 1) ? 0 : ni;
    int ea_n = 0, slot_n = 0;
    long long node_n = 0;
    if ((t >> 2) < nloop) {
        ea_n = item_ea[i0 + (t >> 2)];
        slot_n = item_slot[((long long)i0 + (t >> 2)) * 4 + b];
        node_n = ien[(long long)(ea_n >> 2) * 4 + b];
    }
    for (int base = 0; base < nloop; base += LBLK / 4) {
        const int it = base + (t >> 2);
        const bool valid = it < ni;  // whole quads are valid or not
        const int ea = ea_n;
        const int slot = slot_n;
        const long long node_b = node_n;
        const long long e = ea >> 2;
        const int a = ea & 3;
        if (it + LBLK / 4 < ni) {
            ea_n = item_ea[i0 + it + LBLK / 4];
            slot_n = item_slot[((long long)i0 + it + LBLK / 4) * 4 + b];
            node_n = ien[(long long)(ea_n >> 2) * 4 + b];
        }
        const double* nrec = nodep + node_b * 16 + 3;  // packed node record: u at [3..5]
        double ub[3] = {nrec[0], nrec[1], nrec[2]};
        const double* ge = egeo + e * 16;
        const double2* grec = reinterpret_cast<const double2*>(ge);
        const double g3 = ge[3];  // rows of J^-1 = shape gradients of nodes 1..3
        const double2 g45 = grec[2], g67 = grec[3], g89 = grec[4], gab = grec[5], gs = grec[6];
        const double itr = ge[14];
        double ga[3], gb[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ga[d] = ge[a * 3 + d];
            gb[d] = ge[b * 3 + d];
        }
        // u at this lane's quadrature point q = b (qr_wgalpha, :1648-1655): shl(c,q) = SHB + (SHA-SHB)[c == q]
        double uq[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double sum = ub[d] + dpp_quad<0xB1>(ub[d]);
            sum += dpp_quad<0x4E>(sum);
            uq[d] = SHB * sum + (SHA - SHB) * ub[d];
        }
        // |J^-1 u|^2 and the stabilisation parameters at q (:587-603); 1/sqrt and sqrt through v_rsq_f64
        const double c1 = g3 * uq[0] + g45.x * uq[1] + g45.y * uq[2];
        const double c2 = g67.x * uq[0] + g67.y * uq[1] + g89.x * uq[2];
        const double c3 = g89.y * uq[0] + gab.x * uq[1] + gab.y * uq[2];
        const double knu = kMU / kRHO;
        const double y = c1 * c1 + c2 * c2 + c3 * c3 + (3.0 * knu * knu) * gs.y;
        const double tau0 = rsqrt(4.0 / (kDT * kDT) + y) * (1.0 / kRHO);
        const double tau1 = y * rsqrt(y) * itr;
        const double ca_own = ga[0] * uq[0] + ga[1] * uq[1] + ga[2] * uq[2];
        // quadrature-point values of the whole quad in every lane
        double t0[4], t1[4], ca[4], cb[4];
#define DFL_QUAD_POINT(IQ, CTRL)                                             \
        {                                                                    \
            const double q0 = quad_bcast<CTRL>(uq[0]), q1 = quad_bcast<CTRL>(uq[1]), q2 = quad_bcast<CTRL>(uq[2]); \
            ca[IQ] = ga[0] * q0 + ga[1] * q1 + ga[2] * q2; /* shconv (:574-583) */ \
            cb[IQ] = gb[0] * q0 + gb[1] * q1 + gb[2] * q2;                   \
            t0[IQ] = quad_bcast<CTRL>(tau0);                                 \
            t1[IQ] = quad_bcast<CTRL>(tau1);                                 \
        }
        DFL_QUAD_POINT(0, 0x00)
        DFL_QUAD_POINT(1, 0x55)
        DFL_QUAD_POINT(2, 0xAA)
        DFL_QUAD_POINT(3, 0xFF)
#undef DFL_QUAD_POINT
        // block (a, b) with b == this lane's quadrature point: the sums over q that carry shl(b, q) collapse to
        // SHB * (plain sum) + (SHA - SHB) * (own term); same terms as lhs_block_eval, different association
        double Bk[16];
        {
            const double fact1 = kALPHAM;
            const double fact2 = kDT * kALPHAF * kGAMMA;
            const double eK = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
            double S_t0 = 0.0, S_t1 = 0.0, S_t0ca = 0.0, S_sacb = 0.0, S_t0cacb = 0.0, S_t0cb = 0.0;
#pragma unroll
            for (int iq = 0; iq < 4; ++iq) {
                const double t0ca = t0[iq] * ca[iq];
                S_t0 += t0[iq];
                S_t1 += t1[iq];
                S_t0ca += t0ca;
                S_sacb += (iq == a ? SHA : SHB) * cb[iq];
                S_t0cacb += t0ca * cb[iq];
                S_t0cb += t0[iq] * cb[iq];
            }
            const double S_t0casb = SHB * S_t0ca + (SHA - SHB) * (tau0 * ca_own);
            const double S_t0sb = SHB * S_t0 + (SHA - SHB) * tau0;
            const double S_sasb = (a == b) ? (SHA * SHA + 3.0 * SHB * SHB) : (2.0 * SHA * SHB + 2.0 * SHB * SHB);
            const double S_one = SHA + 3.0 * SHB;  // sum of the shape functions over the quadrature points
            const double w = gs.x * GW;
            const double diag = w * (fact1 * kRHO * S_sasb + fact1 * kRHO * kRHO * S_t0casb + fact2 * kRHO * S_sacb +
                                     fact2 * kRHO * kRHO * S_t0cacb + 4.0 * fact2 * kMU * eK);
            const double cK = 4.0 * fact2 * kMU * w, cT = fact2 * kRHO * S_t1 * w;
            double kgb[3], tgb[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                kgb[d] = cK * gb[d];
                tgb[d] = cT * gb[d];
            }
#pragma unroll
            for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) Bk[ii * 4 + jj] = ga[jj] * kgb[ii] + ga[ii] * tgb[jj];
            Bk[0] += diag;
            Bk[5] += diag;
            Bk[10] += diag;
            const double cP0 = w * S_one, cP1 = w * kRHO * S_t0ca;
            const double cU0 = w * (fact1 * kRHO * S_t0sb + fact2 * kRHO * S_t0cb), cU1 = w * fact2 * S_one;
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                Bk[ii * 4 + 3] = cP1 * gb[ii] - cP0 * ga[ii];  // dRM/dP
                Bk[12 + ii] = cU0 * ga[ii] + cU1 * gb[ii];     // dRC/dU
            }
            Bk[15] = w * S_t0 * eK;  // dRC/dP
        }
        if (valid) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&tab[i * nsp + slot], Bk[i]);
        }
    }
    __syncthreads();
    if (dbg & 2) return;
    // stream the owned rows out: 8 lanes per 128-byte block line
    const int l8 = t & 7;
    if (beta == 0.0 || (dbg & 8)) {
        for (int s = t >> 3; s < ns; s += LBLK / 8) {
            const long long addr = (long long)slot_nz[s0 + s] * 16 + 2 * l8;
            double2 nv;
            nv.x = tab[(2 * l8) * nsp + s];
            nv.y = tab[(2 * l8 + 1) * nsp + s];
            *reinterpret_cast<double2*>(val + addr) = nv;
        }
    } else {
        for (int s = t >> 3; s < ns; s += LBLK / 8) {
            const long long addr = (long long)slot_nz[s0 + s] * 16 + 2 * l8;
            double2 nv = *reinterpret_cast<const double2*>(val + addr);
            nv.x = beta * nv.x + tab[(2 * l8) * nsp + s];
            nv.y = beta * nv.y + tab[(2 * l8 + 1) * nsp + s];
            *reinterpret_cast<double2*>(val + addr) = nv;
        }
    }
}

}  // namespace

extern "C" {

int g_patch_dbg = 0;
void dfl_tune_asm(int v) { g_patch_dbg = v; }
int dfl_tune_asm_flags(void) { return g_patch_dbg; }
void dfl_elem_geometry(I T_, const I* ien_x, const T* xg, T* egeo, void* stream) {
    if (T_ <= 0) return;
    elem_geometry_kernel<<<ceil_div(T_, 256), 256, 0, S(stream)>>>(T_, ien_x, xg, egeo);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_tet_lhs(I B, const I* ien_b, const I* nzmap_b, const T* egeo_b, const T* nodep, T* val, void* stream) {
    if (B <= 0) return;
    tet_lhs_kernel<<<ceil_div(B, EPB), LBLK, 0, S(stream)>>>(B, ien_b, nzmap_b, egeo_b, nodep, val);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_tet_lhs_patch(I npatch, I patch_base, const I* p_eoff, const I* p_boff, const I* ien_p,
                                const unsigned short* lslot, const I* blk_nz, const T* egeo_p, const T* nodep, T* val,
                                I max_slots, void* stream) {
    if (npatch <= 0) return;
    const int NS = (int)max_slots | 1;  // odd stride: conflict-free flush reads
    const size_t shmem = (size_t)16 * NS * sizeof(double);
    tet_lhs_patch_kernel<<<npatch, LBLK, shmem, S(stream)>>>(p_eoff, p_boff, patch_base, ien_p, lslot, blk_nz, egeo_p, nodep, val, NS, g_patch_dbg);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_tet_lhs_rowpatch(I npatch, const I* p_ioff, const I* p_soff, const I* item_ea, const unsigned short* item_slot,
                                   const I* slot_nz, const I* ien, const T* egeo, const T* nodep, T* val, T beta,
                                   I max_slots, void* stream) {
    if (npatch <= 0) return;
    const size_t lds = (size_t)16 * (size_t)(max_slots | 1) * sizeof(double);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        DFL_GUARD(hipFuncSetAttribute((const void*)tet_lhs_rowpatch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DFL_GUARD(hipFuncSetAttribute((const void*)tet_lhs_rowpatch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    const int grid = 8 * ((npatch + 7) / 8);
    if (g_patch_dbg)
        tet_lhs_rowpatch_kernel<true><<<grid, LBLK, lds, S(stream)>>>(npatch, p_ioff, p_soff, item_ea, item_slot, slot_nz, ien, egeo,
                                                                      nodep, val, beta, g_patch_dbg);
    else
        tet_lhs_rowpatch_kernel<false><<<grid, LBLK, lds, S(stream)>>>(npatch, p_ioff, p_soff, item_ea, item_slot, slot_nz, ien, egeo,
                                                                       nodep, val, beta, 0);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_tet_rhs_patch(I npatch, const I* p_eoff, const I* p_noff, const I* pnode, const unsigned char* lien,
                                 const unsigned short* adj, const unsigned short* adj_start, const T* nodep, T* partial,
                                 void* stream) {
    if (npatch <= 0) return;
    tet_rhs_patch_kernel<<<8 * ((npatch + 7) / 8), RBLK, 0, S(stream)>>>(p_eoff, p_noff, pnode, lien, adj, adj_start, nodep,
                                                                          partial, npatch, ((g_patch_dbg & 32) ? 0 : 1) | (g_patch_dbg & 64));
    DFL_LAUNCH_CHECK();
}

void dfl_rhs_node_sum(I N, const I* goff, const I* gidx, const T* partial, T* F, void* stream) {
    if (N <= 0) return;
    // (three lanes per node, one 16-byte third of every partial record each: 0.147 against 0.138 ms -- not kept)
    rhs_node_sum_kernel<<<ceil_div(N, 256), 256, 0, S(stream)>>>(N, goff, gidx, partial, F);
    DFL_LAUNCH_CHECK();
}

int dfl_rhs_patch_max_nodes(void) { return RP_MAXN; }
int dfl_rhs_patch_max_tets(void) { return RP_MAXT; }

void dfl_assemble_tet_rhs(I B, const I* ien_b, const T* nodep, T* Fp, void* stream) {
    if (B <= 0) return;
    tet_rhs_kernel<<<ceil_div(B, REPB), RBLK, 0, S(stream)>>>(B, ien_b, nodep, Fp);
    DFL_LAUNCH_CHECK();
}

void dfl_pack_nodes2(I N, const T* xg, const T* wg, const T* dwg, T* nodep, T* nodexu, void* stream) {
    if (N <= 0 || (!nodep && !nodexu)) return;
    static const bool direct = getenv("DFL_PACK_LDS") && atoi(getenv("DFL_PACK_LDS")) == 0;  // developer A/B
    if (direct) pack_nodes_kernel<<<ceil_div(N, 256), 256, 0, S(stream)>>>(N, xg, wg, dwg, nodep, nodexu);
    else pack_nodes_lds_kernel<<<ceil_div(N, 256), 256, 0, S(stream)>>>(N, xg, wg, dwg, nodep, nodexu);
    DFL_LAUNCH_CHECK();
}
void dfl_pack_nodes(I N, const T* xg, const T* wg, const T* dwg, T* nodep, void* stream) {
    dfl_pack_nodes2(N, xg, wg, dwg, nodep, nullptr, stream);
}

void dfl_unpack_rhs(I N, T* Fp, T* F, void* stream) {
    if (N <= 0) return;
    unpack_rhs_kernel<<<ceil_div(N, 256), 256, 0, S(stream)>>>(N, Fp, F);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_face(I nf, const I* face_list, const I* f2e, const I* forn, const I* ien, I N, const T* xg, const T* wg,
                       const T* dwg, T* F, const I* rp, const I* ci, T* val, void* stream) {
    if (nf <= 0) return;
    face_kernel<<<ceil_div(nf, 64), 64, 0, S(stream)>>>(nf, face_list, f2e, forn, ien, N, xg, wg, dwg, F, rp, ci, val, nullptr,
                                                         nullptr);
    DFL_LAUNCH_CHECK();
}

void dfl_assemble_face_park(I nf, const I* f2e, const I* forn, const I* ien, I N, const T* xg, const T* wg, const T* dwg,
                            T* pF, T* pJ, void* stream) {
    if (nf <= 0) return;
    // F / val are only used as "is this part wanted" switches inside the kernel
    face_kernel<<<ceil_div(nf, 64), 64, 0, S(stream)>>>(nf, nullptr, f2e, forn, ien, N, xg, wg, dwg, pF, nullptr, nullptr, nullptr, pF,
                                                         pJ);
    DFL_LAUNCH_CHECK();
}
void dfl_face_sum_F(I nn, const I* fnode, const I* off, const I* ent, const T* pF, I N, T* F, void* stream) {
    if (nn <= 0) return;
    face_sum_F_kernel<<<ceil_div((long long)nn * 4, 256), 256, 0, S(stream)>>>(nn, fnode, off, ent, pF, N, F);
    DFL_LAUNCH_CHECK();
}
void dfl_face_sum_J(I nz_count, const I* fnz, const I* off, const I* ent, const T* pJ, T* val, void* stream) {
    if (nz_count <= 0) return;
    face_sum_J_kernel<<<ceil_div((long long)nz_count * 16, 256), 256, 0, S(stream)>>>(nz_count, fnz, off, ent, pJ, val);
    DFL_LAUNCH_CHECK();
}

void dfl_elem_nzmap(I T_, const I* ien_b, const I* rp, const I* ci, I* nzmap_b, void* stream) {
    if (T_ <= 0) return;
    nzmap_kernel<<<ceil_div((long long)T_ * 16, 256), 256, 0, S(stream)>>>(T_, ien_b, rp, ci, nzmap_b);
    DFL_LAUNCH_CHECK();
}

void dfl_gather_ien(I T_, const I* ien, const I* batch_ind, I* ien_b, void* stream) {
    if (T_ <= 0) return;
    gather_ien_kernel<<<ceil_div((long long)T_ * 4, 256), 256, 0, S(stream)>>>(T_, ien, batch_ind, ien_b);
    DFL_LAUNCH_CHECK();
}

}  // extern "C"
