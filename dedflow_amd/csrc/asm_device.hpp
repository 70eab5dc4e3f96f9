// Device-side pieces shared by the assembly kernels (k_assemble.hip, k_assemble2.hip): constants and quadrature
// tables of src/assemble.cu:23-52,65-118, closed-form tet geometry, the (a,b) block of the element Jacobian
// (assemble.cu:618-661) and the element residual at one quadrature point per lane (assemble.cu:761-924).
#pragma once
#include "dfl_common.hpp"
#include <cmath>

namespace {

// ---- constants: src/assemble.cu:23-52,65-118 -------------------------------------
#define kRHOC (0.5)
#define kDT (5e-2)
#define kALPHAM ((3.0 - kRHOC) / (1.0 + kRHOC))
#define kALPHAF (1.0 / (1.0 + kRHOC))
#define kGAMMA (0.5 + kALPHAM - kALPHAF)
#define kRHO (1.0e3)
#define kCP (1.0)
#define kKAPPA (0.66)
#define kMU (10.0 / 3.0)
#define GW (0.0416666666666667)
#define GWB (0.1666666666666667)
#define SHA (0.5854101966249685)
#define SHB (0.1381966011250105)
#define FB0 (0.0)
#define FB1 (0.0)
#define FB2 (-9.81 * 0.0)

__device__ __forceinline__ double shl(int a, int q) { return a == q ? SHA : SHB; }

// All lanes that cooperate on one element (16 in the LHS kernel, 4 in the RHS kernel) sit in ONE
// wave, and every LDS slot is written and read by the same wave, so no workgroup barrier is
// needed: LDS operations of a wave complete in order once lgkmcnt drains.  Waves of a block then
// run their gather / compute / scatter phases independently (better latency hiding than four
// s_barrier rendezvous per block).
#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

__constant__ double c_shlub[48] = {
    0.0, GWB, GWB, 0.6666666666666667, 0.0, GWB, 0.6666666666666667, GWB, 0.0, 0.6666666666666667, GWB, GWB,
    GWB, 0.0, GWB, 0.6666666666666667, GWB, 0.0, 0.6666666666666667, GWB, 0.6666666666666667, 0.0, GWB, GWB,
    0.6666666666666667, GWB, 0.0, GWB, GWB, 0.6666666666666667, 0.0, GWB, GWB, GWB, 0.0, 0.6666666666666667,
    GWB, 0.6666666666666667, GWB, 0.0, GWB, GWB, 0.6666666666666667, 0.0, 0.6666666666666667, GWB, GWB, 0.0};
__constant__ double c_nv2[12] = {1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, -1.0};

// Geometry of one tet from its 4 vertices (x[a*3+d]):
//   invJ column-major (invJ[i+3j] = dxi_i/dx_j), detJ = |det J|, shg[a*3+d],
//   G = metric of assemble.cu:1586-1593 (G(i,j) = sum_r dxi_i/dx_r dxi_j/dx_r).
// Closed-form 3x3 inverse replaces the batched pivoted LU (differs by rounding only, Q8).
__device__ __forceinline__ void tet_geometry(const double* x, double* invJ, double& detJ, double* shg) {
    const double j00 = x[3] - x[0], j10 = x[4] - x[1], j20 = x[5] - x[2];   // column 0 = x1 - x0
    const double j01 = x[6] - x[0], j11 = x[7] - x[1], j21 = x[8] - x[2];   // column 1 = x2 - x0
    const double j02 = x[9] - x[0], j12 = x[10] - x[1], j22 = x[11] - x[2]; // column 2 = x3 - x0
    const double c00 = j11 * j22 - j12 * j21;
    const double c01 = j12 * j20 - j10 * j22;
    const double c02 = j10 * j21 - j11 * j20;
    const double det = j00 * c00 + j01 * c01 + j02 * c02;
    const double id = 1.0 / det;
    detJ = fabs(det);
    // inverse(i,j) = cof(j,i)/det
    invJ[0 + 3 * 0] = c00 * id;
    invJ[0 + 3 * 1] = (j02 * j21 - j01 * j22) * id;
    invJ[0 + 3 * 2] = (j01 * j12 - j02 * j11) * id;
    invJ[1 + 3 * 0] = c01 * id;
    invJ[1 + 3 * 1] = (j00 * j22 - j02 * j20) * id;
    invJ[1 + 3 * 2] = (j02 * j10 - j00 * j12) * id;
    invJ[2 + 3 * 0] = c02 * id;
    invJ[2 + 3 * 1] = (j01 * j20 - j00 * j21) * id;
    invJ[2 + 3 * 2] = (j00 * j11 - j01 * j10) * id;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) shg[i * 3 + j + 3] = invJ[i + j * 3];  // GetShapeGradKernel, :1308-1328
    shg[0] = -shg[3] - shg[6] - shg[9];
    shg[1] = -shg[4] - shg[7] - shg[10];
    shg[2] = -shg[5] - shg[8] - shg[11];
}

// 1 / sqrt(x) and 1 / x for the per-tet part of the slot-owner Jacobian kernel (x > 0 and finite there: sums of squares plus
// 4 / dt^2, a determinant of a valid tet, a trace of a positive definite metric): the hardware estimate (v_rsq_f64 /
// v_rcp_f64, about 2^-23 relative) refined by Newton steps -- ONE for the reciprocal square root (relative error
// (3/2) e^2 ~ 2e-14, the stabilisation parameters enter the blocks linearly) and TWO for the reciprocal (full precision:
// every shape gradient carries it) -- instead of the library's IEEE expansions with their special-case selects
// (~11 and ~13 instructions each; ten of them per tet were a sixth of phase 1).  Deterministic, so the kernel stays
// bitwise reproducible; the deviation from the oracle is bounded by the 1e-10 parity tests.
__device__ __forceinline__ double rsqrt_nr1(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);  // 1 - x y^2
    return fma(0.5 * y, e, y);
}
__device__ __forceinline__ double rcp_nr2(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

template <bool FAST = false>
__device__ __forceinline__ void tet_geometry_t(const double* x, double* invJ, double& detJ, double* shg) {
    const double j00 = x[3] - x[0], j10 = x[4] - x[1], j20 = x[5] - x[2];   // column 0 = x1 - x0
    const double j01 = x[6] - x[0], j11 = x[7] - x[1], j21 = x[8] - x[2];   // column 1 = x2 - x0
    const double j02 = x[9] - x[0], j12 = x[10] - x[1], j22 = x[11] - x[2]; // column 2 = x3 - x0
    const double c00 = j11 * j22 - j12 * j21;
    const double c01 = j12 * j20 - j10 * j22;
    const double c02 = j10 * j21 - j11 * j20;
    const double det = j00 * c00 + j01 * c01 + j02 * c02;
    const double id = FAST ? rcp_nr2(det) : 1.0 / det;
    detJ = fabs(det);
    invJ[0 + 3 * 0] = c00 * id;
    invJ[0 + 3 * 1] = (j02 * j21 - j01 * j22) * id;
    invJ[0 + 3 * 2] = (j01 * j12 - j02 * j11) * id;
    invJ[1 + 3 * 0] = c01 * id;
    invJ[1 + 3 * 1] = (j00 * j22 - j02 * j20) * id;
    invJ[1 + 3 * 2] = (j02 * j10 - j00 * j12) * id;
    invJ[2 + 3 * 0] = c02 * id;
    invJ[2 + 3 * 1] = (j01 * j20 - j00 * j21) * id;
    invJ[2 + 3 * 2] = (j00 * j11 - j01 * j10) * id;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) shg[i * 3 + j + 3] = invJ[i + j * 3];
    shg[0] = -shg[3] - shg[6] - shg[9];
    shg[1] = -shg[4] - shg[7] - shg[10];
    shg[2] = -shg[5] - shg[8] - shg[11];
}

__device__ __forceinline__ void tet_metric(const double* shg, double* G) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) s += shg[3 + r + 3 * i] * shg[3 + r + 3 * j];
            G[i + 3 * j] = s;
        }
}

// The 4x4 (u,p) block of node pair (aa,bb) from the shape gradients of the two nodes, |det J| and the
// per-quadrature-point stabilisation parameters / convective shape derivatives (assemble.cu:618-661).
// Every term of the reference's per-quadrature-point update is (quadrature-dependent scalar) x
// (quadrature-independent geometry product) x detJ*gw with equal weights, so the 4-point loop reduces to
// eleven scalar sums followed by ONE pass over the 16 block entries: ~5x fewer fp64 operations than the
// literal loop.  Same terms, different association: results differ from the oracle by rounding only
// (tests bound it at 1e-10).
__device__ __forceinline__ void lhs_block_eval(int aa, int bb, const double* ga, const double* gb, double detJ,
                                               const double* t0, const double* t1, const double* cav,
                                               const double* cbv, double* Bk) {
    const double fact1 = kALPHAM;
    const double fact2 = kDT * kALPHAF * kGAMMA;
    const double eK = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
    double S_t0 = 0.0, S_t1 = 0.0, S_sa = 0.0, S_sb = 0.0, S_sasb = 0.0, S_t0ca = 0.0, S_t0casb = 0.0, S_sacb = 0.0,
           S_t0cacb = 0.0, S_t0sb = 0.0, S_t0cb = 0.0;
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) {
        const double tau0 = t0[iq], tau1 = t1[iq];
        const double ca = cav[iq], cb = cbv[iq];
        const double sa = shl(aa, iq), sb = shl(bb, iq);
        const double t0ca = tau0 * ca;
        S_t0 += tau0;
        S_t1 += tau1;
        S_sa += sa;
        S_sb += sb;
        S_sasb += sa * sb;
        S_t0ca += t0ca;
        S_t0casb += t0ca * sb;
        S_sacb += sa * cb;
        S_t0cacb += t0ca * cb;
        S_t0sb += tau0 * sb;
        S_t0cb += tau0 * cb;
    }
    const double w = detJ * GW;
    const double diag = w * (fact1 * kRHO * S_sasb + fact1 * kRHO * kRHO * S_t0casb + fact2 * kRHO * S_sacb +
                             fact2 * kRHO * kRHO * S_t0cacb + 4.0 * fact2 * kMU * eK);
    const double cK = 4.0 * fact2 * kMU * w, cT = fact2 * kRHO * S_t1 * w;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) Bk[ii * 4 + jj] = cK * ga[jj] * gb[ii] + cT * ga[ii] * gb[jj];
    Bk[0] += diag;
    Bk[5] += diag;
    Bk[10] += diag;
    const double cP0 = w * S_sb, cP1 = w * kRHO * S_t0ca;
    const double cU0 = w * (fact1 * kRHO * S_t0sb + fact2 * kRHO * S_t0cb), cU1 = w * fact2 * S_sa;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        Bk[ii * 4 + 3] = cP1 * gb[ii] - cP0 * ga[ii];  // dRM/dP
        Bk[12 + ii] = cU0 * ga[ii] + cU1 * gb[ii];     // dRC/dU
    }
    Bk[15] = w * S_t0 * eK;  // dRC/dP
}

// acc += the 4x4 (u,p) block of node pair (aa, bb), from what the slot-owner kernel keeps per tet (shape gradients of the
// two nodes, convective shape derivatives ca[q] / cb[q] and tauM t0[q] at the four quadrature points, and four numbers
// that are the same for the sixteen blocks of the tet: w = |det J| GW, cK = 4 fact2 mu w, cT = fact2 rho (sum of tauC) w,
// S_t0 = sum of tauM) plus three single entries the caller reads by index (cb[aa], ca[bb], t0[bb]).  Same terms as
// lhs_block_eval; the sums that carry a shape-function factor collapse through shl(c, q) = SHB + (SHA - SHB)[c == q], and
// every entry is accumulated by fused multiply-adds straight into `acc` (about 85 fp64 operations instead of ~140).
__device__ __forceinline__ void lhs_block_accumulate(bool same, const double* ga, const double* gb, double w, double cK, double cT,
                                                     double S_t0, const double* t0, const double* ca, const double* cb, double cb_a,
                                                     double ca_b, double t0_b, double* acc) {
    const double fact1 = kALPHAM;
    const double fact2 = kDT * kALPHAF * kGAMMA;
    const double tc0 = t0[0] * ca[0], tc1 = t0[1] * ca[1], tc2 = t0[2] * ca[2], tc3 = t0[3] * ca[3];
    const double S_t0ca = (tc0 + tc1) + (tc2 + tc3);
    const double S_t0cacb = fma(tc3, cb[3], fma(tc2, cb[2], fma(tc1, cb[1], tc0 * cb[0])));
    const double S_t0cb = fma(t0[3], cb[3], fma(t0[2], cb[2], fma(t0[1], cb[1], t0[0] * cb[0])));
    const double C_b = (cb[0] + cb[1]) + (cb[2] + cb[3]);
    const double S_sacb = fma(SHA - SHB, cb_a, SHB * C_b);
    const double S_t0casb = fma(SHA - SHB, t0_b * ca_b, SHB * S_t0ca);
    const double S_t0sb = fma(SHA - SHB, t0_b, SHB * S_t0);
    const double S_sasb = same ? (SHA * SHA + 3.0 * SHB * SHB) : (2.0 * SHA * SHB + 2.0 * SHB * SHB);
    const double S_one = SHA + 3.0 * SHB;  // sum of the shape functions over the quadrature points
    const double eK = fma(ga[2], gb[2], fma(ga[1], gb[1], ga[0] * gb[0]));
    double d = (4.0 * fact2 * kMU) * eK;
    d = fma(fact2 * kRHO * kRHO, S_t0cacb, d);
    d = fma(fact2 * kRHO, S_sacb, d);
    d = fma(fact1 * kRHO * kRHO, S_t0casb, d);
    d = fma(fact1 * kRHO, S_sasb, d);
    const double diag = w * d;
    double kgb[3], tgb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        kgb[k] = cK * gb[k];
        tgb[k] = cT * gb[k];
    }
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) acc[ii * 4 + jj] = fma(ga[ii], tgb[jj], fma(ga[jj], kgb[ii], acc[ii * 4 + jj]));
    acc[0] += diag;
    acc[5] += diag;
    acc[10] += diag;
    const double cP0 = w * S_one, cP1 = (w * kRHO) * S_t0ca;
    const double cU0 = (w * kRHO) * fma(fact1, S_t0sb, fact2 * S_t0cb), cU1 = (w * fact2) * S_one;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        acc[ii * 4 + 3] = fma(cP1, gb[ii], fma(-cP0, ga[ii], acc[ii * 4 + 3]));  // dRM/dP
        acc[12 + ii] = fma(cU1, gb[ii], fma(cU0, ga[ii], acc[12 + ii]));          // dRC/dU
    }
    acc[15] = fma(w * S_t0, eK, acc[15]);  // dRC/dP
}

// the same block with the sum of tauC over the quadrature points formed by the caller (tauC enters only through it)
__device__ __forceinline__ void lhs_block_eval_s(int aa, int bb, const double* ga, const double* gb, double detJ,
                                                 const double* t0, double S_t1, const double* cav,
                                               const double* cbv, double* Bk) {
    const double fact1 = kALPHAM;
    const double fact2 = kDT * kALPHAF * kGAMMA;
    const double eK = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
    double S_t0 = 0.0, S_sa = 0.0, S_sb = 0.0, S_sasb = 0.0, S_t0ca = 0.0, S_t0casb = 0.0, S_sacb = 0.0,
           S_t0cacb = 0.0, S_t0sb = 0.0, S_t0cb = 0.0;
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) {
        const double tau0 = t0[iq];
        const double ca = cav[iq], cb = cbv[iq];
        const double sa = shl(aa, iq), sb = shl(bb, iq);
        const double t0ca = tau0 * ca;
        S_t0 += tau0;
        S_sa += sa;
        S_sb += sb;
        S_sasb += sa * sb;
        S_t0ca += t0ca;
        S_t0casb += t0ca * sb;
        S_sacb += sa * cb;
        S_t0cacb += t0ca * cb;
        S_t0sb += tau0 * sb;
        S_t0cb += tau0 * cb;
    }
    const double w = detJ * GW;
    const double diag = w * (fact1 * kRHO * S_sasb + fact1 * kRHO * kRHO * S_t0casb + fact2 * kRHO * S_sacb +
                             fact2 * kRHO * kRHO * S_t0cacb + 4.0 * fact2 * kMU * eK);
    const double cK = 4.0 * fact2 * kMU * w, cT = fact2 * kRHO * S_t1 * w;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) Bk[ii * 4 + jj] = cK * ga[jj] * gb[ii] + cT * ga[ii] * gb[jj];
    Bk[0] += diag;
    Bk[5] += diag;
    Bk[10] += diag;
    const double cP0 = w * S_sb, cP1 = w * kRHO * S_t0ca;
    const double cU0 = w * (fact1 * kRHO * S_t0sb + fact2 * kRHO * S_t0cb), cU1 = w * fact2 * S_sa;
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        Bk[ii * 4 + 3] = cP1 * gb[ii] - cP0 * ga[ii];  // dRM/dP
        Bk[12 + ii] = cU0 * ga[ii] + cU1 * gb[ii];     // dRC/dU
    }
    Bk[15] = w * S_t0 * eK;  // dRC/dP
}

constexpr int RBLK = 256;
constexpr int REPB = RBLK / 4;
constexpr int NV = 14;  // x(3) u(3) phi T du(3) p dphi dT per node

// value of lane (l ^ 1) [CTRL 0xB1] or (l ^ 2) [CTRL 0x4E] inside each quad: DPP quad_perm, no LDS round trip
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// node record of the packed gather layout (pack_nodes_kernel): one 128-byte line per node
//   [0..2] x  [3..5] u  [6] phi  [7] T  [8..10] du  [11] p (from the rate vector, Q9)  [12] dphi  [13] dT
constexpr int NREC = 16;
constexpr int FREC = 8;  // packed residual record: F_u0 F_u1 F_u2 F_p F_phi F_T pad pad (64 bytes)

// Residual of one tet on 4 lanes (lane a: node a for the result, quadrature point a for the weak form).
// r[b] = node record of vertex b (x[3] u[3] phi T du[3] p dphi dT, in LDS); result: mine[0..5] = the 6 residual
// components of node a summed over the 4 quadrature points (DPP quad reduce-scatter).
__device__ __forceinline__ void rhs_quad(const double* const* r, int a, double* mine) {
    const int iq = a;
    double shg[12], detJ, gg, itr;
    {  // geometry from the vertex coordinates in the node records (already staged in LDS: no extra HBM stream; the
       // cached geometry records were measured: equal time with prefetch, slower without, 1.3 GB more traffic)
        double x[12], invJ[9], G[9];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int d = 0; d < 3; ++d) x[b * 3 + d] = r[b][d];
        tet_geometry(x, invJ, detJ, shg);
        tet_metric(shg, G);
        gg = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) gg += G[k] * G[k];
        itr = 1.0 / (G[0] + G[4] + G[8]);
    }

    // buffer[comp][b] of LoadElementValueKernel: wg: u0 u1 u2 p phi T ; dwg: du0 du1 du2 p dphi dT
    const int wsrc[6] = {3, 4, 5, 11, 6, 7};
    const int dsrc[6] = {8, 9, 10, 11, 12, 13};
    double grad[18], qw[6], qd[6];
#pragma unroll
    for (int comp = 0; comp < 6; ++comp) {
        double vb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) vb[b] = r[b][wsrc[comp]];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s += shg[d + 3 * b] * vb[b];
            grad[d + 3 * comp] = s;  // qr_wggradalpha, :1628-1635
        }
        double s = 0.0, sd = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s += shl(b, iq) * vb[b];
            sd += shl(b, iq) * r[b][dsrc[comp]];
        }
        qw[comp] = s;   // qr_wgalpha[comp][iq]
        qd[comp] = sd;  // qr_dwgalpha[comp][iq]
    }

    const double fb[3] = {FB0, FB1, FB2};
    const double divu = grad[0] + grad[4] + grad[8];
    double uadv[3] = {qw[0], qw[1], qw[2]};
    double rLi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double r = 0.0;
        r += kRHO * (qd[i] - fb[i]);
        r += kRHO * uadv[0] * grad[3 * i + 0];
        r += kRHO * uadv[1] * grad[3 * i + 1];
        r += kRHO * uadv[2] * grad[3 * i + 2];
        r += grad[3 * 3 + i];
        rLi[i] = r;
    }
    // GetStabTau, :444-484.  u.G.u = |J^-1 u|^2 with the rows of J^-1 = shape gradients of nodes 1..3; sum G_ij^2 and
    // 1/tr G come from the geometry record; 1/sqrt and sqrt through v_rsq_f64 (no fp64 divisions left)
    double tau[4];
    {
        const double t0 = 4.0 / (kDT * kDT);
        double t1 = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double v = shg[3 + r] * uadv[0] + shg[6 + r] * uadv[1] + shg[9 + r] * uadv[2];
            t1 += v * v;
        }
        const double t2 = gg;
        const double mu = kMU / kRHO, kappa = kKAPPA / (kRHO * kCP);
        const double y = t1 + 3.0 * mu * mu * t2;
        tau[0] = rsqrt(t0 + y) * (1.0 / kRHO);
        tau[1] = y * rsqrt(y) * itr;
        tau[2] = rsqrt(t0 + t1);
        tau[3] = rsqrt(t0 + t1 + 3.0 * kappa * kappa * t2) * (1.0 / (kRHO * kCP));
    }
    double shconv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double c = 0.0;
        c += uadv[0] * shg[b * 3 + 0];
        c += uadv[1] * shg[b * 3 + 1];
        c += uadv[2] * shg[b * 3 + 2];
        shconv[b] = c;
    }
    double tmp0[3], tmp1[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double v = 0.0;
        v += kRHO * (qd[i] - fb[i]);
        v += kRHO * (uadv[0] - tau[0] * rLi[0]) * grad[3 * i + 0];
        v += kRHO * (uadv[1] - tau[0] * rLi[1]) * grad[3 * i + 1];
        v += kRHO * (uadv[2] - tau[0] * rLi[2]) * grad[3 * i + 2];
        tmp0[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            v += kMU * (grad[3 * i + j] + grad[3 * j + i]);
            v += kRHO * tau[0] * rLi[i] * uadv[j];
            v -= kRHO * tau[0] * tau[0] * rLi[i] * rLi[j];
            tmp1[i * 3 + j] = v;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp1[i * 3 + i] += -qw[3] + kRHO * tau[1] * divu;

    const double wq = GW * detJ;
    const double bp = qd[4] + uadv[0] * grad[3 * 4 + 0] + uadv[1] * grad[3 * 4 + 1] + uadv[2] * grad[3 * 4 + 2];
    const double btc = kRHO * kCP * (qd[5] + uadv[0] * grad[3 * 5 + 0] + uadv[1] * grad[3 * 5 + 1] + uadv[2] * grad[3 * 5 + 2]);
    // Each lane holds one quadrature point's contribution to all 4 rows; lane `a` needs row a
    // summed over the 4 points: reduce-scatter inside the quad with two DPP exchanges
    // (xor 2 keeps the row pair of the own half, xor 1 keeps the own row).
    (void)wq;
    const bool hi2 = (a >> 1) != 0, hi1 = (a & 1) != 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double c[4];
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            if (j < 3) {
                double bm = 0.0;
                bm += shl(aa, iq) * tmp0[j];
                bm += shg[aa * 3 + 0] * tmp1[j * 3 + 0];
                bm += shg[aa * 3 + 1] * tmp1[j * 3 + 1];
                bm += shg[aa * 3 + 2] * tmp1[j * 3 + 2];
                c[aa] = bm * GW * detJ;
            } else if (j == 3) {
                double bc = 0.0;
                bc += shl(aa, iq) * divu;
                bc += tau[0] * rLi[0] * shg[aa * 3 + 0];
                bc += tau[0] * rLi[1] * shg[aa * 3 + 1];
                bc += tau[0] * rLi[2] * shg[aa * 3 + 2];
                c[aa] = bc * GW * detJ;
            } else if (j == 4) {
                c[aa] = bp * (shl(aa, iq) + tau[2] * shconv[aa]) * GW * detJ;
            } else {
                double bt = btc * (shl(aa, iq) + kRHO * kCP * tau[3] * shconv[aa]);
                bt += kKAPPA * (grad[3 * 5 + 0] * shg[aa * 3 + 0] + grad[3 * 5 + 1] * shg[aa * 3 + 1] + grad[3 * 5 + 2] * shg[aa * 3 + 2]);
                c[aa] = bt * GW * detJ;
            }
        }
        const double k0 = hi2 ? c[2] : c[0], k1 = hi2 ? c[3] : c[1];
        const double s0 = hi2 ? c[0] : c[2], s1 = hi2 ? c[1] : c[3];
        const double r0 = k0 + dpp_quad<0x4E>(s0);
        const double r1 = k1 + dpp_quad<0x4E>(s1);
        const double keep = hi1 ? r1 : r0, send = hi1 ? r0 : r1;
        mine[j] = keep + dpp_quad<0xB1>(send);
    }
}

template <int CTRL>
__device__ __forceinline__ double quad_bcast(double v) {  // CTRL = 0x00 / 0x55 / 0xAA / 0xFF: lane 0..3 of the quad
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

}  // namespace
