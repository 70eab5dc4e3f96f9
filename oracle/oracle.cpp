// =============================================================================
//  ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
//  CPU restatement of the DEDFlow hot path (zexxzhao/DEDFlow @ 2024-10-16),
//  written from the cited lines of the reference, to serve as the checker in
//  tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
//  under dedflow_amd/ may include, link, import or execute this file.
//
//  PARITY STATUS: "parity unpinned" against the real reference.  The reference
//  is CUDA-only (nvcc/cuBLAS/cuSPARSE/cuRAND/Thrust), cannot be built or run
//  here, and ships no tests, fixtures or golden vectors.
//  This restatement is pinned only by (a) line-by-line review against the
//  citations below and (b) the independent analytic cross-checks in
//  tests/test_oracle_*.py (volume sums, linear-field gradients, scipy SpMV /
//  spsolve, coloring validity, XORWOW self-consistency).
//
//  Layouts follow the reference exactly (global vectors [u:Nx3 AoS|p|phi|T],
//  src/main.c:108-118; four row-expanded scalar CSR value arrays,
//  src/csr_impl.cu:24-59, src/matrix_impl.cu:370-453).
//  Deliberate deviations (each covered by a test that names it):
//    Q3  the last row_ptr entry of expanded patterns is set to nnz
//        (src/csr_impl.cu:24-36 leaves it 0).
//    Q1  JPL ties: `tie_break_by_index` != 0 breaks equal priorities by element
//        id; with 0 the reference's strict `<` (src/color_impl.cu:87) is kept.
// =============================================================================
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef int32_t i32;
typedef double f64;

// ---- constants: src/assemble.cu:23-52, 65-118 --------------------------------
static const f64 kRHOC = 0.5;
static const f64 kDT = 5e-2;
static const f64 kALPHAM = (3.0 - kRHOC) / (1.0 + kRHOC);
static const f64 kALPHAF = 1.0 / (1.0 + kRHOC);
static const f64 kGAMMA = 0.5 + kALPHAM - kALPHAF;
static const f64 kRHO = 1.0e3;
static const f64 kCP = 1.0;
static const f64 kKAPPA = 0.66;
static const f64 kMU = 10.0 / 3.0;
static const f64 fb[3] = {0.0, 0.0, -9.81 * 0.0};
static const f64 gw[4] = {0.0416666666666667, 0.0416666666666667, 0.0416666666666667, 0.0416666666666667};
#define SA 0.5854101966249685
#define SB 0.1381966011250105
static const f64 shlu[16] = {SA, SB, SB, SB, SB, SA, SB, SB, SB, SB, SA, SB, SB, SB, SB, SA};
static const f64 gwb[3] = {0.1666666666666667, 0.1666666666666667, 0.1666666666666667};
#define F6 0.1666666666666667
#define F3 0.6666666666666667
// shlub[forn][q][a]   (src/assemble.cu:68-83)
static const f64 shlub[48] = {
    0.0, F6, F6, F3, 0.0, F6, F3, F6, 0.0, F3, F6, F6,
    F6, 0.0, F6, F3, F6, 0.0, F3, F6, F3, 0.0, F6, F6,
    F3, F6, 0.0, F6, F6, F3, 0.0, F6, F6, F6, 0.0, F3,
    F6, F3, F6, 0.0, F6, F6, F3, 0.0, F3, F6, F6, 0.0};
// reference-element face normals (src/assemble.cu:114-118)
static const f64 nv2[12] = {1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.0, 0.0, -1.0};

enum { NSHL = 4, NQR = 4, BS = 6, NQRB = 3 };

// -----------------------------------------------------------------------------
//  3x3 pivoted LU + inverse: restates cublasD{getrf,getri}Batched(3) as used in
//  GetElemInvJ3D (src/assemble.cu:1245-1291) with LAPACK dgetf2 semantics.
//  a: column-major 3x3, overwritten by LU.  inv: column-major inverse.
// -----------------------------------------------------------------------------
static void lu3_inverse(f64* a, f64* inv) {
    int piv[3];
    for (int j = 0; j < 3; ++j) {
        int p = j;
        f64 best = std::fabs(a[j + 3 * j]);
        for (int i = j + 1; i < 3; ++i) {
            if (std::fabs(a[i + 3 * j]) > best) { best = std::fabs(a[i + 3 * j]); p = i; }
        }
        piv[j] = p;
        if (p != j) for (int c = 0; c < 3; ++c) std::swap(a[j + 3 * c], a[p + 3 * c]);
        f64 r = 1.0 / a[j + 3 * j];
        for (int i = j + 1; i < 3; ++i) a[i + 3 * j] *= r;
        for (int c = j + 1; c < 3; ++c)
            for (int i = j + 1; i < 3; ++i) a[i + 3 * c] -= a[i + 3 * j] * a[j + 3 * c];
    }
    for (int c = 0; c < 3; ++c) {
        f64 b[3] = {0.0, 0.0, 0.0};
        b[c] = 1.0;
        for (int j = 0; j < 3; ++j) if (piv[j] != j) std::swap(b[j], b[piv[j]]);
        for (int i = 1; i < 3; ++i) for (int k = 0; k < i; ++k) b[i] -= a[i + 3 * k] * b[k];
        for (int i = 2; i >= 0; --i) {
            for (int k = i + 1; k < 3; ++k) b[i] -= a[i + 3 * k] * b[k];
            b[i] /= a[i + 3 * i];
        }
        inv[0 + 3 * c] = b[0]; inv[1 + 3 * c] = b[1]; inv[2 + 3 * c] = b[2];
    }
}

struct ElemGeom {
    f64 invJ[9];     // column-major J^{-1}  (elem_metric[0..8] after Dgeam)
    f64 detJ;        // elem_metric[9] = |U00*U11*U22|   (src/assemble.cu:350-357)
    f64 shgrad[12];  // shgradg[a*3+d]                   (src/assemble.cu:1308-1328)
    f64 G[9];        // metric gemm, overwrites J^{-1}   (src/assemble.cu:1586-1593)
};

static void elem_geometry(const f64* xg, const i32* nodes, ElemGeom& g) {
    const f64* x0 = xg + 3 * (size_t)nodes[0];
    const f64* x1 = xg + 3 * (size_t)nodes[1];
    const f64* x2 = xg + 3 * (size_t)nodes[2];
    const f64* x3 = xg + 3 * (size_t)nodes[3];
    f64 lu[9];  // GetElemJ3DKernel, src/assemble.cu:321-348
    lu[0] = x1[0] - x0[0]; lu[1] = x1[1] - x0[1]; lu[2] = x1[2] - x0[2];
    lu[3] = x2[0] - x0[0]; lu[4] = x2[1] - x0[1]; lu[5] = x2[2] - x0[2];
    lu[6] = x3[0] - x0[0]; lu[7] = x3[1] - x0[1]; lu[8] = x3[2] - x0[2];
    lu3_inverse(lu, g.invJ);
    g.detJ = std::fabs(lu[0] * lu[4] * lu[8]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g.shgrad[i * 3 + j + 3] = g.invJ[i + j * 3];
    g.shgrad[0] = -g.shgrad[3] - g.shgrad[6] - g.shgrad[9];
    g.shgrad[1] = -g.shgrad[4] - g.shgrad[7] - g.shgrad[10];
    g.shgrad[2] = -g.shgrad[5] - g.shgrad[8] - g.shgrad[11];
    // C = A^T A with A(r,c) = shgrad[3 + r + 3c]  (OP_T, OP_N; lda 3)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            f64 s = 0.0;
            for (int r = 0; r < 3; ++r) s += g.shgrad[3 + r + 3 * i] * g.shgrad[3 + r + 3 * j];
            g.G[i + 3 * j] = s;
        }
}

// LoadElementValueKernel x4 (src/assemble.cu:135-154, 1601-1619 / 1663-1678):
// buffer[comp*NSHL + a]; the pressure always comes from dwgalpha + 3N (Q9).
static void gather_fields(const i32* nodes, i32 N, const f64* vec, const f64* dwg, f64* buffer) {
    for (int a = 0; a < NSHL; ++a) {
        size_t n = (size_t)nodes[a];
        buffer[0 * NSHL + a] = vec[n * 3 + 0];
        buffer[1 * NSHL + a] = vec[n * 3 + 1];
        buffer[2 * NSHL + a] = vec[n * 3 + 2];
        buffer[3 * NSHL + a] = dwg[(size_t)N * 3 + n];
        buffer[4 * NSHL + a] = vec[(size_t)N * 4 + n];
        buffer[5 * NSHL + a] = vec[(size_t)N * 5 + n];
    }
}

// GetStabTau, src/assemble.cu:444-484
static void stab_tau(const f64* Ginv, const f64* uadv, f64 rho, f64 cp, f64 mu, f64 kappa, f64 dt, f64* tau) {
    f64 t[3] = {0.0, 0.0, 0.0};
    t[0] = 4.0 / (dt * dt);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            t[1] += Ginv[i * 3 + j] * uadv[i] * uadv[j];
            t[2] += Ginv[i * 3 + j] * Ginv[i * 3 + j];
        }
    mu /= rho;
    kappa /= rho * cp;
    tau[0] = (1.0 / std::sqrt(t[0] + t[1] + 3.0 * mu * mu * t[2])) / rho;
    tau[1] = std::sqrt(t[1] + 3.0 * mu * mu * t[2]) / (Ginv[0] + Ginv[4] + Ginv[8]);
    tau[2] = 1.0 / std::sqrt(t[0] + t[1]);
    tau[3] = (1.0 / std::sqrt(t[0] + t[1] + 3.0 * kappa * kappa * t[2])) / (rho * cp);
}

#define M2D(aa, ii) ((aa) * BS + (ii))

// AssembleWeakFormKernel<..,TENSOR=1>, src/assemble.cu:761-924
static void elem_rhs(const ElemGeom& g, const f64* qr_wg, const f64* qr_dwg, const f64* qr_grad, f64* elem_F) {
    const f64* elem_G = g.G;
    const f64 detJ = g.detJ;
    const f64* shgradg = g.shgrad;
    for (int i = 0; i < NSHL * BS; ++i) elem_F[i] = 0.0;
    f64 tau[4] = {0.0, 0.0, 0.0, 0.0};
    f64 divu = qr_grad[0] + qr_grad[4] + qr_grad[8];
    f64 rLi[3], uadv[3], shconv[NSHL];
    for (int iq = 0; iq < NQR; ++iq) {
        uadv[0] = qr_wg[NQR * 0 + iq];
        uadv[1] = qr_wg[NQR * 1 + iq];
        uadv[2] = qr_wg[NQR * 2 + iq];
        for (int i = 0; i < 3; ++i) {
            rLi[i] = 0.0;
            rLi[i] += kRHO * (qr_dwg[NQR * i + iq] - fb[i]);
            rLi[i] += kRHO * uadv[0] * qr_grad[3 * i + 0];
            rLi[i] += kRHO * uadv[1] * qr_grad[3 * i + 1];
            rLi[i] += kRHO * uadv[2] * qr_grad[3 * i + 2];
            rLi[i] += qr_grad[3 * 3 + i];
        }
        stab_tau(elem_G, uadv, kRHO, kCP, kMU, kKAPPA, kDT, tau);
        for (int aa = 0; aa < NSHL; ++aa) {
            shconv[aa] = 0.0;
            shconv[aa] += uadv[0] * shgradg[aa * 3 + 0];
            shconv[aa] += uadv[1] * shgradg[aa * 3 + 1];
            shconv[aa] += uadv[2] * shgradg[aa * 3 + 2];
        }
        f64 tmp0[3], tmp1[9];
        for (int i = 0; i < 3; ++i) {
            tmp0[i] = 0.0;
            tmp0[i] += kRHO * (qr_dwg[NQR * i + iq] - fb[i]);
            tmp0[i] += kRHO * (uadv[0] - tau[0] * rLi[0]) * qr_grad[3 * i + 0];
            tmp0[i] += kRHO * (uadv[1] - tau[0] * rLi[1]) * qr_grad[3 * i + 1];
            tmp0[i] += kRHO * (uadv[2] - tau[0] * rLi[2]) * qr_grad[3 * i + 2];
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                tmp1[i * 3 + j] = 0.0;
                tmp1[i * 3 + j] += kMU * (qr_grad[3 * i + j] + qr_grad[3 * j + i]);
                tmp1[i * 3 + j] += kRHO * tau[0] * rLi[i] * uadv[j];
                tmp1[i * 3 + j] -= kRHO * tau[0] * tau[0] * rLi[i] * rLi[j];
            }
        for (int i = 0; i < 3; ++i) tmp1[i * 3 + i] += -qr_wg[NQR * 3 + iq] + kRHO * tau[1] * divu;
        for (int aa = 0; aa < NSHL; ++aa)
            for (int ii = 0; ii < 3; ++ii) {
                f64 bm = 0.0;
                bm += shlu[aa * NQR + iq] * tmp0[ii];
                bm += shgradg[aa * 3 + 0] * tmp1[ii * 3 + 0];
                bm += shgradg[aa * 3 + 1] * tmp1[ii * 3 + 1];
                bm += shgradg[aa * 3 + 2] * tmp1[ii * 3 + 2];
                elem_F[M2D(aa, ii)] += bm * gw[iq] * detJ;
            }
        for (int aa = 0; aa < NSHL; ++aa) {
            f64 bc = 0.0;
            bc += shlu[aa * NQR + iq] * divu;
            bc += tau[0] * rLi[0] * shgradg[aa * 3 + 0];
            bc += tau[0] * rLi[1] * shgradg[aa * 3 + 1];
            bc += tau[0] * rLi[2] * shgradg[aa * 3 + 2];
            elem_F[M2D(aa, 3)] += bc * gw[iq] * detJ;
        }
        for (int aa = 0; aa < NSHL; ++aa) {
            f64 bp = qr_dwg[NQR * 4 + iq] + uadv[0] * qr_grad[3 * 4 + 0] + uadv[1] * qr_grad[3 * 4 + 1] +
                     uadv[2] * qr_grad[3 * 4 + 2];
            elem_F[M2D(aa, 4)] += bp * (shlu[aa * NQR + iq] + tau[2] * shconv[aa]) * gw[iq] * detJ;
        }
        for (int aa = 0; aa < NSHL; ++aa) {
            f64 bt = kRHO * kCP *
                     (qr_dwg[NQR * 5 + iq] + uadv[0] * qr_grad[3 * 5 + 0] + uadv[1] * qr_grad[3 * 5 + 1] +
                      uadv[2] * qr_grad[3 * 5 + 2]) *
                     (shlu[aa * NQR + iq] + kRHO * kCP * tau[3] * shconv[aa]);
            bt += kKAPPA * (qr_grad[3 * 5 + 0] * shgradg[aa * 3 + 0] + qr_grad[3 * 5 + 1] * shgradg[aa * 3 + 1] +
                            qr_grad[3 * 5 + 2] * shgradg[aa * 3 + 2]);
            elem_F[M2D(aa, 5)] += bt * gw[iq] * detJ;
        }
    }
}

// AssembleWeakFormLHSKernel (the live "shared" variant, Q6), src/assemble.cu:495-759.
// elem_J[(aa*4+bb)*36 + i*6 + j]
static void elem_lhs(const ElemGeom& g, const f64* qr_wg, f64* elem_J) {
    const f64 fact1 = kALPHAM;
    const f64 fact2 = kDT * kALPHAF * kGAMMA;
    const f64* shgradl = g.shgrad;
    const f64 detJ = g.detJ;
    const f64 knu = kMU / kRHO;
    f64 gg = 0.0, tr = 0.0;
    for (int i = 0; i < 9; ++i) {  // :528-533
        f64 gij = g.G[i];
        gg += gij * gij;
        tr += gij * (f64)(!(i & 0x3));
    }
    const f64 tc0 = gg, tc1 = 1.0 / tr;
    f64 buf[16][16];
    for (int p = 0; p < 16; ++p) for (int i = 0; i < 16; ++i) buf[p][i] = 0.0;
    for (int iq = 0; iq < NQR; ++iq) {
        f64 shconv[4];
        for (int lane = 0; lane < NSHL; ++lane) {  // :574-583
            shconv[lane] = 0.0;
            shconv[lane] += shgradl[lane * 3 + 0] * qr_wg[0 * NQR + iq];
            shconv[lane] += shgradl[lane * 3 + 1] * qr_wg[1 * NQR + iq];
            shconv[lane] += shgradl[lane * 3 + 2] * qr_wg[2 * NQR + iq];
        }
        f64 tmp = 0;  // :592-602
        tmp += shconv[1] * shconv[1];
        tmp += shconv[2] * shconv[2];
        tmp += shconv[3] * shconv[3];
        f64 tau0 = (1.0 / std::sqrt(4.0 / (kDT * kDT) + tmp + 3.0 * knu * knu * tc0)) / kRHO;
        f64 tau1 = std::sqrt(tmp + 3.0 * knu * knu * tc0) * tc1;
        for (int aa = 0; aa < NSHL; ++aa)
            for (int bb = 0; bb < NSHL; ++bb) {
                f64* B = buf[aa * 4 + bb];
                f64 eK = shgradl[aa * 3 + 0] * shgradl[bb * 3 + 0] + shgradl[aa * 3 + 1] * shgradl[bb * 3 + 1] +
                         shgradl[aa * 3 + 2] * shgradl[bb * 3 + 2];
                f64 detJgw = detJ * gw[iq];
                f64 t = 0.0;
                t += fact1 * kRHO * shlu[aa * NQR + iq] * shlu[bb * NQR + iq];
                t += fact1 * kRHO * kRHO * tau0 * shconv[aa] * shlu[bb * NQR + iq];
                t += fact2 * shlu[aa * NQR + iq] * kRHO * shconv[bb];
                t += fact2 * tau0 * kRHO * shconv[aa] * kRHO * shconv[bb];
                t += fact2 * kMU * eK;
                B[0 * 4 + 0] += t * detJgw;
                B[1 * 4 + 1] += t * detJgw;
                B[2 * 4 + 2] += t * detJgw;
                for (int ii = 0; ii < 3; ++ii)
                    for (int jj = 0; jj < 3; ++jj) {
                        B[ii * 4 + jj] += fact2 * kMU * shgradl[aa * 3 + jj] * shgradl[bb * 3 + ii] * detJgw;
                        B[ii * 4 + jj] += fact2 * kRHO * tau1 * shgradl[aa * 3 + ii] * shgradl[bb * 3 + jj] * detJgw;
                    }
                for (int ii = 0; ii < 3; ++ii) {  // dRM/dP
                    B[ii * 4 + 3] -= shgradl[aa * 3 + ii] * shlu[bb * NQR + iq] * detJgw;
                    B[ii * 4 + 3] += kRHO * tau0 * shconv[aa] * shgradl[bb * 3 + ii] * detJgw;
                }
                for (int ii = 0; ii < 3; ++ii) {  // dRC/dU
                    B[3 * 4 + ii] += fact1 * kRHO * tau0 * shgradl[aa * 3 + ii] * shlu[bb * NQR + iq] * detJgw;
                    B[3 * 4 + ii] += fact2 * shlu[aa * NQR + iq] * shgradl[bb * 3 + ii] * detJgw;
                    B[3 * 4 + ii] += fact2 * tau0 * shgradl[aa * 3 + ii] * kRHO * shconv[bb] * detJgw;
                }
                B[3 * 4 + 3] += tau0 * eK * detJgw;  // dRC/dP
            }
    }
    for (int i = 0; i < 16 * 36; ++i) elem_J[i] = 0.0;  // cudaMemsetAsync, :1366
    for (int p = 0; p < 16; ++p) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) elem_J[p * 36 + i * BS + j] += buf[p][i * 4 + j];
        int aa = p / 4, bb = p % 4;
        elem_J[p * 36 + M2D(4, 4)] += (f64)(aa == bb);
        elem_J[p * 36 + M2D(5, 5)] += (f64)(aa == bb);
    }
}

// FaceAssemblyKernel, src/assemble.cu:1038-1214.  elem_invJ here is J^{-1}
// (no metric gemm on the face path), nv from GetElemFaceNVKernel (:279-319).
static void face_kernel(const f64* elem_invJ, const f64* nv, const f64* shgradg, i32 iorn, const f64* qr_wg,
                        const f64* qr_grad, f64* elem_F, f64* elem_J) {
    f64 hinv = 0.0, detJb = 0.0;
    f64 uadv[3];
    for (int i = 0; i < 3; ++i) {
        uadv[i] = elem_invJ[i + 3 * 0] * nv[0] + elem_invJ[i + 3 * 1] * nv[1] + elem_invJ[i + 3 * 2] * nv[2];
        hinv += uadv[i] * uadv[i];
        detJb += nv[i] * nv[i];
    }
    detJb = std::sqrt(detJb);
    (void)detJb;
    hinv = std::sqrt(hinv);
    f64 tau_b = 4.0 * kMU * hinv;
    const f64* sb = shlub + NQRB * NSHL * iorn;
    if (elem_F) {
        for (int i = 0; i < NSHL * BS; ++i) elem_F[i] = 0.0;
        f64 tmp0[3], tmp1[9];
        for (int iq = 0; iq < NQRB; ++iq) {
            uadv[0] = qr_wg[NQRB * 0 + iq];
            uadv[1] = qr_wg[NQRB * 1 + iq];
            uadv[2] = qr_wg[NQRB * 2 + iq];
            f64 unor = uadv[0] * nv[0] + uadv[1] * nv[1] + uadv[2] * nv[2];
            f64 uneg = (unor - std::fabs(unor)) * 0.5;
            for (int i = 0; i < 3; ++i) {
                tmp0[i] = 0.0;
                tmp0[i] += nv[i] * qr_wg[NQRB * 3 + iq];
                tmp0[i] -= kMU * (nv[0] * qr_grad[3 * i + 0] + nv[1] * qr_grad[3 * i + 1] + nv[2] * qr_grad[3 * i + 2]);
                tmp0[i] -= kMU * (nv[0] * qr_grad[3 * 0 + i] + nv[1] * qr_grad[3 * 1 + i] + nv[2] * qr_grad[3 * 2 + i]);
                tmp0[i] -= kRHO * uneg * uadv[i];
                tmp0[i] += tau_b * uadv[i];
            }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) tmp1[i * 3 + j] = -kMU * (nv[i] * uadv[j] + nv[j] * uadv[i]);
            for (int aa = 0; aa < NSHL; ++aa) {
                for (int ii = 0; ii < 3; ++ii) {
                    f64 bm = 0.0;
                    bm += sb[iq * NSHL + aa] * tmp0[ii];
                    bm += shgradg[aa * 3 + 0] * tmp1[ii * 3 + 0];
                    bm += shgradg[aa * 3 + 1] * tmp1[ii * 3 + 1];
                    bm += shgradg[aa * 3 + 2] * tmp1[ii * 3 + 2];
                    elem_F[M2D(aa, ii)] += bm * gwb[iq];
                }
                elem_F[M2D(aa, 3)] -= sb[iq * NSHL + aa] * unor * gwb[iq];
            }
        }
    }
    if (elem_J) {
        const f64 fact2 = kDT * kALPHAF * kGAMMA;
        f64 tmp0;
        for (int i = 0; i < NSHL * NSHL * BS * BS; ++i) elem_J[i] = 0.0;
        f64 shnorm[NSHL];
        for (int aa = 0; aa < NSHL; ++aa) {
            shnorm[aa] = 0.0;
            shnorm[aa] += shgradg[aa * 3 + 0] * nv[0];
            shnorm[aa] += shgradg[aa * 3 + 1] * nv[1];
            shnorm[aa] += shgradg[aa * 3 + 2] * nv[2];
        }
#define M4D(aa, bb, ii, jj) (((aa) * NSHL + (bb)) * (BS * BS) + (ii) * BS + (jj))
        for (int iq = 0; iq < NQRB; ++iq) {
            uadv[0] = qr_wg[NQRB * 0 + iq];
            uadv[1] = qr_wg[NQRB * 1 + iq];
            uadv[2] = qr_wg[NQRB * 2 + iq];
            f64 unor = uadv[0] * nv[0] + uadv[1] * nv[1] + uadv[2] * nv[2];
            f64 uneg = (unor - std::fabs(unor)) * 0.5;
            for (int aa = 0; aa < NSHL; ++aa)
                for (int bb = 0; bb < NSHL; ++bb) {
                    f64 sa = sb[iq * NSHL + aa], sbb = sb[iq * NSHL + bb];
                    tmp0 = 0.0;
                    tmp0 -= kMU * (shnorm[bb] * sa + shnorm[aa] * sbb);
                    tmp0 -= kRHO * sa * sbb * uneg;
                    tmp0 += tau_b * sa * sbb;
                    elem_J[M4D(aa, bb, 0, 0)] += fact2 * tmp0 * gwb[iq];
                    elem_J[M4D(aa, bb, 1, 1)] += fact2 * tmp0 * gwb[iq];
                    elem_J[M4D(aa, bb, 2, 2)] += fact2 * tmp0 * gwb[iq];
                    for (int ii = 0; ii < 3; ++ii)
                        for (int jj = 0; jj < 3; ++jj) {
                            tmp0 = 0.0;
                            tmp0 -= kMU * sa * shgradg[bb * 3 + ii] * nv[jj];
                            tmp0 -= kMU * sbb * shgradg[aa * 3 + jj] * nv[ii];
                            elem_J[M4D(aa, bb, ii, jj)] += fact2 * tmp0 * gwb[iq];
                        }
                    tmp0 = sa * sbb;
                    for (int ii = 0; ii < 3; ++ii) {
                        elem_J[M4D(aa, bb, 3, ii)] -= fact2 * tmp0 * nv[ii] * gwb[iq];
                        elem_J[M4D(aa, bb, ii, 3)] += tmp0 * nv[ii] * gwb[iq];
                    }
                }
        }
    }
}

// Scatter of one (a,b) 6x6 block into the four reference-layout value arrays:
// SetBlockValueToSubmatKernel, src/matrix_impl.cu:370-453 (alpha = beta = 1,
// offsets {0,3,4,5,6}, only (0,0),(0,1),(1,0),(1,1) populated, src/main.c:385-391).
struct FSMat {
    const i32* row_ptr;  // nodal 1x1 pattern
    const i32* col_ind;
    f64* m[2][2];        // A00 (3x3), A01 (3x1), A10 (1x3), A11 (1x1)
};
static const int fs_off[3] = {0, 3, 4};

static void scatter_block(const FSMat& A, i32 row, i32 col, const f64* val /*6x6, lda 6*/) {
    i32 start = A.row_ptr[row], end = A.row_ptr[row + 1], len = end - start, k;
    for (k = start; k < end; ++k) if (A.col_ind[k] == col) break;
    for (int i = 0; i < 2; ++i) {
        int br = fs_off[i + 1] - fs_off[i];
        for (int j = 0; j < 2; ++j) {
            int bc = fs_off[j + 1] - fs_off[j];
            f64* m = A.m[i][j];
            if (!m) continue;
            m += (size_t)start * br * bc + (size_t)(k - start) * bc;
            for (int ii = 0; ii < br; ++ii)
                for (int jj = 0; jj < bc; ++jj)
                    m[(size_t)ii * len * bc + jj] = 1.0 * m[(size_t)ii * len * bc + jj] +
                                                    1.0 * val[(fs_off[i] + ii) * BS + (fs_off[j] + jj)];
        }
    }
}

static void interp_volume(const ElemGeom& g, const f64* buffer, const f64* dbuffer, f64* qr_wg, f64* qr_dwg, f64* qr_grad) {
    // qr_wggradalpha: Dgemm 3xBSxNSHL (src/assemble.cu:1628-1635)
    for (int comp = 0; comp < BS; ++comp)
        for (int d = 0; d < 3; ++d) {
            f64 s = 0.0;
            for (int a = 0; a < NSHL; ++a) s += g.shgrad[d + 3 * a] * buffer[a + NSHL * comp];
            qr_grad[d + 3 * comp] = s;
        }
    // qr_wgalpha: A = d_shlu lda NQR, OP_N (:1648-1655)
    for (int comp = 0; comp < BS; ++comp)
        for (int q = 0; q < NQR; ++q) {
            f64 s = 0.0;
            for (int a = 0; a < NSHL; ++a) s += shlu[q + NQR * a] * buffer[a + NSHL * comp];
            qr_wg[q + NQR * comp] = s;
        }
    if (qr_dwg) {  // OP_T, lda NSHL (:1685-1693)
        for (int comp = 0; comp < BS; ++comp)
            for (int q = 0; q < NQR; ++q) {
                f64 s = 0.0;
                for (int a = 0; a < NSHL; ++a) s += shlu[a + NSHL * q] * dbuffer[a + NSHL * comp];
                qr_dwg[q + NQR * comp] = s;
            }
    }
}

extern "C" {

// ---- element-level probes (used by analytic tests) ----------------------------
void orc_elem_geometry(const f64* xg, const i32* nodes, f64* invJ, f64* detJ, f64* shgrad, f64* G) {
    ElemGeom g;
    elem_geometry(xg, nodes, g);
    memcpy(invJ, g.invJ, sizeof g.invJ);
    *detJ = g.detJ;
    memcpy(shgrad, g.shgrad, sizeof g.shgrad);
    memcpy(G, g.G, sizeof g.G);
}

void orc_elem_tensors(const f64* xg, const i32* nodes, i32 N, const f64* wg, const f64* dwg, f64* elem_F, f64* elem_J,
                      f64* qr_wg_out, f64* qr_dwg_out, f64* qr_grad_out) {
    ElemGeom g;
    elem_geometry(xg, nodes, g);
    f64 buffer[24], dbuffer[24], qr_wg[24], qr_dwg[24], qr_grad[18];
    gather_fields(nodes, N, wg, dwg, buffer);
    gather_fields(nodes, N, dwg, dwg, dbuffer);
    interp_volume(g, buffer, dbuffer, qr_wg, qr_dwg, qr_grad);
    if (elem_F) elem_rhs(g, qr_wg, qr_dwg, qr_grad, elem_F);
    if (elem_J) elem_lhs(g, qr_wg, elem_J);
    if (qr_wg_out) memcpy(qr_wg_out, qr_wg, sizeof qr_wg);
    if (qr_dwg_out) memcpy(qr_dwg_out, qr_dwg, sizeof qr_dwg);
    if (qr_grad_out) memcpy(qr_grad_out, qr_grad, sizeof qr_grad);
}

// AssemleWeakFormKernelHeat (dead in the reference, src/assemble.cu:377-443) --
// kept only as the independent Poisson/heat known-answer probe of SURVEY 8(c)-(4).
void orc_elem_heat(const f64* xg, const i32* nodes, f64* elem_J /*4x4*/) {
    ElemGeom g;
    elem_geometry(xg, nodes, g);
    const f64 fact1 = kALPHAM, fact2 = kDT * kALPHAF * kGAMMA;
    for (int lane = 0; lane < 4; ++lane) {
        for (int aa = 0; aa < 4; ++aa) elem_J[aa * 4 + lane] = 0.0;
        for (int aa = 0; aa < 4; ++aa)
            for (int q = 0; q < 4; ++q)
                elem_J[aa * 4 + lane] += fact1 * g.detJ * gw[q] * shlu[NQR * aa + q] * shlu[lane * NQR + q];
        for (int aa = 0; aa < 4; ++aa)
            elem_J[aa * 4 + lane] += fact2 * g.detJ * (1.0 / 6.0) *
                                     (g.shgrad[aa * 3 + 0] * g.shgrad[lane * 3 + 0] + g.shgrad[aa * 3 + 1] * g.shgrad[lane * 3 + 1] +
                                      g.shgrad[aa * 3 + 2] * g.shgrad[lane * 3 + 2]);
    }
}

// ---- volume assembly: AssembleSystemTet, src/assemble.cu:1467-1762 ------------
// F (6N) and/or the four value arrays are accumulated batch by batch in
// batch_ind order (colored, non-atomic scatter: :188-208, matrix_impl.cu:447).
void orc_assemble_tet(const f64* xg, const i32* ien, i32 N, i32 num_batch, const i32* batch_offset, const i32* batch_ind,
                      const f64* wg, const f64* dwg, f64* F, const i32* row_ptr, const i32* col_ind, f64* A00, f64* A01,
                      f64* A10, f64* A11) {
    FSMat A;
    A.row_ptr = row_ptr; A.col_ind = col_ind;
    A.m[0][0] = A00; A.m[0][1] = A01; A.m[1][0] = A10; A.m[1][1] = A11;
    const bool doJ = (A00 != nullptr);
    for (i32 b = 0; b < num_batch; ++b) {
        i32 bsz = batch_offset[b + 1] - batch_offset[b];
        if (bsz == 0) break;  // :1565-1567
        const i32* bidx = batch_ind + batch_offset[b];
        // elements of one color share no node: the loop is race-free and its result independent of the
        // thread count (cpu_baseline's multi-core leg; orc_set_threads(1) = the serial restatement)
#pragma omp parallel for schedule(static)
        for (i32 e = 0; e < bsz; ++e) {
            f64 elem_J[16 * 36];
            i32 iel = bidx[e];
            const i32* nodes = ien + (size_t)iel * 4;
            ElemGeom g;
            elem_geometry(xg, nodes, g);
            f64 buffer[24], dbuffer[24], qr_wg[24], qr_dwg[24], qr_grad[18];
            gather_fields(nodes, N, wg, dwg, buffer);
            gather_fields(nodes, N, dwg, dwg, dbuffer);
            interp_volume(g, buffer, dbuffer, qr_wg, qr_dwg, qr_grad);
            if (F) {
                f64 eF[24];
                elem_rhs(g, qr_wg, qr_dwg, qr_grad, eF);
                for (int a = 0; a < 4; ++a) {  // ElemRHSLocal2Global x4, :1709-1724
                    size_t n = (size_t)nodes[a];
                    for (int j = 0; j < 3; ++j) F[n * 3 + j] += eF[a * BS + j];
                    F[(size_t)N * 3 + n] += eF[a * BS + 3];
                    F[(size_t)N * 4 + n] += eF[a * BS + 4];
                    F[(size_t)N * 5 + n] += eF[a * BS + 5];
                }
            }
            if (doJ) {
                elem_lhs(g, qr_wg, elem_J);
                for (int aa = 0; aa < 4; ++aa)
                    for (int bb = 0; bb < 4; ++bb) scatter_block(A, nodes[aa], nodes[bb], elem_J + (aa * 4 + bb) * 36);
            }
        }
    }
}

// ---- face assembly: AssembleSystemTetFace, src/assemble.cu:1764-1964 -----------
// One boundary group (the reference only runs b == 4, :1826-1828); scatter is
// repeated per color with a parent-element mask (:1916-1945).
void orc_assemble_face(const f64* xg, const i32* ien, i32 N, i32 num_face, const i32* f2e, const i32* forn, const i32* color,
                       i32 num_color, const f64* wg, const f64* dwg, f64* F, const i32* row_ptr, const i32* col_ind, f64* A00,
                       f64* A01, f64* A10, f64* A11) {
    FSMat A;
    A.row_ptr = row_ptr; A.col_ind = col_ind;
    A.m[0][0] = A00; A.m[0][1] = A01; A.m[1][0] = A10; A.m[1][1] = A11;
    const bool doJ = (A00 != nullptr);
    std::vector<f64> eF((size_t)num_face * 24), eJ(doJ ? (size_t)num_face * 576 : 0);
    for (i32 f = 0; f < num_face; ++f) {
        const i32* nodes = ien + (size_t)f2e[f] * 4;
        ElemGeom g;
        elem_geometry(xg, nodes, g);
        i32 iorn = forn[f];
        f64 nv[3], b[3] = {0.0, 0.0, 0.0};  // GetElemFaceNVKernel (Nanson), :306-317
        for (int k = 0; k < 3; ++k)
            for (int n = 0; n < 3; ++n) b[n] += g.invJ[n * 3 + k] * nv2[iorn * 3 + k];
        nv[0] = b[0] * g.detJ; nv[1] = b[1] * g.detJ; nv[2] = b[2] * g.detJ;
        f64 buffer[24];
        for (int i = 0; i < 24; ++i) buffer[i] = 0.0;  // zero-filled allocation, src/alloc.c:23-30
        for (int a = 0; a < 4; ++a) {                   // :1841-1848 (u and p only)
            size_t n = (size_t)nodes[a];
            buffer[0 * 4 + a] = wg[n * 3 + 0];
            buffer[1 * 4 + a] = wg[n * 3 + 1];
            buffer[2 * 4 + a] = wg[n * 3 + 2];
            buffer[3 * 4 + a] = dwg[(size_t)N * 3 + n];
        }
        f64 qr_grad[18], qb[18];
        for (int comp = 0; comp < BS; ++comp)
            for (int d = 0; d < 3; ++d) {
                f64 s = 0.0;
                for (int a = 0; a < 4; ++a) s += g.shgrad[d + 3 * a] * buffer[a + 4 * comp];
                qr_grad[d + 3 * comp] = s;
            }
        for (int comp = 0; comp < BS; ++comp)  // DgemmBatched OP_T, :1867-1875
            for (int q = 0; q < NQRB; ++q) {
                f64 s = 0.0;
                for (int a = 0; a < 4; ++a) s += shlub[iorn * 12 + a + 4 * q] * buffer[a + 4 * comp];
                qb[q + NQRB * comp] = s;
            }
        face_kernel(g.invJ, nv, g.shgrad, iorn, qb, qr_grad, F ? &eF[(size_t)f * 24] : nullptr, doJ ? &eJ[(size_t)f * 576] : nullptr);
    }
    for (i32 c = 0; c < num_color; ++c)
        for (i32 f = 0; f < num_face; ++f) {
            if (color[f2e[f]] != c) continue;  // SetupMaskKernel, :1293-1299
            const i32* nodes = ien + (size_t)f2e[f] * 4;
            if (F) {
                const f64* e = &eF[(size_t)f * 24];
                for (int a = 0; a < 4; ++a) {
                    size_t n = (size_t)nodes[a];
                    for (int j = 0; j < 3; ++j) F[n * 3 + j] += e[a * BS + j];
                    F[(size_t)N * 3 + n] += e[a * BS + 3];
                    F[(size_t)N * 4 + n] += e[a * BS + 4];
                    F[(size_t)N * 5 + n] += e[a * BS + 5];
                }
            }
            if (doJ)
                for (int aa = 0; aa < 4; ++aa)
                    for (int bb = 0; bb < 4; ++bb) scatter_block(A, nodes[aa], nodes[bb], &eJ[(size_t)f * 576 + (aa * 4 + bb) * 36]);
        }
}

// ---- CSR pattern: src/csr.c:36-190 (sorted fixed-width rows, PREALLOC 64) ------
// returns nnz, or -1 on row overflow (the reference ASSERTs, csr.c:63)
i32 orc_csr_pattern(const i32* ien, i32 T, i32 N, i32* row_ptr, i32* col_ind /*capacity 64*N or NULL*/) {
    const int PRE = 64;
    std::vector<i32> buff((size_t)N * PRE, 0), row_len(N, 0);
    auto push = [&](i32 key, i32 value) -> bool {
        i32* row = &buff[(size_t)key * PRE];
        i32 len = row_len[key];
        i32* it = std::lower_bound(row, row + len, value);
        if (it != row + len && *it == value) return true;
        if (len >= PRE) return false;
        memmove(it + 1, it, (size_t)(row + len - it) * sizeof(i32));
        *it = value;
        row_len[key]++;
        return true;
    };
    for (i32 k = 0; k < T; ++k) {
        const i32* e = ien + (size_t)k * 4;
        for (int i = 0; i < 4; ++i) {
            if (!push(e[i], e[i])) return -1;
            for (int j = 0; j < 4; ++j)
                if (i != j && !push(e[i], e[j])) return -1;
        }
    }
    row_ptr[0] = 0;
    for (i32 i = 0; i < N; ++i) row_ptr[i + 1] = row_ptr[i] + row_len[i];
    if (col_ind)
        for (i32 i = 0; i < N; ++i) memcpy(col_ind + row_ptr[i], &buff[(size_t)i * PRE], (size_t)row_len[i] * sizeof(i32));
    return row_ptr[N];
}

// ExpandCSRByBlockSize: SetRowLength + SetColIndex, src/csr_impl.cu:24-59 (Q3 fixed)
void orc_csr_expand(const i32* row_ptr, const i32* col_ind, i32 N, i32 br, i32 bc, i32* new_row_ptr, i32* new_col_ind) {
    for (i32 i = 0; i < N; ++i) {
        i32 start = row_ptr[i], len = row_ptr[i + 1] - start;
        for (i32 j = 0; j < br; ++j) new_row_ptr[i * br + j] = start * br * bc + j * bc * len;
    }
    new_row_ptr[N * br] = row_ptr[N] * br * bc;  // Q3: the reference never writes this entry
    for (i32 i = 0; i < N; ++i) {
        i32 start = row_ptr[i], len = row_ptr[i + 1] - start;
        for (i32 j = 0; j < br; ++j)
            for (i32 k = 0; k < len; ++k)
                for (i32 l = 0; l < bc; ++l) new_col_ind[new_row_ptr[i * br + j] + k * bc + l] = col_ind[start + k] * bc + l;
    }
}

// ---- vertex -> element map: src/color_impl.cu:17-61 (order inside a row is a set;
// the oracle emits ascending element ids)
void orc_v2e(const i32* ien, i32 T, i32 N, i32* row_ptr, i32* col) {
    for (i32 i = 0; i <= N; ++i) row_ptr[i] = 0;
    for (i32 e = 0; e < T; ++e) for (int j = 0; j < 4; ++j) row_ptr[ien[(size_t)e * 4 + j] + 1] += 1;
    for (i32 i = 0; i < N; ++i) row_ptr[i + 1] += row_ptr[i];
    std::vector<i32> cnt(N, 0);
    for (i32 e = 0; e < T; ++e)
        for (int j = 0; j < 4; ++j) {
            i32 n = ien[(size_t)e * 4 + j];
            col[row_ptr[n] + cnt[n]++] = e;
        }
}

// priorities: GenerateRandomColor, src/color_impl.cu:185-192,225-237
void orc_priorities_from_u32(const uint32_t* raw, i32 T, i32* prio) {
    const int ub = INT32_MAX / 2, lb = 0;
    for (i32 i = 0; i < T; ++i) prio[i] = (i32)(raw[i] % (uint32_t)(ub - lb) + lb);
}

// Jones-Plassmann-Luby, synchronous rounds: src/color_impl.cu:64-183.
// color[] holds priorities on entry, colors 0..num_color-1 on exit.
// Returns num_color; *num_ties counts adjacent uncolored pairs with equal priority
// seen while deciding maxima (Q1).
i32 orc_color_jpl(const i32* ien, i32 T, i32 N, const i32* v2e_row, const i32* v2e_col, i32* color, i32 max_color,
                  i32 tie_break_by_index, i32* num_ties) {
    // One round = ColorElementJPLKernel over every element, then ReverseColorKernel / SetUpFlagKernel (synchronous: a round
    // reads the colors of the previous one).  Colored elements return at once in the kernel (ec < 0), so only the still
    // uncolored ones are visited here -- an `active` list compacted after each round, same result -- and the visits of a
    // round are independent of each other (OpenMP over the list; orc_set_threads(1) = serial).
    const i32 MARK = INT32_MAX / 2 + 1;
    std::vector<i32> active(T), nxt(T);
    for (i32 i = 0; i < T; ++i) active[i] = i;
    i32 nact = T;
    long long ties = 0;
    i32 c = 0;
    for (; c < max_color && nact > 0; ++c) {
        long long round_ties = 0;
#pragma omp parallel for schedule(static) reduction(+ : round_ties)
        for (i32 q = 0; q < nact; ++q) {
            const i32 i = active[q];
            const i32 ec = color[i];
            bool found_max = true;
            for (int j = 0; j < 4; ++j) {
                i32 node = ien[(size_t)i * 4 + j];
                for (i32 k = v2e_row[node]; k < v2e_row[node + 1]; ++k) {
                    i32 el = v2e_col[k];
                    if (el == i) continue;
                    i32 oc = color[el];
                    if (oc < 0) continue;
                    if (ec < oc) found_max = false;
                    else if (ec == oc) {
                        ++round_ties;
                        if (tie_break_by_index && i < el) found_max = false;
                    }
                }
            }
            nxt[q] = found_max ? MARK : ec;
        }
        ties += round_ties;
        i32 keep = 0;
        for (i32 q = 0; q < nact; ++q) {
            const i32 i = active[q];
            if (nxt[q] == MARK) color[i] = -1 - c;  // ReverseColorKernel
            else active[keep++] = i;                 // SetUpFlagKernel + cub Max: anything left?
        }
        nact = keep;
    }
    for (i32 i = 0; i < T; ++i) color[i] = color[i] * (-1) - 1;  // RecoverColorKernel
    if (num_ties) *num_ties = (i32)(ties > INT32_MAX ? INT32_MAX : ties);
    (void)N;
    return c;
}

// The same coloring without the rounds (bench.py's cpu_baseline set-up at 10M tets, where 141 rounds over every uncolored
// element take minutes): an element becomes a local maximum among its uncolored neighbours in the round after its last
// BLOCKING neighbour was colored -- blocking = strictly higher priority (color_impl.cu:87: `ec < color[elem]`), or equal
// priority and higher index under the tie break -- so color(i) = 1 + max color over its blocking neighbours (0 without any),
// evaluated in descending (priority, index) order `order[T]`.  Checked against orc_color_jpl in tests/test_oracle_cpu.py.
i32 orc_color_jpl_sorted(const i32* ien, i32 T, const i32* v2e_row, const i32* v2e_col, const i32* prio, const i32* order,
                         i32* color, i32 tie_break_by_index) {
    i32 num_color = 0;
    for (i32 q = 0; q < T; ++q) {
        const i32 i = order[q];
        const i32 pi = prio[i];
        i32 c = 0;
        for (int j = 0; j < 4; ++j) {
            const i32 node = ien[(size_t)i * 4 + j];
            for (i32 k = v2e_row[node]; k < v2e_row[node + 1]; ++k) {
                const i32 el = v2e_col[k];
                if (el == i) continue;
                const i32 pe = prio[el];
                if (pe > pi || (pe == pi && tie_break_by_index && el > i)) {
                    if (color[el] + 1 > c) c = color[el] + 1;
                }
            }
        }
        color[i] = c;
        if (c + 1 > num_color) num_color = c + 1;
    }
    return num_color;
}

// per-color batches: src/Mesh.c:165-206, src/indexing.cu:92-102 (count + stable copy_if)
void orc_batches(const i32* color, i32 T, i32 num_color, i32* batch_offset, i32* batch_ind) {
    batch_offset[0] = 0;
    for (i32 c = 0; c < num_color; ++c) {
        i32 n = 0;
        for (i32 i = 0; i < T; ++i) n += (color[i] == c);
        batch_offset[c + 1] = batch_offset[c] + n;
    }
    for (i32 c = 0; c < num_color; ++c) {
        i32 p = batch_offset[c];
        for (i32 i = 0; i < T; ++i) if (color[i] == c) batch_ind[p++] = i;
    }
}

// ---- Dirichlet: src/dirichlet.c:31-61, dirichlet_impl.cu:15-36, matrix.c:449-469,
//      matrix_impl.cu:6-23 ------------------------------------------------------
void orc_dirichlet_vec(f64* b, i32 n_bnode, const i32* bnode, i32 shape, const i32* bctype) {
    for (i32 ic = 0; ic < shape; ++ic)
        if (bctype[ic] == 1)
            for (i32 i = 0; i < n_bnode; ++i) b[(size_t)bnode[i] * shape + ic] = 0.0;
}

// rows of A00 <- unit rows, rows of A01 <- 0; the pressure block-row call is a
// no-op (negative count, matrix.c:464).  row_ptr3x3 / col3x3 etc. are the expanded patterns.
void orc_dirichlet_mat(i32 n_bnode, const i32* bnode, i32 shape, const i32* bctype, i32 N, const i32* rp33, const i32* ci33,
                       f64* A00, const i32* rp31, const i32* ci31, f64* A01) {
    for (i32 ic = 0; ic < shape; ++ic) {
        if (bctype[ic] != 1) continue;
        for (i32 i = 0; i < n_bnode; ++i) {
            i32 ir = bnode[i] * shape + ic;
            if (ir < 0 || ir >= 3 * N) continue;
            for (i32 j = rp33[ir]; j < rp33[ir + 1]; ++j) A00[j] = 1.0 * (f64)(ci33[j] == ir);
            for (i32 j = rp31[ir]; j < rp31[ir + 1]; ++j) A01[j] = 0.0 * (f64)(ci31[j] == ir);
        }
    }
}

// ---- scalar CSR SpMV y = alpha*A*x + beta*y (cusparseSpMV semantics, matrix.c:101-165)
void orc_csr_spmv(i32 nrow, const i32* rp, const i32* ci, const f64* val, f64 alpha, const f64* x, f64 beta, f64* y) {
#pragma omp parallel for schedule(static)
    for (i32 i = 0; i < nrow; ++i) {
        f64 s = 0.0;
        for (i32 j = rp[i]; j < rp[i + 1]; ++j) s += val[j] * x[ci[j]];
        y[i] = alpha * s + beta * y[i];
    }
}

struct FSPat {
    i32 N;
    const i32 *rp33, *ci33, *rp31, *ci31, *rp13, *ci13, *rp11, *ci11;
    const f64 *A00, *A01, *A10, *A11;
};

// MatrixFSAMVPBY, src/matrix.c:471-497: scal of y[0:4N] by beta, then 4 SpMVs with beta = 1
static void fs_amvpby(const FSPat& P, f64 alpha, const f64* x, f64 beta, f64* y) {
    i32 N = P.N;
    for (i32 i = 0; i < 4 * N; ++i) y[i] *= beta;
    orc_csr_spmv(3 * N, P.rp33, P.ci33, P.A00, alpha, x, 1.0, y);
    orc_csr_spmv(3 * N, P.rp31, P.ci31, P.A01, alpha, x + 3 * (size_t)N, 1.0, y);
    orc_csr_spmv(N, P.rp13, P.ci13, P.A10, alpha, x, 1.0, y + 3 * (size_t)N);
    orc_csr_spmv(N, P.rp11, P.ci11, P.A11, alpha, x + 3 * (size_t)N, 1.0, y + 3 * (size_t)N);
}

void orc_fs_amvpby(i32 N, const i32* rp33, const i32* ci33, const i32* rp31, const i32* ci31, const i32* rp13, const i32* ci13,
                   const i32* rp11, const i32* ci11, const f64* A00, const f64* A01, const f64* A10, const f64* A11, f64 alpha,
                   const f64* x, f64 beta, f64* y) {
    FSPat P = {N, rp33, ci33, rp31, ci31, rp13, ci13, rp11, ci11, A00, A01, A10, A11};
    fs_amvpby(P, alpha, x, beta, y);
}

// ---- preconditioner: Decomposition{Jacobi(A00,bs=3), Jacobi(A11,bs=1), None, None}
//      src/krylov.c:439-453, src/pc.c:44-147, src/matrix_impl.cu:25-44,642-683 --------
// dinv33: 9N (memory image after getri: inv(D)^T stored row-major == inv(D^T) col-major, Q7)
void orc_pc_setup(i32 N, const i32* rp1, const i32* ci1, const f64* A00, const f64* A11, f64* dinv33, f64* dinv1) {
    for (i32 idx = 0; idx < N; ++idx) {
        i32 start = rp1[idx], end = rp1[idx + 1], len = end - start, k;
        for (k = start; k < end; ++k) if (ci1[k] == idx) break;
        const f64* mv = A00 + (size_t)start * 9 + (size_t)(k - start) * 3;
        f64 blk[9], inv[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) blk[i * 3 + j] = mv[(size_t)i * len * 3 + j];  // row-major D
        lu3_inverse(blk, inv);  // LAPACK reads the same 9 numbers column-major => inverts D^T
        memcpy(dinv33 + (size_t)idx * 9, inv, sizeof inv);
        dinv1[idx] = 1.0 / A11[k];
    }
}

void orc_pc_apply(i32 N, const f64* dinv33, const f64* dinv1, const f64* x, f64* y) {
    for (i32 n = 0; n < N; ++n) {  // DgemvStridedBatched OP_N on the column-major image
        const f64* A = dinv33 + (size_t)n * 9;
        for (int r = 0; r < 3; ++r) {
            f64 s = 0.0;
            for (int c = 0; c < 3; ++c) s += A[r + 3 * c] * x[(size_t)n * 3 + c];
            y[(size_t)n * 3 + r] = s;
        }
    }
    for (i32 n = 0; n < N; ++n) y[(size_t)3 * N + n] = x[(size_t)3 * N + n] * dinv1[n];
    for (i32 n = 4 * N; n < 6 * N; ++n) y[n] = x[n];
}

static void drotg(f64* a, f64* b, f64* c, f64* s) {  // reference BLAS drotg (cublasDrotg)
    f64 roe = *b;
    if (std::fabs(*a) > std::fabs(*b)) roe = *a;
    f64 scale = std::fabs(*a) + std::fabs(*b), r, z;
    if (scale == 0.0) { *c = 1.0; *s = 0.0; r = 0.0; z = 0.0; }
    else {
        r = scale * std::sqrt((*a / scale) * (*a / scale) + (*b / scale) * (*b / scale));
        r = (roe < 0.0 ? -1.0 : 1.0) * r;
        *c = *a / r; *s = *b / r; z = 1.0;
        if (std::fabs(*a) > std::fabs(*b)) z = *s;
        if (std::fabs(*b) >= std::fabs(*a) && *c != 0.0) z = 1.0 / *c;
    }
    *a = r; *b = z;
}

// dot product as a sum of fixed 4096-element chunk sums: the value does not depend on the thread count
static f64 chunked_dot(const f64* a, const f64* b, size_t n) {
    const size_t CH = 4096, nch = (n + CH - 1) / CH;
    std::vector<f64> part(nch);
#pragma omp parallel for schedule(static)
    for (long long c = 0; c < (long long)nch; ++c) {
        size_t lo = (size_t)c * CH, hi = std::min(n, lo + CH);
        f64 s = 0.0;
        for (size_t i = lo; i < hi; ++i) s += a[i] * b[i];
        part[c] = s;
    }
    f64 s = 0.0;
    for (size_t c = 0; c < nch; ++c) s += part[c];
    return s;
}

// ---- GMRES: GMRESSolvePrivate, src/krylov.c:56-334 + krylov_util.cu:5-19 --------
// n = 6N (Q5).  pc_kind: 0 = none (identity), 1 = the reference tree above.
// res_hist[k] = |beta[k+1]| after iteration k (the reference prints every 20th).
// Returns the number of iterations performed.
i32 orc_gmres(i32 N, const i32* rp33, const i32* ci33, const i32* rp31, const i32* ci31, const i32* rp13, const i32* ci13,
              const i32* rp11, const i32* ci11, const f64* A00, const f64* A01, const f64* A10, const f64* A11, i32 pc_kind,
              f64* x, const f64* b, i32 maxit, f64 atol, f64 rtol, f64* res_hist, f64* rnrm_init_out) {
    FSPat P = {N, rp33, ci33, rp31, ci31, rp13, ci13, rp11, ci11, A00, A01, A10, A11};
    const size_t n = (size_t)6 * N;
    const i32 ldh = ((maxit + 1 + 31) / 32) * 32;
    std::vector<f64> Q(n * (maxit + 1), 0.0), H((size_t)ldh * maxit, 0.0), tmp(2 * n, 0.0), gv(2 * (size_t)maxit, 0.0), beta(maxit + 1, 0.0);
    std::vector<f64> d33, d1;
    if (pc_kind == 1) {
        d33.resize((size_t)9 * N); d1.resize(N);
        orc_pc_setup(N, rp11, ci11, A00, A11, d33.data(), d1.data());
    }
    auto pc_apply = [&](const f64* in, f64* out) {
        if (pc_kind == 1) orc_pc_apply(N, d33.data(), d1.data(), in, out);
        else memcpy(out, in, n * sizeof(f64));
    };
#define QCOL(c) (&Q[(size_t)(c) * n])
#define HCOL(c) (&H[(size_t)(c) * ldh])
    memcpy(QCOL(0), b, n * sizeof(f64));
    fs_amvpby(P, -1.0, x, 1.0, QCOL(0));
    f64 rnrm_init = std::sqrt(chunked_dot(QCOL(0), QCOL(0), n));
    if (rnrm_init_out) *rnrm_init_out = rnrm_init;
    beta[0] = rnrm_init;
    f64 rnrm = 1.0 / rnrm_init;
    for (size_t i = 0; i < n; ++i) QCOL(0)[i] *= rnrm;
    bool converged = false;
    i32 iter = 0;
    while (!converged && iter < maxit) {
        pc_apply(QCOL(iter), tmp.data());
        fs_amvpby(P, 1.0, tmp.data(), 0.0, QCOL(iter + 1));
        f64* w = QCOL(iter + 1);
        for (i32 j = 0; j <= iter; ++j) HCOL(iter)[j] = chunked_dot(QCOL(j), w, n);  // Dgemv OP_T
        for (i32 j = 0; j <= iter; ++j) {  // Dgemv OP_N, alpha = -1
            const f64* q = QCOL(j);
            f64 h = HCOL(iter)[j];
#pragma omp parallel for schedule(static)
            for (long long i = 0; i < (long long)n; ++i) w[i] -= q[i] * h;
        }
        f64 nr = std::sqrt(chunked_dot(w, w, n));
        HCOL(iter)[iter + 1] = nr;
        rnrm = 1.0 / nr;
        for (size_t i = 0; i < n; ++i) w[i] *= rnrm;
        for (i32 i = 0; i < iter; ++i) {  // Drot n = 1
            f64 c = gv[2 * i], s = gv[2 * i + 1];
            f64 xx = HCOL(iter)[i], yy = HCOL(iter)[i + 1];
            HCOL(iter)[i] = c * xx + s * yy;
            HCOL(iter)[i + 1] = c * yy - s * xx;
        }
        drotg(&HCOL(iter)[iter], &HCOL(iter)[iter + 1], &gv[2 * iter], &gv[2 * iter + 1]);
        HCOL(iter)[iter + 1] = 0.0;
        {  // GMRESUpdateResidualUpdateKernel
            f64 b0 = beta[iter];
            beta[iter + 1] = -gv[2 * iter + 1] * b0;
            beta[iter] = b0 * gv[2 * iter];
        }
        if (res_hist) res_hist[iter] = std::fabs(beta[iter + 1]);
        if ((iter + 1) % 20 == 0) {
            rnrm = std::fabs(beta[iter + 1]);
            if (rnrm < atol || rnrm < (rnrm_init + 1e-16) * rtol) converged = true;
        }
        iter++;
    }
    if (iter) {
        for (i32 i = iter - 1; i >= 0; --i) {  // Dtrsv upper, non-unit
            f64 s = beta[i];
            for (i32 j = i + 1; j < iter; ++j) s -= HCOL(j)[i] * beta[j];
            beta[i] = s / HCOL(i)[i];
        }
        for (size_t i = 0; i < n; ++i) tmp[i] = 0.0;
        for (i32 j = 0; j < iter; ++j) {
            const f64* q = QCOL(j);
            for (size_t i = 0; i < n; ++i) tmp[i] += q[i] * beta[j];
        }
        pc_apply(tmp.data(), tmp.data() + n);
        for (size_t i = 0; i < n; ++i) x[i] += tmp[n + i];
    }
#undef QCOL
#undef HCOL
    return iter;
}

}  // extern "C"
