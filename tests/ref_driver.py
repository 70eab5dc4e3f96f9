"""Test-side restatement of SolveFlowSystem / the time-step algebra (src/main.c:77-283, 535-565)
with the CPU oracle as the compute backend.  Test infrastructure only."""
import numpy as np

kRHOC, kDT = 0.5, 5e-2
kALPHAM = (3.0 - kRHOC) / (1.0 + kRHOC)
kALPHAF = 1.0 / (1.0 + kRHOC)
kGAMMA = 0.5 + kALPHAM - kALPHAF


def alpha_states(N, wgold, dwgold, dwg):
    dwga = (1.0 - kALPHAM) * dwgold + kALPHAM * dwg
    dwga[3 * N:4 * N] = dwg[3 * N:4 * N]
    wga = wgold + kDT * kALPHAF * (1.0 - kGAMMA) * dwgold + kDT * kALPHAF * kGAMMA * dwg
    wga[3 * N:4 * N] = 0.0
    return wga, dwga


def norms(N, F):
    return np.array([np.linalg.norm(F[:3 * N]), np.linalg.norm(F[3 * N:4 * N]), np.linalg.norm(F[4 * N:5 * N]),
                     np.linalg.norm(F[5 * N:])])


def solve_flow_system(S, wgold, dwgold, dwg, maxit=4, tol=0.5e-3):
    N = S.N
    dwg = dwg.copy()
    wga, dwga = alpha_states(N, wgold, dwgold, dwg)
    F, _ = S.assemble_system(wga, dwga, True, False)
    rinit = norms(N, F)
    r0 = rinit + 1e-16
    it, rn, gm_its = 0, np.zeros(4), []
    converged = False
    while not converged and it < maxit:
        _, vals = S.assemble_system(wga, dwga, False, True)
        dx, hist, _, nit = S.gmres(vals, F)
        gm_its.append(nit)
        dwg -= dx
        wga, dwga = alpha_states(N, wgold, dwgold, dwg)
        F, _ = S.assemble_system(wga, dwga, True, False)
        rn = norms(N, F)
        converged = bool(np.all(rn < tol * r0))
        it += 1
    return it, rn, rinit, dwg, F, gm_its


def time_step(S, wgold, dwgold, dwg, maxit=4):
    N = S.N
    dwg = dwg.copy()
    fac_pred = (kGAMMA - 1.0) / kGAMMA
    dwg[:3 * N] *= fac_pred
    dwg[4 * N:] *= fac_pred
    it, rn, rinit, dwg, F, gm = solve_flow_system(S, wgold, dwgold, dwg, maxit)
    wgold = wgold.copy()
    for sl in (slice(0, 3 * N), slice(4 * N, 6 * N)):
        wgold[sl] += kDT * (1.0 - kGAMMA) * dwgold[sl]
    for sl in (slice(0, 3 * N), slice(4 * N, 6 * N)):
        wgold[sl] += kDT * kGAMMA * dwg[sl]
    return it, rn, rinit, wgold, dwg.copy(), dwg
