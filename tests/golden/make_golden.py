"""Generates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so).

The reference ships no golden vectors and cannot be run here, so
these fixtures are REGRESSION fixtures of the build's own restatement -- they pin
the oracle against accidental change and give the GPU box data to compare with;
they do not pin the oracle to the reference ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dedflow_amd.meshgen import kuhn_cube, synthetic_fields  # noqa: E402
from oracle import orc  # noqa: E402


def full_case(M, jitter):
    m = kuhn_cube(M, jitter=jitter)
    S = orc.System(m)
    wg, dwg = synthetic_fields(m)
    F, vals = S.assemble_system(wg, dwg, True, True)
    d33, d1 = S.pc_setup(vals)
    x, hist, r0, it = S.gmres(vals, F)
    return m, S, dict(M=M, jitter=jitter, color=S.color, num_color=S.num_color, num_ties=S.num_ties,
                      batch_offset=S.batch_offset, batch_ind=S.batch_ind, row_ptr=S.rp11, col_ind=S.ci11,
                      prio=orc.priorities(S.T), F=F, A00=vals[0], A01=vals[1], A10=vals[2], A11=vals[3],
                      dinv33=d33, dinv1=d1, gmres_x=x, gmres_hist=hist, gmres_r0=r0, gmres_it=it)


def main():
    _, _, small = full_case(4, 0.2)
    np.savez_compressed(os.path.join(HERE, "cube_M4.npz"), **small)
    m, S, big = full_case(12, 0.2)
    # 10k-tet case: integer structure in full, values as checksums + strided samples
    keep = {k: big[k] for k in ("M", "jitter", "color", "num_color", "num_ties", "batch_offset", "batch_ind", "row_ptr",
                                "col_ind", "gmres_hist", "gmres_r0", "gmres_it")}
    keep["F"] = big["F"]
    for k in ("A00", "A01", "A10", "A11", "dinv33", "gmres_x"):
        v = big[k]
        keep[k + "_sum"] = np.array([v.sum(), np.abs(v).sum(), np.dot(v, v)])
        keep[k + "_sample"] = v[::37].copy()
    keep["xorwow_head"] = orc.xorwow_legacy(8192 + 8)[[0, 1, 2, 3, 4095, 4096, 4097, 8192, 8199]]
    np.savez_compressed(os.path.join(HERE, "cube_M12.npz"), **keep)
    for f in ("cube_M4.npz", "cube_M12.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
