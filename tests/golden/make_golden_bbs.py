"""Generates tests/golden/bbs_*.npz by calling the reference's own Thirdparty/BBS/bbs.cc, compiled unmodified into
oracle/_ref/libbbs_ref.so by oracle/Makefile (run in the build container: python tests/golden/make_golden_bbs.py).
The fixtures hold inputs and the reference's outputs only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, bbs, n, seed):
    rng = np.random.default_rng(seed)
    umin, umax, nptsu, vmin, vmax, nptsv, valdim = bbs
    ctrl = rng.normal(size=(nptsu * nptsv, valdim))
    u = rng.uniform(umin, umax, n)
    v = rng.uniform(vmin, vmax, n)
    # boundary sites: x == xmax takes the special branch of normalize_with_inter (bbs.cc:77-79)
    u[0], v[0] = umax, vmax
    u[1], v[1] = umin, vmin
    u[2], v[2] = umax, vmin
    knots_u = umin + (umax - umin) * np.arange(nptsu - 2) / (nptsu - 3)
    u[3:3 + min(5, len(knots_u))] = knots_u[:5]      # exactly on knots
    out = dict(bbs=np.asarray(bbs, float), ctrl=ctrl, u=u, v=v)
    for du, dv in [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2)]:
        out[f"val_{du}{dv}"] = oracle.ref_bbs_eval(bbs, ctrl, u, v, du, dv)
        A, ret = oracle.ref_bbs_coloc_dense(bbs, u, v, du, dv)
        assert ret == 0
        out[f"coloc_{du}{dv}"] = A.astype(np.float64)
    for order in range(3):
        out[f"basis_{order}"] = np.stack([oracle.ref_bbs_basis(order, t) for t in np.linspace(0, 1, 11)])
    np.savez_compressed(os.path.join(HERE, f"bbs_{name}.npz"), **out)
    print(name, "sites", n)


if __name__ == "__main__":
    assert oracle.ref_bbs_lib() is not None, "build oracle/_ref first (make -C oracle ref)"
    case("13x15", (-0.85, 0.9, 13, -0.7, 0.75, 15, 2), 60, 0)     # the reference's warp grid (bbs_MAC.h: 13 x 15), valdim 2
    case("7x5_v1", (0.0, 1.0, 7, -2.0, 3.0, 5, 1), 40, 1)         # odd grid, scalar spline (depth surface)
    case("4x4_v3", (-1.0, 1.0, 4, -1.0, 1.0, 4, 3), 30, 2)        # single knot interval
