"""CPU: the Shape-from-Normals oracle (oracle/sfn_oracle.c).  The bending matrix is pinned against the reference's own
bending_ur (oracle/_ref/libbbs_ref.so, built from /root/reference by oracle/Makefile) and against a golden vector made
from it; the least-squares solve is cross-checked against numpy."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bending_13x15.npz")


def test_bending_matches_golden_vector(oracle_mod):
    g = np.load(GOLD)
    bbs = tuple(g["bbs"][:2]) + (int(g["bbs"][2]),) + tuple(g["bbs"][3:5]) + (int(g["bbs"][5]), 1)
    B = oracle_mod.sfn_bending(bbs, float(g["lam"]))
    np.testing.assert_allclose(B, g["bending"], rtol=0, atol=4e-16 * np.abs(g["bending"]).max())


@pytest.mark.parametrize("bbs,lam", [((-0.6, 0.62, 13, -0.45, 0.5, 15, 1), 0.7), ((0.0, 1.0, 4, 0.0, 2.0, 4, 1), 1.0), ((-1.0, 3.0, 7, 2.0, 2.5, 5, 1), 1e-3)])
def test_bending_matches_reference_build(oracle_mod, bbs, lam):
    if oracle_mod.ref_bbs_lib() is None:
        pytest.skip("oracle/_ref/libbbs_ref.so not built (reference absent)")
    B = oracle_mod.sfn_bending(bbs, lam)
    R = oracle_mod.ref_bbs_bending(bbs, lam)
    np.testing.assert_allclose(B, R, rtol=0, atol=4e-16 * np.abs(R).max())
    np.testing.assert_array_equal(B, B.T)
    # bending energy vanishes on affine depth maps: constants are in the null space
    assert np.abs(B.sum(1)).max() < 1e-12 * np.abs(B).max()


def test_host_library_bending_equals_oracle(oracle_mod):
    from defslam_amd import nrsfm
    bbs = (-0.6, 0.62, 13, -0.45, 0.5, 15, 1)
    np.testing.assert_array_equal(nrsfm.bbs_bending(nrsfm.Bbs(*bbs), 0.7), oracle_mod.sfn_bending(bbs, 0.7))


def test_rows_and_least_squares(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_sfn_scene(400, seed=2)
    M = oracle_mod.sfn_rows(sc["bbs"], sc["u"], sc["v"], sc["normals"])
    n = sc["u"].shape[0]
    assert M.shape == (2 * n, 195) and ((M != 0).sum(1) <= 16).all()
    # a depth map z(u,v) whose surface has normal n satisfies n . (z_u eta + z e_u) = 0: plane z = 1/(a u + b v + c)
    ok, raw, ctrl, pts = oracle_mod.sfn_estimate(sc["bbs"], sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
    assert ok
    B = oracle_mod.sfn_bending(sc["bbs"], 1e-3)
    A = np.vstack([M, B, np.ones((1, 195))])
    b = np.zeros(A.shape[0])
    b[-1] = 195 * sc["mean_depth"]
    ref = np.linalg.lstsq(A, b, rcond=None)[0]
    np.testing.assert_allclose(raw, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    med = np.sort(raw.astype(np.float32))[195 // 2]
    np.testing.assert_allclose(ctrl, raw * (np.float32(1) / med), rtol=1e-15)
    # the reconstructed depth is the true one up to the global scale fixed by the median
    d = pts[:, 2].astype(float)
    s = np.median(d / sc["depth_true"])
    assert np.abs(d / (s * sc["depth_true"]) - 1).max() < 0.05
    np.testing.assert_allclose(pts[:, 0], (sc["u_all"] * d).astype(np.float32), rtol=2e-7)


def test_no_key_points_fails_like_the_reference(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_sfn_scene(50, seed=3)
    ok, *_ = oracle_mod.sfn_estimate(sc["bbs"], sc["u"], sc["v"], sc["normals"], 1e-3, 1.0, np.zeros(0), np.zeros(0))
    assert not ok
