"""CPU tests: the C oracle against the committed golden vectors, the NumPy restatement and known answers."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import oracle_args

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sft_*.npz")) if "sft_C5_" not in os.path.basename(p))


def _load(path, oracle):
    g = np.load(path)
    tc = oracle.template_build(g["xyz0"], g["facets"])
    args = (tc, g["Tcw"], g["K"], int(g["n_frame"]), g["obs_nodes"], g["obs_bary"], g["obs_uv"], g["obs_invsig2"], g["xyz"]) + tuple(g["regs"])
    return g, tc, args


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("mode", [0, 1])
def test_c_oracle_matches_golden(oracle_mod, path, mode):
    g, tc, args = _load(path, oracle_mod)
    r = oracle_mod.sft_solve(*args, layers=int(g["layers"]), ldlt_mode=mode)
    assert r.iters == g["out_trace"].shape[0]
    assert r.ret == int(g["out_inliers"])
    # per-iteration (chi2, lambda, trials, accepted) trace
    np.testing.assert_array_equal(r.trace[:, 2], g["out_trace"][:, 2])
    np.testing.assert_array_equal(r.trace[:, 6], g["out_trace"][:, 6])
    np.testing.assert_allclose(r.trace[:, [0, 1, 3, 4]], g["out_trace"][:, [0, 1, 3, 4]], rtol=1e-9)
    np.testing.assert_allclose(r.xyz, g["out_xyz"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r.pose7, g["out_pose7"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(r.outlier.astype(bool), g["out_outlier"])
    np.testing.assert_allclose(r.rep_error, float(g["out_rep_error"]), rtol=1e-10)


def test_c_oracle_matches_numpy_restatement_live(oracle_mod):
    from defslam_amd import synth
    from oracle import sft_oracle_np as onp
    tmpl = synth.make_grid_template(7, 9)
    fr = synth.make_frame(tmpl, 150, 11)
    tc, args = oracle_args(oracle_mod, tmpl, fr)
    r = oracle_mod.sft_solve(*args)
    rn = onp.solve(*args)
    assert r.iters == rn["iters"]
    np.testing.assert_allclose(r.xyz, rn["xyz"], atol=1e-11)
    np.testing.assert_allclose(r.pose7, rn["pose7"], atol=1e-11)


def test_system_matches_numpy_jacobian(oracle_mod):
    """H = J^T W J and b = -J^T W e built two different ways."""
    from defslam_amd import synth
    from oracle import sft_oracle_np as onp
    tmpl = synth.make_grid_template(6, 6)
    fr = synth.make_frame(tmpl, 80, 2)
    tc, args = oracle_args(oracle_mod, tmpl, fr)
    H, b, chi = oracle_mod.sft_system(*args)
    g = onp.Graph(*args)
    res = g.residuals()
    Hn, bn = g.system(res)
    np.testing.assert_allclose(H, Hn, rtol=1e-11, atol=1e-12 * np.abs(Hn).max())
    np.testing.assert_allclose(b, bn, rtol=1e-11, atol=1e-12 * np.abs(bn).max())
    assert chi == pytest.approx(g.robust_chi2(res), rel=1e-13)


def test_huber_known_answers(oracle_mod):
    L = oracle_mod.lib()
    rho = (C.c_double * 3)()
    d = float(np.float32(np.sqrt(5.991)))
    for e2 in [0.0, 1.0, d * d, d * d * (1 + 1e-12), 10.0, 1e6]:
        L.sft_oracle_huber(C.c_double(d), C.c_double(e2), rho)
        if e2 <= d * d:
            assert (rho[0], rho[1], rho[2]) == (e2, 1.0, 0.0)
        else:
            assert rho[0] == pytest.approx(2 * np.sqrt(e2) * d - d * d, rel=1e-15)
            assert rho[1] == pytest.approx(d / np.sqrt(e2), rel=1e-15)
    # continuity at the threshold
    L.sft_oracle_huber(C.c_double(d), C.c_double(d * d * (1 + 1e-14)), rho)
    assert rho[0] == pytest.approx(d * d, rel=1e-12)


def test_se3_exp_against_scipy(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(0)
    out = (C.c_double * 7)()
    for scale in [1e-9, 1e-6, 1e-3, 0.3, 2.5]:
        u = rng.normal(size=6) * scale
        L.sft_oracle_se3_exp((C.c_double * 6)(*u), out)
        p = np.array(out[:])
        if np.linalg.norm(u[:3]) >= 1e-5:
            q = Rotation.from_rotvec(u[:3]).as_quat()
            q = q if q[3] >= 0 else -q
            np.testing.assert_allclose(p[3:], q, atol=1e-13)
            th = np.linalg.norm(u[:3])
            K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
            V = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (th - np.sin(th)) / th**3 * K @ K
            np.testing.assert_allclose(p[:3], V @ u[3:], atol=1e-13)
        else:  # small-angle branch: R = I + Om + Om^2, V = R (se3quat.h:236-242)
            assert abs(np.linalg.norm(p[3:]) - 1) < 1e-15
            np.testing.assert_allclose(p[:3], u[3:], atol=2 * scale**2 + 1e-18)


def test_pose_from_float32_matrix(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(3)
    out = (C.c_double * 7)()
    for _ in range(20):
        R = Rotation.from_rotvec(rng.normal(size=3) * 2.0)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R.as_matrix()
        T[:3, 3] = rng.normal(size=3)
        L.sft_oracle_pose_from_f32(T.ctypes.data_as(C.POINTER(C.c_float)), out)
        p = np.array(out[:])
        q = R.as_quat()
        q = q if q[3] >= 0 else -q
        np.testing.assert_allclose(p[3:], q, atol=5e-7)  # float32 input
        np.testing.assert_array_equal(p[:3], T[:3, 3].astype(np.float64))


@pytest.mark.parametrize("n", [1, 2, 7, 64, 193])
def test_ldlt_modes(oracle_mod, n):
    L = oracle_mod.lib()
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    S = A @ A.T + n * np.eye(n)
    b = rng.normal(size=n)
    Sf = np.asfortranarray(S)
    for mode in (0, 1):
        x = np.zeros(n)
        ok = L.sft_oracle_ldlt_solve(mode, n, Sf.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                     x.ctypes.data_as(C.POINTER(C.c_double)))
        assert ok == 1
        np.testing.assert_allclose(x, np.linalg.solve(S, b), rtol=1e-9)
    # an indefinite matrix is reported as "not positive" (linear_solver_dense.h:107-112)
    if n >= 2:
        Sn = np.asfortranarray(S - 3 * np.abs(np.linalg.eigvalsh(S)).max() * np.outer(np.eye(n)[0], np.eye(n)[0]))
        x = np.zeros(n)
        assert L.sft_oracle_ldlt_solve(0, n, Sn.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                       x.ctypes.data_as(C.POINTER(C.c_double))) == 0


def test_camera_jacobian_is_the_true_derivative_node_jacobian_is_not(oracle_mod):
    """EdgeNodesCamera: J_cam equals the numeric derivative (g2o's central differences, delta=1e-9), while
    J_node is the reference's per-node-depth approximation (sft_types.h:176-205) and is pinned by formula."""
    from defslam_amd import synth
    tmpl = synth.make_grid_template(4, 4)
    fr = synth.make_frame(tmpl, 1, 1, outlier_frac=0.0)
    tc, args = oracle_args(oracle_mod, tmpl, fr, regs=(0.0, 0.0, 0.0))
    H, b, chi = oracle_mod.sft_system(*args)
    # with a single observation and no regulariser: b_cam = -J_cam^T w e  => recover J_cam^T e and compare to finite differences of chi2
    w = fr.obs_invsig2[0] / fr.n_frame

    def chi_at(delta6):
        from oracle import sft_oracle_np as onp
        g = onp.Graph(*args)
        g.apply(np.concatenate([delta6, np.zeros(g.D - 6)]))
        return (g.chi2_parts(g.residuals())[0]).sum()

    grad = np.zeros(6)
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        grad[k] = (chi_at(d) - chi_at(-d)) / 2e-6
    # d chi / d delta = 2 J^T w e = -2 b_cam
    np.testing.assert_allclose(-2 * b[:6], grad, rtol=2e-5, atol=1e-9)
    assert w > 0


def test_curvature_and_stretch_gradients_match_finite_differences(oracle_mod):
    from defslam_amd import synth
    from oracle import sft_oracle_np as onp
    tmpl = synth.make_grid_template(5, 5)
    fr = synth.make_frame(tmpl, 60, 4)
    rng = np.random.default_rng(0)
    fr.xyz = fr.xyz + rng.normal(scale=0.003, size=fr.xyz.shape)
    tc, args = oracle_args(oracle_mod, tmpl, fr)
    H, b, chi = oracle_mod.sft_system(*args)

    def reg_cost(dx):
        g = onp.Graph(*args)
        g.xyz[g.opt] += dx.reshape(-1, 3)
        parts = g.chi2_parts(g.residuals())
        return parts[1].sum() + parts[2].sum() + parts[3].sum()

    g0 = onp.Graph(*args)
    nact = int(g0.opt.sum())
    # gradient of the regularisers wrt node coordinates == -2 * (b without the observation part)
    args_noobs = list(args)
    Hn, bn_full = g0.system(g0.residuals())
    # isolate regulariser part of b by zeroing observation weights
    g1 = onp.Graph(*args)
    g1.w_obs = g1.w_obs * 0
    _, b_reg = g1.system(g1.residuals())
    for k in rng.choice(3 * nact, size=12, replace=False):
        d = np.zeros(3 * nact)
        d[k] = 1e-7
        fd = (reg_cost(d) - reg_cost(-d)) / 2e-7
        assert -2 * b_reg[6 + k] == pytest.approx(fd, rel=1e-4, abs=1e-7)


def test_c5_fixture_inputs_are_still_what_synth_generates():
    """tests/golden/sft_C5_p*.npz hold oracle OUTPUTS for problems whose inputs are regenerated from seeds: a change of the
    generator must fail here (CPU), not silently compare different problems on the GPU box."""
    import sys
    here = os.path.join(os.path.dirname(__file__), "golden")
    sys.path.insert(0, here)
    from make_golden_c5 import input_digest
    from defslam_amd import synth
    paths = sorted(glob.glob(os.path.join(here, "sft_C5_p*.npz")))
    assert paths, "full-size C5 fixtures are missing (tests/golden/make_golden_c5.py)"
    for p in paths:
        g = np.load(p)
        tmpl, fr = synth.make_problem("C5", int(g["problem_id"]))
        assert input_digest(tmpl, fr) == str(g["input_sha256"])
        assert int(g["out_dims"][0]) == 6006 and g["out_xyz"].shape == (2000, 3) and g["out_chi2_obs"].shape == (4000,)
