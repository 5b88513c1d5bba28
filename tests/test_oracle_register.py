"""CPU checks of the surface registration oracle (oracle/horn_oracle.c): the reference has no tests or vectors for this
path (g2o needs Eigen, the call site OpenCV), so the restatement is cross-checked against independent closed forms."""
import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation as Rot


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


@pytest.mark.parametrize("u", [
    [0.1, -0.2, 0.3, 0.5, -0.4, 0.2, 0.15],      # general branch
    [0.1, -0.2, 0.3, 0.5, -0.4, 0.2, 1e-7],      # |sigma| < eps
    [1e-7, 0, -1e-7, 0.5, -0.4, 0.2, 0.3],       # theta < eps
    [0, 0, 0, 0.5, -0.4, 0.2, 0],                # both small
])
def test_sim3_exp_is_the_matrix_exponential(oracle_mod, u):
    u = np.array(u, float)
    s = oracle_mod.sim3_exp(u)
    G = np.zeros((4, 4))
    G[:3, :3] = _skew(u[:3]) + u[6] * np.eye(3)
    G[:3, 3] = u[3:6]
    E = expm(G)
    sR = s[7] * Rot.from_quat(s[:4]).as_matrix()
    # below eps = 1e-5 the reference switches to the zeroth-order coefficients: exact only to that order
    tol = 1e-9 if (abs(u[6]) > 1e-5 and np.linalg.norm(u[:3]) > 1e-5) else 1e-6
    np.testing.assert_allclose(sR, E[:3, :3], rtol=0, atol=tol)
    np.testing.assert_allclose(s[4:7], E[:3, 3], rtol=0, atol=tol)
    assert abs(s[7] - np.exp(u[6])) < 1e-15


def test_numeric_jacobian_system_matches_the_analytic_one(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_register_scene(200, seed=3)
    sim3 = np.array([0.01, -0.02, 0.005, 1.0, 0.02, 0.0, -0.01, 1.2])
    sim3[:4] /= np.linalg.norm(sim3[:4])
    H, b, chi = oracle_mod.horn_system(sc["surface"], sc["map"], sim3, huber=0.01)
    R = Rot.from_quat(sim3[:4]).as_matrix()
    y = sim3[7] * (sc["surface"].astype(float) @ R.T) + sim3[4:7]
    e = sc["map"].astype(float) - y
    e2 = (e * e).sum(1)
    delta = float(np.float32(np.sqrt(0.01)))
    w = np.where(e2 <= delta * delta, 1.0, delta / np.sqrt(np.maximum(e2, 1e-300)))
    rho = np.where(e2 <= delta * delta, e2, 2 * np.sqrt(e2) * delta - delta * delta)
    Ha = np.zeros((7, 7)); ba = np.zeros(7)
    for i in range(y.shape[0]):
        J = -np.hstack([-_skew(y[i]), np.eye(3), y[i][:, None]])     # d(z - exp(d) y)/dd at 0
        Ha += w[i] * J.T @ J
        ba -= w[i] * J.T @ e[i]
    assert abs(chi - rho.sum()) < 1e-12 * rho.sum()
    np.testing.assert_allclose(np.tril(H), np.tril(Ha), rtol=0, atol=2e-6 * np.abs(Ha).max())   # delta 1e-9 differences: ~1e-7 noise
    np.testing.assert_allclose(b, ba, rtol=0, atol=2e-6 * np.abs(ba).max())


def test_optimize_horn_recovers_a_known_similarity(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_register_scene(500, seed=11, noise=0.0, outliers=0.0, scale=1.3)
    r = oracle_mod.optimize_horn(sc["surface"], sc["map"], [0, 0, 0, 1, 0, 0, 0, 1.2], chi=0.05 ** 2)
    assert r["ok"] and r["count"] == 500
    np.testing.assert_allclose(r["sim3"][7], 1.3, rtol=1e-6)
    np.testing.assert_allclose(Rot.from_quat(r["sim3"][:4]).as_matrix(), sc["R"], atol=1e-6)
    np.testing.assert_allclose(r["sim3"][4:7], sc["t"], atol=1e-6)
    # with noise and gross outliers the Huber estimate stays close and the inlier count drops
    sc = synth.make_register_scene(500, seed=12, noise=2e-3, outliers=0.1, scale=1.3)
    r = oracle_mod.optimize_horn(sc["surface"], sc["map"], [0, 0, 0, 1, 0, 0, 0, 1.2], chi=0.05 ** 2)
    assert abs(r["sim3"][7] - 1.3) < 0.02 and 350 < r["count"] < 500
    assert r["iters"][0] >= 3 and r["trials"].sum() >= r["iters"].sum()


def _smm_numpy(mono, stereo, u):
    """Independent restatement with numpy sorting (float32 where the reference has float)."""
    n = mono.shape[0]
    k = 0
    min_med = np.float32(10000.0)
    best = 0.0
    final_points = 0
    for i in range(n):
        r = u[k]; k += 1
        if r > 0.25:
            continue
        scale = float(np.float32(stereo[i, 2]) / np.float32(mono[i, 2]))
        draws = u[k:k + n - 1]; k += n - 1
        sel = np.ones(n, bool); sel[i] = False
        sel[np.arange(n) != i] = ~(draws > 0.25)
        d = scale * mono[sel].astype(np.float64) - stereo[sel].astype(np.float64)
        r2 = np.zeros(d.shape[0], np.float32)
        for c in range(3):
            r2 = (r2.astype(np.float64) + d[:, c] * d[:, c]).astype(np.float32)
        res = np.sort(np.sqrt(r2))
        final_points += 1
        if res.shape[0] <= 1:
            return 0.0, k, 2
        tail = res[1:]
        med = tail[tail.shape[0] // 2]
        if med < min_med:
            min_med = med; best = scale
    desv = np.float32(1.4826 * (1.0 - (5.0 / (final_points - 1.0))) * float(np.sqrt(min_med)))
    num = np.float32(0); den = np.float32(0)
    for i in range(n):
        d = best * mono[i].astype(np.float64) - stereo[i].astype(np.float64)
        r = np.float32(0)
        for c in range(3):
            r = np.float32(float(r) + d[c] * d[c])
        r = np.sqrt(r)
        if float(np.float32(r / desv)) < 2.5:
            num = np.float32(num + np.float32(stereo[i, 2] * mono[i, 2]))
            den = np.float32(den + np.float32(mono[i, 2] * mono[i, 2]))
    return float(np.float32(num / den)), k, 0


@pytest.mark.parametrize("n,seed", [(120, 1), (400, 2)])
def test_scale_min_median_equals_numpy_restatement_bit_for_bit(oracle_mod, n, seed):
    from defslam_amd import synth
    sc = synth.make_register_scene(n, seed=seed)
    o = oracle_mod.scale_min_median(sc["surface"], sc["map"], sc["u"])
    s, k, st = _smm_numpy(sc["surface"], sc["map"], sc["u"])
    assert (o["status"], o["consumed"]) == (st, k)
    assert np.float32(o["scale"]) == np.float32(s)
    assert abs(o["scale"] - sc["scale"]) < 0.05


def test_scale_min_median_edge_cases(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_register_scene(40, seed=3)
    # a candidate whose own selection is empty: the reference's early `return 0.0`
    u = np.ones(40 + 40 * 40)
    u[0] = 0.1
    o = oracle_mod.scale_min_median(sc["surface"], sc["map"], u)
    assert o["status"] == 2 and o["scale"] == 0.0
    # stream too short
    o = oracle_mod.scale_min_median(sc["surface"], sc["map"], np.full(10, 0.1))
    assert o["status"] == 1


def test_compose_inverts_the_scaled_pose(oracle_mod):
    from defslam_amd import synth
    sc = synth.make_register_scene(100, seed=8)
    q = Rot.from_matrix(sc["R"]).as_quat()
    sim3 = np.concatenate([q, sc["t"], [sc["scale"]]])
    s22, Tcw = oracle_mod.horn_compose(sim3, sc["Twc"])
    assert abs(s22 - sc["scale"]) < 1e-6 * sc["scale"]
    S = np.eye(4); S[:3, :3] = sc["scale"] * sc["R"]; S[:3, 3] = sc["t"]
    Twc_new = S @ sc["Twc"].astype(float)
    Twc_new[:3, :3] /= s22
    np.testing.assert_allclose(Tcw.astype(float) @ Twc_new, np.eye(4), atol=2e-6)
