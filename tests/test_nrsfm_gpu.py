"""GPU parity tests of the mapping-side kernels, through the C ABI, against the CPU oracles and the golden vectors
produced by the reference's own bbs.cc."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BBS_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bbs_*.npz")))
ORDERS = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2)]


@pytest.mark.parametrize("path", BBS_GOLDEN, ids=[os.path.basename(p) for p in BBS_GOLDEN])
def test_bbs_eval_and_coloc_match_reference_golden(gpu_ctx, path):
    from defslam_amd import nrsfm
    g = np.load(path)
    b = g["bbs"]
    bbs = nrsfm.Bbs(float(b[0]), float(b[1]), int(b[2]), float(b[3]), float(b[4]), int(b[5]), int(b[6]))
    for du, dv in ORDERS:
        val, out = nrsfm.bbs_eval(gpu_ctx, bbs, g["ctrl"], g["u"], g["v"], du, dv)
        assert not out.any()
        # same expression order, no FMA contraction: equal to the reference's bbs.cc up to the last bit of pow()
        np.testing.assert_allclose(val, g[f"val_{du}{dv}"], rtol=1e-15, atol=0)
        cols, w, n_out = nrsfm.bbs_coloc(gpu_ctx, bbs, g["u"], g["v"], du, dv)
        assert n_out == 0
        A = np.zeros_like(g[f"coloc_{du}{dv}"])
        np.add.at(A, (np.repeat(np.arange(cols.shape[0]), 16), cols.ravel()), w.ravel())
        ref = g[f"coloc_{du}{dv}"]
        np.testing.assert_array_equal(A != 0, ref != 0)           # bit-exact tap indexing
        np.testing.assert_allclose(A, ref, rtol=1e-15, atol=0)


def test_bbs_large_and_edge_cases(gpu_ctx, oracle_mod):
    from defslam_amd import nrsfm
    rng = np.random.default_rng(3)
    bbs_t = (-0.85, 0.9, 13, -0.7, 0.75, 15, 2)
    bbs = nrsfm.Bbs(*bbs_t)
    ctrl = rng.normal(size=(195, 2))
    n = 200_003                                               # ragged size, > one grid sweep
    u, v = rng.uniform(-0.85, 0.9, n), rng.uniform(-0.7, 0.75, n)
    u[:4] = [-0.85, 0.9, 0.9, -0.85]
    v[:4] = [-0.7, 0.75, -0.7, 0.75]
    for du, dv in [(0, 0), (1, 1), (0, 2)]:
        val, out = nrsfm.bbs_eval(gpu_ctx, bbs, ctrl, u, v, du, dv)
        ref, _ = oracle_mod.bbs_eval(bbs_t, ctrl, u, v, du, dv)
        assert not out.any()
        np.testing.assert_allclose(val, ref, rtol=1e-15, atol=0)
    # outside sites are flagged, empty input is fine
    val, out = nrsfm.bbs_eval(gpu_ctx, bbs, ctrl, np.array([-1.0, 0.0, 2.0]), np.array([0.0, 0.0, 0.0]))
    assert out.tolist() == [True, False, True] and (val[[0, 2]] == 0).all()
    _, _, n_out = nrsfm.bbs_coloc(gpu_ctx, bbs, np.array([-1.0, 0.0, 2.0]), np.array([0.0, 0.0, 0.0]))
    assert n_out == 2
    val, out = nrsfm.bbs_eval(gpu_ctx, bbs, ctrl, np.zeros(0), np.zeros(0))
    assert val.shape == (0, 2)
    # a control grid too large for LDS takes the global-memory path
    big = nrsfm.Bbs(0.0, 1.0, 120, 0.0, 1.0, 100, 1)
    cb = rng.normal(size=(12000, 1))
    ub, vb = rng.uniform(0, 1, 1000), rng.uniform(0, 1, 1000)
    vg, _ = nrsfm.bbs_eval(gpu_ctx, big, cb, ub, vb, 1, 0)
    vr, _ = oracle_mod.bbs_eval((0.0, 1.0, 120, 0.0, 1.0, 100, 1), cb, ub, vb, 1, 0)
    np.testing.assert_allclose(vg, vr, rtol=1e-15, atol=0)


@pytest.mark.parametrize("n_points,n_views,seed", [(300, 4, 7), (5000, 6, 1), (1, 1, 2)])
def test_normals_match_oracle(gpu_ctx, oracle_mod, n_points, n_views, seed):
    from defslam_amd import nrsfm, synth
    sc = synth.make_normals_scene(n_points, n_views, seed)
    keys = ["rec_ptr", "recs", "rec_is_ref", "rec_first_normal", "rec_has_first_normal", "x0", "has_x0", "ref_uv"]
    o = oracle_mod.normals(*[sc[k] for k in keys])
    g = nrsfm.ObtainK1K2(gpu_ctx, *[sc[k] for k in keys])
    np.testing.assert_array_equal(g.status, o["status"])            # which points are solved / skipped / rejected
    np.testing.assert_array_equal(g.rec_written, o["rec_written"])  # which keyframe normals are written
    np.testing.assert_array_equal(g.iters, o["iters"])              # same accept/reject sequence
    solved = o["status"] == 0
    np.testing.assert_allclose(g.k1k2[solved], o["k1k2"][solved], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g.cov[solved], o["cov"][solved], rtol=1e-7)
    np.testing.assert_allclose(g.normal_ref[solved], o["normal_ref"][solved], rtol=0, atol=1e-6)
    wr = o["rec_written"].astype(bool)
    np.testing.assert_allclose(g.normal_rec[wr], o["normal_rec"][wr], rtol=0, atol=1e-6)


def test_normals_points_with_more_than_a_hundred_reference_records(gpu_ctx, oracle_mod):
    """NormalEstimator.cc:77-118 adds one residual block per record of the reference keyframe, however many: neither side may cap them."""
    from defslam_amd import nrsfm, synth
    sc = synth.make_normals_scene(12, 170, 3, nonref_frac=0.3, min_views=150)
    nref = np.add.reduceat(sc["rec_is_ref"].astype(np.int64), sc["rec_ptr"][:-1])
    assert nref.min() >= 100
    keys = ["rec_ptr", "recs", "rec_is_ref", "rec_first_normal", "rec_has_first_normal", "x0", "has_x0", "ref_uv"]
    o = oracle_mod.normals(*[sc[k] for k in keys])
    g = nrsfm.ObtainK1K2(gpu_ctx, *[sc[k] for k in keys])
    np.testing.assert_array_equal(g.status, o["status"])
    np.testing.assert_array_equal(g.rec_written, o["rec_written"])
    np.testing.assert_array_equal(g.iters, o["iters"])
    assert (o["status"] == 0).all()
    np.testing.assert_allclose(g.k1k2, o["k1k2"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(g.cov, o["cov"], rtol=1e-7)
    np.testing.assert_allclose(g.normal_ref, o["normal_ref"], rtol=0, atol=1e-6)
    wr = o["rec_written"].astype(bool)
    np.testing.assert_allclose(g.normal_rec[wr], o["normal_rec"][wr], rtol=0, atol=1e-6)


def test_normals_degenerate_inputs(gpu_ctx, oracle_mod):
    from defslam_amd import nrsfm
    # point 0: no records at all; point 1: rank-deficient (identity warp); point 2: only non-reference records
    recs = np.zeros((3, 18), np.float32)
    recs[:, 4] = 1.0
    recs[:, 7] = 1.0
    recs[:, 8] = 1.0
    recs[:, 11] = 1.0
    rec_ptr = np.array([0, 0, 1, 3], np.int32)
    is_ref = np.array([1, 0, 0], np.uint8)
    args = (rec_ptr, recs, is_ref, np.full((3, 2), 0.25, np.float32), np.array([0, 1, 0], np.uint8), np.zeros((3, 2), np.float32), np.zeros(3, np.uint8),
            np.zeros((3, 2), np.float32))
    o = oracle_mod.normals(*args)
    g = nrsfm.ObtainK1K2(gpu_ctx, *args)
    assert g.status.tolist() == o["status"].tolist() == [1, 2, 1]
    assert g.rec_written.tolist() == o["rec_written"].tolist() == [0, 1, 0]
    np.testing.assert_array_equal(g.normal_rec[1], o["normal_rec"][1])


@pytest.mark.parametrize("P,seed,lam", [(300, 3, 0.1), (57, 8, 1.0), (1200, 1, 0.3)])
def test_schwarp_residuals_and_jacobian_match_oracle(gpu_ctx, oracle_mod, P, seed, lam):
    """Warps::Warp::Evaluate + Warps::Schwarzian::Evaluate (rows B1a, B1b), including the reference's overwritten warp y-rows."""
    from defslam_amd import nrsfm, synth
    pr = synth.make_warp_problem(P, seed)
    x = pr["x0"] + np.random.default_rng(seed).normal(scale=5e-3, size=pr["x0"].shape)
    ro, Jo = oracle_mod.schwarp_eval(pr["bbs"], pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, x)
    rg, Jg = nrsfm.schwarp_eval(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, x)
    np.testing.assert_array_equal(Jg != 0, Jo != 0)                       # sparsity: bit-exact tap indexing
    np.testing.assert_allclose(rg, ro, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Jg, Jo, rtol=1e-13, atol=1e-15)
    N = pr["bbs"][2] * pr["bbs"][5]
    np.testing.assert_array_equal(Jg[:P], Jg[P:2 * P])                     # quirk: y rows are copies of the x rows
    assert (Jg[:2 * P, N:] == 0).all()
    # the Schwarzian block is the true derivative (central differences)
    for k in [3, N + 7, 2 * N - 1]:
        d = np.zeros_like(x)
        d[k] = 1e-6
        fd = (nrsfm.schwarp_eval(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, x + d, False)[0] -
              nrsfm.schwarp_eval(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, x - d, False)[0]) / 2e-6
        np.testing.assert_allclose(fd[2 * P:], Jg[2 * P:, k], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("P,seed,lam,outl,iters", [(300, 3, 0.1, 0.0, 3), (300, 3, 1.0, 0.05, 3), (150, 5, 1.0, 0.0, 12), (40, 9, 0.5, 0.1, 3)])
def test_schwarp_fit_matches_oracle(gpu_ctx, oracle_mod, P, seed, lam, outl, iters):
    """SchwarpDatabase::calculateSchwarps (row B1c): same accept/reject sequence, same control points, same DiffProp records."""
    from defslam_amd import nrsfm, synth
    pr = synth.make_warp_problem(P, seed, outliers=outl)
    xo, do, dro, io, co = oracle_mod.schwarp_fit(pr["bbs"], pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, pr["fx"], pr["fy"], pr["x0"], iters)
    xg, dg, drg, ig, cg = nrsfm.calculateSchwarps(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, pr["fx"], pr["fy"],
                                                  pr["x0"], iters)
    np.testing.assert_array_equal(ig, io)                                   # iterations and accepted steps
    np.testing.assert_allclose(cg, co, rtol=1e-10)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-9 * max(1.0, np.abs(xo).max()))
    np.testing.assert_array_equal(drg, dro)                                 # which matches are dropped (> 10 px)
    np.testing.assert_allclose(dg, do, rtol=2e-6, atol=1e-6)                # float32 DiffProp fields
    # J21 fields exactly as SchwarpDatabase.cc:322-329 assigns them (note: b and c trade places w.r.t. the matrix inverse)
    a, b, c, d = dg[:, 4], dg[:, 5], dg[:, 6], dg[:, 7]
    det = a * d - c * b
    good = np.abs(det) > 0.2
    np.testing.assert_allclose(dg[good, 8], (d / det)[good], rtol=1e-5)
    np.testing.assert_allclose(dg[good, 9], (-c / det)[good], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dg[good, 10], (-b / det)[good], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dg[good, 11], (a / det)[good], rtol=1e-5)


def test_schwarp_fit_batch_equals_single_fits_and_the_oracle(gpu_ctx, oracle_mod):
    """dsh_schwarp_fit_batch (one warp per anchor keyframe of a new keyframe, SchwarpDatabase.cc:50-128): fits of different sizes,
    regularisation, iteration limits and outlier rates advance together with the trust-region control on the device; every
    result is bit-identical to the single call and follows the oracle's accept / reject sequence."""
    from defslam_amd import nrsfm, synth
    cases = [(300, 3, 0.1, 0.0, 3), (150, 5, 1.0, 0.0, 12), (40, 9, 0.5, 0.1, 3), (500, 21, 1e-2, 0.05, 3), (80, 22, 1e-2, 0.0, 0), (300, 3, 1.0, 0.05, 3),
             (200, 31, 5.0, 0.0, 8), (120, 32, 0.1, 0.3, 6), (60, 23, 10.0, 0.0, 60)]
    probs, prs = [], []
    for P, seed, lam, outl, iters in cases:
        pr = synth.make_warp_problem(P, seed, outliers=outl)
        if seed >= 31:   # a poor start far from the fitted warp: the trust region has steps to reject while the other fits of the batch go on
            pr["x0"] = pr["x0"] + np.random.default_rng(seed).normal(scale=0.05, size=pr["x0"].size)
        prs.append(pr)
        probs.append(dict(bbs=nrsfm.Bbs(*pr["bbs"]), kp1=pr["kp1"], kp2=pr["kp2"], invsig=pr["invsig"], fx_slot=pr["fy"], fy_slot=pr["fx"], lam=lam, fx=pr["fx"], fy=pr["fy"],
                          x0=pr["x0"], max_iters=iters))
    res = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs)
    for (P, seed, lam, outl, iters), pr, q, (xb, db, drb, ib, cb) in zip(cases, prs, probs, res):
        xs, ds, drs, is_, cs = nrsfm.calculateSchwarps(gpu_ctx, q["bbs"], pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, pr["fx"], pr["fy"], pr["x0"], iters)
        np.testing.assert_array_equal(xb, xs)
        np.testing.assert_array_equal(db.view(np.uint32), ds.view(np.uint32))
        np.testing.assert_array_equal(drb, drs)
        np.testing.assert_array_equal(ib, is_)
        np.testing.assert_array_equal(cb, cs)
        xo, do, dro, io, co = oracle_mod.schwarp_fit(pr["bbs"], pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, pr["fx"], pr["fy"], pr["x0"], iters)
        np.testing.assert_array_equal(ib, io)
        np.testing.assert_allclose(cb, co, rtol=1e-10)
        np.testing.assert_allclose(xb, xo, rtol=0, atol=1e-9 * max(1.0, np.abs(xo).max()))
        np.testing.assert_array_equal(drb, dro)
    # the batch is heterogeneous where the device-side trust-region control matters: fits that stop before their iteration limit next to
    # fits that use all of it, fits whose every step is rejected next to fits that accept some (each already equal to the ORACLE above,
    # which runs every fit on its own -- the independent reference of the control, not the single-fit call, which is a batch of one)
    infos = np.array([r[3] for r in res])
    limits = np.array([c[4] for c in cases])
    assert (infos[:, 0] < limits).any() and (infos[:, 0] == limits)[limits > 0].any()
    assert (infos[:, 1] < infos[:, 0]).any() and (infos[:, 1] > 0).any() and (infos[:, 1] == 0)[limits > 0].any()


def test_schwarp_fit_batch_with_the_initialisation_inside(gpu_ctx):
    """dsh_schwarp_problem.init_lambda: Warp::initialize runs as the first stage of the batch, on the device, for the fits that ask
    for it (the others keep their start value): the same control points as dsh_warp_initialize followed by the same fit, bit for
    bit, with mixed sizes and a fit without initialisation in the middle of the batch."""
    from defslam_amd import nrsfm, synth
    cases = [(300, 3, 0.1, 1e-2, True), (150, 5, 1.0, 1e-2, True), (500, 21, 1e-2, 0.0, False), (80, 22, 1e-2, 1.0, True), (25, 4, 0.5, 1e-4, True)]
    probs, ref = [], []
    for P, seed, lam, ilam, init in cases:
        pr = synth.make_warp_problem(P, seed)
        b = nrsfm.Bbs(*pr["bbs"])
        q = dict(bbs=b, kp1=pr["kp1"], kp2=pr["kp2"], invsig=pr["invsig"], fx_slot=pr["fy"], fy_slot=pr["fx"], lam=lam, fx=pr["fx"], fy=pr["fy"], max_iters=3)
        if init:
            ok, x0 = nrsfm.WarpInitialize(gpu_ctx, b, pr["kp1"], pr["kp2"], ilam)
            q["init_lam"] = ilam
        else:
            ok, x0 = True, pr["x0"]
            q["x0"] = x0
        probs.append(q)
        ref.append((ok, nrsfm.calculateSchwarps(gpu_ctx, b, pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], lam, pr["fx"], pr["fy"], x0, 3)))
    res = nrsfm.calculateSchwarpsBatch(gpu_ctx, probs)
    for (P, seed, lam, ilam, init), r, (ok, (xs, ds, drs, is_, cs)) in zip(cases, res, ref):
        if init:
            assert r[5] == ok
        np.testing.assert_array_equal(r[0], xs)
        np.testing.assert_array_equal(r[1].view(np.uint32), ds.view(np.uint32))
        np.testing.assert_array_equal(r[2], drs)
        np.testing.assert_array_equal(r[3], is_)
        np.testing.assert_array_equal(r[4], cs)


@pytest.mark.parametrize("n,seed,lam", [(600, 4, 1e-3), (150, 7, 0.05), (2500, 9, 1e-4)])
def test_shape_from_normals_matches_oracle(gpu_ctx, oracle_mod, n, seed, lam):
    """ShapeFromNormals::estimate: the device solves the stacked least squares by corrected semi-normal
    equations (MFMA A^T A, tile Cholesky, two refinement steps); the oracle by Householder QR like the reference."""
    from defslam_amd import nrsfm, synth
    sc = synth.make_sfn_scene(n, seed=seed)
    oko, rawo, ctrlo, ptso = oracle_mod.sfn_estimate(sc["bbs"], sc["u"], sc["v"], sc["normals"], lam, sc["mean_depth"], sc["u_all"], sc["v_all"])
    okg, rawg, ctrlg, ptsg = nrsfm.ShapeFromNormals(gpu_ctx, nrsfm.Bbs(*sc["bbs"]), sc["u"], sc["v"], sc["normals"], lam, sc["mean_depth"], sc["u_all"], sc["v_all"])
    assert oko and okg
    scale = np.abs(rawo).max()
    np.testing.assert_allclose(rawg, rawo, rtol=0, atol=1e-8 * scale)       # north-star tolerance is 1e-4; QR vs refined normal equations agree far better
    np.testing.assert_allclose(ctrlg, ctrlo, rtol=0, atol=1e-6 * np.abs(ctrlo).max())   # float32 median in the scale factor
    np.testing.assert_allclose(ptsg, ptso, rtol=2e-6, atol=1e-6)
    assert ptsg.dtype == np.float32


@pytest.mark.parametrize("n,lam", [(12, 1e-3), (5, 1e-2), (40, 1e-5)])
def test_shape_from_normals_few_clustered_normals_against_numpy_lstsq(gpu_ctx, oracle_mod, n, lam):
    """The reference solves the stacked system [M; Bend; 1] with Eigen's HouseholderQR (ShapeFromNormals.cc:95); the device forms
    A^T A (which squares the condition number of the already ill-conditioned bending block) and repairs that with two
    refinement steps on residuals taken from A itself.  Worst case for that deviation: a handful of normals clustered in one
    corner of the domain -- almost all of the surface is determined by the bending energy alone (its null space, affine depth
    maps, is pinned only by those few rows and the mean-depth row).  Compared against an SVD least-squares solve of the same
    stacked system (numpy.linalg.lstsq), independent of both the oracle's QR and the device's normal equations."""
    from defslam_amd import nrsfm, synth
    sc = synth.make_sfn_scene(400, seed=11)
    bbs = sc["bbs"]
    # keep only the n sites closest to one corner of the definition domain
    d = (sc["u"] - bbs[0]) ** 2 + (sc["v"] - bbs[3]) ** 2
    keep = np.argsort(d)[:n]
    u, v, nr = sc["u"][keep], sc["v"][keep], sc["normals"][keep]
    N = bbs[2] * bbs[5]
    M = oracle_mod.sfn_rows(bbs, u, v, nr)
    A = np.vstack([M, oracle_mod.sfn_bending(bbs, lam), np.ones((1, N))])
    b = np.zeros(A.shape[0])
    b[-1] = N * sc["mean_depth"]
    ref, _, rank, sv = np.linalg.lstsq(A, b, rcond=None)
    assert rank == N and sv[0] / sv[-1] > 1e3          # full rank, but badly conditioned: that is the point of the case
    ok, raw, ctrl, pts = nrsfm.ShapeFromNormals(gpu_ctx, nrsfm.Bbs(*bbs), u, v, nr, lam, sc["mean_depth"], sc["u_all"], sc["v_all"])
    assert ok
    # north-star tolerance 1e-4 relative; the refined semi-normal equations stay orders of magnitude inside it even here
    np.testing.assert_allclose(raw, ref, rtol=0, atol=1e-6 * np.abs(ref).max())
    oko, rawo, *_ = oracle_mod.sfn_estimate(bbs, u, v, nr, lam, sc["mean_depth"], sc["u_all"], sc["v_all"])
    assert oko
    np.testing.assert_allclose(rawo, ref, rtol=0, atol=1e-6 * np.abs(ref).max())


def test_shape_from_normals_edge_cases(gpu_ctx, oracle_mod):
    from defslam_amd import nrsfm, synth
    sc = synth.make_sfn_scene(80, seed=5)
    b = nrsfm.Bbs(*sc["bbs"])
    # no key points to place: estimate() returns false
    ok, *_ = nrsfm.ShapeFromNormals(gpu_ctx, b, sc["u"], sc["v"], sc["normals"], 1e-3, 1.0, np.zeros(0), np.zeros(0))
    assert not ok
    # no normals at all: bending + mean-depth row only -> rank deficient (affine depth maps are free).  Like the reference's
    # QR the result is then meaningless; the call must come back with a flag and finite-or-flagged numbers, not hang or crash.
    ok, raw, *_ = nrsfm.ShapeFromNormals(gpu_ctx, b, np.zeros(0), np.zeros(0), np.zeros((0, 3), np.float32), 1e-3, 1.0, sc["u_all"], sc["v_all"])
    assert (not ok) or np.isfinite(raw).all()
    # sites outside the definition domain add no constraint (and do not crash)
    u = np.r_[sc["u"], 5.0]
    v = np.r_[sc["v"], -7.0]
    nr = np.vstack([sc["normals"], [[0, 0, 1]]]).astype(np.float32)
    ok1, raw1, *_ = nrsfm.ShapeFromNormals(gpu_ctx, b, u, v, nr, 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
    ok0, raw0, *_ = nrsfm.ShapeFromNormals(gpu_ctx, b, sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
    assert ok0 and ok1
    np.testing.assert_allclose(raw1, raw0, rtol=0, atol=1e-9 * np.abs(raw0).max())


@pytest.mark.parametrize("P,seed,lam", [(400, 3, 1e-2), (60, 8, 1.0), (1500, 1, 1e-4)])
def test_warp_initialize_matches_oracle(gpu_ctx, oracle_mod, P, seed, lam):
    """Warps::Warp::initialize (first half): regularised linear fit of the warp control points."""
    from defslam_amd import nrsfm, synth
    pr = synth.make_warp_problem(P, seed)
    oko, xo = oracle_mod.warp_initialize(pr["bbs"], pr["kp1"], pr["kp2"], lam)
    okg, xg = nrsfm.WarpInitialize(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], lam)
    assert oko and okg
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-9 * np.abs(xo).max())
    # the fitted warp maps kp1 close to kp2 (smooth synthetic warp, small lambda)
    if lam <= 1e-2:
        b = nrsfm.Bbs(*pr["bbs"])
        N = b.nptsu * b.nptsv
        val, _ = nrsfm.bbs_eval(gpu_ctx, nrsfm.Bbs(b.umin, b.umax, b.nptsu, b.vmin, b.vmax, b.nptsv, 2), np.stack([xg[:N], xg[N:]]).T.reshape(-1),
                                pr["kp1"][:, 0].astype(float), pr["kp1"][:, 1].astype(float))
        assert np.abs(val - pr["kp2"]).max() < 0.02


@pytest.mark.parametrize("nq,nx,seed", [(600, 900, 2), (50, 0, 7), (1500, 4000, 9)])
def test_search_by_schwarp_matches_oracle_bit_exact(gpu_ctx, oracle_mod, nq, nx, seed):
    """DefORBmatcher::searchBySchwarp (second half): index work, the bar is bit-exact -- including distance
    ties (first candidate in the reference's grid visiting order), candidates with a map point, points outside image / grid."""
    from defslam_amd import nrsfm, synth
    sc = synth.make_match_scene(nq, nx, seed=seed)
    mo = oracle_mod.search_by_schwarp(sc["bbs"], sc["x"], sc["kp1"], sc["desc1"], sc["cam2"], sc["bounds2"], sc["kp2"], sc["desc2"], sc["has_mp2"])
    mg = nrsfm.searchBySchwarp(gpu_ctx, nrsfm.Bbs(*sc["bbs"]), sc["x"], sc["kp1"], sc["desc1"], sc["cam2"], sc["bounds2"], sc["kp2"], sc["desc2"], sc["has_mp2"])
    np.testing.assert_array_equal(mg, mo)
    assert (mo >= 0).sum() >= nq // 10
    # no candidates at all / every candidate taken
    none = nrsfm.searchBySchwarp(gpu_ctx, nrsfm.Bbs(*sc["bbs"]), sc["x"], sc["kp1"], sc["desc1"], sc["cam2"], sc["bounds2"], sc["kp2"], sc["desc2"],
                                 np.ones_like(sc["has_mp2"]))
    assert (none == -1).all()
