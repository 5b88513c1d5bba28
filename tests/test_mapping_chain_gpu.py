"""The mapping loop end to end on the GPU, stage by stage against the oracles (DefLocalMapping::NRSfM, DefLocalMapping.cc:160-234;
SchwarpDatabase::add, SchwarpDatabase.cc:50-128; the template it hands to tracking, DefTracking.cc:109-115,175):

    dsh_warp_initialize -> dsh_search_by_schwarp -> dsh_schwarp_fit (per keyframe pair)
      -> dsh_normals_estimate -> dsh_sfn_estimate -> dsh_surface_register -> dsh_template_build + embedding -> dsh_sft_solve

and, beside it, the device-resident route through the handles (dsh_schwarp_fit_batch_store -> dsh_normals_estimate_db ->
dsh_sfn_estimate_db: key points in, normals / control points out, no DiffProp record or normal crosses PCIe), which has to give the
same bits.

Every stage consumes what the previous GPU stage produced; the oracle of the stage is run on the same inputs and compared (index
work bit-exact, floating point to the stage's tolerance), so a disagreement is pinned to the stage that caused it, and the chain
as a whole has to recover the scene (scale, surface, tracking pose)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 23])
def test_mapping_loop_chain_matches_the_oracles_stage_by_stage(gpu_ctx, oracle_mod, seed):
    from defslam_amd import nrsfm, register, sft, synth
    sc = synth.make_mapping_scene(seed=seed)
    P, nt = sc["kp0"].shape[0], sc["n_tracked"]
    b2, b1 = nrsfm.Bbs(*sc["bbs2"]), nrsfm.Bbs(*sc["bbs1"])
    fx, fy = float(sc["cam"][0]), float(sc["cam"][1])
    lam_init, lam_fit = 1e-2, 0.1
    recs_per_point = [[] for _ in range(P)]
    db = nrsfm.DiffDatabase(gpu_ctx, 4 * P)
    for ikf, kf in enumerate(sc["kfs"]):
        # ---- Warp::initialize on the tracked matches
        kp1, kp2 = sc["kp0"][:nt], kf["kp_norm"][:nt]
        okg, x0 = nrsfm.WarpInitialize(gpu_ctx, b2, kp1, kp2, lam_init)
        oko, x0o = oracle_mod.warp_initialize(sc["bbs2"], kp1, kp2, lam_init)
        assert okg and oko
        np.testing.assert_allclose(x0, x0o, rtol=0, atol=1e-9 * np.abs(x0o).max())
        # ---- DefORBmatcher::searchBySchwarp for the key points without a match yet (bit-exact index work)
        q = np.arange(nt, P)
        mg = nrsfm.searchBySchwarp(gpu_ctx, b2, x0, sc["kp0"][q], sc["desc0"][q], sc["cam"], sc["bounds"], kf["pix"], kf["desc"], kf["has_mp"], radius=8.0)
        mo = oracle_mod.search_by_schwarp(sc["bbs2"], x0, sc["kp0"][q], sc["desc0"][q], sc["cam"], sc["bounds"], kf["pix"], kf["desc"], kf["has_mp"], radius=8.0)
        np.testing.assert_array_equal(mg, mo)
        found = mg >= 0
        assert found.sum() > 0.6 * q.size                                # the warp guides the search to most of them ...
        assert (mg[found] == kf["index_of_point"][q[found]]).mean() > 0.97   # ... and to the right key points
        # ---- SchwarpDatabase::calculateSchwarps on tracked + new matches (the reference passes (fy, fx) in Warp's (fx, fy) slots)
        sel = np.r_[np.arange(nt), q[found]]
        kp2_pix = kf["pix"][np.r_[kf["index_of_point"][:nt], mg[found]]]
        kp2n = ((kp2_pix - sc["cam"][2:]) / sc["cam"][:2]).astype(np.float32)
        args = (sc["kp0"][sel], kp2n, sc["invsig"][sel], fy, fx, lam_fit, fx, fy, x0, 3)
        xg, dg, drg, ig, cg = nrsfm.calculateSchwarps(gpu_ctx, b2, *args)
        xo, do, dro, io, co = oracle_mod.schwarp_fit(sc["bbs2"], *args)
        np.testing.assert_array_equal(ig, io)
        np.testing.assert_array_equal(drg, dro)
        np.testing.assert_allclose(cg, co, rtol=1e-9)
        np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-9 * max(1.0, np.abs(xo).max()))
        np.testing.assert_allclose(dg, do, rtol=2e-6, atol=1e-6)
        assert cg[1] <= cg[0]
        # the same fit through the handle: the records stay in HBM, only the drop flags come back
        idx2 = np.r_[kf["index_of_point"][:nt], mg[found]].astype(np.int32)
        (xs, ds, drs, is_, cs), = nrsfm.calculateSchwarpsBatch(gpu_ctx, [dict(bbs=b2, kp1=args[0], kp2=kp2n, invsig=args[2], fx_slot=fy, fy_slot=fx, lam=lam_fit, fx=fx, fy=fy,
                                                                               x0=x0, max_iters=3, point_id=sel.astype(np.int32), idx2=idx2, tag=ikf)],
                                                               db=db, want_records=False)
        np.testing.assert_array_equal(xs, xg)
        np.testing.assert_array_equal(drs, drg)
        assert not ds.any()
        for k, p in enumerate(sel):
            if not drg[k]:
                recs_per_point[p].append(dg[k])
    # ---- NormalEstimator::ObtainK1K2: every record's first keyframe is the point's reference keyframe, no previous normal
    pts = [p for p in range(P) if recs_per_point[p]]
    rec_ptr = np.r_[0, np.cumsum([len(recs_per_point[p]) for p in pts])].astype(np.int32)
    recs = np.concatenate([np.stack(recs_per_point[p]) for p in pts]).astype(np.float32)
    R = recs.shape[0]
    nargs = (rec_ptr, recs, np.ones(R, np.uint8), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8), np.zeros((len(pts), 2), np.float32),
             np.zeros(len(pts), np.uint8), sc["kp0"][pts])
    ng = nrsfm.ObtainK1K2(gpu_ctx, *nargs)
    no = oracle_mod.normals(*nargs)
    np.testing.assert_array_equal(ng.status, no["status"])
    assert len(db) == R
    nd = nrsfm.ObtainK1K2Database(gpu_ctx, db, np.array(pts, np.int32), nargs[5], nargs[6], nargs[7])
    for k in ["k1k2", "cov", "status", "normal_ref", "iters", "normal_rec", "rec_written"]:
        np.testing.assert_array_equal(getattr(nd, k), getattr(ng, k), err_msg=k)
    assert set(nd.rec_tag.tolist()) == set(range(len(sc["kfs"]))) and (nd.rec_point == np.repeat(np.arange(len(pts)), np.diff(rec_ptr))).all()
    okn = ng.status == 0
    assert okn.mean() > 0.9
    np.testing.assert_allclose(ng.k1k2[okn], no["k1k2"][okn], rtol=0, atol=1e-7)
    np.testing.assert_allclose(ng.normal_ref[okn], no["normal_ref"][okn], rtol=2e-6, atol=1e-6)
    # the normals are the scene's: n ~ X_u x X_v of the true surface (noise of the matches allowed for)
    u, v = sc["kp0"][pts][okn, 0].astype(float), sc["kp0"][pts][okn, 1].astype(float)
    nrm = ng.normal_ref[okn].astype(float)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    du = 0.12 + 0.05 * 2.0 * np.cos(2.0 * u) * np.cos(1.5 * v)
    dv = -0.08 - 0.05 * 1.5 * np.sin(2.0 * u) * np.sin(1.5 * v)
    d = sc["depth"][pts][okn]
    tn = np.cross(np.stack([du * u + d, du * v, du], 1), np.stack([dv * u, dv * v + d, dv], 1))
    tn /= np.linalg.norm(tn, axis=1, keepdims=True)
    assert np.median(np.abs((nrm * tn).sum(1))) > 0.9                   # two views, the reference's warp quirks: ~20 degrees, not a parity matter
    # ---- ShapeFromNormals: depth B-spline of the keyframe from those normals, surface points for every key point
    mean_depth = float(sc["depth"].mean())
    sargs = (u, v, ng.normal_ref[okn], 1e-3, mean_depth, sc["kp0"][:, 0].astype(float), sc["kp0"][:, 1].astype(float))
    okg, rawg, ctrlg, surf = nrsfm.ShapeFromNormals(gpu_ctx, b1, *sargs)
    oko, rawo, ctrlo, surfo = oracle_mod.sfn_estimate(sc["bbs1"], *sargs)
    assert okg and oko
    okd, rawd, ctrld, surfd = nrsfm.ShapeFromNormalsDatabase(gpu_ctx, b1, db, np.flatnonzero(okn), *sargs[:2], *sargs[3:])   # normals picked on the device
    assert okd
    np.testing.assert_array_equal(rawd, rawg)
    np.testing.assert_array_equal(ctrld, ctrlg)
    np.testing.assert_array_equal(surfd.view(np.uint32), surf.view(np.uint32))
    np.testing.assert_allclose(rawg, rawo, rtol=0, atol=1e-7 * np.abs(rawo).max())
    np.testing.assert_allclose(surf, surfo, rtol=5e-6, atol=2e-6)
    shape_err = np.abs(surf[:, 2] / np.median(surf[:, 2]) - sc["depth"] / np.median(sc["depth"]))
    assert np.median(shape_err) < 0.08                                  # the surface up to scale
    # ---- SurfaceRegistration: the surface (keyframe frame -> world) against the map points: scale and pose
    Twc = sc["Twc"].astype(np.float64)
    surf_w = (surf.astype(np.float64) @ Twc[:3, :3].T + Twc[:3, 3]).astype(np.float32)
    rg = register.registerSurfaces(gpu_ctx, surf_w, sc["map_pts"], sc["u_stream"], sc["Twc"], chi_limit=0.2)
    s0 = oracle_mod.scale_min_median(surf_w, sc["map_pts"], sc["u_stream"])
    ro = oracle_mod.optimize_horn(surf_w, sc["map_pts"], [0, 0, 0, 1, 0, 0, 0, s0["scale"]], chi=0.2 ** 2)
    s22, Tcw_new = oracle_mod.horn_compose(ro["sim3"], sc["Twc"])
    assert rg["registered"] and rg["acceptable"] == ro["ok"]
    assert np.float32(rg["scale0"]) == np.float32(s0["scale"])
    np.testing.assert_allclose(rg["sim3"], ro["sim3"], rtol=0, atol=1e-6)
    assert abs(rg["s22"] - s22) < 1e-5 * s22
    np.testing.assert_allclose(rg["Tcw"], Tcw_new, rtol=0, atol=1e-5)
    true_over_surface = sc["scale_true"] * np.median(sc["depth"]) / np.median(surf[:, 2])
    assert abs(rg["s22"] / true_over_surface - 1.0) < 0.1               # the recovered scale is the scene's
    # ---- the new template: the registered surface sampled on a regular grid, map points embedded in it (DefMap / Template)
    gu, gv = np.meshgrid(np.linspace(sc["kp0"][:, 0].min(), sc["kp0"][:, 0].max(), 14), np.linspace(sc["kp0"][:, 1].min(), sc["kp0"][:, 1].max(), 12))
    gd, _ = nrsfm.bbs_eval(gpu_ctx, b1, ctrlg, gu.ravel().astype(float), gv.ravel().astype(float))
    gd = gd.ravel() * rg["s22"]
    nodes_kf = np.stack([gu.ravel() * gd, gv.ravel() * gd, gd], 1)     # keyframe frame, metric
    Tcw = rg["Tcw"].astype(np.float64)
    Rcw, tcw = Tcw[:3, :3], Tcw[:3, 3]
    nodes_w = (nodes_kf - tcw) @ Rcw                                    # world: R^T (x - t)
    facets = synth.regular_triangulation(12, 14)
    gpu_ctx.template_build(nodes_w, facets)
    tc = oracle_mod.template_build(nodes_w, facets)
    pts_w = ((surf.astype(np.float64) * rg["s22"] - tcw) @ Rcw).astype(np.float32)
    fid, enodes, bary = gpu_ctx.template_embed_device(pts_w)
    fid_h, enodes_h, bary_h = gpu_ctx.template_embed(pts_w)
    np.testing.assert_array_equal(fid, fid_h)
    np.testing.assert_array_equal(enodes, enodes_h)
    np.testing.assert_array_equal(bary.view(np.uint32), bary_h.view(np.uint32))
    inside = fid >= 0
    assert inside.mean() > 0.8
    # ---- tracking the next frame against that template: DefPoseOptimization, GPU vs oracle
    rng = np.random.default_rng(seed + 1)
    Rn = synth._rodrigues(np.array([0.01, -0.02, 0.015]))
    Tn = np.eye(4)
    Tn[:3, :3] = Rn @ Rcw
    Tn[:3, 3] = Rn @ tcw + np.array([0.01, 0.005, -0.01])
    Xw_true = (sc["X"] * sc["scale_true"]) @ sc["Twc"][:3, :3].astype(np.float64).T + sc["Twc"][:3, 3].astype(np.float64)
    pc = Xw_true[inside] @ Tn[:3, :3].T + Tn[:3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + sc["cam"][2], fy * pc[:, 1] / pc[:, 2] + sc["cam"][3]], 1) + rng.normal(scale=0.4, size=(int(inside.sum()), 2))
    f = sft.Frame(Tcw=rg["Tcw"].copy(), K=sc["cam"].astype(np.float64), N=1200, obs_nodes=enodes[inside], obs_bary=bary[inside].astype(np.float64),
                  obs_uv=uv.astype(np.float32).astype(np.float64), obs_invsig2=(sc["invsig"][inside].astype(np.float64)) ** 2, nodes_xyz=nodes_w.copy())
    inl = sft.DefPoseOptimization(gpu_ctx, f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    r = oracle_mod.sft_solve(tc, rg["Tcw"], f.K, f.N, f.obs_nodes, f.obs_bary, f.obs_uv, f.obs_invsig2, nodes_w, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP,
                             ldlt_mode=1)
    assert f.status == 0 and f.iters == r.iters and f.trials == r.trials and inl == r.ret
    assert np.abs(f.nodes_xyz - r.xyz).max() <= 1e-7 * np.abs(r.xyz).max()
    assert np.abs(f.pose7 - r.pose7).max() <= 1e-8
    np.testing.assert_array_equal(f.mvbOutlier, r.outlier.astype(bool))
    assert inl > 0.9 * inside.sum()
    # the tracked pose is the frame's pose (the template is the reconstruction, not the truth: centimetres, not nanometres)
    assert np.abs(f.Tcw[:3, 3] - Tn[:3, 3]).max() < 0.3
