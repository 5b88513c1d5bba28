"""Tracking and mapping interleaved in one sequence, on the C ABI (BASELINE.json configs[2] substitute `synth.SEQMAP`).

The loop the reference runs (host side: DefTracking::Track, DefTracking.cc:78-231 and DefLocalMapping::insideTheLoop / NRSfM /
updateTemplate, DefLocalMapping.cc:115-234), with every numeric stage on the GPU through the mirrors of this package:

    every frame        DefPoseOptimization(frame, template, RegLap, RegInex, RegTemp)      DefTracking.cc:244  (dsh_sft_solve)
    every 10th frame   keyframe (DefTracking.cc:175) -> SchwarpDatabase::add: Warp::initialize, searchBySchwarp, calculateSchwarps
                       -> NormalEstimator::ObtainK1K2 -> ShapeFromNormals -> SurfaceRegistration -> createTemplate
    the frame after    updateTemplate + DefPoseOptimization(..., RegTemp = 0) on the NEW template  DefTracking.cc:109-117
                       and then, on the SAME frame, TrackLocalMap -> DefPoseOptimization(..., RegTemp)    DefTracking.cc:123, 244-247
                       without the observations the first solve flagged (pFrame->mvbOutlier, DefOptimizer.cc:295)

`hooks` (optional) is called with the inputs and outputs of every stage -- tests/test_seqmap_gpu.py passes a checker that runs the
oracle of the stage on the same inputs; bench.py passes nothing and times the loop.

Two routes for the DiffProp records between the mapping stages (`route`):
    "host"    the host-buffer calls: the fit returns the records, the host keeps them per map point like WarpDatabase::mapPointsDB_
              (WarpDatabase.h:61) and hands them to the normal solve, whose normals it hands to Shape-from-Normals;
    "device"  the records stay in HBM (dsh_diffdb): dsh_schwarp_fit_batch_store -> dsh_normals_estimate_db -> dsh_sfn_estimate_db; what comes
              back are the drop flags of a fit, the per-point normals / status of the solve and the surface -- no DiffProp record crosses PCIe.
The two routes give bit-identical results (tests/test_seqmap_gpu.py runs both).
"""
from __future__ import annotations

import time

import numpy as np

from . import nrsfm, register, sft, synth


class _NoHooks:
    def __getattr__(self, name):
        return lambda *a, **k: None


def run(ctx, seq, regs=None, hooks=None, lam_init=1e-2, lam_fit=0.1, bending=1e-3, chi_limit=0.2, route="host", record=None):
    """Runs the whole sequence.  Returns a dict of counters and timings (seconds of host wall clock inside the C-ABI calls).
    record (optional list): every stage output that the next stage or the result depends on is appended to it (the route comparison)."""
    assert route in ("host", "device")
    dev = route == "device"
    hooks = hooks or _NoHooks()
    rec_out = record.append if record is not None else (lambda *a: None)
    regs = regs or (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    P, nt = seq["kp0"].shape[0], seq["n_tracked"]
    b2, b1 = nrsfm.Bbs(*seq["bbs2"]), nrsfm.Bbs(*seq["bbs1"])
    cam = seq["cam"]
    fx, fy = float(cam[0]), float(cam[1])
    K = cam.astype(np.float64)
    rows, cols = seq["mesh"]
    gu, gv = seq["grid_uv"]
    facets = seq["facets"]
    t_track = t_map = 0.0
    # ---- the first template and the embedding of the map points in it
    nodes_rest = seq["nodes0"].copy()
    ctx.template_build(nodes_rest, facets)
    pts_w = seq["frames"][0]["Xw"].astype(np.float32)
    fid, enodes, bary = ctx.template_embed_device(pts_w)
    hooks.template(nodes_rest, facets, pts_w, fid, enodes, bary, 0)
    inside = fid >= 0
    T = seq["frames"][0]["Tcw_gt"].astype(np.float32)
    x = nodes_rest.copy()
    recs_per_point = [[] for _ in range(P)]
    n_recs = np.zeros(P, np.int64)                                         # records per map point (device route: the records themselves are in HBM)
    db = nrsfm.DiffDatabase(ctx, (len(seq["kfs"]) + 1) * P) if dev else None
    prev_normal = np.zeros((P, 2), np.float32)
    has_prev = np.zeros(P, np.uint8)
    mean_depth = float(seq["depth"].mean())
    switch = False

    def add_keyframe(key):
        """SchwarpDatabase::add for the keyframe `key` against the anchor (SchwarpDatabase.cc:50-128): Warp::initialize on the tracked matches,
        searchBySchwarp for the rest, calculateSchwarps on all of them; the DiffProp records go into the database.  Returns the seconds spent."""
        kfd = seq["kfs"][key]
        dt = 0.0
        kp1, kp2 = seq["kp0"][:nt], kfd["kp_norm"][:nt]
        t0 = time.perf_counter()
        ok_init, x0 = nrsfm.WarpInitialize(ctx, b2, kp1, kp2, lam_init)
        dt += time.perf_counter() - t0
        hooks.warp_init(key, kp1, kp2, lam_init, ok_init, x0)
        q = np.arange(nt, P)
        t0 = time.perf_counter()
        mg = nrsfm.searchBySchwarp(ctx, b2, x0, seq["kp0"][q], seq["desc0"][q], cam, seq["bounds"], kfd["pix"], kfd["desc"], kfd["has_mp"], radius=8.0)
        dt += time.perf_counter() - t0
        hooks.search(key, x0, q, kfd, mg)
        found = mg >= 0
        sel = np.r_[np.arange(nt), q[found]]
        kp2_pix = kfd["pix"][np.r_[kfd["index_of_point"][:nt], mg[found]]]
        kp2n = ((kp2_pix - cam[2:]) / cam[:2]).astype(np.float32)
        fit_args = (seq["kp0"][sel], kp2n, seq["invsig"][sel], fy, fx, lam_fit, fx, fy, x0, 3)
        if dev:
            idx2 = np.r_[kfd["index_of_point"][:nt], mg[found]].astype(np.int32)
            prob = dict(bbs=b2, kp1=fit_args[0], kp2=kp2n, invsig=fit_args[2], fx_slot=fy, fy_slot=fx, lam=lam_fit, fx=fx, fy=fy, x0=x0, max_iters=3,
                        point_id=sel.astype(np.int32), idx2=idx2, tag=stats["schwarp_fits"])
            t0 = time.perf_counter()
            (xg, dg, drop, info, costs), = nrsfm.calculateSchwarpsBatch(ctx, [prob], db=db, want_records=False)
            dt += time.perf_counter() - t0
        else:
            t0 = time.perf_counter()
            xg, dg, drop, info, costs = nrsfm.calculateSchwarps(ctx, b2, *fit_args)
            dt += time.perf_counter() - t0
            hooks.schwarp(key, fit_args, xg, dg, drop, info, costs)
            for j, p in enumerate(sel):
                if not drop[j]:
                    recs_per_point[p].append(dg[j])
        stats["schwarp_fits"] += 1
        n_recs[sel[~np.asarray(drop, bool)]] += 1
        rec_out(("fit", key, xg.copy(), np.asarray(drop).copy(), info.copy(), costs.copy()))
        return dt

    stats = dict(frames=0, keyframes=0, templates=1, iters=0, trials=0, inliers=[], switch_frames=[], switch_solves=0, switch_dropped=0, schwarp_fits=0, normals=0)
    t_map += add_keyframe(-1)                                              # the keyframe the map was bootstrapped with
    for k in range(1, seq["n_frames"]):
        fk = seq["frames"][k]
        # ---- tracking: observations of the embedded map points in frame k (ground truth + pixel noise), warm start from frame k-1
        Xc = fk["Xc"][inside]
        uv = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cam[2], fy * Xc[:, 1] / Xc[:, 2] + cam[3]], 1) + seq["noise"][k][inside]
        f = sft.Frame(Tcw=T.copy(), K=K, N=1200, obs_nodes=enodes[inside], obs_bary=bary[inside].astype(np.float64), obs_uv=uv.astype(np.float32).astype(np.float64),
                      obs_invsig2=(seq["invsig"][inside].astype(np.float64)) ** 2, nodes_xyz=x.copy())
        if switch:
            # The frame right behind a template switch is solved TWICE (DefTracking.cc:109-123): first against the new template with RegTemp = 0
            # (it is at its rest shape: no temporal term) ...
            t0 = time.perf_counter()
            inl = sft.DefPoseOptimization(ctx, f, regs[0], regs[1], 0.0)
            t_track += time.perf_counter() - t0
            hooks.tracking(k, T, x, f, inl, (regs[0], regs[1], 0.0), 1)
            stats["switch_solves"] += 1
            stats["iters"] += f.iters
            stats["trials"] += f.trials
            # ... then TrackLocalMap runs the regular solve on the same frame: from the pose and the mesh the first one left, WITHOUT the
            # observations it flagged (DefOptimizer.cc:295 skips key points with mvbOutlier set; their flags stay set)
            keep = ~f.mvbOutlier
            T1, x1, flagged = f.Tcw.copy(), f.nodes_xyz.copy(), f.mvbOutlier.copy()
            f = sft.Frame(Tcw=T1.copy(), K=K, N=1200, obs_nodes=f.obs_nodes[keep], obs_bary=f.obs_bary[keep], obs_uv=f.obs_uv[keep],
                          obs_invsig2=f.obs_invsig2[keep], nodes_xyz=x1.copy())
            t0 = time.perf_counter()
            inl = sft.DefPoseOptimization(ctx, f, regs[0], regs[1], regs[2])
            t_track += time.perf_counter() - t0
            hooks.tracking(k, T1, x1, f, inl, regs, 2)
            stats["switch_solves"] += 1
            stats["switch_dropped"] += int(flagged.sum())
        else:
            t0 = time.perf_counter()
            inl = sft.DefPoseOptimization(ctx, f, regs[0], regs[1], regs[2])
            t_track += time.perf_counter() - t0
            hooks.tracking(k, T, x, f, inl, regs, 0)
        if switch:
            stats["switch_frames"].append(k)
        switch = False
        stats["frames"] += 1
        stats["iters"] += f.iters
        stats["trials"] += f.trials
        stats["inliers"].append(inl / max(int(inside.sum()), 1))
        T, x = f.Tcw.copy(), f.nodes_xyz.copy()
        rec_out(("frame", k, T.copy(), x.copy(), int(inl), int(f.iters), int(f.trials), f.mvbOutlier.copy()))
        if k not in seq["kfs"]:
            continue
        # ---- mapping: frame k is a keyframe (every 10th frame, DefTracking.cc:175)
        kf = seq["kfs"][k]
        stats["keyframes"] += 1
        t_map += add_keyframe(k)
        # NormalEstimator::ObtainK1K2 over every point with records (reference keyframe = the anchor), previous normals as start values
        pts = [p for p in range(P) if n_recs[p]]
        if dev:
            t0 = time.perf_counter()
            ng = nrsfm.ObtainK1K2Database(ctx, db, np.asarray(pts, np.int32), prev_normal[pts].copy(), has_prev[pts].copy(), seq["kp0"][pts], per_record=False)
            t_map += time.perf_counter() - t0
        else:
            rec_ptr = np.r_[0, np.cumsum([len(recs_per_point[p]) for p in pts])].astype(np.int32)
            recs = np.concatenate([np.stack(recs_per_point[p]) for p in pts]).astype(np.float32)
            R = recs.shape[0]
            nargs = (rec_ptr, recs, np.ones(R, np.uint8), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8), prev_normal[pts].copy(), has_prev[pts].copy(), seq["kp0"][pts])
            t0 = time.perf_counter()
            ng = nrsfm.ObtainK1K2(ctx, *nargs)
            t_map += time.perf_counter() - t0
            hooks.normals(k, nargs, ng)
        stats["normals"] += len(pts)
        rec_out(("normals", k, ng.k1k2.copy(), ng.status.copy(), ng.normal_ref.copy(), ng.iters.copy()))
        okn = ng.status == 0
        pts = np.asarray(pts)
        prev_normal[pts[okn]] = ng.normal_ref[okn][:, :2]
        has_prev[pts[okn]] = 1
        # ShapeFromNormals of the anchor keyframe, then SurfaceRegistration against the map points (their tracked world positions)
        un, vn = seq["kp0"][pts[okn], 0].astype(float), seq["kp0"][pts[okn], 1].astype(float)
        sargs = (un, vn, ng.normal_ref[okn], bending, mean_depth, seq["kp0"][:, 0].astype(float), seq["kp0"][:, 1].astype(float))
        t0 = time.perf_counter()
        if dev:   # the normals are picked on the device from the solve above (sel = index of the requested point)
            ok_sfn, raw, ctrl, surf = nrsfm.ShapeFromNormalsDatabase(ctx, b1, db, np.flatnonzero(okn), sargs[0], sargs[1], *sargs[3:])
        else:
            ok_sfn, raw, ctrl, surf = nrsfm.ShapeFromNormals(ctx, b1, *sargs)
        t_map += time.perf_counter() - t0
        if not dev:
            hooks.sfn(k, sargs, ok_sfn, raw, ctrl, surf)
        rec_out(("sfn", k, ok_sfn, raw.copy(), ctrl.copy(), surf.copy()))
        if not ok_sfn:
            continue
        Twc = seq["Twc"].astype(np.float64)
        surf_w = (surf.astype(np.float64) @ Twc[:3, :3].T + Twc[:3, 3]).astype(np.float32)
        map_pts = np.zeros((P, 3), np.float32)                             # DefMapPoint::RecalculatePosition of the tracked points (float32)
        map_pts[inside] = f.mappoints
        both = inside.copy()
        t0 = time.perf_counter()
        rg = register.registerSurfaces(ctx, surf_w[both], map_pts[both], kf["u_stream"], seq["Twc"], chi_limit=chi_limit)
        t_map += time.perf_counter() - t0
        hooks.registration(k, surf_w[both], map_pts[both], kf["u_stream"], seq["Twc"], chi_limit, rg)
        rec_out(("registration", k, {kk: (np.array(vv).copy() if isinstance(vv, np.ndarray) else vv) for kk, vv in rg.items()}))
        if not rg["registered"]:
            continue
        # createTemplate (DefMap.cc:55-64): the registered surface sampled on the regular grid, map points embedded again
        t0 = time.perf_counter()
        gd, _ = nrsfm.bbs_eval(ctx, b1, ctrl, gu.ravel().astype(float), gv.ravel().astype(float))
        gd = gd.ravel() * rg["s22"]
        nodes_kf = np.stack([gu.ravel() * gd, gv.ravel() * gd, gd], 1)
        Tn = rg["Tcw"].astype(np.float64)
        Rcw, tcw = Tn[:3, :3], Tn[:3, 3]
        nodes_rest = (nodes_kf - tcw) @ Rcw
        ctx.template_build(nodes_rest, facets)
        pts_w = ((surf.astype(np.float64) * rg["s22"] - tcw) @ Rcw).astype(np.float32)
        fid, enodes, bary = ctx.template_embed_device(pts_w)
        t_map += time.perf_counter() - t0
        hooks.template(nodes_rest, facets, pts_w, fid, enodes, bary, k)
        inside = fid >= 0
        x = nodes_rest.copy()                                              # the new template starts at its rest shape, the camera where tracking left it
        stats["templates"] += 1
        switch = True
    if db is not None:
        stats["db_records"] = len(db)
        db.close()
    stats["t_track"], stats["t_map"], stats["route"] = t_track, t_map, route
    return stats
