"""Host integration shim (integration/): the reference's call sites compiled over the C ABI.

CPU: the shim and the writers compile (g++, stand-in types); Matches.txt / ErrorGTs files have the format scripts/Twiddle.py
reads back (parsed here the way Twiddle.py:38-131 does).  GPU: the compiled driver runs DefPoseOptimizationHIP on stand-in
Frame / DefMap objects and every in-place mutation of SURVEY 8b "Ownership" is checked against the Python mirror of the operator."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

INTEG = os.path.join(ROOT, "integration")


def _build():
    subprocess.run(["make", "-C", INTEG], check=True, capture_output=True)
    return os.path.join(INTEG, "build", "shim_test")


def test_shim_and_writers_compile_against_the_c_abi():
    exe = _build()
    assert os.path.exists(exe) and os.path.exists(os.path.join(INTEG, "build", "result_writers.o"))
    # the shim header only names the public ABI
    src = open(os.path.join(INTEG, "defslam_hip_shim.h")).read()
    assert "defslam_hip_debug.h" not in src and "dsh_lab" not in src


def test_writers_produce_what_twiddle_reads(tmp_path):
    import pandas as pd
    _build()
    prog = tmp_path / "w.cc"
    prog.write_text('#include "integration/result_writers.h"\n'
                    'int main(int, char** argv) { defslam_hip::MatchesWriter m(std::string(argv[1]) + "/Matches.txt");\n'
                    '  m.add_row(3, 410, 12, 640); m.add_row(12, 388, 40, 655); m.add_row(104, 0, 0, 700);\n'
                    '  std::vector<std::vector<float>> mono = {{0.1f, 0.2f, 1.0f}, {0.0f, -0.1f, 1.2f}, {0.3f, 0.1f, 0.9f}};\n'
                    '  std::vector<std::vector<float>> stereo = {{0.13f, 0.26f, 1.31f}, {0.0f, -0.13f, 1.55f}, {0.4f, 0.12f, 1.2f}};\n'
                    '  for (unsigned t : {3u, 12u}) { auto e = defslam_hip::surface_errors(mono, stereo, 1.3);\n'
                    '    if (!defslam_hip::save_results(e, defslam_hip::error_gts_name(argv[1], t))) return 1; }\n'
                    '  return m.ok() ? 0 : 1; }\n')
    exe = tmp_path / "w"
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", ROOT, str(prog), os.path.join(INTEG, "build", "result_writers.o"), "-o", str(exe)], check=True)
    subprocess.run([str(exe), str(tmp_path)], check=True)
    assert open(tmp_path / "Matches.txt").read() == "00003 410 12 640\n00012 388 40 655\n00104 0 0 700\n"
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ErrorGTs")) == ["ErrorGTs00003.txt", "ErrorGTs00012.txt"]
    # ---- read back exactly like scripts/Twiddle.py:38-131 (rms_per_sequence)
    df, names = None, []
    for file in sorted(os.listdir(tmp_path)):
        if file.startswith("ErrorGTs"):
            d = pd.read_csv(str(tmp_path / file), header=None).transpose()
            names.append(file.strip("ErrorGTs").strip(".txt"))
            df = d if df is None else pd.concat([df, d])
    df["frame"] = names
    df["frame"] = df["frame"].astype("int32")
    assert sorted(df["frame"].tolist()) == [3, 12]
    vals = df.iloc[:, 0:-1].to_numpy(dtype=float)
    mono = np.array([[0.1, 0.2, 1.0], [0.0, -0.1, 1.2], [0.3, 0.1, 0.9]], np.float32).astype(float)
    stereo = np.array([[0.13, 0.26, 1.31], [0.0, -0.13, 1.55], [0.4, 0.12, 1.2]], np.float32).astype(float)
    expect = np.linalg.norm(stereo - 1.3 * mono, axis=1)
    np.testing.assert_allclose(vals[0], expect, rtol=2e-6)              # six significant digits like Eigen's default stream format
    dm = pd.read_csv(str(tmp_path / "Matches.txt"), sep=" ", header=None, names=["frame", "inliers", "outliers", "possibleMatches"])
    assert dm["frame"].astype("int32").tolist() == [3, 12, 104]
    assert dm["inliers"].sum() / dm["possibleMatches"].sum() == pytest.approx((410 + 388) / (640 + 655 + 700))
    lines = open(tmp_path / "ErrorGTs00003.txt").read().split("\n")
    assert len(lines) == 3 and len({len(ln) for ln in lines}) == 1         # right-aligned to one width, no trailing newline


@pytest.mark.gpu
def test_def_pose_optimization_hip_mutates_frame_map_and_template_like_the_reference(gpu_ctx, oracle_mod, tmp_path):
    from defslam_amd import sft, synth
    exe = _build()
    tmpl, fr = synth.make_problem("smoke", 3)
    rng = np.random.default_rng(5)
    M = fr.obs_nodes.shape[0]
    facets_sorted = np.sort(tmpl.facets, axis=1)
    # key points of the frame: the matches (kind 1) interleaved with key points the reference skips: no map point (0), bad map
    # point (2), map point without facet (3), flagged outlier (4)
    kinds = np.r_[np.ones(M, int), np.zeros(40, int), np.full(15, 2), np.full(10, 3), np.full(12, 4)]
    src = np.r_[np.arange(M), rng.integers(0, M, 77)]
    perm = rng.permutation(kinds.size)
    kinds, src = kinds[perm], src[perm]
    N = kinds.size
    levels = (1.2 ** (-2.0 * np.arange(8))).astype(np.float32)
    octave = np.round(-np.log(fr.obs_invsig2) / (2 * np.log(1.2))).astype(int)
    xyz_now = fr.xyz + rng.normal(scale=1e-3, size=fr.xyz.shape)
    with open(tmp_path / "in.txt", "w") as f:
        f.write(f"{tmpl.n} {facets_sorted.shape[0]}\n")
        for row in tmpl.xyz0:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in xyz_now:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in facets_sorted:
            f.write(" ".join(str(int(v)) for v in row) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.K) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.Tcw.ravel()) + "\n")
        f.write(f"{N} 37\n{levels.size}\n" + " ".join(repr(float(v)) for v in levels) + "\n")
        for i in range(N):
            m = src[i]
            f.write(f"{float(fr.obs_uv[m, 0])!r} {float(fr.obs_uv[m, 1])!r} {octave[m]} {kinds[i]} {fr.obs_facet[m]} "
                    + " ".join(repr(float(v)) for v in fr.obs_bary[m]) + "\n")
        f.write(f"{synth.REG_LAP!r} {synth.REG_INEX!r} {synth.REG_TEMP!r} 1\n640\n")
    r = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt"), str(tmp_path / "Matches.txt"), "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tok = open(tmp_path / "out.txt").read().split()
    it = iter(tok)
    inliers, rep, pose_sets, locks, held_after, moved_under_lock = int(next(it)), float(next(it)), int(next(it)), int(next(it)), int(next(it)), int(next(it))
    Tcw = np.array([float(next(it)) for _ in range(16)], np.float32).reshape(4, 4)
    node_rows = np.array([[float(next(it)) for _ in range(7)] for _ in range(tmpl.n)])
    outl = np.array([int(next(it)) for _ in range(N)], bool)
    n_mp = int((kinds != 0).sum())
    mp_rows = np.array([[float(next(it)) for _ in range(4)] for _ in range(n_mp)])
    # ---- the same call through the Python mirror: the key points the reference takes, in key point order
    taken = np.nonzero(kinds == 1)[0]
    f = sft.Frame(Tcw=fr.Tcw.copy(), K=fr.K.copy(), N=N, obs_nodes=facets_sorted[fr.obs_facet[src[taken]]].astype(np.int32), obs_bary=fr.obs_bary[src[taken]],
                  obs_uv=fr.obs_uv[src[taken]], obs_invsig2=levels[octave[src[taken]]].astype(np.float64), nodes_xyz=xyz_now.copy())
    gpu_ctx.template_build(tmpl.xyz0, facets_sorted)
    inl = sft.DefPoseOptimization(gpu_ctx, f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    assert inliers == inl and pose_sets == 1                                 # return value; SetPose called once
    # MapPoint::mGlobalMutex (DefOptimizer.cc:287): taken once, held while every map point is moved, released on return
    assert locks == 1 and held_after == 0 and moved_under_lock == int((kinds[kinds != 0] != 3).sum())
    # the same frame through the ORACLE (the CPU restatement of the reference's g2o path): what the compiled shim wrote into the
    # stand-in Frame / Node objects is what the reference algorithm computes
    tc = oracle_mod.template_build(tmpl.xyz0, facets_sorted)
    ro = oracle_mod.sft_solve(tc, fr.Tcw, fr.K, N, facets_sorted[fr.obs_facet[src[taken]]].astype(np.int32), fr.obs_bary[src[taken]], fr.obs_uv[src[taken]],
                              levels[octave[src[taken]]].astype(np.float64), xyz_now, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    assert inliers == ro.ret
    np.testing.assert_allclose(node_rows[:, :3], ro.xyz, rtol=0, atol=1e-7)
    np.testing.assert_allclose(Tcw, ro.Tcw, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(outl[taken], ro.outlier.astype(bool))
    np.testing.assert_array_equal(Tcw, f.Tcw)                                # pFrame->mTcw
    assert np.float32(rep) == np.float32(f.repError)                         # pFrame->repError
    np.testing.assert_array_equal(node_rows[:, :3], f.nodes_xyz)            # Node::x,y,z of every node
    np.testing.assert_array_equal(node_rows[:, 3], np.arange(1, tmpl.n + 1)) # Node::indx = vertex id (setMeshNodes)
    assert (node_rows[:, 6] == 0).all()                                      # roles reset after the update (updateNodes)
    viewed = np.zeros(tmpl.n, bool)
    viewed[np.unique(f.obs_nodes)] = True
    np.testing.assert_array_equal(node_rows[:, 4].astype(bool), viewed)      # Node::viewed
    assert ((node_rows[:, 5] == 1) & viewed).sum() == 0 and (node_rows[:, 5] == 1).sum() > 0   # 1-ring of the viewed zone is LOCAL
    # pFrame->mvbOutlier: written for the key points that entered the graph, untouched for every other key point
    np.testing.assert_array_equal(outl[taken], f.mvbOutlier)
    assert outl[kinds == 4].all() and not outl[kinds == 0].any() and not outl[kinds == 2].any() and not outl[kinds == 3].any()
    # DefMapPoint::mWorldPos: RecalculatePosition of every map point with a facet (also bad ones and flagged ones), exactly once
    has_facet = kinds[kinds != 0] != 3
    assert (mp_rows[has_facet, 3] == 1).all() and (mp_rows[~has_facet, 3] == 0).all()
    mp_src = src[kinds != 0][has_facet]
    expect = (fr.obs_bary[mp_src][:, :, None] * f.nodes_xyz[facets_sorted[fr.obs_facet[mp_src]]]).sum(1).astype(np.float32)
    np.testing.assert_allclose(mp_rows[has_facet, :3], expect, rtol=0, atol=2e-7)
    # Matches.txt row of the frame (DefTracking.cc:299-328): inliers / outliers among key points with a good map point
    good = (kinds == 1) | (kinds == 3) | (kinds == 4)
    mI, mO = int((~outl[good]).sum()), int(outl[good].sum())
    assert open(tmp_path / "Matches.txt").read() == f"00037 {mI} {mO} 640\n"


@pytest.mark.gpu
def test_schwarp_database_hip_and_normal_estimator_hip_follow_the_reference_flow(gpu_ctx, oracle_mod, tmp_path):
    """SchwarpDatabaseHIP::add for three keyframes and ObtainK1K2HIP through the compiled shim (stand-in KeyFrame / MapPoint /
    WarpDatabase classes) against the same sequence of C-ABI calls issued from Python with the reference's bookkeeping:
    which matches are removed after the initial warp (including the residual-index quirk of DefORBmatcher.cc:167-175), which are
    found through the warp, which DiffProp records are stored, the normals and covariances written back."""
    from defslam_amd import nrsfm, synth
    _build()
    exe = os.path.join(INTEG, "build", "mapping_shim_test")
    sc = synth.make_mapping_scene(seed=31)
    P, nt = sc["kp0"].shape[0], sc["n_tracked"]
    cam, lam = sc["cam"], 0.1
    # a few gross mismatches among the tracked points so that the initial-warp test removes something
    rng = np.random.default_rng(2)
    levels = (1.2 ** (-2.0 * np.arange(8))).astype(np.float32)
    octave = np.round(-np.log(sc["invsig"].astype(np.float64) ** 2) / (2 * np.log(1.2))).astype(int)
    kfs = []
    pix0 = sc["kp0"] * cam[:2] + cam[2:]
    kfs.append(dict(N=P, pix=pix0.astype(np.float32), norm=sc["kp0"], desc=sc["desc0"], mp=np.arange(P), octave=octave))
    for kf in sc["kfs"]:
        N = kf["pix"].shape[0]
        pix = kf["pix"].copy()
        bad = kf["index_of_point"][rng.choice(nt, 6, replace=False)]
        pix[bad] += rng.uniform(25, 40, size=(6, 2)).astype(np.float32)
        mp = -np.ones(N, int)
        mp[kf["index_of_point"][:nt]] = np.arange(nt)
        kfs.append(dict(N=N, pix=pix, norm=((pix - cam[2:]) / cam[:2]).astype(np.float32), desc=kf["desc"], mp=mp, octave=rng.integers(0, 6, N)))
    bb = sc["bbs2"]
    with open(tmp_path / "in.txt", "w") as f:
        f.write(f"{len(kfs)} {P} {bb[2]} {bb[5]} {lam!r} 8\n" + " ".join(repr(float(v)) for v in levels) + "\n")
        for kf in kfs:
            f.write(f"{kf['N']}\n{bb[0]!r} {bb[1]!r} {bb[3]!r} {bb[4]!r} {float(cam[0])!r} {float(cam[1])!r} {float(cam[2])!r} {float(cam[3])!r} 0 640 0 480\n")
            for i in range(kf["N"]):
                f.write(f"{float(kf['pix'][i, 0])!r} {float(kf['pix'][i, 1])!r} {int(kf['octave'][i])} {float(kf['norm'][i, 0])!r} {float(kf['norm'][i, 1])!r} {int(kf['mp'][i])} "
                        + " ".join(str(int(b)) for b in kf["desc"][i]) + "\n")
    r = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt"), "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tok = iter(open(tmp_path / "out.txt").read().split())
    nrec = int(next(tok))
    db = np.array([[float(next(tok)) for _ in range(22)] for _ in range(nrec)])
    slots, related = [], []
    for kf in kfs:
        related.append(int(next(tok)))
        slots.append(np.array([int(next(tok)) for _ in range(kf["N"])]))
    solved = int(next(tok))
    surf = []
    for kf in kfs:
        writes = int(next(tok))
        surf.append((writes, np.array([[float(next(tok)) for _ in range(4)] for _ in range(kf["N"])])))
    cov = np.array([[float(next(tok)) for _ in range(4)] for _ in range(P)])
    pending = int(next(tok))

    # ---- the reference's flow in Python, numeric steps through the same C ABI
    b2 = nrsfm.Bbs(*bb)
    fx, fy = float(cam[0]), float(cam[1])
    kf_mp = [kf["mp"].copy() for kf in kfs]                      # KeyFrame::mvpMapPoints as ids
    obs = [dict() for _ in range(P)]                             # MapPoint::mObservations: keyframe -> index
    for k, kf in enumerate(kfs):
        for i in np.nonzero(kf["mp"] >= 0)[0]:
            obs[kf["mp"][i]][k] = int(i)
    exp_db = []
    for k in range(1, len(kfs)):
        kf = kfs[k]
        m = [(obs[p][0], i) for i, p in enumerate(kf_mp[k]) if p >= 0 and 0 in obs[p] and k in obs[p]]
        assert len(m) >= 20
        i1, i2 = np.array([a for a, _ in m]), np.array([b for _, b in m])
        isg = np.sqrt(levels[kfs[0]["octave"][i1]])
        x, out_g, res_g = nrsfm.CalculateInitialSchwarp(gpu_ctx, b2, kfs[0]["norm"][i1], kf["norm"][i2], isg, fx, fy, lam)
        # the ORACLE's CalculateInitialSchwarp (independent code: dense Cholesky initialisation, CPU B-spline evaluation, Ceres' Corrector
        # of the Huber block): the same control points, the same loss-corrected residuals, the same matches removed
        ok_o, x_o = oracle_mod.warp_initialize(bb, kfs[0]["norm"][i1], kf["norm"][i2], lam)
        assert ok_o
        np.testing.assert_allclose(x, x_o, rtol=0, atol=1e-8)
        res_o, _ = oracle_mod.schwarp_eval_initial(bb, kfs[0]["norm"][i1], kf["norm"][i2], isg, fx, fy, x_o)
        np.testing.assert_allclose(res_g, res_o, rtol=1e-7, atol=1e-7)
        raw, _ = oracle_mod.schwarp_eval(bb, kfs[0]["norm"][i1], kf["norm"][i2], isg, fx, fy, 0.0, x_o, want_jacobian=False)
        assert np.sum(raw[:2 * len(m)] ** 2) > 5.77 ** 2 and np.abs(res_o).max() < np.abs(raw[:2 * len(m)]).max()   # the loss correction is active here
        err_o = res_o[2 * np.arange(len(m))] ** 2 + res_o[2 * np.arange(len(m)) + 1] ** 2
        np.testing.assert_array_equal(out_g, err_o > 20)
        assert out_g.sum() >= 1                                     # the scene makes the initial-warp test bite
        for j in np.nonzero(out_g)[0]:
            kf_mp[k][i2[j]] = -1                                    # EraseMapPointMatch (the map point keeps its observation)
        m = [mm for mm, e in zip(m, out_g) if not e]
        cand = [i for i in range(kfs[0]["N"]) if kf_mp[0][i] >= 0 and k not in obs[kf_mp[0][i]]]
        mg = nrsfm.searchBySchwarp(gpu_ctx, b2, x, kfs[0]["norm"][cand], kfs[0]["desc"][cand], cam, np.array([0, 640, 0, 480], np.float32), kf["pix"], kf["desc"],
                                   (kf_mp[k] >= 0).astype(np.uint8), radius=2.0)
        new = [(cand[q], int(mg[q])) for q in range(len(cand)) if mg[q] >= 0]
        assert len(new) > 20
        for a, b in new:
            obs[kf_mp[0][a]][k] = b
            kf_mp[k][b] = kf_mp[0][a]
        m = m + new
        i1, i2 = np.array([a for a, _ in m]), np.array([b for _, b in m])
        isg = np.sqrt(levels[kfs[0]["octave"][i1]])
        xg, dg, drop, info, costs = nrsfm.calculateSchwarps(gpu_ctx, b2, kfs[0]["norm"][i1], kf["norm"][i2], isg, fy, fx, lam, fx, fy, x, 3)
        # the fit the shim stores records from, against the oracle's restated Ceres LM on the same matches
        xo, do, drop_o, info_o, costs_o = oracle_mod.schwarp_fit(bb, kfs[0]["norm"][i1], kf["norm"][i2], isg, fy, fx, lam, fx, fy, x, 3)
        np.testing.assert_array_equal(info, info_o)
        np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-8)
        np.testing.assert_array_equal(drop.astype(bool), drop_o)
        np.testing.assert_allclose(dg, do, rtol=2e-5, atol=2e-6)
        for j, (a, b) in enumerate(m):
            p1, p2 = kf_mp[0][a], kf_mp[k][b]
            if p1 < 0 or p2 < 0:
                continue
            if drop[j]:
                obs[p2].pop(k, None)
                kf_mp[k][b] = -1
                continue
            exp_db.append(np.r_[p1, k, a, b, dg[j].astype(np.float64)])
    exp_db = np.array(exp_db)
    # the database: same records (per map point in insertion order), float32 fields printed with 9 digits
    order_g = np.lexsort((db[:, 1], db[:, 0]))
    order_e = np.lexsort((exp_db[:, 1], exp_db[:, 0]))
    np.testing.assert_array_equal(db[order_g][:, :4], exp_db[order_e][:, :4])
    np.testing.assert_allclose(db[order_g][:, 4:], exp_db[order_e][:, 4:], rtol=2e-7, atol=1e-12)
    for k in range(len(kfs)):
        np.testing.assert_array_equal(slots[k], kf_mp[k])          # matches erased / registered exactly like the reference's bookkeeping
    assert related == [0, 1, 1]
    # ---- NormalEstimator: every point with records, reference keyframe 0, no previous normals
    pts = sorted(set(exp_db[:, 0].astype(int)))
    recs_by_pt = {p: exp_db[exp_db[:, 0] == p] for p in pts}
    rec_ptr = np.r_[0, np.cumsum([len(recs_by_pt[p]) for p in pts])].astype(np.int32)
    recs = np.concatenate([recs_by_pt[p][:, 4:] for p in pts]).astype(np.float32)
    R = recs.shape[0]
    ng = nrsfm.ObtainK1K2(gpu_ctx, rec_ptr, recs, np.ones(R, np.uint8), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8), np.zeros((len(pts), 2), np.float32),
                          np.zeros(len(pts), np.uint8), kfs[0]["norm"][pts])
    assert solved == int((ng.status == 0).sum()) and pending == 0
    # the normals the compiled shim wrote into the surfaces, against the oracle's per-point solve of the same records
    no = oracle_mod.normals(rec_ptr, recs, np.ones(R, np.uint8), np.zeros((R, 2), np.float32), np.zeros(R, np.uint8), np.zeros((len(pts), 2), np.float32),
                            np.zeros(len(pts), np.uint8), kfs[0]["norm"][pts])
    np.testing.assert_array_equal(ng.status, no["status"])
    okp = ng.status == 0
    np.testing.assert_allclose(ng.k1k2[okp], no["k1k2"][okp], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ng.normal_ref[okp], no["normal_ref"][okp], rtol=2e-6, atol=1e-7)
    w0, s0 = surf[0]
    assert w0 == solved
    for q, p in enumerate(pts):
        if ng.status[q] == 0:
            assert s0[p, 0] == 1
            np.testing.assert_allclose(s0[p, 1:], ng.normal_ref[q], rtol=2e-7)
            np.testing.assert_allclose(cov[p], ng.cov[q].ravel(), rtol=1e-12)
    allrec = np.concatenate([recs_by_pt[p] for p in pts])
    for k in range(1, len(kfs)):
        wk, sk = surf[k]
        sel = (allrec[:, 1] == k) & (ng.rec_written == 1)
        assert wk == int(sel.sum())
        np.testing.assert_allclose(sk[allrec[sel, 3].astype(int), 1:], ng.normal_rec[sel], rtol=2e-7)
    # ---- the same run with the DiffProp records resident in HBM (SchwarpDatabaseHIP::enable_device_records + ObtainK1K2DeviceHIP: dsh_diffdb,
    # dsh_schwarp_fit_batch_store, dsh_normals_estimate_db): no record on the host, and everything the calls change comes out the same
    r2 = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out_dev.txt"), "0", "devrec"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    host_lines = open(tmp_path / "out.txt").read().split("\n")
    dev_lines = open(tmp_path / "out_dev.txt").read().split("\n")
    assert dev_lines[0].split() == [str(nrec), "0"]                # as many records on the device as the host map held; none on the host
    assert dev_lines[1:] == host_lines[1 + nrec:]                  # bookkeeping, normals written into the surfaces, covariances, pending flags
    # ---- a point whose reference keyframe changed after its records were stored (MapPoint::EraseObservation can reassign mpRefKF): the stored
    # records are anchored in the OLD keyframe, the device mode has no first-keyframe normals for them -> it must skip the point (and say so),
    # not solve it against the new keyframe's key point.  The first 3 points with records are re-anchored before the normals are solved.
    r3 = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out_dev3.txt"), "0", "devrec", "3"], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    assert "reanchored 3 skipped 3" in r3.stderr
    l3 = open(tmp_path / "out_dev3.txt").read().split("\n")
    nkf = len(kfs)
    i_solved = 1 + nkf
    assert l3[:i_solved] == dev_lines[:i_solved]                   # storing is untouched
    first3 = pts[:3]
    lost = int(sum(ng.status[q] == 0 for q in range(3)))
    assert int(l3[i_solved]) == solved - lost
    i_cov = i_solved + 1 + sum(1 + int(k["N"]) for k in kfs)
    for p in range(len(cov)):
        if p in first3:
            assert [float(v) for v in l3[i_cov + p].split()] == [0.0, 0.0, 0.0, 0.0]        # never solved: covNorm untouched
        else:
            assert l3[i_cov + p] == dev_lines[i_cov + p]


def _parse_shim_out(path, n_nodes, N, n_mp):
    it = iter(open(path).read().split())
    head = dict(inliers=int(next(it)), rep=float(next(it)), pose_sets=int(next(it)), locks=int(next(it)), held=int(next(it)), moved_under_lock=int(next(it)))
    head["Tcw"] = np.array([float(next(it)) for _ in range(16)], np.float32).reshape(4, 4)
    head["nodes"] = np.array([[float(next(it)) for _ in range(7)] for _ in range(n_nodes)])
    head["outl"] = np.array([int(next(it)) for _ in range(N)], bool)
    head["mp"] = np.array([[float(next(it)) for _ in range(4)] for _ in range(n_mp)])
    return head


@pytest.mark.gpu
def test_switch_frame_two_calls_through_the_shim_follow_the_oracle(oracle_mod, tmp_path):
    """The frame right behind a template switch is solved twice (DefTracking.cc:109-123, then TrackLocalMap :244-247): RegTemp = 0 first,
    then the regular call on the SAME frame object -- from the pose and the nodes the first call wrote, without the key points it flagged
    (DefOptimizer.cc:295).  The compiled shim executable makes both calls; each is checked against the oracle on the inputs it saw."""
    from defslam_amd import synth
    exe = _build()
    tmpl, fr = synth.make_problem("smoke", 5)
    M = fr.obs_nodes.shape[0]
    facets_sorted = np.sort(tmpl.facets, axis=1)
    N = M
    levels = (1.2 ** (-2.0 * np.arange(8))).astype(np.float32)
    octave = np.round(-np.log(fr.obs_invsig2) / (2 * np.log(1.2))).astype(int)
    with open(tmp_path / "in.txt", "w") as f:
        f.write(f"{tmpl.n} {facets_sorted.shape[0]}\n")
        for row in tmpl.xyz0:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in fr.xyz:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        for row in facets_sorted:
            f.write(" ".join(str(int(v)) for v in row) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.K) + "\n")
        f.write(" ".join(repr(float(v)) for v in fr.Tcw.ravel()) + "\n")
        f.write(f"{N} 41\n{levels.size}\n" + " ".join(repr(float(v)) for v in levels) + "\n")
        for m in range(M):
            f.write(f"{float(fr.obs_uv[m, 0])!r} {float(fr.obs_uv[m, 1])!r} {octave[m]} 1 {fr.obs_facet[m]} " + " ".join(repr(float(v)) for v in fr.obs_bary[m]) + "\n")
        f.write(f"{synth.REG_LAP!r} {synth.REG_INEX!r} {synth.REG_TEMP!r} 1\n640\n")
    r = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt"), str(tmp_path / "Matches.txt"), "0", "switch"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    first = _parse_shim_out(str(tmp_path / "out.txt") + ".first", tmpl.n, N, M)
    second = _parse_shim_out(tmp_path / "out.txt", tmpl.n, N, M)
    obs_nodes = facets_sorted[fr.obs_facet].astype(np.int32)
    isig = levels[octave].astype(np.float64)
    tc = oracle_mod.template_build(tmpl.xyz0, facets_sorted)
    # ---- first call: every match, RegTemp = 0
    r1 = oracle_mod.sft_solve(tc, fr.Tcw, fr.K, N, obs_nodes, fr.obs_bary, fr.obs_uv, isig, fr.xyz, synth.REG_LAP, synth.REG_INEX, 0.0)
    assert first["inliers"] == r1.ret and first["pose_sets"] == 1 and first["locks"] == 1
    np.testing.assert_allclose(first["nodes"][:, :3], r1.xyz, rtol=0, atol=1e-7)
    np.testing.assert_allclose(first["Tcw"], r1.Tcw, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(first["outl"], r1.outlier.astype(bool))
    assert first["outl"].any(), "the case must flag something, otherwise the second call proves nothing"
    # ---- second call: the regular regulariser, the first call's state (float32 pose round trip like Frame::mTcw), its inliers only
    keep = ~first["outl"]
    r2 = oracle_mod.sft_solve(tc, first["Tcw"], fr.K, N, obs_nodes[keep], fr.obs_bary[keep], fr.obs_uv[keep], isig[keep], first["nodes"][:, :3].copy(),
                              synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    assert second["inliers"] == r2.ret and second["pose_sets"] == 2 and second["locks"] == 2 and second["held"] == 0
    np.testing.assert_allclose(second["nodes"][:, :3], r2.xyz, rtol=0, atol=1e-7)
    np.testing.assert_allclose(second["Tcw"], r2.Tcw, rtol=0, atol=2e-6)
    # flags: what the first call flagged stays flagged (those key points never entered the second graph), the rest is the second call's verdict
    assert second["outl"][~keep].all()
    np.testing.assert_array_equal(second["outl"][keep], r2.outlier.astype(bool))
    # Matches.txt counts the final flags
    mI, mO = int((~second["outl"]).sum()), int(second["outl"].sum())
    assert open(tmp_path / "Matches.txt").read() == f"00041 {mI} {mO} 640\n"
