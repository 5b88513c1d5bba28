"""Tracking AND mapping interleaved in one sequence (BASELINE.json configs[2] substitute `synth.SEQMAP`), whole and on the GPU:

    frame k:            DefPoseOptimization against the current template                          (DefTracking.cc:244)
    k = 10, 20, 30:     keyframe (DefTracking.cc:175) -> Warp::initialize -> searchBySchwarp -> calculateSchwarps -> ObtainK1K2 (all
                        records so far, previous normals as start values) -> ShapeFromNormals -> SurfaceRegistration against the TRACKED
                        map points -> createTemplate + embedding                                  (DefLocalMapping.cc:138-153,172-234)
    k = 11, 21, 31:     DefPoseOptimization on the NEW template with RegTemp = 0                  (DefTracking.cc:109-115)

The loop is defslam_amd/seqmap.py; this test hangs the ORACLE of every stage into it (same inputs, the stage's tolerance; index work
bit-exact), at every keyframe, and checks tracking against the oracle on the frame after each switch, on every keyframe and on the
last frame.  A disagreement is pinned to the stage and frame that caused it; the loop as a whole has to keep tracking."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class OracleHooks:
    def __init__(self, oracle, seq):
        self.o, self.seq = oracle, seq
        self.tc = None
        self.counts = dict(template=0, tracking=0, tracking_switch=0, tracking_switch_second=0, warp_init=0, search=0, schwarp=0, normals=0, sfn=0, registration=0)

    def template(self, nodes, facets, pts_w, fid, enodes, bary, k):
        self.tc = self.o.template_build(nodes, facets)
        fo, no, bo = self.o.template_embed(self.tc, pts_w) if hasattr(self.o, "template_embed") else (None, None, None)
        if fo is not None:
            np.testing.assert_array_equal(fid, fo)
            np.testing.assert_array_equal(enodes[fid >= 0], no[fid >= 0])
            np.testing.assert_array_equal(bary.view(np.uint32)[fid >= 0], bo.view(np.uint32)[fid >= 0])
        assert (fid >= 0).mean() > 0.8
        self.counts["template"] += 1

    def tracking(self, k, T_prev, x_prev, f, inl, regs, switch):
        seq = self.seq
        if not (switch or k % seq["kf_every"] == 0 or k == seq["n_frames"] - 1 or k == 1):
            assert f.status == 0 and inl > 0.9 * f.obs_nodes.shape[0]
            return
        # the frame the GPU just solved, from the same previous state, through the oracle (the reference's g2o path restated)
        r = self.o.sft_solve(self.tc, T_prev, f.K, f.N, f.obs_nodes, f.obs_bary, f.obs_uv, f.obs_invsig2, x_prev, *regs, ldlt_mode=1)
        assert f.status == 0 and (f.iters, f.trials, inl) == (r.iters, r.trials, r.ret), (k, f.iters, r.iters, f.trials, r.trials, inl, r.ret)
        np.testing.assert_array_equal(f.trace[:f.iters, [2, 6]], r.trace[:, [2, 6]])
        assert np.abs(f.nodes_xyz - r.xyz).max() <= 1e-7 * np.abs(r.xyz).max()
        assert np.abs(f.pose7 - r.pose7).max() <= 1e-8
        np.testing.assert_array_equal(f.mvbOutlier, r.outlier.astype(bool))
        self.counts["tracking"] += 1
        self.counts["tracking_switch"] += int(bool(switch))
        self.counts["tracking_switch_second"] += int(switch == 2)
        if switch == 2:   # the second solve of a switch frame starts where the first one ended and sees only its inliers
            assert regs[2] == seq_reg_temp()

    def warp_init(self, k, kp1, kp2, lam, ok, x0):
        oko, x0o = self.o.warp_initialize(self.seq["bbs2"], kp1, kp2, lam)
        assert ok and oko
        np.testing.assert_allclose(x0, x0o, rtol=0, atol=1e-9 * np.abs(x0o).max())
        self.counts["warp_init"] += 1

    def search(self, k, x0, q, kf, mg):
        seq = self.seq
        mo = self.o.search_by_schwarp(seq["bbs2"], x0, seq["kp0"][q], seq["desc0"][q], seq["cam"], seq["bounds"], kf["pix"], kf["desc"], kf["has_mp"], radius=8.0)
        np.testing.assert_array_equal(mg, mo)                              # index work: bit-exact
        found = mg >= 0
        assert found.mean() > 0.6 and (mg[found] == kf["index_of_point"][q[found]]).mean() > 0.97
        self.counts["search"] += 1

    def schwarp(self, k, args, xg, dg, drop, info, costs):
        xo, do, dro, io, co = self.o.schwarp_fit(self.seq["bbs2"], *args)
        np.testing.assert_array_equal(info, io)
        np.testing.assert_array_equal(drop.astype(bool), dro)
        np.testing.assert_allclose(costs, co, rtol=1e-9)
        np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-9 * max(1.0, np.abs(xo).max()))
        np.testing.assert_allclose(dg, do, rtol=2e-6, atol=1e-6)
        self.counts["schwarp"] += 1

    def normals(self, k, nargs, ng):
        no = self.o.normals(*nargs)
        np.testing.assert_array_equal(ng.status, no["status"])
        okn = ng.status == 0
        assert okn.mean() > 0.9
        np.testing.assert_allclose(ng.k1k2[okn], no["k1k2"][okn], rtol=0, atol=1e-7)
        np.testing.assert_allclose(ng.normal_ref[okn], no["normal_ref"][okn], rtol=2e-6, atol=1e-6)
        self.counts["normals"] += 1

    def sfn(self, k, sargs, ok, raw, ctrl, surf):
        oko, rawo, ctrlo, surfo = self.o.sfn_estimate(self.seq["bbs1"], *sargs)
        assert ok and oko
        np.testing.assert_allclose(raw, rawo, rtol=0, atol=1e-7 * np.abs(rawo).max())
        np.testing.assert_allclose(surf, surfo, rtol=5e-6, atol=2e-6)
        self.counts["sfn"] += 1

    def registration(self, k, surf_w, map_pts, u_stream, Twc, chi_limit, rg):
        s0 = self.o.scale_min_median(surf_w, map_pts, u_stream)
        ro = self.o.optimize_horn(surf_w, map_pts, [0, 0, 0, 1, 0, 0, 0, s0["scale"]], chi=chi_limit ** 2)
        s22, Tcw_new = self.o.horn_compose(ro["sim3"], Twc)
        assert rg["registered"] and rg["acceptable"] == ro["ok"]
        assert np.float32(rg["scale0"]) == np.float32(s0["scale"])
        np.testing.assert_allclose(rg["sim3"], ro["sim3"], rtol=0, atol=1e-6)
        assert abs(rg["s22"] - s22) < 1e-5 * s22
        np.testing.assert_allclose(rg["Tcw"], Tcw_new, rtol=0, atol=1e-5)
        self.counts["registration"] += 1


def seq_reg_temp():
    from defslam_amd import synth
    return synth.REG_TEMP


def test_tracking_and_mapping_interleaved_in_one_sequence_against_the_oracles(gpu_ctx, oracle_mod):
    from defslam_amd import seqmap, synth
    seq = synth.make_interleaved_sequence(**synth.SEQMAP)
    hooks = OracleHooks(oracle_mod, seq)
    st = seqmap.run(gpu_ctx, seq, hooks=hooks)
    kf_frames = sorted(k for k in seq["kfs"] if k > 0)                      # (key -1: the keyframe the map was bootstrapped with)
    n_kf = len(kf_frames)
    assert st["frames"] == seq["n_frames"] - 1 and st["keyframes"] == n_kf == 4
    assert st["templates"] == 1 + n_kf                                      # every keyframe handed tracking a new template ...
    assert st["switch_frames"] == [k + 1 for k in kf_frames if k + 1 < seq["n_frames"]]   # ... which the NEXT frame was solved against with RegTemp = 0
    c = hooks.counts
    assert c["template"] == 1 + n_kf and c["normals"] == c["sfn"] == c["registration"] == n_kf
    assert c["warp_init"] == c["search"] == c["schwarp"] == n_kf + 1
    # a switch frame is solved twice (DefTracking.cc:109-123 then :244-247): both solves went through the oracle
    assert st["switch_solves"] == 2 * len(st["switch_frames"])
    assert c["tracking_switch"] == 2 * len(st["switch_frames"]) and c["tracking_switch_second"] == len(st["switch_frames"]) and c["tracking"] >= 2 * n_kf
    assert min(st["inliers"]) > 0.9                                         # tracking holds through every template switch
    assert st["iters"] / st["frames"] < 12


def _same(a, b, what):
    if isinstance(a, dict):
        assert a.keys() == b.keys(), what
        for k in a:
            _same(a[k], b[k], f"{what}.{k}")
    elif isinstance(a, np.ndarray):
        np.testing.assert_array_equal(a, b, err_msg=what)
    elif isinstance(a, (tuple, list)):
        assert len(a) == len(b), what
        for i, (u, v) in enumerate(zip(a, b)):
            _same(u, v, f"{what}[{i}]")
    else:
        assert a == b, what


@pytest.mark.parametrize("mesh", [None, (20, 25)])
def test_device_resident_record_chain_gives_the_sequence_of_the_host_record_route_bit_for_bit(gpu_ctx, mesh):
    """The same sequence twice: DiffProp records handed from stage to stage through host buffers (what the oracle hooks above check), and
    resident in HBM (dsh_schwarp_fit_batch_store -> dsh_normals_estimate_db -> dsh_sfn_estimate_db).  Every stage output either route produces --
    warp fits, drop flags, normals, status, surfaces, registrations, and every tracked frame's pose, vertices, inliers, iterations, outlier
    flags -- must be the same bits; also on the 500-node template of BASELINE configs[1] (20 x 25)."""
    from defslam_amd import seqmap, synth
    cfg = dict(synth.SEQMAP)
    if mesh is not None:
        cfg["mesh"] = mesh
    seq = synth.make_interleaved_sequence(**cfg)
    rec_h, rec_d = [], []
    st_h = seqmap.run(gpu_ctx, seq, route="host", record=rec_h)
    st_d = seqmap.run(gpu_ctx, seq, route="device", record=rec_d)
    assert st_d["route"] == "device" and st_d["db_records"] > 0
    assert len(rec_h) == len(rec_d) and len(rec_h) > seq["n_frames"]
    for i, (a, b) in enumerate(zip(rec_h, rec_d)):
        assert a[0] == b[0] and a[1] == b[1], (i, a[0], b[0])
        _same(a[2:], b[2:], f"{a[0]} {a[1]}")
    for k in ("frames", "keyframes", "templates", "iters", "trials", "switch_frames", "switch_solves", "switch_dropped", "schwarp_fits", "normals"):
        assert st_h[k] == st_d[k], k
    assert st_h["templates"] == 1 + st_h["keyframes"] and min(st_d["inliers"]) > 0.9
