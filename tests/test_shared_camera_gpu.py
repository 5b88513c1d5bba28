"""Shared-camera Shape-from-Template across ranks (BASELINE north star: patches sharded over GPUs, RCCL all-reduce of the camera-pose
normal equations; include/defslam_hip.h dsh_sft_shared_solve*).  One GPU is available to these tests, so the protocol is checked
(a) with one rank against the ordinary solve, (b) with two / three ranks inside one process -- every rank a context of its own, the
all-reduce a summation kernel -- against the ORACLE's single solve of the union of the patches (one template made of disconnected
meshes, one camera), and (c) through a real RCCL communicator of size one (ncclCommInitRank + ncclAllReduce on the rank's stream).
The N-process RCCL path is the same driver with the reducer swapped; tests/test_shard_gloo.py checks the exchange protocol
between two real processes on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _split_template(tmpl, col_cuts):
    """Cut a rows x cols grid template into vertical strips (disconnected patches): facets that cross a cut are dropped.
    Returns the joint facets and, per patch, (node ids in the joint template, local facets)."""
    cols = tmpl.cols
    strip_of = np.searchsorted(np.asarray(col_cuts), np.arange(cols), side="right")      # strip index of every column
    node_strip = strip_of[np.arange(tmpl.n) % cols]
    keep = (node_strip[tmpl.facets[:, 0]] == node_strip[tmpl.facets[:, 1]]) & (node_strip[tmpl.facets[:, 1]] == node_strip[tmpl.facets[:, 2]])
    facets = tmpl.facets[keep]
    patches = []
    for s in range(len(col_cuts) + 1):
        ids = np.nonzero(node_strip == s)[0]
        local = -np.ones(tmpl.n, np.int64)
        local[ids] = np.arange(ids.size)
        fs = facets[node_strip[facets[:, 0]] == s]
        patches.append((ids, local[fs].astype(np.int32)))
    return facets, patches


def _patch_problem(ctx, tmpl, fr, ids, local_facets, median_L):
    """Template of one patch in `ctx` (with the JOINT template's median edge length: the temporal weight divides by it,
    DefOptimizer.cc:378) and the frame restricted to the observations whose facet lies in the patch."""
    from defslam_amd import sft
    ctx.template_build(tmpl.xyz0[ids], local_facets)
    t = ctx.template_get()
    ctx.template_set(tmpl.xyz0[ids], t["boundary"], t["nbr_ptr"], t["nbr_idx"], t["nbr_w"], t["k0"], t["edge_nodes"], t["edge_L0"], median_L)
    local = -np.ones(tmpl.n, np.int64)
    local[ids] = np.arange(ids.size)
    sel = np.all(local[fr.obs_nodes] >= 0, axis=1)
    f = sft.Frame(Tcw=fr.Tcw.copy(), K=fr.K.copy(), N=fr.n_frame, obs_nodes=local[fr.obs_nodes[sel]].astype(np.int32), obs_bary=fr.obs_bary[sel],
                  obs_uv=fr.obs_uv[sel], obs_invsig2=fr.obs_invsig2[sel], nodes_xyz=fr.xyz[ids].copy())
    return f, sel


def test_one_rank_equals_the_ordinary_solve(gpu_ctx):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem("C2", 4)
    gpu_ctx.template_build(tmpl.xyz0, tmpl.facets)
    ref = sft.frame_from_synth(fr)
    inl_ref = sft.DefPoseOptimization(gpu_ctx, ref, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    f = sft.frame_from_synth(fr)
    inl = sft.SharedCameraPoseOptimizationGroup([gpu_ctx], [f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)[0]
    assert (inl, f.iters, f.trials, f.status) == (inl_ref, ref.iters, ref.trials, 0)
    np.testing.assert_array_equal(f.trace[:, [2, 6]], ref.trace[:, [2, 6]])
    np.testing.assert_allclose(f.trace[:, [0, 1, 3, 4]], ref.trace[:, [0, 1, 3, 4]], rtol=1e-9)
    np.testing.assert_allclose(f.nodes_xyz, ref.nodes_xyz, rtol=0, atol=1e-10)
    np.testing.assert_allclose(f.pose7, ref.pose7, rtol=0, atol=1e-11)
    np.testing.assert_array_equal(f.mvbOutlier, ref.mvbOutlier)
    assert f.rep_error_f64 == pytest.approx(ref.rep_error_f64, rel=1e-10)


@pytest.mark.parametrize("rows,cols,cuts,m,pid", [(10, 20, [10], 800, 1), (12, 24, [8, 16], 1200, 2)])
def test_patches_on_separate_ranks_equal_the_joint_solve_of_the_oracle(oracle_mod, rows, cols, cuts, m, pid):
    """The union of the patches as ONE problem for the oracle (one template of disconnected meshes, all observations, one camera)
    against the shared-camera protocol with one rank per patch: same LM trajectory, joint pose, every patch's vertices."""
    from defslam_amd import sft, synth
    tmpl = synth.make_grid_template(rows, cols)
    fr = synth.make_frame(tmpl, m, pid)
    facets, patches = _split_template(tmpl, cuts)
    on_patch = np.isin(np.sort(tmpl.facets[fr.obs_facet], axis=1).view([("", np.int32)] * 3).ravel(), np.sort(facets, axis=1).view([("", np.int32)] * 3).ravel())
    for k in ["obs_facet", "obs_nodes", "obs_bary", "obs_uv", "obs_invsig2"]:      # observations on the facets that were cut away have no patch
        setattr(fr, k, getattr(fr, k)[on_patch])
    tc = oracle_mod.template_build(tmpl.xyz0, facets)
    regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    r = oracle_mod.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, ldlt_mode=1)
    ctxs = [sft.Context(0) for _ in patches]
    try:
        frames, sels = [], []
        for ctx, (ids, lf) in zip(ctxs, patches):
            f, sel = _patch_problem(ctx, tmpl, fr, ids, lf, tc.median_L)
            frames.append(f)
            sels.append(sel)
        assert sum(int(s.sum()) for s in sels) == fr.obs_nodes.shape[0]
        inl = sft.SharedCameraPoseOptimizationGroup(ctxs, frames, *regs)
        assert sum(inl) == r.ret
        for f, (ids, _), sel in zip(frames, patches, sels):
            assert f.status == 0 and f.iters == r.iters and f.trials == r.trials
            np.testing.assert_array_equal(f.trace[:, [2, 6]], r.trace[:, [2, 6]])
            np.testing.assert_allclose(f.trace[:, [0, 1, 3, 4]], r.trace[:, [0, 1, 3, 4]], rtol=1e-8)
            assert np.abs(f.nodes_xyz - r.xyz[ids]).max() <= 1e-7 * np.abs(r.xyz).max()
            assert np.abs(f.pose7 - r.pose7).max() <= 1e-8                      # every rank ends with the joint pose
            np.testing.assert_array_equal(f.mvbOutlier, r.outlier[sel].astype(bool))
            np.testing.assert_allclose(f.chi2_obs, r.chi2_obs[sel], rtol=1e-7, atol=1e-12)
        np.testing.assert_array_equal(frames[0].pose7, frames[-1].pose7)           # bit-identical decisions and pose on every rank
    finally:
        for c in ctxs:
            c.close()


def test_rccl_communicator_of_one_rank(gpu_ctx):
    """The RCCL path itself (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the context's stream between the phase kernels)."""
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem("smoke", 5)
    gpu_ctx.template_build(tmpl.xyz0, tmpl.facets)
    ref = sft.frame_from_synth(fr)
    sft.SharedCameraPoseOptimizationGroup([gpu_ctx], [ref], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    comm = sft.Comm(gpu_ctx, 1, 0, sft.comm_unique_id())
    try:
        f = sft.frame_from_synth(fr)
        sft.SharedCameraPoseOptimization(gpu_ctx, comm, f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    finally:
        comm.close()
    assert (f.iters, f.trials) == (ref.iters, ref.trials) and f.iters > 2
    np.testing.assert_array_equal(f.nodes_xyz, ref.nodes_xyz)
    np.testing.assert_array_equal(f.pose7, ref.pose7)


# ---- one CONNECTED template cut across two ranks (dsh_sft_connected_solve*) -----------------------------------------------------
@pytest.mark.parametrize("cfg,pid", [("smoke", 2), ("C2", 1), ("W12", 2), ("W16", 5)])
def test_connected_mesh_cut_across_two_ranks_equals_the_oracle_solve_of_the_whole_mesh(oracle_mod, cfg, pid):
    """No facet is dropped at the cut: the template is the connected grid, every curvature / stretching / observation edge that crosses
    the cut is in the system through the separator (one bandwidth of unknowns, the 2-ring halo of the cut).  Two ranks (two contexts,
    the all-reduces are summation kernels), rank g factors part g: the Levenberg-Marquardt trajectory, pose and vertices are the
    oracle's solve of the whole connected mesh; both ranks end with bit-identical results.  Narrow bands (kd <= 128: smoke, C2) and wide
    ones (kd = 182, 248) alike."""
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem(cfg, pid)
    tc = oracle_mod.template_build(tmpl.xyz0, tmpl.facets)
    regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    r = oracle_mod.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, ldlt_mode=1)
    ctxs = [sft.Context(0), sft.Context(0)]
    try:
        for c in ctxs:
            c.template_build(tmpl.xyz0, tmpl.facets)
        frames = [sft.frame_from_synth(fr), sft.frame_from_synth(fr)]
        inl = sft.ConnectedPoseOptimizationGroup(ctxs[0], ctxs[1], frames, *regs)
        _, counts = ctxs[0].problem_info(0)
        cut = sft.two_sided_cut(int(counts[5]) - 6, int(counts[6]))
        assert cut is not None and cut[1] >= counts[6]                       # the separator is at least one half-bandwidth wide
        for f, i in zip(frames, inl):
            assert f.status == 0 and i == r.ret and f.iters == r.iters and f.trials == r.trials
            np.testing.assert_array_equal(f.trace[:, [2, 6]], r.trace[:, [2, 6]])
            np.testing.assert_allclose(f.trace[:, [0, 1, 3, 4]], r.trace[:, [0, 1, 3, 4]], rtol=1e-8)
            assert np.abs(f.nodes_xyz - r.xyz).max() <= 1e-7 * np.abs(r.xyz).max()
            assert np.abs(f.pose7 - r.pose7).max() <= 1e-8
            np.testing.assert_array_equal(f.mvbOutlier, r.outlier.astype(bool))
            np.testing.assert_allclose(f.chi2_obs, r.chi2_obs, rtol=1e-7, atol=1e-12)
        np.testing.assert_array_equal(frames[0].nodes_xyz, frames[1].nodes_xyz)      # replicated control: bit-identical on both ranks
        np.testing.assert_array_equal(frames[0].pose7, frames[1].pose7)
        np.testing.assert_array_equal(frames[0].trace, frames[1].trace)
    finally:
        for c in ctxs:
            c.close()


def test_connected_mesh_c5_full_size_against_the_golden_fixture():
    """BASELINE.json configs[4], the mesh the connected mode exists for: the 2000-node template (half-bandwidth 248) cut across two ranks
    against the oracle's solve of the whole connected mesh (tests/golden/sft_C5_p0.npz: 15 minutes of CPU, done once in the build
    container -- the inputs are regenerated from the seed and their digest is checked)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_c5 import input_digest
    from defslam_amd import sft, synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sft_C5_p0.npz"))
    tmpl, fr = synth.make_problem("C5", int(g["problem_id"]))
    assert input_digest(tmpl, fr) == str(g["input_sha256"])
    regs = tuple(g["regs"])
    ctxs = [sft.Context(0), sft.Context(0)]
    try:
        for c in ctxs:
            c.template_build(tmpl.xyz0, tmpl.facets)
        frames = [sft.frame_from_synth(fr), sft.frame_from_synth(fr)]
        inl = sft.ConnectedPoseOptimizationGroup(ctxs[0], ctxs[1], frames, *regs)
        _, counts = ctxs[0].problem_info(0)
        assert int(counts[5]) == 6006 and 128 < int(counts[6]) <= 256
        for f, i in zip(frames, inl):
            assert f.status == 0 and i == int(g["out_inliers"]) and f.iters == int(g["out_iters"]) and f.trials == int(g["out_trials"])
            np.testing.assert_array_equal(f.trace[:, [2, 6]], g["out_trace"][:, [2, 6]])
            assert np.abs(f.nodes_xyz - g["out_xyz"]).max() <= 1e-7 * np.abs(g["out_xyz"]).max()
            assert np.abs(f.pose7 - g["out_pose7"]).max() <= 1e-8
            np.testing.assert_array_equal(f.mvbOutlier, np.asarray(g["out_outlier"], bool))
        np.testing.assert_array_equal(frames[0].nodes_xyz, frames[1].nodes_xyz)
        np.testing.assert_array_equal(frames[0].pose7, frames[1].pose7)
    finally:
        for c in ctxs:
            c.close()


def test_connected_mode_refuses_what_it_cannot_cut_on_every_rank():
    """A band too short for two parts next to a separator, and a half-bandwidth beyond 256: both contexts get the same error code and
    nobody is left waiting inside a collective."""
    from defslam_amd import sft, synth
    ctxs = [sft.Context(0), sft.Context(0)]
    try:
        for rows, cols, m in [(6, 6, 120), (5, 45, 400)]:
            tmpl = synth.make_grid_template(rows, cols)
            fr = synth.make_frame(tmpl, m, 0)
            for c in ctxs:
                c.template_build(tmpl.xyz0, tmpl.facets)
            with pytest.raises(sft.DshError):
                sft.ConnectedPoseOptimizationGroup(ctxs[0], ctxs[1], [sft.frame_from_synth(fr), sft.frame_from_synth(fr)], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
        # the contexts are still usable
        tmpl, fr = synth.make_problem("smoke", 0)
        ctxs[0].template_build(tmpl.xyz0, tmpl.facets)
        f = sft.frame_from_synth(fr)
        assert sft.DefPoseOptimization(ctxs[0], f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP) > 0
    finally:
        for c in ctxs:
            c.close()
