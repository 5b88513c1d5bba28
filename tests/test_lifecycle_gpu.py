"""GPU: context life cycle -- repeated create/destroy, two live contexts, scratch reuse across calls of different sizes,
re-upload of differently sized batches.  Guards the grow-only device scratch and the batch arena against stale state."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve_one(ctx, cfg="smoke", pid=0):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem(cfg, pid)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    f = sft.frame_from_synth(fr)
    inl = sft.DefPoseOptimization(ctx, f, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    return f, inl


def test_create_destroy_many_contexts():
    from defslam_amd import sft
    ref = None
    for _ in range(6):
        ctx = sft.Context(0)
        f, inl = _solve_one(ctx)
        ctx.close()
        if ref is None:
            ref = (f.nodes_xyz.copy(), inl)
        else:
            np.testing.assert_array_equal(f.nodes_xyz, ref[0])
            assert inl == ref[1]


def test_two_contexts_do_not_share_state():
    from defslam_amd import sft, nrsfm, synth
    a, b = sft.Context(0), sft.Context(0)
    fa, ia = _solve_one(a, "smoke", 1)
    fb, ib = _solve_one(b, "smoke", 2)
    # interleave one-shot mapping calls (they use each context's scratch) with SfT solves
    sc = synth.make_sfn_scene(200, seed=1)
    ok1, raw1, *_ = nrsfm.ShapeFromNormals(a, nrsfm.Bbs(*sc["bbs"]), sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
    fb2, ib2 = _solve_one(b, "smoke", 2)
    ok2, raw2, *_ = nrsfm.ShapeFromNormals(b, nrsfm.Bbs(*sc["bbs"]), sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
    fa2, ia2 = _solve_one(a, "smoke", 1)
    assert ok1 and ok2
    np.testing.assert_array_equal(raw1, raw2)
    np.testing.assert_array_equal(fa.nodes_xyz, fa2.nodes_xyz)
    np.testing.assert_array_equal(fb.nodes_xyz, fb2.nodes_xyz)
    assert (ia, ib) == (ia2, ib2)
    a.close()
    b.close()


def test_scratch_and_arena_reuse_across_sizes(gpu_ctx):
    """Big call, small call, big call again: results of the repeated call are bit-identical (no stale scratch contents leak in)."""
    from defslam_amd import sft, nrsfm, synth
    rng = np.random.default_rng(0)
    b = nrsfm.Bbs(-0.7, 0.7, 13, -0.55, 0.55, 15, 2)
    ctrl = rng.normal(size=(2, 195))
    u, v = rng.uniform(-0.69, 0.69, 200000), rng.uniform(-0.54, 0.54, 200000)
    big1, _ = nrsfm.bbs_eval(gpu_ctx, b, ctrl, u, v)
    small, _ = nrsfm.bbs_eval(gpu_ctx, b, ctrl, u[:7], v[:7])
    pr = synth.make_warp_problem(300, 3)
    fit1 = nrsfm.calculateSchwarps(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], 0.1, pr["fx"], pr["fy"], pr["x0"], 3)
    big2, _ = nrsfm.bbs_eval(gpu_ctx, b, ctrl, u, v)
    fit2 = nrsfm.calculateSchwarps(gpu_ctx, nrsfm.Bbs(*pr["bbs"]), pr["kp1"], pr["kp2"], pr["invsig"], pr["fy"], pr["fx"], 0.1, pr["fx"], pr["fy"], pr["x0"], 3)
    np.testing.assert_array_equal(big1, big2)
    np.testing.assert_array_equal(small, big1[:7])
    np.testing.assert_array_equal(fit1[0], fit2[0])
    np.testing.assert_array_equal(fit1[1], fit2[1])
    # SfT batches of different sizes through the same context
    tmpl = synth.make_grid_template(10, 10)
    gpu_ctx.template_build(tmpl.xyz0, tmpl.facets)
    frames = [sft.frame_from_synth(synth.make_frame(tmpl, 150 + 5 * p, p)) for p in range(12)]
    inl12 = sft.DefPoseOptimizationBatch(gpu_ctx, frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    x12 = [f.nodes_xyz.copy() for f in frames]
    few = [sft.frame_from_synth(synth.make_frame(tmpl, 150 + 5 * p, p)) for p in (3, 9)]
    inl2 = sft.DefPoseOptimizationBatch(gpu_ctx, few, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    assert inl2 == [inl12[3], inl12[9]]
    np.testing.assert_array_equal(few[0].nodes_xyz, x12[3])
    np.testing.assert_array_equal(few[1].nodes_xyz, x12[9])
