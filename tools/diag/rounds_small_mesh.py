import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle; oracle.build()
from defslam_amd import sft, synth
B = 512; rows, cols, m = 3, 3, 60
tmpl = synth.make_grid_template(rows, cols)
regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
def mk(p):
    fr = synth.make_frame(tmpl, m, p)
    return fr
syn = [mk(p) for p in range(B)]
tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
ctx = sft.Context(0)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(fr) for fr in syn]
ctx.batch_upload(frames, *regs, 1, 50)
print('counts7', int(ctx.problem_info(0)[1][7]))
ctx.batch_run(); inl = ctx.batch_download()
bad = []
for p in range(B):
    fr = syn[p]
    r = oracle.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, ldlt_mode=1)
    f = frames[p]
    same = f.iters == r.trace.shape[0] and np.array_equal(f.trace[:, 2], r.trace[:, 2]) and np.array_equal(f.trace[:, 6], r.trace[:, 6])
    degen = r.trace[-1, 2] == 10 and r.trace[-1, 4] > 1e12 * r.trace[0, 1]
    if not same or degen:
        bad.append((p, same, degen))
        # the same problem alone: latency path (8 wavefronts, B=1)
        f1 = sft.frame_from_synth(fr)
        i1 = sft.DefPoseOptimization(ctx, f1, *regs, 1, 50)
        ctx.batch_upload(frames, *regs, 1, 50)   # restore the batch for the next one
        print('p', p, 'same', same, 'degen', degen, 'iters r/rounds/lat', r.trace.shape[0], f.iters, f1.iters,
              'last trials r/rounds/lat', r.trace[-1, 2], f.trace[-1, 2], f1.trace[-1, 2],
              'lam_last r', '%.3g' % r.trace[-1, 4], 'dxyz rounds', '%.2e' % np.abs(f.nodes_xyz - r.xyz).max(), 'lat', '%.2e' % np.abs(f1.nodes_xyz - r.xyz).max())
print('n flagged', len(bad), 'mismatch', sum(1 for b in bad if not b[1]), 'degenerate', sum(1 for b in bad if b[2]))
ctx.close()
# one-wave vs four-wave factor at the damping of the disputed iteration (lab build)
lab = sft.Context(0, lab=True)
lab.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(fr) for fr in syn]
lab.batch_upload(frames, *regs, 1, 50)
lab.batch_run()
for rel in (1e-5, 1.0, 1e6, 1e12, 1e18):
    xr, xn, ok, ms = lab.wave_check(rel, 1)
    worst = 0.0; wp = -1
    for b in range(B):
        d = np.abs(xr[b] - xn[b]).max() / max(np.abs(xr[b]).max(), 1e-300)
        if not np.isfinite(d) or d > worst: worst, wp = d, b
    print('wave_check rel %.0e: ok four/one %d/%d of %d, worst rel diff of x %.3e at problem %d' % (rel, ok[:, 0].sum(), ok[:, 1].sum(), B, worst, wp))
lab.close()
