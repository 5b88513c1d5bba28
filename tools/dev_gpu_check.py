import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import oracle
from defslam_amd import synth, sft
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
for cfg in ["smoke", "C2"]:
    tmpl, fr = synth.make_problem(cfg)
    tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    tg = ctx.template_get()
    print(cfg, "template parity:", all(np.array_equal(tg[k], getattr(tc, k)) for k in ["boundary","nbr_ptr","nbr_idx","edge_nodes"]),
          np.abs(tg["nbr_w"]-tc.nbr_w).max(), np.abs(tg["k0"]-tc.k0).max(), np.abs(tg["edge_L0"]-tc.edge_L0).max(), tg["median_L"]-tc.median_L)
    args = (tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    Ho, bo, chio = oracle.sft_system(*args)
    f = sft.frame_from_synth(fr)
    ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    Hg, bg, chig = ctx.debug_system(0, Ho.shape[0])
    print(" D", Ho.shape[0], "chi", chio, chig, "H rel", np.abs(Hg-Ho).max()/np.abs(Ho).max(), "b rel", np.abs(bg-bo).max()/np.abs(bo).max())
    t=time.time(); ctx.batch_run(); inl = ctx.batch_download()[0]; dt=time.time()-t
    r = oracle.sft_solve(*args, ldlt_mode=1)
    print(" gpu time", dt, "iters", f.iters, r.iters, "trials", f.trials, r.trials, "inliers", inl, r.ret, "kd", f.half_bandwidth, "status", f.status)
    k = min(len(f.trace), len(r.trace))
    print(" trace rel diff", np.abs(f.trace[:k,:7]-r.trace[:k,:7]).max(axis=0))
    print(" xyz diff", np.abs(f.nodes_xyz-r.xyz).max(), "pose diff", np.abs(f.pose7-r.pose7).max(), "rep", f.rep_error_f64, r.rep_error, "outl eq", (f.mvbOutlier==r.outlier.astype(bool)).all())
