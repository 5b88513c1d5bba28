"""Per-problem phase times under load (needs a build with EXTRA=-DSFT_PHASE_TIMERS): is one problem slower when all CUs are busy?"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import synth, sft
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
Bmax = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(Bmax)]
for B in (1, 256, 512, Bmax):
    ctx.batch_upload(frames[:B], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run(); ctx.synchronize()
    ms = ctx.batch_run_timed(1)
    import copy
    fr2 = [copy.copy(f) for f in frames[:B]]
    ctx._frames = fr2
    ctx.batch_download()
    tr = np.array([f.trials for f in fr2]); it = np.array([f.iters for f in fr2])
    tot = np.array([sum(ctx.phase_ms(b).values()) for b in range(B)])
    ph0 = ctx.phase_ms(0)
    print(f"B={B}: launch {ms:.2f} ms; trials min/mean/max {tr.min()}/{tr.mean():.1f}/{tr.max()}; per-problem ms min/mean/max {tot.min():.2f}/{tot.mean():.2f}/{tot.max():.2f}; "
          f"ms per trial mean {np.mean(tot / tr):.3f}; sum(problem ms)/256 = {tot.sum() / 256:.2f}")
    print("   problem 0:", {k: round(v, 2) for k, v in ph0.items()})
