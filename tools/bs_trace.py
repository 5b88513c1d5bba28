import sys, numpy as np, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import synth, sft, _lib
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, 0))]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
# per-wave shader-clock stamps of factorisation step 40 (needs: make -C defslam_amd/csrc lab EXTRA=-DSFT_STEP_TRACE)
t = ctx.step_trace(0)
t0 = t[t > 0].min() if (t > 0).any() else 0.0
for w in range(8):
    print(f"wave {w}:", " ".join(f"{(v - t0) if v > 0 else -1:8.0f}" for v in t[w]))
