import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from defslam_amd import sft, synth
B = 16384; reps = 5
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
frs = [synth.make_frame(tmpl, m, p) for p in range(B)]
for waves in (0, 4, 8):
    ctx = sft.Context(0, lab=True)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    if waves: ctx.set_option("waves", waves)
    frames = [sft.frame_from_synth(fr) for fr in frs]
    ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    print("waves", waves, "counts7", int(ctx.problem_info(0)[1][7]), flush=True)
    ctx.batch_run(); ctx.synchronize()
    ms = ctx.batch_assemble_timed(reps) / reps
    print(f"waves={waves}: assembly-only pass {ms:.3f} ms", flush=True)
    ctx.close()
