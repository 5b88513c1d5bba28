import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from defslam_amd import synth, sft, _lib
ctx = sft.Context(0)
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, 0))]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
import os
os.environ["DSH_STEP_TRACE"] = "1"
