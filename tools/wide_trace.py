import sys, os
sys.path.insert(0, '.')
from defslam_amd import synth, sft
ctx = sft.Context(0)
rows, cols, m = synth.CONFIGS["C5"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
ctx.batch_upload([sft.frame_from_synth(synth.make_frame(tmpl, m, 0))], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
os.environ["DSH_STEP_TRACE"] = "1"; os.environ["DSH_SFT_DATAFLOW"] = "1"
ctx.phase_ms(0)
