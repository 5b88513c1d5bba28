"""Assembly-only launches of a C2 batch (dsh_lab_sft_assemble_timed), for rocprofv3 PMC passes: one full run, then `reps` launches
that do one linearisation + normal-equation assembly per problem."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import sft, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
ms = ctx.batch_assemble_timed(reps) / reps
nbytes = sum(ctx.problem_info(b)[0] for b in range(B))
print(f"C2 B={B}: assembly-only pass {ms:.3f} ms, algorithmic {nbytes / 1e9:.3f} GB per pass = {nbytes / ms / 1e6:.0f} GB/s")
