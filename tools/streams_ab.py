#!/usr/bin/env python
"""Sub-batches of the throughput shape on streams of their own: ms per step of the C2 batch for 1, 2, 3, 4 sub-batches (lab option "streams")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defslam_amd import sft, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ref = None
for S in (1, 2, 3, 4):
    ctx.set_option("streams", S)
    ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ms = ctx.batch_run_timed(3) / 3
    it, tr = ctx.batch_counts()
    print(f"B={B} sub-batches {S}: {ms:.1f} ms per step, {it / (ms * 1e-3):.0f} it/s ({it} iterations, {tr} trials)", flush=True)
ctx.close()
