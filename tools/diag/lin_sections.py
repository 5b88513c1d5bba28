#!/usr/bin/env python
"""Where a linearisation of the LIN kernel spends its cycles (lab build with -DSFT_PHASE_TIMERS: tools/ab_build.sh timers "-DSFT_PHASE_TIMERS"):
the assembly's section timers (wave 0, shader clock) accumulated over the linearisations of a problem in a full batched run.
  usage (GPU box): python tools/diag/lin_sections.py [B = 16384]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defslam_amd import _lib  # noqa: E402
_lib.LAB_LIB_PATH = os.path.join(ROOT, "tools", "_ab", "timers.so")
from defslam_amd import sft, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run()
ctx.synchronize()
ph, nr = ctx.rounds_timed()
ctx.batch_download(only=list(range(0, B, max(1, B // 64))))
ids = list(range(0, B, max(1, B // 64)))
d = np.array([ctx.dump(b, 7, 64) for b in ids])
its = np.array([frames[b].iters for b in ids], float)
names = {32: "corner reduction + set-up", 33: "diagonal gather", 34: "butterfly + diagonal finish (stores)", 35: "off-diagonal blocks", 36: "round overhead"}
print(f"LIN {ph['lin']:.2f} ms per step; per linearisation of a sampled problem (wave 0, cycles):")
tot = 0.0
for s, n in names.items():
    v = (d[:, s] / its).mean()
    tot += v
    print(f"  {n:40s} {v:9.0f}")
print(f"  {'assembly, sum':40s} {tot:9.0f}   rounds of wave 0 per linearisation: {(d[:, 37] / its).mean():.1f}")
ctx.close()
