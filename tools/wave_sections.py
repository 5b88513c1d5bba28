#!/usr/bin/env python
"""Where the cycles of a one-wavefront factor step go (sft_wave.h section timers): build the lab variant with
   tools/ab_build.sh wvtrace "-DWV_STEP_TRACE"   and run   python tools/wave_sections.py [B] [variant = wvtrace] [tail option]
Prints, for the LAST factorisation of a sample of problems of a full batched run (deferred back substitution riding along), the mean
shader-clock cycles per factor step of every section."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import _lib  # noqa: E402

_lib.LAB_LIB_PATH = os.path.join(ROOT, "tools", "_ab", (sys.argv[2] if len(sys.argv) > 2 else "wvtrace") + ".so")
from defslam_amd import sft, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
if len(sys.argv) > 3:
    ctx.set_option("tail", int(sys.argv[3]))     # 0: rounds to the end (small batches: the factor kernel at a fraction of the device)
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run()
ctx.synchronize()
d = np.array([ctx.dump(b, 7, 64) for b in range(0, B, max(1, B // 48))])
print(os.path.basename(_lib.LAB_LIB_PATH))
names = ["head requests + diag read", "tile Cholesky", "W transposition", "TRSM", "L stores + row fetch", "corner + rows 1-7 + LDS run 1",
         "wait for memory", "deferred back substitution", "row 8 + LDS run 2"]
nT = 94
tot = 0.0
for e, n in enumerate(names):
    v = d[:, 40 + e].mean() / nT
    tot += v
    print(f"  {n:34s} {v:8.0f} cycles per step")
print(f"  {'sum':34s} {tot:8.0f} cycles per step; prologue {d[:, 5].mean() / 1e3:.1f} k, loop {d[:, 6].mean() / 1e3:.1f} k cycles")
if d[:, 4].mean() > 0:
    us = d[:, 4].mean() / 100.0     # wall_clock64 ticks at 100 MHz
    print(f"  the factor loop took {us:.1f} us of wall time: shader clock {d[:, 6].mean() / us / 1e3:.3f} GHz while it ran")
ctx.close()
