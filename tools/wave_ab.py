#!/usr/bin/env python
"""A/B of the one-wavefront factorisation (sft_wave.h) against the four-wavefront solver on the GPU: agreement of the solutions and the
device time per launch.  python tools/wave_ab.py [config] [B] [launches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defslam_amd import sft, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows, cols, m = synth.CONFIGS[cfg]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run()
ctx.synchronize()
for rel in (1.0, 100.0):
    xr, xn, ok, ms = ctx.wave_check(rel, launches)
    err = max(float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-300)) for a, b in zip(xr, xn))
    nan = sum(int(not np.isfinite(b).all()) for b in xn)
    print(f"{cfg} B={B} rel={rel}: max rel diff {err:.3e}, non-finite solutions {nan}, ok flags ref/new {ok[:, 0].min()}/{ok[:, 1].min()}, "
          f"ms per launch: four-wavefront {ms[0]:.3f}, one-wavefront {ms[1]:.3f} (x{ms[0] / ms[1]:.2f})", flush=True)
    d = np.array([ctx.dump(b, 7, 8) for b in range(0, B, max(1, B // 16))])
    print(f"   one-wavefront sections, kcycles (mean of {len(d)} problems): prologue {d[:, 5].mean() / 1e3:.1f}, factor loop {d[:, 6].mean() / 1e3:.1f}, "
          f"corner + back substitution {d[:, 7].mean() / 1e3:.1f}", flush=True)
ctx.close()
