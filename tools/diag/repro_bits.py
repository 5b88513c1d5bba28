#!/usr/bin/env python
"""Run-to-run reproducibility of the throughput shape: B C2 problems, N runs of the same uploaded batch, every result compared bit for bit with
the first run's -- product library, lab library with the product's tail rule, lab library with rounds to the end (tail = 0: the one-wavefront
factorisation alone).  A timing-dependent hazard in hand-scheduled code shows up here as a handful of differing problems.
  usage (GPU box): python tools/diag/repro_bits.py [B = 1024] [N = 5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
syn = [synth.make_frame(tmpl, m, p) for p in range(B)]
for name, lab, tail in (("product", False, None), ("lab, tail rule of the product", True, -1), ("lab, rounds to the end", True, 0)):
    ctx = sft.Context(0, lab=lab)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    if tail is not None:
        ctx.set_option("tail", tail)
    frames = [sft.frame_from_synth(fr) for fr in syn]
    ctx.batch_upload(frames, *regs, 1, 50)
    first = None
    for r in range(N):
        ctx.batch_run()
        ctx.batch_download()
        snap = [(f.iters, f.trials, f.nodes_xyz.copy(), f.pose7.copy()) for f in frames]
        if first is None:
            first = snap
            continue
        bad = [p for p in range(B) if snap[p][:2] != first[p][:2] or not np.array_equal(snap[p][2], first[p][2]) or not np.array_equal(snap[p][3], first[p][3])]
        worst = max([float(np.abs(snap[p][2] - first[p][2]).max()) for p in bad], default=0.0)
        print(f"{name:32s} run {r}: {len(bad)} of {B} problems differ from run 0 (ids {bad[:8]}, max vertex difference {worst:.2e})", flush=True)
    if lab:
        ctx.set_option("tail", -1)
    ctx.close()
