#!/usr/bin/env python
"""A/B of the tail kernel of the batched rounds (lab option "tail"): ms per step, LM totals and the largest difference of the results with the
last problems of a step run to their end by sftb_tail_kernel (eight-wavefront solver) against rounds to the end (one-wavefront solver).
  usage (GPU box): python tools/tail_ab.py [512 2048 ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

args = sys.argv[1:]
if "--lib" in args:      # an A/B build of the lab library (tools/ab_build.sh)
    i = args.index("--lib")
    from defslam_amd import _lib
    _lib.LAB_LIB_PATH = os.path.abspath(args[i + 1])
    del args[i:i + 2]
tails = (0, 1, 2, 3, 4, 6)
if "--tail" in args:     # thresholds to try (default: all)
    i = args.index("--tail")
    tails = tuple(int(v) for v in args[i + 1].split(","))
    del args[i:i + 2]
sizes = [int(a) for a in args] or [512, 2048]
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
for B in sizes:
    syn = [synth.make_frame(tmpl, m, p) for p in range(B)]
    ref = None
    for tail in tails:
        ctx.set_option("tail", tail)
        fs = [sft.frame_from_synth(fr) for fr in syn]
        ctx.batch_upload(fs, *regs, 1, 50)
        ctx.batch_run()
        ctx.synchronize()
        ctx.batch_run()
        ctx.synchronize()
        ms = ctx.lab_run_timed(3) / 3
        it, tr = ctx.batch_counts()
        ph, nr = ctx.rounds_timed()
        ctx.batch_download()
        res = [(f.nodes_xyz.copy(), f.iters, f.trials, f.mvbOutlier.copy()) for f in fs]
        msg = ""
        if ref is None:
            ref = res
        else:
            dv = max(float(np.abs(a[0] - b[0]).max()) for a, b in zip(ref, res))
            same = sum(a[1:3] == b[1:3] and np.array_equal(a[3], b[3]) for a, b in zip(ref, res))
            msg = f"; max vertex difference to tail=0 {dv:.2e}, {same} of {B} problems with the same iterations / trials / outliers"
        print(f"C2 x{B} tail={tail}: {ms:.2f} ms per step, {it / ms * 1e3:.0f} it/s, {it} iterations, {tr} trials, {nr} rounds, phases {ph}{msg}", flush=True)
ctx.set_option("tail", -1)
ctx.close()
