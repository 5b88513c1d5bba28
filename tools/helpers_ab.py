#!/usr/bin/env python
"""Latency-mode A/B of the helper workgroups of the two-sided factorisation (lab build): kernel time per frame for 0 / automatic / 1 / 2 / 3
helpers per part, and the results of every setting against the setting without helpers -- they must be bit-identical (sft_wide.h: the owner
forms a far sum itself, in the same order, whenever a helper's is not there).
  usage (GPU box): python tools/helpers_ab.py [C5 C2 W16 ...] [--batch 16]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

args = sys.argv[1:]
batch = 1
if "--lib" in args:      # an A/B build of the lab library (tools/ab_build.sh)
    i = args.index("--lib")
    from defslam_amd import _lib
    _lib.LAB_LIB_PATH = os.path.abspath(args[i + 1])
    del args[i:i + 2]
if "--batch" in args:
    i = args.index("--batch")
    batch = int(args[i + 1])
    del args[i:i + 2]
spec = 0
if "--speculate" in args:   # lanes per problem (0: automatic)
    i = args.index("--speculate")
    spec = int(args[i + 1])
    del args[i:i + 2]
ow = (8,)
if "--owner-waves" in args:   # wavefronts of a FACTOR workgroup with helpers: 8 (default), 16 or "8,16"
    i = args.index("--owner-waves")
    ow = tuple(int(v) for v in args[i + 1].split(","))
    del args[i:i + 2]
cfgs = args or ["C5", "C2"]
ctx = sft.Context(0, lab=True)
for cfg in cfgs:
    rows, cols, m = synth.CONFIGS[cfg]
    tmpl = synth.make_grid_template(rows, cols)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    ref = None
    for nh, waves in [(0, 8)] + [(h, w) for w in ow for h in (-1, 1, 2, 3)]:
        ctx.set_option("helpers", nh)
        ctx.set_option("owner_waves", waves)
        ctx.set_option("speculate", spec)
        fs = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(batch)]
        ctx.batch_upload(fs, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
        ctx.batch_run()
        ctx.synchronize()
        ms = ctx.lab_run_timed(5) / 5
        it, tr = ctx.batch_counts()
        ctx.batch_download()
        res = [(f.nodes_xyz.copy(), f.pose7.copy(), f.iters, f.trials) for f in fs]
        same = "-"
        if ref is None:
            ref = res
        else:
            same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:] for a, b in zip(ref, res))
        info = ctx.solver_info(0)
        print(f"{cfg} x{batch} helpers={nh} waves={waves}: {ms:.3f} ms per step, {it} iterations, {tr} trials, two-sided {info['split']}, lanes {info['lanes']}, bit-identical to helpers=0: {same}", flush=True)
        if nh != 0 and info['split']:
            for g in (0, 1):   # statistics of lane 0's part g over the 6 runs since the upload (lab build): owner / helper counters, 100 MHz ticks
                try:
                    w = ctx.dump(0, 8 + g, 8).view(np.int32)
                    if w[5]:
                        print("      helper segments, us per column: stage + barrier %.2f, pass 1 (+ next requests) %.2f, pass 2 %.2f, drain + barrier %.2f" % tuple(w[9 + i] * 1e-2 / w[5] for i in range(4)))
                except Exception as e:  # noqa: BLE001
                    print("   no statistics:", e)
                    break
                print(f"   part {g}: owner took {w[1]} columns from helpers, formed {w[2]} itself, {w[3]} polls, {w[4] * 1e-2:.0f} us looking | helpers formed {w[5]} columns, skipped {w[6]}, waited {w[7] * 1e-2:.0f} us, worked {w[8] * 1e-2:.0f} us", flush=True)
ctx.set_option("helpers", -1)
ctx.set_option("owner_waves", 8)
ctx.close()
