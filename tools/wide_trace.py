#!/usr/bin/env python
"""Where a block column of the left-looking wide-band factorisation spends its time, per role (A/B build with -DSFT_WIDE_TRACE:
tools/ab_build.sh wtrace "-DSFT_WIDE_TRACE"): 100 MHz stamps summed over the columns of a part's last factorisation, printed as microseconds
per column.  Segments: 0 staging loads + diagonal tile / Cholesky (role 0), 1 rows (tile loads + products), 2 look-ahead (role 1), 3 border
(role 7) + LDS staging, 4 wait for W, 5 TRSM + stores, 6 look for the helper + store drain, 7 last barrier.
  usage (GPU box): python tools/wide_trace.py --lib tools/_ab/wtrace.so [C5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import _lib, sft, synth  # noqa: E402

args = sys.argv[1:]
if "--lib" in args:
    i = args.index("--lib")
    _lib.LAB_LIB_PATH = os.path.abspath(args[i + 1])
    del args[i:i + 2]
cfg = args[0] if args else "C5"
rows, cols, m = synth.CONFIGS[cfg]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
for nh, waves in ((0, 8), (-1, 8), (-1, 16)):
    ctx.set_option("helpers", nh)
    ctx.set_option("owner_waves", waves)
    f = sft.frame_from_synth(synth.make_frame(tmpl, m, 0))
    ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ms = ctx.lab_run_timed(3) / 3
    info = ctx.solver_info(0)
    d = ctx.dump(0, 7, 128).reshape(2, 8, 8)
    print(f"{cfg} helpers={nh} waves={waves}: {ms:.3f} ms per frame; two-sided {info['split']}")
    if waves == 16 and nh != 0:   # part 0 only, 16 roles (the last role's segments 6, 7 hold the clocks)
        raw = ctx.dump(0, 7, 128)
        ncol = 204
        print(f"  part 0: us per column by role and segment; shader clock {raw[126] / raw[127] * 100:.0f} MHz ({raw[127] * 1e-2:.0f} us)")
        for r in range(16):
            us = raw[8 * r:8 * r + (6 if r == 15 else 8)] * 1e-2 / ncol
            print("    role %2d: " % r + " ".join(f"{v:6.2f}" for v in us) + f"   {us.sum():6.2f}")
        continue
    for g in (0, 1):
        ncol = (info["c0"] if g == 0 else info["n1p"]) // 16 + info["s"] // 16 if "c0" in info else 204
        print(f"  part {g} ({ncol} columns): us per column by role (rows) and segment (columns 0..7), last column: sum")
        raw = d[g].ravel()
        if raw[63] > 0:
            print(f"    shader clock over the factorisation: {raw[62] / raw[63] * 100:.0f} MHz ({raw[63] * 1e-2:.0f} us)")
        for r in range(8):
            us = d[g, r] * 1e-2 / ncol
            print("    role %d: " % r + " ".join(f"{v:6.2f}" for v in us) + f"   {us.sum():6.2f}")
ctx.set_option("helpers", -1)
ctx.close()
