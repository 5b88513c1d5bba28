#!/usr/bin/env python
"""Where a SOLVE launch of the latency mode spends its time (A/B build with -DSFT_SOLVE_TRACE: tools/ab_build.sh strace "-DSFT_SOLVE_TRACE"):
100 MHz stamps of lane 0's two workgroups in the LAST round: sum of the parts' Schur contributions, factorisation of the reduced problem,
its back substitution, the part's back substitution.   usage (GPU box): python tools/solve_trace.py --lib tools/_ab/strace.so [C5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import _lib, sft, synth  # noqa: E402

args = sys.argv[1:]
if "--lib" in args:
    i = args.index("--lib")
    _lib.LAB_LIB_PATH = os.path.abspath(args[i + 1])
    del args[i:i + 2]
ctx = sft.Context(0, lab=True)
for cfg in args or ["C5"]:
    rows, cols, m = synth.CONFIGS[cfg]
    tmpl = synth.make_grid_template(rows, cols)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    f = sft.frame_from_synth(synth.make_frame(tmpl, m, 0))
    ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ms = ctx.lab_run_timed(3) / 3
    d = ctx.dump(0, 7, 128)
    print(f"{cfg}: {ms:.3f} ms per frame")
    for g in (0, 1):
        v = d[96 + 8 * g:96 + 8 * g + 4] * 1e-2
        print(f"  part {g}: shader clock {d[96 + 8 * g + 4] * 100:.0f} MHz; sum {v[0]:.1f} us, reduced factorisation {v[1]:.1f} us, reduced back substitution {v[2]:.1f} us, part back substitution {v[3]:.1f} us")
ctx.close()
