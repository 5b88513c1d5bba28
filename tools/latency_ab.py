#!/usr/bin/env python
"""Latency-mode A/B on one problem (lab build): kernel time per frame for the settings of the "split" option."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

cfgs = sys.argv[1:] or ["C2", "C5", "smoke", "W16"]
ctx = sft.Context(0, lab=True)
for cfg in cfgs:
    tmpl, fr = synth.make_problem(cfg, 0)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    for split in (0, 1, 2):
        ctx.set_option("split", split)
        f = sft.frame_from_synth(fr)
        ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
        ctx.batch_run()
        ctx.synchronize()
        ms = ctx.lab_run_timed(5) / 5
        it, tr = ctx.batch_counts()
        info = ctx.solver_info(0)
        print(f"{cfg} split={split}: {ms:.3f} ms per frame, {it} iterations, {tr} trials, {it / ms * 1e3:.0f} it/s, tile_mode {info['tile_mode']}, two-sided {info['split']}, lanes {info['lanes']}")
ctx.set_option("split", 2)
ctx.close()
