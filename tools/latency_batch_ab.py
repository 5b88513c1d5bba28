#!/usr/bin/env python
"""Latency mode with several problems per launch (lab build): kernel time per launch for the settings of the "split" option.
usage: python tools/latency_batch_ab.py CFG B [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

cfg = sys.argv[1]
ctx = sft.Context(0, lab=True)
for B in [int(a) for a in sys.argv[2:]]:
    frames = []
    for pid in range(B):
        tmpl, fr = synth.make_problem(cfg, pid)
        if not frames:
            ctx.template_build(tmpl.xyz0, tmpl.facets)
        frames.append(sft.frame_from_synth(fr))
    for split in (0, 2):
        ctx.set_option("split", split)
        ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
        ctx.batch_run()
        ctx.synchronize()
        ms = ctx.lab_run_timed(3) / 3
        it, tr = ctx.batch_counts()
        info = ctx.solver_info(0)
        print(f"{cfg} B={B} split={split}: {ms:.3f} ms per launch, {it / ms * 1e3:.0f} it/s, tile_mode {info['tile_mode']}, two-sided {info['split']}, lanes {info['lanes']}")
ctx.set_option("split", 2)
ctx.close()
