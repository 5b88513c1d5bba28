#!/usr/bin/env python
"""The throughput of the PRODUCT library over the batch size (C2 problems, inputs resident, HIP events on the library's stream): where the
launch shapes hand over to each other -- latency mode with four / two speculative lanes, one persistent workgroup per problem, tail kernel only,
rounds + tail -- and whether the curve is monotone.
  usage (GPU box): python tools/batch_curve.py [B ...]      (default: 1 ... 4096 with the hand-over points bracketed)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 13, 16, 32, 48, 64, 65, 96, 128, 129, 192, 256, 257, 384, 448, 511, 512, 640, 768, 1024, 1536, 2048, 4096]
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
ctx = sft.Context(0)
ctx.template_build(tmpl.xyz0, tmpl.facets)
syn = [synth.make_frame(tmpl, m, p) for p in range(max(sizes))]
prev = None
for B in sizes:
    fs = [sft.frame_from_synth(fr) for fr in syn[:B]]
    ctx.batch_upload(fs, *regs, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ctx.batch_run()
    ctx.synchronize()
    n = 5 if B <= 512 else 3
    ms = ctx.batch_run_timed(n) / n
    it, tr = ctx.batch_counts()
    _, cc = ctx.problem_info(0)
    rate = it / ms * 1e3
    flag = "" if prev is None or rate >= prev else "   <-- slower than the next-smaller batch"
    print(f"C2 x{B:5d}: {ms:8.3f} ms per step, {rate:9.0f} it/s, {ms / B * 1e3:8.1f} us per problem, {it} iterations, {tr} trials, wavefronts per problem {int(cc[7])}{flag}", flush=True)
    prev = rate
ctx.close()
