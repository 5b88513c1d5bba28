#!/usr/bin/env python
"""Where a connected-mesh solve (two contexts on one GPU) spends its wall clock: per-call wall time, and -- under
`rocprofv3 --kernel-trace` -- the kernel time by kernel.  usage (GPU box): python tools/diag/connected_timing.py [C2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defslam_amd import sft, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
if "--torch" in sys.argv:   # the process bench.py measures in: torch imported, its CUDA context alive
    import torch
    torch.cuda.set_device(0)
    torch.zeros(4, device="cuda")
if "--big" in sys.argv:     # ... and a third context holding a large batch arena
    big = sft.Context(0)
    t2, _ = synth.make_problem("C2", 0)
    big.template_build(t2.xyz0, t2.facets)
    big.batch_upload([sft.frame_from_synth(synth.make_frame(t2, 1000, p)) for p in range(2048)], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    big.batch_run()
    big.synchronize()
tmpl, fr = synth.make_problem(cfg, 0)
regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
ctxs = [sft.Context(0), sft.Context(0)]
for c in ctxs:
    c.template_build(tmpl.xyz0, tmpl.facets)
for i in range(5):
    fs = [sft.frame_from_synth(fr), sft.frame_from_synth(fr)]
    t0 = time.perf_counter()
    sft.ConnectedPoseOptimizationGroup(ctxs[0], ctxs[1], fs, *regs)
    dt = time.perf_counter() - t0
    print(f"{cfg} connected call {i}: {1e3 * dt:.2f} ms, {fs[0].iters} iterations, {fs[0].trials} trials -> {1e3 * dt / fs[0].trials:.3f} ms per trial")
f = sft.frame_from_synth(fr)
call = ctxs[0].prepare_solve(f, *regs, 1, 50)
call()
t0 = time.perf_counter()
call()
print(f"{cfg} undivided dsh_sft_solve: {1e3 * (time.perf_counter() - t0):.2f} ms")
for c in ctxs:
    c.close()
