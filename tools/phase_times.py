import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import synth, sft, _lib
import os
_ab = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ab', os.environ.get('AB_LIB', 'lab_timers') + '.so')
if os.path.exists(_ab):
    _lib.LAB_LIB_PATH = _ab   # an A/B lab build (tools/ab_build.sh NAME "-DSFT_PHASE_TIMERS ..."), chosen with AB_LIB=NAME
    print("lab library:", _ab)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
WAVES = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
ctx.set_option("waves", WAVES)
if len(sys.argv) > 4:
    ctx.set_option("speculate", int(sys.argv[4]))   # 1 = the one-workgroup kernel also for small batches
rows, cols, m = synth.CONFIGS[cfg]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
ms = ctx.batch_run_timed(3) / 3
it, tr = ctx.batch_counts()
try:
    ph = ctx.phase_ms(0)
except sft.DshError:
    ph = {}   # built without EXTRA=-DSFT_PHASE_TIMERS
print(f"{cfg} B={B}: {ms:.2f} ms/launch, iters {it} trials {tr}, per-trial {ms/ (tr/B):.3f} ms; single-problem it/s {it/B/(ms*1e-3):.0f}")
print(" phases of problem 0 (ms):", {k: round(v, 2) for k, v in ph.items()}, "sum", round(sum(ph.values()), 2))
