"""Stress: barrier-free factor steps vs the barrier version on many problems, bit for bit, repeated."""
import os, sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import synth, sft
ctx = sft.Context(0, lab=True)   # lab build: timers, test hooks, A/B switches (include/defslam_hip_debug.h)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
waves = sys.argv[4] if len(sys.argv) > 4 else "8"
rows, cols, m = synth.CONFIGS[cfg]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
def run(df):
    ctx.set_option("waves", int(waves))
    ctx.set_option("dataflow", int(df))
    frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
    inl = sft.DefPoseOptimizationBatch(ctx, frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    return np.stack([f.nodes_xyz for f in frames]), np.array([f.trials for f in frames]), np.array(inl)
ref = run("0")
bad = 0
for r in range(reps):
    cur = run("1")
    same = np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2])
    nd = np.where(np.abs(cur[0] - ref[0]).reshape(B, -1).max(1) > 0)[0]
    print(f"rep {r}: dataflow == barrier bit for bit: {same}; max |diff| {np.abs(cur[0] - ref[0]).max():.3e}; differing problems: {nd[:12]} ({len(nd)})")
    bad += not same
sys.exit(1 if bad else 0)
