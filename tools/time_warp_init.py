"""Wall clock of dsh_warp_initialize (host buffers in/out), single calls: tools/time_warp_init.py [P]"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import nrsfm, synth, sft
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = sft.Context(0)
pr = synth.make_warp_problem(P, 3)
bbs = nrsfm.Bbs(*pr["bbs"])
for _ in range(3):
    nrsfm.WarpInitialize(ctx, bbs, pr["kp1"], pr["kp2"], 1e-2)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    ok, x = nrsfm.WarpInitialize(ctx, bbs, pr["kp1"], pr["kp2"], 1e-2)
print(f"dsh_warp_initialize, {P} matches: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (ok={ok})")
