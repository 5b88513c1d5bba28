"""Shape-from-Normals timing: device path vs the CPU oracle (Householder QR, 1 core)."""
import sys, time, json
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import sft, nrsfm, synth
import oracle
ctx = sft.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = synth.make_sfn_scene(n, seed=4)
b = nrsfm.Bbs(*sc["bbs"])
f = lambda: nrsfm.ShapeFromNormals(ctx, b, sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
f()
t = time.perf_counter()
for _ in range(10):
    f()
dt = (time.perf_counter() - t) / 10
g = lambda: oracle.sfn_estimate(sc["bbs"], sc["u"], sc["v"], sc["normals"], 1e-3, sc["mean_depth"], sc["u_all"], sc["v_all"])
g()
t = time.perf_counter()
g()
dc = time.perf_counter() - t
print(json.dumps({"metric": "Shape-from-Normals estimates/s (13x15 grid)", "value": 1 / dt, "unit": "estimates/s", "key_points": n, "normals": int(sc["u"].shape[0]),
                  "ms_per_call": 1e3 * dt, "cpu_baseline": {"value": 1 / dc, "unit": "estimates/s", "cores": 1, "kind": "port",
                                                            "sample": "same scene, oracle/sfn_oracle.c (unpivoted Householder QR of the stacked system)"}}))
