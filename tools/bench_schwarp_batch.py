"""The batched Schwarp fit alone (dsh_schwarp_fit_batch, B keyframe pairs per call): wall time per call and fits/s; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.  usage: python tools/bench_schwarp_batch.py [B] [matches] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defslam_amd import _lib, nrsfm, sft, synth

_ab = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", os.environ.get("AB_LIB", "") + ".so")
if os.environ.get("AB_LIB") and os.path.exists(_ab):
    _lib.LIB_PATH = _ab   # an A/B build (tools/ab_build.sh)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = sft.Context(0)
probs = []
for b in range(B):
    q = synth.make_warp_problem(n_matches=M, seed=100 + b)
    probs.append(dict(bbs=nrsfm.Bbs(*q["bbs"]), kp1=q["kp1"], kp2=q["kp2"], invsig=q["invsig"], fx_slot=q["fy"], fy_slot=q["fx"], lam=1e-2, fx=q["fx"], fy=q["fy"], x0=q["x0"]))
nrsfm.calculateSchwarpsBatch(ctx, probs, 3)
# the C call alone (the rest of the wall time is the Python mirror: argument conversion, result arrays)
_real = ctx._L.dsh_schwarp_fit_batch
tc = []
class _Timed:
    def __call__(self, *a):
        t0 = time.perf_counter()
        r = _real(*a)
        tc.append(time.perf_counter() - t0)
        return r
class _L:
    def __getattr__(self, k):
        return _Timed() if k == "dsh_schwarp_fit_batch" else getattr(ctx.__dict__["_Lreal"], k)
ctx.__dict__["_Lreal"] = ctx._L
ctx._L = _L()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    nrsfm.calculateSchwarpsBatch(ctx, probs, 3)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts)
print(f"  C call dsh_schwarp_fit_batch alone: median {1e3 * np.median(tc):.3f} ms")
print(f"B={B} matches={M}: median {1e3 * np.median(ts):.3f} ms/call (min {1e3 * ts.min():.3f}), {B / np.median(ts):.0f} fits/s")
