"""How many problems are still running in round r of the throughput shape (a round = one damping trial of every running problem), and what
a speculative FACTOR (K dampings of a rejection run side by side once K * active <= waves) would save.  GPU: python tools/diag/rounds_histogram.py [B]"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from defslam_amd import sft, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.batch_download()
T = np.array([f.trials for f in frames]); I = np.array([f.iters for f in frames])
print("problems", B, "iterations", I.sum(), "trials", T.sum(), "max trials", T.max())
R = T.max()
active = np.array([(T >= r).sum() for r in range(1, R + 1)])
waves = 1024
sets = np.ceil(active / waves)
print("rounds", R, "factor wave-sets", int(sets.sum()), "ideal", T.sum() / waves)
lin = np.zeros(R + 1, int)
for f in frames:
    r = 1
    for q in f.trace[:f.iters, 2].astype(int):
        lin[r - 1] += 1
        r += q
print("round: problems in FACTOR / TRIAL, factor wave-sets, problems linearised")
for r in range(R):
    print(r + 1, active[r], int(sets[r]), lin[r])
# speculation: per problem the trial sequence per iteration (trace[:, 2]); in rounds where K * active <= waves a rejection run of n trials takes ceil(n / K) rounds
def rounds_with_spec(Kmax):
    # simulate: every problem advances through its list of runs; global rounds; K chosen per round from the active count
    runs = [list(f.trace[:f.iters, 2].astype(int)) for f in frames]
    pos = [0] * B; left = [r[0] if r else 0 for r in runs]
    act = set(p for p in range(B) if runs[p])
    nr = 0; cost = 0.0
    while act:
        nr += 1
        K = 1
        while K * 2 <= Kmax and K * 2 * len(act) <= waves: K *= 2
        cost += np.ceil(len(act) * K / waves)
        done = []
        for p in act:
            left[p] -= min(K, left[p])
            if left[p] == 0:
                pos[p] += 1
                if pos[p] < len(runs[p]): left[p] = runs[p][pos[p]]
                else: done.append(p)
        for p in done: act.discard(p)
    return nr, cost
for K in (1, 2, 4, 8):
    print("Kmax", K, "rounds, factor wave-sets:", rounds_with_spec(K))
ctx.close()
