#!/usr/bin/env python
"""Debug aid for sft_wave.h: first block columns of L of both solvers, tile by tile."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from defslam_amd import sft, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "smoke"
B = 64
rows, cols, m = synth.CONFIGS[cfg]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run()
ctx.synchronize()
_, counts = ctx.problem_info(0)
Dn = int(counts[5]) - 6
Dnp = ((Dn + 31) // 32) * 32
nT = Dnp // 16
print("Dn", Dn, "nT", nT, "kd", counts[6])


def acc_to_mat(t):   # accumulator order: lane l = (g, c), register q -> [g + 4q][c]; stored lane-major, 4 registers per lane
    M = np.zeros((16, 16))
    for l in range(64):
        g, c = l >> 4, l & 15
        for q in range(4):
            M[g + 4 * q, c] = t[4 * l + q]
    return M


ctx.wave_check(1.0, 1, only=1)
Lr = ctx.dump(0, 0, nT * 9 * 256).reshape(nT, 9, 256)
Wr = ctx.dump(0, 1, nT * 256).reshape(nT, 256)
Br = ctx.dump(0, 2, 8 * Dnp).reshape(8, Dnp)
xr = ctx.dump(0, 5, Dnp + 6)
ctx.wave_check(1.0, 1, only=2)
Ln = ctx.dump(0, 0, nT * 9 * 256).reshape(nT, 9, 256)
Wn = ctx.dump(0, 1, nT * 256).reshape(nT, 256)
xn = ctx.dump(0, 5, Dnp + 6)
dbg = ctx.dump(0, 7, 64)
print("dbg", dbg[:8])
Hcn = ctx.dump(0, 6, 49).reshape(7, 7)
lam = dbg[1]
Cref = np.tril(Hcn) + lam * np.diag([1.0] * 6 + [0.0])
for k in range(nT):
    Xb = Br[:7, 16 * k:16 * k + 16]
    Cref -= np.tril(Xb @ Xb.T)
Cn = dbg[8:57].reshape(7, 7)
print("corner: reference (host, from the reference border rows)\n", np.array_str(np.tril(Cref), precision=4, max_line_width=200))
print("corner: one-wavefront\n", np.array_str(np.tril(Cn), precision=4, max_line_width=200))
for k in range(min(nT, int(sys.argv[2]) if len(sys.argv) > 2 else 12)):
    w_r, w_n = acc_to_mat(Wr[k]), acc_to_mat(Wn[k])
    line = f"col {k}: W diff {np.abs(w_r - w_n).max():.2e} (|W| {np.abs(w_r).max():.2e})"
    for i in range(1, 9):
        if k + i >= nT:
            break
        X = acc_to_mat(Lr[k, i])            # reference: X(k+i, k)
        Y = acc_to_mat(Ln[k, i])            # one-wavefront: X^T
        line += f" | i={i}: {np.abs(X - Y.T).max():.1e}/{np.abs(X).max():.1e}"
    Yb = acc_to_mat(Ln[k, 0])               # Xb^T: [b][c] = Xb[c][b]
    Xb = Br[:7, 16 * k:16 * k + 16]
    line += f" | border {np.abs(Xb - Yb.T[:7]).max():.1e}/{np.abs(Xb).max():.1e}"
    print(line)
print("x diff", np.abs(xr - xn).max(), np.abs(xr).max())
ctx.close()
