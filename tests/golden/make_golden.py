"""Generates tests/golden/sft_*.npz with the NumPy restatement (oracle/sft_oracle_np.py).

Run in the build container:  python tests/golden/make_golden.py
The fixtures hold inputs and expected outputs only (data, no reference source).  They pin the C
oracle and the HIP path to the independent NumPy restatement; the reference itself ships no vectors
for this path and cannot be built here, so parity stays "unpinned" in the
sense of the task statement.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import sft_oracle_np as onp  # noqa: E402
from defslam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, rows, cols, m, pid, regs, keep_nodes=None, layers=1):
    tmpl = synth.make_grid_template(rows, cols)
    fr = synth.make_frame(tmpl, m, pid)
    if keep_nodes is not None:  # partial view: only observations whose facet lies in a corner of the mesh
        sel = np.all(np.isin(fr.obs_nodes, keep_nodes), axis=1)
        for k in ["obs_facet", "obs_nodes", "obs_bary", "obs_uv", "obs_invsig2"]:
            setattr(fr, k, getattr(fr, k)[sel])
    tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
    r = onp.solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz, *regs, layers=layers)
    np.savez_compressed(os.path.join(HERE, f"sft_{name}.npz"), xyz0=tmpl.xyz0, facets=tmpl.facets, Tcw=fr.Tcw, K=fr.K, n_frame=fr.n_frame,
                        obs_nodes=fr.obs_nodes, obs_bary=fr.obs_bary, obs_uv=fr.obs_uv, obs_invsig2=fr.obs_invsig2, xyz=fr.xyz,
                        regs=np.asarray(regs), layers=layers, out_pose7=r["pose7"], out_xyz=r["xyz"], out_outlier=r["outlier"],
                        out_rep_error=r["rep_error"], out_trace=r["trace"], out_inliers=r["ret"], out_chi2_obs=r["chi2_obs"])
    print(name, "iters", r["iters"], "inliers", r["ret"], "M", fr.obs_nodes.shape[0])


if __name__ == "__main__":
    regs = (synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP)
    case("grid10", 10, 10, 300, 0, regs)
    case("grid8x12_notemporal", 12, 8, 250, 3, (25.0, 240.0, 0.0))       # webcam yaml regularisers, RegTemp = 0
    corner = [c + 10 * r for r in range(5) for c in range(5)]
    case("grid10_partial", 10, 10, 600, 5, regs, keep_nodes=corner)       # fixed nodes outside the viewed 1-ring
