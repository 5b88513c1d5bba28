"""Generates tests/golden/sft_C5_p<id>.npz: full-size stress problems (BASELINE.json configs[4]: 2000-node template
40x50, 4000 matches, D = 6006) solved by the C oracle (oracle/sft_oracle.c, dense (6+3n)^2 system, blocked unpivoted
LDLT = ldlt_mode 1; the Eigen-style pivoted one is O(D^3) unblocked and takes hours at this size).

Run in the build container (about 10-20 minutes of CPU per problem):

    python tests/golden/make_golden_c5.py [problem ids ...]        # default: 0 1

The inputs are NOT stored: they are regenerated from the seeds by defslam_amd/synth.py (`make_problem("C5", id)`); the
fixture keeps a checksum of every input array so that a change of the generator is detected instead of silently
comparing different problems.  Outputs only: final pose, vertices, per-observation chi2, outlier flags, the LM trace.
The reference itself ships no vectors for this path and cannot be built here: parity stays unpinned.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from defslam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
INPUT_KEYS = ["Tcw", "K", "obs_nodes", "obs_bary", "obs_uv", "obs_invsig2", "xyz"]


def input_digest(tmpl, fr) -> str:
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(tmpl.xyz0, np.float64).tobytes())
    h.update(np.ascontiguousarray(tmpl.facets, np.int32).tobytes())
    for k in INPUT_KEYS:
        a = getattr(fr, k)
        h.update(np.ascontiguousarray(a, np.float32 if k == "Tcw" else (np.int32 if k == "obs_nodes" else np.float64)).tobytes())
    h.update(str(int(fr.n_frame)).encode())
    return h.hexdigest()


def case(pid: int):
    tmpl, fr = synth.make_problem("C5", pid)
    tc = oracle.template_build(tmpl.xyz0, tmpl.facets)
    t0 = time.perf_counter()
    r = oracle.sft_solve(tc, fr.Tcw, fr.K, fr.n_frame, fr.obs_nodes, fr.obs_bary, fr.obs_uv, fr.obs_invsig2, fr.xyz,
                         synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, layers=1, max_iters=50, ldlt_mode=1)
    dt = time.perf_counter() - t0
    np.savez_compressed(os.path.join(HERE, f"sft_C5_p{pid}.npz"), problem_id=pid, input_sha256=input_digest(tmpl, fr),
                        regs=np.asarray([synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP]), out_pose7=r.pose7, out_Tcw=r.Tcw, out_xyz=r.xyz,
                        out_outlier=r.outlier, out_rep_error=r.rep_error, out_trace=r.trace, out_inliers=r.ret, out_chi2_obs=r.chi2_obs,
                        out_iters=r.iters, out_trials=r.trials, out_dims=r.dims, oracle_seconds=dt)
    print(f"C5 problem {pid}: {r.iters} LM iterations, {r.trials} trials, inliers {r.ret}, D={int(r.dims[0])}, {dt:.0f} s", flush=True)


if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[1:]] or [0, 1]
    for pid in ids:
        case(pid)
