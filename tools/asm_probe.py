#!/usr/bin/env python
"""Isolated-assembly timing (lab build): ms per pass of 16384 C2 problems and the algorithmic HBM fraction."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from defslam_amd import _lib, sft, synth  # noqa: E402
if len(sys.argv) > 2:
    _lib.LAB_LIB_PATH = sys.argv[2]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
lab = sft.Context(0, lab=True)
lab.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
lab.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
lab.batch_run()
lab.synchronize()
alg = sum(lab.problem_info(b)[0] for b in range(B))
ms = lab.batch_assemble_timed(12) / 12
print(f"{_lib.LAB_LIB_PATH}: B={B} assembly {ms:.3f} ms per pass, {alg / (ms * 1e-3) / 1e9:.0f} GB/s algorithmic = {alg / (ms * 1e-3) / 8e12:.3f} of 8 TB/s")
lab.close()
