#!/usr/bin/env python
"""Per-phase device time of one step of the throughput shape (dsh_lab_sft_rounds_timed) for a lab variant built by tools/ab_build.sh:
   python tools/phases_ab.py [variant ...]     (variant = name under tools/_ab/, "" = the in-tree lab library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--one"):
    for v in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, WV_VARIANT=v))
    sys.exit(0)
sys.path.insert(0, ROOT)
from defslam_amd import _lib  # noqa: E402

v = os.environ.get("WV_VARIANT", "")
if v and v != "intree":
    _lib.LAB_LIB_PATH = os.path.join(ROOT, "tools", "_ab", v + ".so")
from defslam_amd import sft, synth  # noqa: E402

B = 16384
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx = sft.Context(0, lab=True)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run()
ctx.synchronize()
ph, r = ctx.rounds_timed()
it, tr = ctx.batch_counts()
ph = {k: v for k, v in ph.items() if not k.endswith("_in_rounds")}
asm_ms = ctx.batch_assemble_timed(5) / 5
tot = sum(ph.values())
print(f"{v or 'intree':12s} isolated assembly pass {asm_ms:6.3f} ms", flush=True)
print(f"{v or 'intree':12s} lin {ph['lin']:7.2f}  factor {ph['factor']:7.2f}  trial {ph['trial']:6.2f}  sum {tot:7.2f} ms  ({it / (tot * 1e-3):.0f} it/s, {r} rounds)", flush=True)
ctx.close()
