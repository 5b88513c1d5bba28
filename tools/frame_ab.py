#!/usr/bin/env python
"""One latency-mode frame (library defaults) under several lab builds, each in its own process on the same box:
   python tools/frame_ab.py [--cfg C5,C2] variant ...     (variant = name under tools/_ab/ built by tools/ab_build.sh, "intree" = the in-tree lab library)
Prints kernel time per frame (5 timed runs after one warm-up) and a checksum of the result (bit-identical builds print the same)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--one" not in sys.argv:
    args = sys.argv[1:]
    cfg = "C5,C2"
    if "--cfg" in args:
        i = args.index("--cfg")
        cfg = args[i + 1]
        del args[i:i + 2]
    for rep in range(2):
        for v in args or ["intree"]:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, WV_VARIANT=v, WV_CFG=cfg))
    sys.exit(0)
sys.path.insert(0, ROOT)
from defslam_amd import _lib  # noqa: E402

v = os.environ.get("WV_VARIANT", "intree")
if v != "intree":
    _lib.LAB_LIB_PATH = os.path.join(ROOT, "tools", "_ab", v + ".so")
from defslam_amd import sft, synth  # noqa: E402

ctx = sft.Context(0, lab=True)
for kv in filter(None, os.environ.get("WV_OPTS", "").split(",")):   # lab options: WV_OPTS=helpers_wbt=8,helpers=3
    k, val = kv.split("=")
    ctx.set_option(k, int(val))
for cfg in os.environ.get("WV_CFG", "C5").split(","):
    rows, cols, m = synth.CONFIGS[cfg]
    tmpl = synth.make_grid_template(rows, cols)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    nb = int(os.environ.get("WV_BATCH", "1"))   # problems per step (WV_BATCH=16: the 16-problem line of the bench)
    fs = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(nb)]
    f = fs[0]
    ctx.batch_upload(fs, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ms = ctx.lab_run_timed(5) / 5
    ctx.batch_download()
    h = hashlib.sha1(f.nodes_xyz.tobytes() + f.pose7.tobytes()).hexdigest()[:12]
    info = ctx.solver_info(0)
    print(f"{v:>10} {cfg} x{nb}: {ms:.3f} ms per step, {f.iters} iterations, {f.trials} trials, lanes {info['lanes']}, result {h} {os.environ.get('WV_OPTS', '')}", flush=True)
ctx.close()
