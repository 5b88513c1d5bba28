"""Where the wavefronts of the dataflow factorisation wait (lab build with EXTRA=-DSFT_STEP_TRACE): kcycles per wave over the first
factorisation of problem 0, columns: buffers free, W, X1, X2..i, border, store flags, chol, whole loop.
usage: tools/wait_trace.py [B] [waves]"""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from defslam_amd import synth, sft, _lib
import os
_ab = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", os.environ.get("AB_LIB", "lab_trace") + ".so")
if os.path.exists(_ab):
    _lib.LAB_LIB_PATH = _ab
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = sft.Context(0, lab=True)
ctx.set_option("speculate", 1)
ctx.set_option("waves", W)
rows, cols, m = synth.CONFIGS["C2"]
tmpl = synth.make_grid_template(rows, cols)
ctx.template_build(tmpl.xyz0, tmpl.facets)
frames = [sft.frame_from_synth(synth.make_frame(tmpl, m, p)) for p in range(B)]
ctx.batch_upload(frames, synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
ctx.batch_run(); ctx.synchronize()
for b in (0, B // 2):
    t = ctx.step_trace(b)
    print(f"B={B} waves={W} problem {b}: kcycles  free      W     X1   X2..i border  store   chol   loop")
    for w in range(8):
        if t[w, 7] > 0:
            print(f"  wave {w}: " + " ".join(f"{(v - 1) / 1e3:6.1f}" for v in t[w]))
