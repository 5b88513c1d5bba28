#!/usr/bin/env python
"""One problem in latency mode under `rocprofv3 --kernel-trace`: mean duration of the launches of sft_spec_kernel by phase.
The phases share one kernel name; they are told apart by their position in the launch sequence (INIT, then LIN [FACTOR] TRIAL ...)
and by the grid (FACTOR launches two workgroups per lane).
  usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d gpurun_out/spec -- python tools/spec_phase_trace.py run C5
                   python tools/spec_phase_trace.py parse gpurun_out/spec"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(cfg):
    from defslam_amd import sft, synth
    tmpl, fr = synth.make_problem(cfg, 0)
    ctx = sft.Context(0)
    ctx.template_build(tmpl.xyz0, tmpl.facets)
    f = sft.frame_from_synth(fr)
    ctx.batch_upload([f], synth.REG_LAP, synth.REG_INEX, synth.REG_TEMP, 1, 50)
    ctx.batch_run()
    ctx.synchronize()
    ctx.batch_run()
    ctx.synchronize()
    it, tr = ctx.batch_counts()
    print(f"{cfg}: {it} iterations, {tr} trials")
    ctx.close()


def parse(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                if "sft_spec_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
    rows.sort()
    if not rows:
        print("no sft_spec_kernel launches found")
        return
    grids = sorted(set(g for _, _, g in rows))
    print("grid sizes:", grids, "launches:", len(rows))
    by = {}
    for s, e, g in rows:
        by.setdefault(g, []).append((e - s) * 1e-3)
    for g, v in sorted(by.items()):
        live = [x for x in v if x > 20.0]
        print(f"grid {g}: {len(v)} launches, {len(live)} with work: mean {sum(live) / max(len(live), 1):.1f} us, max {max(v):.1f} us, total {sum(v) * 1e-3:.2f} ms")
    # the sequence of the second run: durations in launch order
    half = rows[len(rows) // 2:]
    t0 = half[0][0]
    print("second run, launch order (us):", " ".join(f"{(e - s) * 1e-3:.0f}" for s, e, _ in half))
    print(f"second run span: {(half[-1][1] - t0) * 1e-6:.2f} ms, sum of kernels {sum(e - s for s, e, _ in half) * 1e-6:.2f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        parse(sys.argv[2])
