"""Turns the rocprofv3 PMC passes of tools/profile_round.sh into traffic.json (what bench.py carries as roofline.traffic).
FETCH_SIZE reports half of the bytes of wide streaming reads on gfx950 (MI355X_MICROARCH.md, calibrated with tools/probes/pmc_calib.hip):
x2; WRITE_SIZE is exact; both in KiB.  Per launch = mean over the dispatches of the kernel in the pass."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

out_dir, commit = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384   # problems per launch of both passes (bench.py's C2 default)
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 0         # > 0: throughput shape (rounds of phase kernels): steps + warm-ups of a PMC pass; bytes are per STEP


def per_step(sub, counter, kernel):
    tot = 0.0
    n = 0
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                n += 1
    return (tot / runs, n) if n and runs else (None, 0)


def total_ms(sub, kernel, steps):
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Name"]:
                return float(r["TotalDurationNs"]) * 1e-6 / steps, int(r["Calls"])
    return None, 0



def per_launch(sub, counter, kernel):
    tot = collections.defaultdict(float)
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(tot.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def avg_ms(sub, kernel):
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Name"]:
                return float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    return None, 0


res = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, --kernel-trace only; "
                   "FETCH_SIZE x2, both KiB); bench.py drops these numbers when its device code hash differs",
       "kernel_source_hash": kernel_source_hash(), "commit": commit}
if runs:
    for key, kernel in ((f"C2_B{batch}_factor", "sftb_factor_kernel"), (f"C2_B{batch}_lin", "sftb_lin_kernel"), (f"C2_B{batch}_trial", "sftb_trial_kernel"),
                        (f"C2_B{batch}_tail", "sftb_tail_kernel")):
        f, nf = per_step("pmc_fetch", "FETCH_SIZE", kernel)
        w, nw = per_step("pmc_write", "WRITE_SIZE", kernel)
        ms, calls = total_ms("stats", kernel, 6)      # the stats pass runs 5 steps + 1 warm-up
        if f is None or w is None:
            continue
        res[key] = {"fetch_kib": f, "write_kib": w, "bytes_per_step": int((2 * f + w) * 1024), "counter_rows_in_pass": [nf, nw], "kernel_ms_per_step_rocprof": ms, "calls": calls}
    # the isolated assembly pass (tools/assembly_probe.py under the same three rocprofv3 runs: asm_stats, asm_fetch, asm_write), per launch
    f, nf = per_launch("asm_fetch", "FETCH_SIZE", "sft_assembly_kernel")
    w, nw = per_launch("asm_write", "WRITE_SIZE", "sft_assembly_kernel")
    ms, calls = avg_ms("asm_stats", "sft_assembly_kernel")
    if f is not None and w is not None:
        res[f"C2_B{batch}_assembly"] = {"fetch_kib": f, "write_kib": w, "bytes_per_launch": int((2 * f + w) * 1024), "launches_in_pass": [nf, nw], "kernel_ms_rocprof_avg": ms, "calls": calls}
    # C5 x 16 (latency mode: every launch of sft_spec_kernel of ONE step; the c5 passes run 2 steps + 1 warm-up)
    c5_runs = 3
    tf = tw = 0.0
    nf = nw = 0
    for sub, counter in (("c5_fetch", "FETCH_SIZE"), ("c5_write", "WRITE_SIZE")):
        for fn in glob.glob(os.path.join(out_dir, sub, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                if "sft_spec_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    if counter == "FETCH_SIZE":
                        tf += float(r["Counter_Value"]); nf += 1
                    else:
                        tw += float(r["Counter_Value"]); nw += 1
    if nf and nw:
        res["C5_B16"] = {"fetch_kib": tf / c5_runs, "write_kib": tw / c5_runs, "bytes_per_launch": int((2 * tf + tw) * 1024 / c5_runs), "counter_rows_in_pass": [nf, nw],
                         "what": "all sft_spec_kernel<8> launches of ONE step of 16 C5 problems (latency mode: LIN / FACTOR with helper workgroups / SOLVE / TRIAL launches per round)"}
    print(json.dumps(res, indent=1))
    sys.exit(0)
for key, kernel, fs, ws, st in ((f"C2_B{batch}", "sft_lm_kernel", "pmc_fetch", "pmc_write", "stats"),
                                (f"C2_B{batch}_assembly", "sft_assembly_kernel", "asm_fetch", "asm_write", "asm_stats")):
    f, nf = per_launch(fs, "FETCH_SIZE", kernel)
    w, nw = per_launch(ws, "WRITE_SIZE", kernel)
    ms, calls = avg_ms(st, kernel)
    if f is None or w is None:
        continue
    res[key] = {"fetch_kib": f, "write_kib": w, "bytes_per_launch": int((2 * f + w) * 1024), "launches_in_pass": [nf, nw], "kernel_ms_rocprof_avg": ms, "calls": calls}
print(json.dumps(res, indent=1))
