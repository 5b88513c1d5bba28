#!/bin/bash
# Per-launch durations of the phase kernels of ONE bench step (rocprofv3 kernel trace): tools/diag/lin_launches.sh <tag>
TAG=${1:-lin}; ROOT=$PWD; OUT=$ROOT/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 1 --warmup 0 > $OUT/bench.log 2>&1)
python - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/kt/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for name in ("sftb_lin", "sftb_factor", "sftb_trial"):
    k = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Workgroup_Size_X"]) for r in rows if name in r["Kernel_Name"])
    d = [x[1] for x in k][-40:]
    print(name, "lds", k[0][2], "vgpr", k[0][3], "agpr", k[0][4], "wg", k[0][5], "sum %.2f" % sum(d))
    print("  " + " ".join("%.2f" % x for x in d))
PY
