#!/bin/bash
# Address-unit / L1 counters of the factor kernel (2048 C2 problems, one step): is the scattered gather of H what the step waits for?
ROOT=$PWD; OUT=$ROOT/gpurun_out/ta; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|TD_[A-Z_a-z0-9]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
i=0
# ONLY this set: a second pass with TA_FLAT_*_WAVEFRONTS / TA_ADDR_STALLED_BY_TC_CYCLES aborted inside rocprofv3 and hung the box until the timeout (r04)
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --batch 2048 --steps 1 --warmup 0 > $OUT/p$i.log 2>&1)
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); dur = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "factor" if "sftb_factor" in r["Kernel_Name"] else "lin" if "sftb_lin" in r["Kernel_Name"] else None
        if k: tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
for (k, n), v in sorted(tot.items()): print(f"{k:7s} {n:40s} {v:.4g}")
PY
grep -il "error\|invalid\|not found" $OUT/p*.log | head; cut -c1-1500 $OUT/avail.txt
