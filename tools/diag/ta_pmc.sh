#!/bin/bash
# Address-unit / L1 counters of the factor kernel (2048 C2 problems, one step): is the scattered gather of H what the step waits for?
ROOT=$PWD; OUT=$ROOT/gpurun_out/ta; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|TD_[A-Z_a-z0-9]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
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
