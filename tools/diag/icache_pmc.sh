#!/bin/bash
# Instruction-cache counters of the phase kernels (2048 C2 problems, one step): tools/diag/icache_pmc.sh
ROOT=$PWD; OUT=$ROOT/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SALU SQ_INSTS_VALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --batch 2048 --steps 1 --warmup 0 > $OUT/p$i.log 2>&1)
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "factor" if "sftb_factor" in r["Kernel_Name"] else "lin" if "sftb_lin" in r["Kernel_Name"] else None
        if k: tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
for (k, n), v in sorted(tot.items()): print(f"{k:7s} {n:32s} {v:.4g}")
PY
tail -3 $OUT/p1.log | cut -c1-300
