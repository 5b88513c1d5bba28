#!/bin/bash
# Instruction mix and wait counters of the FACTOR launches of ONE C5 frame in latency mode (owners + helper workgroups of sft_spec_kernel<8>;
# the launches are told apart by their grid), separate rocprofv3 --pmc passes with --kernel-trace only.  usage (GPU box): tools/diag/c5_factor_pmc.sh
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out/c5_factor_pmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $ROOT/tools/spec_phase_trace.py run C5 > $OUT/p$i.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sft_spec_kernel" not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
        tot[(g, r["Counter_Name"])] += float(r["Counter_Value"]); n[(g, r["Counter_Name"])] += 1
print("sft_spec_kernel<8>, one C5 frame run twice, per grid size (threads): sum over all launches of that grid / launches")
for (g, name), v in sorted(tot.items()):
    print(f"grid {g:6d} {name:28s} {v:.4g}  ({n[(g, name)]} rows)")
PY
