#!/bin/bash
# Instruction mix / MFMA-busy counters of sft_lm_kernel (separate rocprofv3 --pmc passes, kernel-trace only).
# usage: tools/pmc_mix.sh <tag> <problems> [waves]
set -u
TAG=${1:-mix}; B=${2:-2048}; W=${3:-}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
SETS=${PMC_SETS:-"SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM SQ_WAVE_CYCLES,SQ_WAVES,SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_MISC,SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_VMEM,SQ_INSTS_BRANCH"}
for set in $SETS; do
  set=${set//,/ }
  i=$((i+1))
  (cd /tmp && PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $OLDPWD/tools/phase_times.py C2 $B $W > $OUT/p$i.log 2>&1)
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sft_lm_kernel" in r["Kernel_Name"]:
            tot[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
agg = collections.defaultdict(list)
for (name, disp), v in tot.items():
    agg[name].append(sum(v))
for name, v in sorted(agg.items()):
    print(f"{name:32s} per launch: {sum(v)/len(v):.4g}  (launches {len(v)})")
PY
