#!/bin/bash
# Round-4 profile of bench.py (throughput shape = rounds of sftb_lin / sftb_factor / sftb_trial kernels) on the GPU box: plain run, kernel
# stats, HBM traffic (separate --pmc passes with --kernel-trace only), instruction mix of the factor kernel.  Everything lands under
# gpurun_out/prof_<tag>/; tools/make_traffic_json.py turns the counter CSVs into traffic.json.
# usage: tools/profile_r04.sh <tag> <commit>
set -u
TAG=${1:-r04}; COMMIT=${2:-unknown}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_plain.log 2> $OUT/bench_plain.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 5 --warmup 1 > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/mix$i -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --batch 2048 --steps 1 --warmup 0 > $OUT/mix$i.log 2>&1
done
cd $ROOT
python tools/make_traffic_json.py $OUT $COMMIT 16384 3 > $OUT/traffic.json
cat $OUT/traffic.json
python - <<PY > $OUT/pmc_instruction_mix.log
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/mix*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "factor" if "sftb_factor" in r["Kernel_Name"] else "lin" if "sftb_lin" in r["Kernel_Name"] else "trial" if "sftb_trial" in r["Kernel_Name"] else None
        if k:
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
print("per step of 2048 C2 problems (all launches of the kernel summed), rocprofv3 --pmc, separate passes")
for (k, name), v in sorted(tot.items()):
    print(f"{k:7s} {name:32s} {v:.4g}")
PY
cat $OUT/pmc_instruction_mix.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv; head -6 $f | cut -c1-200
tail -c 400 $OUT/bench_plain.log
