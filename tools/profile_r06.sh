#!/bin/bash
# Round-6 profile on the GPU box: the default bench (plain), its kernel stats, HBM traffic of its kernels (separate --pmc passes with
# --kernel-trace only), the instruction mix of the round kernels, the isolated assembly pass (stats + traffic), C5 x 16 (plain, stats, traffic),
# one C5 frame by phase, the mapping-side benches.  Everything lands under gpurun_out/prof_<tag>/; tools/collect_profiles_r06.sh copies the
# summaries into profiles/r06/.
# usage: tools/profile_r06.sh <tag> <commit>
set -u
TAG=${1:-r06}; COMMIT=${2:-unknown}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(hostname; rocm-smi --showuniqueid --showserial --showproductname 2>/dev/null | grep -i "unique\|serial\|Card Model\|Node ID"; lscpu | grep "^Model name"; date -u) > $OUT/box.txt
python bench.py > $OUT/bench_plain.log 2> $OUT/bench_plain.err
cd /tmp
BA="--no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py $BA --steps 5 --warmup 1 > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py $BA --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py $BA --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/mix$i -- python $ROOT/bench.py $BA --steps 1 --warmup 0 > $OUT/mix$i.log 2>&1
done
# the isolated assembly pass (lab kernel sft_assembly_kernel<8>, 16384 problems, 5 launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm_stats -- python $ROOT/tools/assembly_probe.py 16384 5 > $OUT/assembly_probe.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/asm_fetch -- python $ROOT/tools/assembly_probe.py 16384 3 > $OUT/asm_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/asm_write -- python $ROOT/tools/assembly_probe.py 16384 3 > $OUT/asm_write.log 2>&1
# C5 x 16
C5="--config C5 --batch 16 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python $ROOT/bench.py $C5 --no-extra-legs --steps 5 --warmup 1 > $OUT/bench_c5_b16_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -- python $ROOT/bench.py $C5 --no-extra-legs --steps 2 --warmup 1 > $OUT/c5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -- python $ROOT/bench.py $C5 --no-extra-legs --steps 2 --warmup 1 > $OUT/c5_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/c5_spec -- python $ROOT/tools/spec_phase_trace.py run C5 > /dev/null 2>&1
cd $ROOT
python tools/spec_phase_trace.py parse $OUT/c5_spec > $OUT/c5_frame_phases.log 2>&1
python tools/helpers_ab.py C5 > $OUT/c5_helpers_ab.log 2>&1
python tools/make_traffic_json.py $OUT $COMMIT 16384 3 > $OUT/traffic.json
python - <<PY > $OUT/pmc_instruction_mix.log
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/mix*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "factor" if "sftb_factor" in r["Kernel_Name"] else "lin" if "sftb_lin" in r["Kernel_Name"] else "trial" if "sftb_trial" in r["Kernel_Name"] else "tail" if "sftb_tail" in r["Kernel_Name"] else None
        if k:
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
print("per step of 16384 C2 problems (the benched batch) (all launches of the kernel summed), rocprofv3 --pmc, separate passes")
for (k, name), v in sorted(tot.items()):
    print(f"{k:7s} {name:32s} {v:.4g}")
PY
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv
f=$(find $OUT/asm_stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/assembly_kernel_stats.csv
f=$(find $OUT/c5_stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/c5_b16_kernel_stats.csv
for n in pmc_fetch pmc_write asm_fetch asm_write c5_fetch c5_write; do f=$(find $OUT/$n -name "*_counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$n.csv; done
# the traffic file is in place now: the bench lines that carry roofline.traffic
mkdir -p profiles/r06; cp $OUT/traffic.json profiles/r06/traffic.json
python bench.py > $OUT/bench_final.log 2> $OUT/bench_final.err
python bench.py $C5 --steps 10 --warmup 2 > $OUT/bench_c5_b16.log 2> $OUT/bench_c5_b16.err
python tools/bench_nrsfm.py > $OUT/nrsfm_plain.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nrsfm_stats -- python $ROOT/tools/bench_nrsfm.py --no-cpu > $OUT/nrsfm_under_rocprof.log 2>&1)
f=$(find $OUT/nrsfm_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/nrsfm_kernel_stats.csv
python tools/bench_schwarp_batch.py 64 1000 20 >> $OUT/nrsfm_plain.log 2>&1
python tools/bench_register.py > $OUT/register_bench.log 2>&1
# r06: the probes behind the factor kernel's changes, the batch curve of the product library, run-to-run reproducibility
timeout 300 tools/probes/mfma4x4_probe > $OUT/mfma4x4_probe.log 2>&1
timeout 120 tools/probes/chol_probe > $OUT/chol_probe.log 2>&1
timeout 900 python tools/batch_curve.py > $OUT/batch_curve.log 2>&1
timeout 600 python tools/diag/repro_bits.py 1024 4 > $OUT/repro_bits.log 2>&1
# keep what travels back small: the raw rocprofv3 directories stay on the box
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/mix? $OUT/asm_stats $OUT/asm_fetch $OUT/asm_write $OUT/c5_stats $OUT/c5_fetch $OUT/c5_write $OUT/c5_spec $OUT/nrsfm_stats
cat $OUT/traffic.json | head -40; cat $OUT/pmc_instruction_mix.log; head -7 $OUT/bench_kernel_stats.csv | cut -c1-170; cat $OUT/c5_frame_phases.log | head -8
