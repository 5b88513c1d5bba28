#!/bin/bash
# Copy the summaries of the last tools/profile_round.sh (+ the extra legs under gpurun_out/prof_<tag>x, pmc_<tag>) into profiles/<tag>/.
# usage: tools/collect_profiles.sh <tag>
set -eu
newest() { ls -t $1 | head -1; }   # gpurun merges every call into the same directories: take the latest pass
T=${1:-r02}; G=gpurun_out; D=profiles/$T
mkdir -p $D
cp $G/prof_$T/bench_plain.log $G/prof_$T/bench_under_rocprof.log $G/prof_$T/traffic.json $D/
cp $(newest "$G/prof_$T/stats/runc/*_kernel_stats.csv") $D/bench_kernel_stats.csv
cp $(newest "$G/prof_$T/pmc_fetch/runc/*_counter_collection.csv") $D/pmc_fetch.csv
cp $(newest "$G/prof_$T/pmc_write/runc/*_counter_collection.csv") $D/pmc_write.csv
cp $(newest "$G/prof_$T/asm_stats/runc/*_kernel_stats.csv") $D/assembly_kernel_stats.csv
cp $G/prof_$T/asm_stats.log $D/assembly_probe.log
cp $(newest "$G/prof_$T/asm_fetch/runc/*_counter_collection.csv") $D/assembly_pmc_fetch.csv
cp $(newest "$G/prof_$T/asm_write/runc/*_counter_collection.csv") $D/assembly_pmc_write.csv
X=$G/prof_${T}x
[ -f $X/bench_c5_b16.log ] && cp $X/bench_c5_b16.log $X/bench_c5_b16_rocprof.log $D/ && cp $(newest "$X/c5_stats/runc/*_kernel_stats.csv") $D/c5_b16_kernel_stats.csv
[ -f $X/nrsfm_plain.log ] && cp $X/nrsfm_plain.log $D/nrsfm_bench.log && cp $(newest "$X/nrsfm_stats/runc/*_kernel_stats.csv") $D/nrsfm_kernel_stats.csv
[ -f $X/pmc_mix.log ] && cp $X/pmc_mix.log $D/pmc_instruction_mix.log
[ -f $X/register_bench.log ] && cp $X/register_bench.log $D/
python tools/scratch_report.py > $D/scratch_report.txt 2>&1
ls $D
