#!/bin/bash
# Copies the summaries of tools/profile_r06.sh from gpurun_out/prof_r05/ into profiles/r06/ (bench_plain.log = the plain bench repeated behind the
# PMC passes, whose line carries roofline.traffic; bench_first.log = the plain bench in front of them).
set -eu
G=gpurun_out/prof_${1:-r06}; D=profiles/r06
mkdir -p $D
cp $G/traffic.json $G/bench_under_rocprof.log $G/pmc_instruction_mix.log $G/bench_kernel_stats.csv $G/box.txt $D/
cp $G/bench_final.log $D/bench_plain.log
cp $G/bench_plain.log $D/bench_first.log
cp $G/pmc_fetch.csv $G/pmc_write.csv $D/
cp $G/assembly_kernel_stats.csv $G/assembly_probe.log $D/
cp $G/asm_fetch.csv $D/assembly_pmc_fetch.csv; cp $G/asm_write.csv $D/assembly_pmc_write.csv
cp $G/bench_c5_b16.log $G/bench_c5_b16_rocprof.log $G/c5_b16_kernel_stats.csv $G/c5_frame_phases.log $G/c5_helpers_ab.log $D/
cp $G/c5_fetch.csv $D/c5_pmc_fetch.csv; cp $G/c5_write.csv $D/c5_pmc_write.csv
cp $G/nrsfm_plain.log $D/nrsfm_bench.log; cp $G/nrsfm_kernel_stats.csv $G/register_bench.log $D/
cp $G/mfma4x4_probe.log $G/chol_probe.log $G/batch_curve.log $G/repro_bits.log $D/
python tools/scratch_report.py > $D/scratch_report.txt 2>&1 || true
ls $D
