#!/bin/bash
# Round profile of bench.py on the GPU box: plain run, kernel stats, HBM traffic (separate --pmc passes, kernel-trace only), the same for
# assembly-only launches; everything lands under gpurun_out/prof_<tag>/ and tools/make_traffic_json.py turns the counter CSVs into traffic.json.
# usage: tools/profile_round.sh <tag> <commit> [bench args...]   (ASM_B: problems per assembly-only launch, default = bench.py's C2 default)
set -u
TAG=${1:-r}; COMMIT=${2:-unknown}; shift; shift || true
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "${ASM_ONLY:-}" ]; then
python bench.py "$@" > $OUT/bench_plain.log 2> $OUT/bench_plain.err
fi
cd /tmp
if [ -z "${ASM_ONLY:-}" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm_stats -- python $ROOT/tools/assembly_probe.py ${ASM_B:-16384} 5 > $OUT/asm_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/asm_fetch -- python $ROOT/tools/assembly_probe.py ${ASM_B:-16384} 3 > $OUT/asm_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/asm_write -- python $ROOT/tools/assembly_probe.py ${ASM_B:-16384} 3 > $OUT/asm_write.log 2>&1
cd $ROOT
python tools/make_traffic_json.py $OUT $COMMIT ${ASM_B:-16384} > $OUT/traffic.json
cat $OUT/traffic.json
find $OUT -name "*kernel_stats.csv" | head
tail -c 600 $OUT/bench_plain.log
