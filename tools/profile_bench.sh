#!/bin/bash
# Round profile of bench.py: kernel stats + HBM traffic (separate PMC passes), written under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py "$@" > $OUT/bench_plain.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $OLDPWD/bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1)
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_plain.log
