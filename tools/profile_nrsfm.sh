#!/bin/bash
# Mapping-side measurements: plain run (wall-clock incl. host copies + CPU legs) and a rocprofv3 kernel-stats pass.
set -u
TAG=${1:-r}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/bench_nrsfm.py "$@" > $OUT/nrsfm_plain.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nrsfm_stats -- python $OLDPWD/tools/bench_nrsfm.py "$@" --no-cpu > $OUT/nrsfm_under_rocprof.log 2>&1)
cat $OUT/nrsfm_plain.log
cat $OUT/nrsfm_stats/*/*_kernel_stats.csv
