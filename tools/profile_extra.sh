#!/bin/bash
# The extra legs of a round profile (tools/collect_profiles.sh picks them up from gpurun_out/prof_<tag>x/): the C5 x 16 bench with its
# kernel stats, the mapping-side bench with its kernel stats, the registration bench.
# usage: tools/profile_extra.sh <tag>
set -u
TAG=${1:-r}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_${TAG}x
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --config C5 --batch 16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_b16.log 2> $OUT/bench_c5_b16.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python $ROOT/bench.py --config C5 --batch 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/bench_c5_b16_rocprof.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -- python $ROOT/bench.py --config C5 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/c5_fetch.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -- python $ROOT/bench.py --config C5 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/c5_write.log 2>&1)
for pr in wave_mfma_probe agpr_tile_probe; do [ -x tools/probes/$pr ] && timeout 120 tools/probes/$pr > $OUT/$pr.log 2>&1; done
python tools/bench_nrsfm.py > $OUT/nrsfm_plain.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nrsfm_stats -- python $ROOT/tools/bench_nrsfm.py --no-cpu > $OUT/nrsfm_under_rocprof.log 2>&1)
python tools/bench_schwarp_batch.py 64 1000 20 >> $OUT/nrsfm_plain.log 2>&1
python tools/bench_register.py > $OUT/register_bench.log 2>&1
tail -c 400 $OUT/bench_c5_b16.log; tail -5 $OUT/nrsfm_plain.log
