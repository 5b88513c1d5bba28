#!/bin/bash
# A/B lab builds next to the in-tree libraries: tools/ab_build.sh NAME "EXTRA HIPCC FLAGS" [SRCROOT] -> tools/_ab/NAME.so (a lab library compiled
# with the extra flags; tools/phases_ab.py, tools/latency_ab.py ... load it through _lib.LAB_LIB_PATH).  SRCROOT: a tree that holds defslam_amd/csrc and
# include (default: this repository; `git archive <commit> defslam_amd/csrc include | tar -x -C tools/_ab/src_<name>` gives the sources of an older
# commit for a before/after pair).  tools/_ab is git-ignored.
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRCROOT=${3:-$ROOT}
SRC=$SRCROOT/defslam_amd/csrc
OBJ=$ROOT/tools/_ab/obj_$NAME
mkdir -p "$OBJ"
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DDSH_LAB $EXTRA"
pids=()
for f in "$SRC"/*.hip; do
  b=$(basename "$f" .hip)
  nc=""; [[ $b == nrsfm_kernels || $b == register_kernels ]] && nc="-ffp-contract=off"
  /opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 $nc -c "$f" -o "$OBJ/$b.o" & pids+=($!)
done
for f in "$SRC"/*.cpp; do
  b=$(basename "$f" .cpp)
  /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$OBJ/$b.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/tools/_ab/$NAME.so" "$OBJ"/*.o
echo "built tools/_ab/$NAME.so"
