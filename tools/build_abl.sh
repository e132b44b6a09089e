#!/bin/bash
# Ablation / A-B variants of the product library built the PRODUCT way -- one translation unit per kernel family, in parallel (~2 min
# each; tools/build_exp.sh builds single-translation-unit variants for the time-stamp tools, which takes much longer):
#   bash tools/build_abl.sh NAME1:"-DFLAG1 -DFLAG2" NAME2:"-DFLAG3" ...   -> build_exp/NAME/liblce_hip.so   (scratch, git-ignored)
# Use with LCE_HIP_LIBRARY=$PWD/build_exp/NAME/liblce_hip.so.  -DLCE_EXPERIMENT is passed here (csrc/lce_experiments.h).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  [ "$flags" == "$spec" ] && flags=""
  mkdir -p $ROOT/build_exp/$name
  make -s -j8 -C $ROOT/compute-engine_amd/csrc OUT=$ROOT/build_exp/$name \
    FLAGS="-DLCE_EXPERIMENT $flags -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I. -Wno-unused-result" 2>&1 | grep -E " error|Error" 
  rm -rf $ROOT/build_exp/$name/obj
  ls -la $ROOT/build_exp/$name/liblce_hip.so
done
