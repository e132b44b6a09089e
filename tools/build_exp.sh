#!/bin/bash
# Build experiment variants of the product library side by side (scratch, git-ignored; single translation unit,
# -DLCE_EXPERIMENT: csrc/lce_experiments.h):
#   bash tools/build_exp.sh NAME1:"-DFLAG1 -DFLAG2" NAME2:"-DFLAG3" ...   -> build_exp/lib_NAME.so
# Use them with LCE_HIP_LIBRARY=$PWD/build_exp/lib_NAME.so (tools/run_one.py, tools/phases.py, tools/abn.sh).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_exp
cd $ROOT/compute-engine_amd/csrc
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  [ "$flags" == "$spec" ] && flags=""
  ( hipcc -DLCE_EXPERIMENT -DLCE_UNITY $flags -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I. -Wno-unused-result -shared \
      -o $ROOT/build_exp/lib_$name.so lce_hip_api.hip lce_plan.cpp lce_plan_stream.cpp lce_plan_cost.cpp lce_prepare.cpp 2>&1 | grep -E " error|Error" ) &
done
wait
ls -la $ROOT/build_exp/
