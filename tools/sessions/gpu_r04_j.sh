#!/bin/bash
TAG=${1:-r04j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for a in "56 256 f32 stream auto 40" "56 256 bp stream auto 40" "14 256 bp stream auto 300"; do
  bash tools/abn.sh 3 "$a" base build_exp/lib_r03kernels.so build_exp/lib_ksplit_commit.so build_exp/lib_upfront.so
done 2>&1 | tee $OUT/ab_bisect.txt
