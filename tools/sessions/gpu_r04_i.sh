#!/bin/bash
# Round 4, GPU call I: same-box A/B of the round-4 streaming kernel against the round-3 kernels (lib_r03kernels.so = commit 2b71693).
TAG=${1:-r04i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for a in "56 256 f32 stream auto 40" "56 256 i8 stream auto 40" "56 256 bp stream auto 40" "14 256 f32 stream auto 300" "14 256 i8 stream auto 300" "14 256 bp stream auto 300" "7 512 f32 auto auto 300" "7 512 i8 auto auto 300"; do
  bash tools/abn.sh 3 "$a" base build_exp/lib_r03kernels.so
done 2>&1 | tee $OUT/ab_r03_kernels.txt
