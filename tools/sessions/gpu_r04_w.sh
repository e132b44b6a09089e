#!/bin/bash
# Float-output store cache policy, in the kernels (round 3 measured plain / sc1 / nt on an earlier streaming kernel; the pure-store probe of round 4
# says write-back stores of L0's pattern are 9 % faster than nt ones): LCE_STORE_AUX = 0 plain, 1 sc0, 2 nt (the product), 3 sc0 nt, 16 sc1, 17 sc0 sc1, 18 nt sc1.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
LIBS="build_exp/lib_aux2.so build_exp/lib_aux0.so build_exp/lib_aux1.so build_exp/lib_aux3.so build_exp/lib_aux16.so build_exp/lib_aux17.so build_exp/lib_aux18.so"
{
for spec in "3 1 56 256 f32 auto auto 40 256" "3 1 56 64 f32 auto auto 100 256" "3 1 28 128 f32 auto auto 100 256" "3 1 14 256 f32 auto auto 200 256"; do
  set -- $spec
  export LCE_K=$1 LCE_STRIDE=$2; shift 2
  bash tools/abn.sh 3 "$*" $LIBS
done
} 2>&1 | tee gpurun_out/r04w/ab.txt
