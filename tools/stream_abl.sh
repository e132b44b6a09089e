#!/bin/bash
# Timing ablations of the streaming kernel (library variants from tools/build_exp.sh; results of the ablated ones are wrong
# by construction, only the clock is read): usage: bash tools/stream_abl.sh <outdir> <variant>...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  echo "== $v" >> $OUT/abl.log
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_$v.so timeout 120 python tools/stream_phases.py 56 256x256 ${DST:-f32} 2>&1 | grep -v amdgpu.ids | grep "launch\|prologue\|tile steps\|block life" >> $OUT/abl.log
done
cat $OUT/abl.log
