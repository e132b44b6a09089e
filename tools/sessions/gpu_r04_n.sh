#!/bin/bash
# Round 4, GPU call N: timing ablations of the streaming kernel on the low-K layers (64 / 128 input channels): what a block step is made of.
# (results of the ablated variants are wrong by construction; only the clock is read)
OUT=gpurun_out/${1:-r04n}
mkdir -p $OUT
export TMPDIR=/tmp
for a in "56 64 i8" "56 64 bp" "28 128 i8" "28 128 bp"; do
  for lib in base build_exp/lib_noepi.so build_exp/lib_noprod.so build_exp/lib_nofrag.so build_exp/lib_bare.so; do
    if [ $lib = base ]; then v=$(python tools/run_one.py $a stream auto 200 2>/dev/null | tail -1); else v=$(LCE_HIP_LIBRARY=$PWD/$lib python tools/run_one.py $a stream auto 200 2>/dev/null | tail -1); fi
    echo "[$a] $(basename $lib .so): $v"
  done
  echo -n "[$a] block GEMM: "; python tools/run_one.py $a direct auto 200 2>/dev/null | tail -1
done | tee $OUT/stream_ablations_low_k.txt
