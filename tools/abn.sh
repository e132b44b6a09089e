#!/bin/bash
# Interleaved comparison of several builds of liblce_hip.so on the same box (clock drift hits all).
# usage: bash tools/abn.sh <rounds> "<run_one args>" lib1.so lib2.so ...   ("base" = the in-tree library)
ROUNDS=$1; ARGS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  line=""
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then v=$(python tools/run_one.py $ARGS 2>/dev/null | tail -1 | awk '{print $(NF-1)}');
    else v=$(LCE_HIP_LIBRARY=$PWD/$lib python tools/run_one.py $ARGS 2>/dev/null | tail -1 | awk '{print $(NF-1)}'); fi
    n=$(basename $lib .so); [ "$n" = "liblce_hip" ] && n=$(basename $(dirname $lib))   # (tools/build_abl.sh: build_exp/NAME/liblce_hip.so)
    line="$line $n=$(printf %.4f $v)"
  done
  echo "[$ARGS]$line"
done
