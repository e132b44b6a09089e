#!/bin/bash
# Interleaved comparison of library BUILDS through tools/ab_opts.py (one process per build and round; same box):
#   bash tools/ab_libs.sh <rounds> "<ab_opts args: HxW CINxCOUT DST BATCH ROUNDS STEPS variants...>" lib1.so lib2.so ...   ("base" = the in-tree library)
ROUNDS=$1; ARGS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then out=$(python tools/ab_opts.py $ARGS 2>/dev/null | grep MEDIAN);
    else out=$(LCE_HIP_LIBRARY=$PWD/$lib python tools/ab_opts.py $ARGS 2>/dev/null | grep MEDIAN); fi
    echo "$(basename $lib .so): $out"
  done
done
