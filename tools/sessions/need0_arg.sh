#!/bin/bash
# Experiment (-DLCE_ST_NEED0_ARG; the switch lived only for this measurement): the first tile step's quota from the kernel arguments on EVERY instance of the
# weight-stationary kernel (the K-split ones already do), instead of the schedule table's first entry -- one dependent scalar load less in front of the first rows' loads.
OUT=gpurun_out/r06; mkdir -p $OUT
{
for spec in "14 256 f32" "14 256 i8" "28 128 f32" "28 128 i8" "56 64 i8" "56 256 f32"; do
  bash tools/abn.sh 3 "$spec stream auto 300" base build_exp/need0/liblce_hip.so
done
} > $OUT/need0_arg.txt 2>&1
cat $OUT/need0_arg.txt
