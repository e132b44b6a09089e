#!/bin/bash
# Experiment (-DLCE_ST_LOWK_EARLYB: a switch that lived only for this measurement, profiles/r06/lowk_early_b.txt; not in the tree): on the nine-K-step instances (64 input channels) phase B's transposed reads ride in K-steps 4 and 5
# (instead of all eight in K-step 6) and phase C spreads over the last three K-steps (instead of two).
OUT=gpurun_out/r06; mkdir -p $OUT
{
for spec in "56 64 i8" "56 64 f32"; do
  bash tools/abn.sh 3 "$spec stream auto 300" base build_exp/earlyb/liblce_hip.so
done
LCE_STRIDE=2 bash tools/abn.sh 3 "56 64x128 i8 stream auto 300" base build_exp/earlyb/liblce_hip.so
LCE_STRIDE=2 bash tools/abn.sh 3 "56 64x128 f32 stream auto 300" base build_exp/earlyb/liblce_hip.so
echo "## parity of the experiment build"
LCE_HIP_LIBRARY=$PWD/build_exp/earlyb/liblce_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream or birealnet or quicknet" 2>&1 | tail -3
} > $OUT/lowk_early_b.txt 2>&1
cat $OUT/lowk_early_b.txt
