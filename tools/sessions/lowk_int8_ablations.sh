#!/bin/bash
# Round 6, after the review's items: what the low-K int8 layers of config 5 (one launch each, batch 256) are made of -- the same
# ablation builds as `gpu_r06.sh abl` (tools/build_abl.sh noepi / noprod / nofrag), on the 64- and 128-channel layers.
OUT=gpurun_out/r06; mkdir -p $OUT
{
for spec in "56 64 i8" "56 64 f32" "56 64 bp" "28 128 i8" "28 128 f32"; do
  bash tools/abn.sh 3 "$spec stream auto 300" base build_exp/nofrag/liblce_hip.so build_exp/noepi/liblce_hip.so build_exp/noprod/liblce_hip.so
done
LCE_STRIDE=2 bash tools/abn.sh 3 "56 64x128 i8 stream auto 300" base build_exp/nofrag/liblce_hip.so build_exp/noepi/liblce_hip.so build_exp/noprod/liblce_hip.so
} > $OUT/lowk_int8_ablations.txt 2>&1
cat $OUT/lowk_int8_ablations.txt
