#!/bin/bash
# Which phase of the int8 epilogue the low-K block steps pay for: product vs builds without phase A (transform + scratch writes), B (the
# transposed reads), C (round + pack + stores), and without all three (tools/build_abl.sh noepia / noepib / noepic / noepi; results wrong).
OUT=gpurun_out/r06; mkdir -p $OUT
{
for spec in "56 64 i8" "28 128 i8" "56 64 f32"; do
  bash tools/abn.sh 3 "$spec stream auto 300" base build_exp/noepia/liblce_hip.so build_exp/noepib/liblce_hip.so build_exp/noepic/liblce_hip.so build_exp/noepi/liblce_hip.so
done
LCE_STRIDE=2 bash tools/abn.sh 3 "56 64x128 i8 stream auto 300" base build_exp/noepia/liblce_hip.so build_exp/noepib/liblce_hip.so build_exp/noepic/liblce_hip.so build_exp/noepi/liblce_hip.so
} > $OUT/lowk_int8_epilogue_phases.txt 2>&1
cat $OUT/lowk_int8_epilogue_phases.txt
