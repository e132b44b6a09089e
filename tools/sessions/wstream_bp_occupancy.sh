#!/bin/bash
# The weight-streaming kernel with bitpacked output: its epilogue needs no transpose scratch, so the launch asks for the images' LDS only
# (28 KiB instead of 60 for a 14x14x256 image) and more blocks are resident per CU (registers: 88 / 124 / 164 / 206 for 1 / 2 / 3 / 4 pixel
# blocks per block).  base = the library before that change (build_exp/basews), new = the tree's; nb1..nb4 = wstream_blocks.
OUT=gpurun_out/r06; mkdir -p $OUT
# (the host-side change this measured -- wstream_lds_extra / wstream_occupancy in lce_plan.h -- was not kept: profiles/r06/wstream_bp_occupancy_ab.txt)
{
for spec in "14 256x256 bp 256 3 40" "14 256x512s2 bp 256 3 40" "7 512x512 bp 256 3 40" "14 128x128 bp 256 3 40"; do
  bash tools/ab_libs.sh 2 "$spec auto:engine=wstream nb1:engine=wstream,wstream_blocks=1 nb2:engine=wstream,wstream_blocks=2 nb3:engine=wstream,wstream_blocks=3 nb4:engine=wstream,wstream_blocks=4 stream:engine=stream" build_exp/basews/liblce_hip.so base
done
} > $OUT/wstream_bp_occupancy.txt 2>&1
cat $OUT/wstream_bp_occupancy.txt
