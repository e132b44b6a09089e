#!/bin/bash
# Two resident blocks per CU of the weight-stationary kernel on 64-input-channel layers.
# profiles/r06/occ2_potential.txt was measured with a scratch build (never committed: -DLCE_ST_OCC2 = launch bounds (256, 2) on ALL KCH = 1
# instances, no "a"-constrained registers in them, no epilogue scratch in the bitpacked instance's LDS) against the product of that moment,
# one = 256 blocks / two = 512 blocks (compute_units=512).  The bitpacked part of it is the product now; this script repeats that
# part on the product through the plan option (the int8 / float instances are compiled for one block per CU and refuse the option).
OUT=gpurun_out/r06; mkdir -p $OUT
{
for spec in "56 64x64 bp 256 3 40" "112 64x64 bp 64 3 20" "56 64x256 bp 256 3 20" "56 64x128s2 bp 256 3 40"; do
  python tools/ab_opts.py $spec one:engine=stream,stream_blocks_per_cu=1 two:engine=stream,stream_blocks_per_cu=2 auto 2>/dev/null | grep "MEDIAN\|^# "
done
} > $OUT/two_blocks_per_cu_product.txt 2>&1
cat $OUT/two_blocks_per_cu_product.txt
