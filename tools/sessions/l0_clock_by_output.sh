#!/bin/bash
# The BASELINE layer on the weight-stationary kernel, per output type, from the kernel's own cycle stamps (build_exp/lib_sph.so = tools/build_exp.sh
# sph:"-DLCE_STREAM_PHASES"): cycles per block step and per MFMA, and the clock the launch's wall time implies -- which of int8's 20 % over
# bitpacked is more cycles (epilogue not hidden) and which a lower granted clock (power).
OUT=gpurun_out/r06; mkdir -p $OUT
{
for dst in bp i8 f32; do
  echo "== 56x56 256->256 batch 256, $dst"
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_sph.so timeout 120 python tools/stream_phases.py 56 256x256 $dst 2>&1 | grep -v amdgpu.ids
done
for dst in bp i8; do
  echo "== 56x56 64->64 batch 256, $dst"
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_sph.so timeout 120 python tools/stream_phases.py 56 64x64 $dst 2>&1 | grep -v amdgpu.ids
done
} > $OUT/l0_clock_by_output.txt 2>&1
cat $OUT/l0_clock_by_output.txt
