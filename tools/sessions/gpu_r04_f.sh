#!/bin/bash
# Round 4, GPU call F: the reordered prologue (rows + context first, bank in three runs) -- correctness, timings, phases.
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== stream tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_kernel or stream" 2>&1 | tail -3
echo "== timings"
for a in "56 256 f32" "56 256 i8" "56 256 bp" "14 256 f32" "14 256 i8" "14 256 bp" "7 512 f32" "7 512 i8" "7 512 bp" "7 256 f32"; do
  set -- $a; n=300; [ $1 = 56 ] && n=40
  echo -n "$a: "; python tools/run_one.py $1 $2 $3 stream auto $n 2>/dev/null | tail -1
done | tee $OUT/timings.txt
if [ -f build_exp/lib_phases.so ]; then
for a in "7 512x512 f32" "14 256x256 f32" "56 256x256 f32"; do
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_phases.so timeout 300 python tools/stream_phases.py $a 2>&1 | grep -v amdgpu.ids; echo
done | tee $OUT/stream_phases.txt
fi
