#!/bin/bash
# Round 4, GPU call O: int8 layers of 64 / 128 input channels on the streaming kernel (auto rule variant), with and without packed f32 math.
OUT=gpurun_out/${1:-r04o}
mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2; do for lib in base build_exp/lib_lowk.so build_exp/lib_lowkpk.so; do
  if [ $lib = base ]; then v=$(python tools/graph_gaps.py run eager 100 birealnet 2>/dev/null | grep "per chain"); else v=$(LCE_HIP_LIBRARY=$PWD/$lib python tools/graph_gaps.py run eager 100 birealnet 2>/dev/null | grep "per chain"); fi
  echo "lib=$lib $v"
done; done | tee $OUT/chains_lowk.txt
for a in "56 64 i8" "28 128 i8"; do for lib in build_exp/lib_lowk.so build_exp/lib_lowkpk.so; do
  echo "[$a] $lib"; LCE_HIP_LIBRARY=$PWD/$lib python tools/dual_check.py $a auto direct 2>&1 | grep -v amdgpu
done; done | tee -a $OUT/chains_lowk.txt
for a in "56 64x128 i8" "28 128x256 i8"; do for lib in build_exp/lib_lowk.so build_exp/lib_lowkpk.so; do
  echo "[$a s2] $lib"; LCE_STRIDE=2 LCE_HIP_LIBRARY=$PWD/$lib python tools/dual_check.py $a auto direct 2>&1 | grep -v amdgpu
done; done | tee -a $OUT/chains_lowk.txt
# correctness of the packed variant: a few stream tests with the library swapped in
LCE_HIP_LIBRARY=$PWD/build_exp/lib_lowkpk.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_kernel_shapes or int8_exact_ties" 2>&1 | tail -2
