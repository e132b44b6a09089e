#!/bin/bash
# ballots one unit late in the low-K streaming kernels too (several epilogue units per K-step; base) vs padded in place there (build_exp/lib_pipe4.so).  Parity first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04pb2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or baseline or dual or bitpacked or both_ways" > gpurun_out/r04pb2/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04pb2/pytest.log
{
for r in 1 2 3; do
  for lib in build_exp/lib_pipe4.so base; do
    if [ $lib = base ]; then unset LCE_HIP_LIBRARY; else export LCE_HIP_LIBRARY=$PWD/$lib; fi
    for spec in "28 128 f32" "28 128 i8" "56 64 f32" "56 64 i8"; do
      echo "$(basename $lib .so) dual_check $spec: $(python tools/dual_check.py $spec stream 2>/dev/null | tail -1)"
    done
    echo "$(basename $lib .so) dual_check stride 2 56 64x128 f32: $(LCE_STRIDE=2 python tools/dual_check.py 56 64x128 f32 stream 2>/dev/null | tail -1)"
    echo "$(basename $lib .so) 28 128 bp stream: $(LCE_K=3 python tools/run_one.py 28 128 bp stream auto 100 256 2>/dev/null | tail -1)"
  done
done
unset LCE_HIP_LIBRARY
} 2>&1 | tee gpurun_out/r04pb2/ab.txt
