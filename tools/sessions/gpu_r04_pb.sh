#!/bin/bash
# streaming kernel: ballots written to their lanes one K-step after the compares (no s_nop 4 per unit; base) vs padded in place (build_exp/lib_prepipe.so).
# Parity first (every stream / dual / bitpacked test), then interleaved timings: bitpacked layers by run_one.py, run_dual by dual_check.py, the chains by bench.py.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04pb
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_runner.py -m gpu -x -q -k "stream or baseline or strips or dual or bitpacked or both_ways or model" > gpurun_out/r04pb/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04pb/pytest.log
{
for spec in "3 1 56 256 bp auto auto 40 256" "3 1 14 256 bp auto auto 200 256" "3 1 7 512 bp auto auto 200 256" "3 1 56 256 f32 auto auto 40 256"; do
  set -- $spec
  export LCE_K=$1 LCE_STRIDE=$2; shift 2
  bash tools/abn.sh 3 "$*" build_exp/lib_prepipe.so base
done
for r in 1 2 3; do
  for lib in build_exp/lib_prepipe.so base; do
    if [ $lib = base ]; then unset LCE_HIP_LIBRARY; else export LCE_HIP_LIBRARY=$PWD/$lib; fi
    for spec in "56 256 f32" "56 256 i8" "14 256 f32" "7 512 f32" "7 512 i8"; do
      echo "$(basename $lib .so) dual_check $spec: $(python tools/dual_check.py $spec stream 2>/dev/null | tail -1)"
    done
  done
done
unset LCE_HIP_LIBRARY
} 2>&1 | tee gpurun_out/r04pb/ab.txt
