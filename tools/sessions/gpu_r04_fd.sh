#!/bin/bash
# streaming kernel: branch-free fastdiv with a planner-provided pass mask (one SGPR per divisor; base) vs the magic == 0 comparison
# (an SGPR pair per divisor live through the K loop; build_exp/lib_prefd.so).  Parity first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04fd
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or baseline or strips" > gpurun_out/r04fd/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04fd/pytest.log
{
for spec in "3 1 224 256 f32 auto auto 40 16" "3 1 56 256 f32 auto auto 40 256" "3 1 56 256 i8 auto auto 40 256" "3 1 56 256 bp auto auto 40 256" "3 1 14 256 f32 auto auto 200 256" "3 1 7 512 f32 auto auto 200 256" "3 1 28 128 f32 auto auto 100 256"; do
  set -- $spec
  export LCE_K=$1 LCE_STRIDE=$2; shift 2
  bash tools/abn.sh 4 "$*" build_exp/lib_prefd.so base
done
} 2>&1 | tee gpurun_out/r04fd/ab.txt
