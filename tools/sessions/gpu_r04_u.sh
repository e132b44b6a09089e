#!/bin/bash
# int8 epilogue: the truncating conversion packs as it goes (cvt_pack8_i8) vs four v_cvt + three v_perm per dword (build_exp/lib_prepack.so =
# the same tree before the change).  Parity first, then interleaved timings on this box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tflite_ops.py -m gpu -x -q -k "int8 or i8 or dual or pointwise or exact or baseline" > gpurun_out/r04u/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04u/pytest.log
{
for spec in "3 1 56 256 i8 auto auto 40 256" "3 1 28 128 i8 auto auto 100 256" "3 1 56 64 i8 auto auto 100 256" "3 1 14 256 i8 auto auto 200 256" "3 1 7 512 i8 auto auto 200 256" \
            "3 2 56 64 i8 auto auto 100 256" "3 1 28 128 i8 direct auto 100 256" "3 1 56 64 i8 direct auto 100 256" "1 1 56 64 i8 auto auto 200 256" "1 1 28 128 i8 auto auto 200 256" "1 1 7 512 i8 auto auto 200 256"; do
  set -- $spec
  export LCE_K=$1 LCE_STRIDE=$2; shift 2
  echo "# filter ${LCE_K}x${LCE_K} stride $LCE_STRIDE"
  bash tools/abn.sh 3 "$*" build_exp/lib_prepack.so base
done
} 2>&1 | tee gpurun_out/r04u/ab.txt
