#!/bin/bash
# after the ballots change: low-K int8 run_dual and bitpacked layers, streaming kernel vs block GEMM on one box (does the auto rule still hold?).  Parity first.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04pb3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or baseline or dual or bitpacked or both_ways" > gpurun_out/r04pb3/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04pb3/pytest.log
{
for r in 1 2; do
  for spec in "28 128 i8" "56 64 i8" "28 128 f32" ; do echo "dual_check $spec:"; python tools/dual_check.py $spec stream direct 2>/dev/null | tail -2; done
  for spec in "56 64x128 i8" "28 128x256 i8"; do echo "dual_check stride 2 $spec:"; LCE_STRIDE=2 python tools/dual_check.py $spec stream direct 2>/dev/null | tail -2; done
  for eng in stream direct; do
    echo "28 128 bp $eng: $(LCE_K=3 python tools/run_one.py 28 128 bp $eng auto 100 256 2>/dev/null | tail -1)"
    echo "56 64 bp $eng: $(LCE_K=3 python tools/run_one.py 56 64 bp $eng auto 100 256 2>/dev/null | tail -1)"
    echo "14 256 bp $eng: $(LCE_K=3 python tools/run_one.py 14 256 bp $eng auto 200 256 2>/dev/null | tail -1)"
  done
done
} 2>&1 | tee gpurun_out/r04pb3/ab.txt
