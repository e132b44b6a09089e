#!/bin/bash
# Round 4, GPU call C: the K-split streaming kernel (512 input channels) and flat pixel blocks against the block GEMM.
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== stream tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "streaming_kernel" 2>&1 | tail -4
echo "== K-split vs block GEMM"
LCE_STEPS=200 timeout 600 python tools/stream_check.py 7x512x512 7x512x512s2 14x512x512 7x512x128 7x512x256 2>&1 | tee $OUT/ksplit_check.txt
echo "== flat vs per-image blocks (7x7x256, 3x3)"
for f in 1 0; do echo "stream_flat=$f"; LCE_OPTS=stream_flat=$f LCE_STEPS=200 timeout 300 python tools/stream_check.py 7x256x256 7x512x512 2>&1; done | tee $OUT/flat_check.txt
echo "== strided 256 -> 512 (config 5)"
LCE_STEPS=200 timeout 300 python tools/stream_check.py 14x256x512s2 28x128x256s2 2>&1 | tee -a $OUT/ksplit_check.txt
