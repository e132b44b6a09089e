#!/bin/bash
# Round 4, GPU call G: column strips of wide images on the streaming kernel.
TAG=${1:-r04g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== strip tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "column_strips or wide_images or streaming_kernel" 2>&1 | tail -4
echo "== 224x224x256, batch 16: strips vs the 2-D tiles of the block GEMM"
for d in f32 i8 bp; do
  echo -n "auto   $d: "; python tools/run_one.py 224 256 $d auto auto 30 16 2>/dev/null | tail -1
  echo -n "direct $d: "; python tools/run_one.py 224 256 $d direct auto 30 16 2>/dev/null | tail -1
  echo -n "strip64 $d: "; LCE_OPTS=stream_strip=64 python tools/run_one.py 224 256 $d stream auto 30 16 2>&1 | tail -1
done | tee $OUT/strips_224.txt
for rows in 7 8 14 16 28; do echo -n "rows=$rows f32: "; LCE_OPTS=stream_rows=$rows python tools/run_one.py 224 256 f32 stream auto 30 16 2>/dev/null | tail -1; done | tee -a $OUT/strips_224.txt
echo -n "112x112x256 b64 auto: "; python tools/run_one.py 112 256 f32 auto auto 30 64 2>/dev/null | tail -1 | tee -a $OUT/strips_224.txt
echo -n "112x112x256 b64 direct: "; python tools/run_one.py 112 256 f32 direct auto 30 64 2>/dev/null | tail -1 | tee -a $OUT/strips_224.txt
echo -n "112x112x256 b64 strip32: "; LCE_OPTS=stream_strip=32 python tools/run_one.py 112 256 f32 stream auto 30 64 2>/dev/null | tail -1 | tee -a $OUT/strips_224.txt
