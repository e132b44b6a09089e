#!/bin/bash
# Round 4, GPU call B: the first block step taking its weights as they arrive (A/B against the round-3 order), the
# 256-channel layers as 2 slices x 2 pixel phases, chains as half batches on two streams, per-layer durations in the chains.
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== residency tests"; timeout 600 python -m pytest tests/test_gpu_tflite_ops.py tests/test_gpu_model_runner.py -q -x 2>&1 | tail -3
echo "== stream tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "stream or l0_batch256 or quicknet_layer or int8_exact" 2>&1 | tail -3
echo "== A/B peeled first step (base) vs bank up front"
for a in "56 256 f32 stream auto 30" "56 256 i8 stream auto 30" "56 256 bp stream auto 30" "14 256 f32 stream auto 200" "14 256 i8 stream auto 200" "14 256 bp stream auto 200"; do
  bash tools/abn.sh 3 "$a" base build_exp/lib_upfront.so
done 2>&1 | tee $OUT/ab_peel.txt
echo "== 14x14x256 pixel phases"
for ph in 0 2 4; do for d in f32 i8 bp; do
  echo -n "phases=$ph $d: "; LCE_OPTS=stream_pixel_phases=$ph python tools/run_one.py 14 256 $d stream auto 300 2>/dev/null | tail -1
done; done 2>&1 | tee $OUT/phases_14.txt
echo "== 7x7x256 / 28x28x256 phases"
for hw in 7 28; do for ph in 0 2; do
  echo -n "hw=$hw phases=$ph f32: "; LCE_OPTS=stream_pixel_phases=$ph python tools/run_one.py $hw 256 f32 stream auto 200 2>/dev/null | tail -1
done; done 2>&1 | tee -a $OUT/phases_14.txt
echo "== chains split over two streams"
for w in quicknet birealnet; do timeout 300 python tools/chain_split.py $w 2 60 2>&1 | tail -5; done | tee $OUT/chain_split.txt
timeout 300 python tools/chain_split.py birealnet 4 60 2>&1 | tail -5 | tee -a $OUT/chain_split.txt
echo "== per-layer durations inside the chains"
bash tools/chain_layers.sh > $OUT/chain_layers.txt 2>&1; tail -60 $OUT/chain_layers.txt
