#!/bin/bash
# Round 4, GPU call D: phase timeline of the K-split kernel, chains with the K-split kernel chosen by the planner, and with
# strided 3x3 layers on the streaming kernel too.
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== phases: K-split 7x7x512 and the 14x14x256 layer"
for a in "7 512x512 f32" "7 512x512 bp" "14 256x256 f32" "7 512x512 f32 64"; do
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_phases.so timeout 300 python tools/stream_phases.py $a 2>&1 | grep -v amdgpu.ids; echo
done | tee $OUT/stream_phases.txt
echo "== chains: base vs strided-on-stream"
for r in 1 2; do for lib in base build_exp/lib_strided.so; do
  for st in quicknet birealnet; do
    if [ $lib = base ]; then v=$(python tools/graph_gaps.py run eager 100 $st 2>/dev/null | grep "per chain"); else v=$(LCE_HIP_LIBRARY=$PWD/$lib python tools/graph_gaps.py run eager 100 $st 2>/dev/null | grep "per chain"); fi
    echo "lib=$lib $v"
  done
done; done | tee $OUT/chains_strided.txt
echo "== per-layer durations inside the chains"
bash tools/chain_layers.sh > $OUT/chain_layers.txt 2>&1; grep -A40 "birealnet eager" $OUT/chain_layers.txt | tail -16; grep -B2 -A24 "quicknet eager" $OUT/chain_layers.txt | grep -E "^ +1[2-5] |sum of"
