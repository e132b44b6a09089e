#!/bin/bash
# Round 4, GPU call Q: the low-K layers (64 / 128 input channels) on the streaming kernel by the planner's own rule (depends on the outputs asked for).
OUT=gpurun_out/${1:-r04q}
mkdir -p $OUT
export TMPDIR=/tmp
for a in "56 64 f32" "28 128 f32" "56 64 i8" "28 128 i8"; do python tools/dual_check.py $a auto stream direct 2>&1 | grep -v amdgpu; done | tee $OUT/low_k_auto.txt
for a in "56 64x128 i8" "28 128x256 i8" "56 64x128 f32" "28 128x256 f32"; do echo "stride 2: $a"; LCE_STRIDE=2 python tools/dual_check.py $a auto stream direct 2>&1 | grep -v amdgpu; done | tee -a $OUT/low_k_auto.txt
echo "== 224x224xC maps, batch 16" | tee -a $OUT/low_k_auto.txt
for c in 64 128; do for e in stream direct; do echo -n "224x224x$c f32 $e: "; python tools/run_one.py 224 $c f32 $e auto 60 16 2>/dev/null | tail -1; done; done | tee -a $OUT/low_k_auto.txt
for c in 64 128; do for e in stream direct; do echo -n "112x112x$c f32 b64 $e: "; python tools/run_one.py 112 $c f32 $e auto 60 64 2>/dev/null | tail -1; done; done | tee -a $OUT/low_k_auto.txt
for st in quicknet birealnet; do python tools/graph_gaps.py run eager 100 $st 2>/dev/null | grep "per chain"; done | tee -a $OUT/low_k_auto.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "quicknet or birealnet or run_dual or golden" 2>&1 | tail -3
