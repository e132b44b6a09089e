#!/bin/bash
# The suite's randomized GPU tests on the round's last build (two blocks per CU in the product; "stream_x2" among the engines), other seeds.
OUT=gpurun_out/r06; mkdir -p $OUT
{
echo "## python -m pytest tests/test_gpu_parity_r6.py -x -q -k 'planner_decides or two_blocks'   (the suite's own 420 draws)"
timeout 900 python -m pytest tests/test_gpu_parity_r6.py -x -q -k "planner_decides or two_blocks" 2>&1 | tail -3
for s in ${FUZZ_SEEDS:-11 12}; do
  echo "## LCE_FUZZ_EXAMPLES=2500 LCE_FUZZ_SEED=$s python -m pytest tests/test_gpu_parity_r6.py -x -q -k planner_decides"
  LCE_FUZZ_EXAMPLES=2500 LCE_FUZZ_SEED=$s timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -x -q -k planner_decides 2>&1 | tail -3
done
echo "## LCE_FUZZ_EXAMPLES=1500 LCE_FUZZ_SEED=11 python -m pytest tests/test_gpu_model_random.py -x -q"
LCE_FUZZ_EXAMPLES=1500 LCE_FUZZ_SEED=11 timeout 900 python -m pytest tests/test_gpu_model_random.py -x -q 2>&1 | tail -3
} > $OUT/fuzz_two_blocks.txt 2>&1
cat $OUT/fuzz_two_blocks.txt
