#!/bin/bash
# rocprofv3 PMC passes (counters in their own runs, kernel-trace only).  Usage: bash tools/gpu_pmc.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
W="python $GRAFT_REPO_ROOT/tools/pmc_workload.py"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pass$i -o p -- $W > $OUT/pass$i.log 2>&1
  echo "pass $i ($ctrs) rc=$?"
done
find $OUT -name "*.csv" | head -30
du -sh $OUT
