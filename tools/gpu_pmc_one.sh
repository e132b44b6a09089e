#!/bin/bash
# PMC passes over ONE configuration: bash tools/gpu_pmc_one.sh <tag> <run_one.py args...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU" \
            "SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
            "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/run_one.py "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$? $(tail -1 $OUT/pass$i.log | cut -c1-100)"
done
rm -f $OUT/pass*/p_kernel_trace.csv
python3 $GRAFT_REPO_ROOT/tools/parse_pmc.py $OUT > $OUT/summary.json; cat $OUT/summary.json
