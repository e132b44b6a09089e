#!/bin/bash
# The round's rocprofv3 evidence in one GPU call:
#   bash tools/gpu_profile_round.sh <tag>            (writes gpurun_out/<tag>/...)
# 1. --kernel-trace --stats over the default bench.py command (the L0 headline + every `extra` line)
# 2. --kernel-trace --stats per single layer named in bench.py's `extra` (one process per layer, so the
#    per-kernel average IS that layer's launch duration)
# 3. PMC passes (separate runs, counters only + kernel trace) over the L0 float layer
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/layers $OUT/bench
PARTS=${PARTS:-123}
export TMPDIR=/tmp
cd /tmp
summ() {  # kernel_stats.csv -> text
python3 - "$1" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].split("(")[0].replace("void ","")
    if "at::native" in n or "rocclr" in n or "elementwise" in n: continue
    print("%-64s calls %5s avg %9.2f us  min %9.2f  max %9.2f  total %10.1f us" % (n[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
}
# 1 -------------------------------------------------------------------------------------------------
if [[ $PARTS == *1* ]]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o t -- \
    python $R/bench.py --no-cpu-baseline > $OUT/bench/stdout.log 2> $OUT/bench/stderr.log
echo "bench rc=$?"
tail -1 $OUT/bench/stdout.log > $OUT/bench_under_rocprof.json
summ $OUT/bench/t_kernel_stats.csv | tee $OUT/bench_kernel_stats.txt
rm -f $OUT/bench/t_kernel_trace.csv $OUT/bench/t_agent_info.csv
fi
# 2 -------------------------------------------------------------------------------------------------
if [[ $PARTS == *2* ]]; then
one() {  # tag K args...
  local t=$1 k=$2; shift 2
  LCE_K=$k timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/layers/$t -o t -- \
      python $R/tools/run_one.py "$@" > $OUT/layers/$t.log 2>/dev/null
  { echo "# $t: run_one.py $* (filter ${k}x${k}) -> $(tail -1 $OUT/layers/$t.log)"; summ $OUT/layers/$t/t_kernel_stats.csv; } | tee -a $OUT/layer_kernel_stats.txt
  cp $OUT/layers/$t/t_kernel_stats.csv $OUT/layers/$t.kernel_stats.csv
  rm -rf $OUT/layers/$t
}
: > $OUT/layer_kernel_stats.txt
one l0_f32            3 56 256 f32 auto auto 30
one l0_int8           3 56 256 i8  auto auto 30
one l0_bitpacked      3 56 256 bp  auto auto 30
one l0_f32_workspace  3 56 256 f32 mfma auto 30
one l0_f32_valu       3 56 256 f32 valu auto 10
one quicknet_56x64    3 56 64  f32 auto auto 50
one quicknet_28x128   3 28 128 f32 auto auto 50
one quicknet_14x256   3 14 256 f32 auto auto 100
one quicknet_7x512    3 7  512 f32 auto auto 100
one pointwise_56x64   1 56 64  i8  auto auto 100
one pointwise_28x128  1 28 128 i8  auto auto 100
one pointwise_14x256  1 14 256 i8  auto auto 100
one pointwise_7x512   1 7  512 i8  auto auto 100
fi
# 3 -------------------------------------------------------------------------------------------------
if [[ $PARTS == *3* ]]; then
bash $R/tools/gpu_pmc_one.sh $TAG/pmc_l0_f32 56 256 f32 auto auto 20 > $OUT/pmc_l0_f32.log 2>&1
tail -5 $OUT/pmc_l0_f32.log
fi
