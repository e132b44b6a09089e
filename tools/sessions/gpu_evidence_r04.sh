#!/bin/bash
# Round-4 evidence in one GPU call (writes gpurun_out/r04/...):
#   1  rocprofv3 --kernel-trace --stats of the default bench command
#   2  one process per layer under rocprofv3 --kernel-trace: per-dispatch durations, all launches AND the timed ones alone
#      (tools/steady_stats.py), next to the HIP-event figure of the same process
#   3  PMC passes (seven separate rocprofv3 --pmc runs each) of the kernels new in this round and of the headline kernel
#   4  the default bench line un-profiled, without the clock spin-up, and with the traffic measured in the run
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04
mkdir -p $OUT/layers
export TMPDIR=/tmp
cd /tmp
PARTS=${PARTS:-1234}
if [[ $PARTS == *1* ]]; then
  PARTS=1 bash $R/tools/gpu_profile_round.sh r04 > $OUT/part1.log 2>&1; tail -3 $OUT/part1.log
fi
if [[ $PARTS == *2* ]]; then
one() {  # tag K stride batch steps args...
  local t=$1 k=$2 st=$3 b=$4 n=$5; shift 5
  LCE_K=$k LCE_STRIDE=$st timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/layers/$t -o t -- \
      python $R/tools/run_one.py "$@" $n $b > $OUT/layers/$t.log 2>/dev/null
  { echo "# $t: run_one.py $* $n $b (filter ${k}x${k}, stride $st) -> HIP events: $(tail -1 $OUT/layers/$t.log)"; python3 $R/tools/steady_stats.py $OUT/layers/$t/t_kernel_trace.csv $n; } | tee -a $OUT/layer_steady_stats.txt
  rm -rf $OUT/layers/$t
}
: > $OUT/layer_steady_stats.txt
one l0_f32            3 1 256 40  56 256 f32 auto auto
one l0_int8           3 1 256 40  56 256 i8  auto auto
one l0_bitpacked      3 1 256 40  56 256 bp  auto auto
one quicknet_56x64    3 1 256 100 56 64  f32 auto auto
one quicknet_28x128   3 1 256 100 28 128 f32 auto auto
one quicknet_14x256   3 1 256 200 14 256 f32 auto auto
one quicknet_7x512    3 1 256 200 7  512 f32 auto auto
one ksplit_7x512_i8   3 1 256 200 7  512 i8  auto auto
one ksplit_7x512_s2   3 2 256 200 7  512 i8  auto auto
one strips_224x256    3 1 16  40  224 256 f32 auto auto
one strips_224x256_i8 3 1 16  40  224 256 i8  auto auto
one pointwise_56x64   1 1 256 200 56 64  i8  auto auto
one pointwise_7x512   1 1 256 200 7  512 i8  auto auto
fi
if [[ $PARTS == *3* ]]; then
for spec in "pmc_l0_f32 3 56 256 f32 256" "pmc_7x512_f32 3 7 512 f32 256" "pmc_7x512_i8 3 7 512 i8 256" "pmc_14x256_f32 3 14 256 f32 256" "pmc_224x256_f32 3 224 256 f32 16"; do
  set -- $spec
  LCE_K=$2 bash $R/tools/gpu_pmc_one.sh r04/$1 $3 $4 $5 auto auto 20 $6 > $OUT/$1.log 2>&1
  tail -2 $OUT/$1.log | cut -c1-160
done
fi
if [[ $PARTS == *4* ]]; then
cd $R && timeout 900 python bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
timeout 600 python bench.py --spinup-ms 0 --no-extra --no-cpu-baseline > $OUT/bench_no_spinup.json 2> $OUT/bench_no_spinup.err
timeout 900 python bench.py --measure-traffic --no-extra --no-cpu-baseline > $OUT/bench_measured_traffic.json 2> $OUT/bench_measured_traffic.err
fi
du -sh $OUT
