#!/bin/bash
# Round-3 evidence in one GPU call (writes gpurun_out/r03/...):
#   1-3  tools/gpu_profile_round.sh: rocprofv3 kernel stats of the whole bench, of every `extra` layer (one process each),
#        PMC passes of the L0 float kernel (now the streaming kernel)
#   4    PMC passes (seven separate rocprofv3 --pmc runs each) of the other kernels bench.py's `extra` names
#   5    the streaming kernel's tile-step timelines (needs build_exp/lib_sph.so: tools/build_exp.sh sph:"-DLCE_STREAM_PHASES")
#   6    the default bench line, un-profiled; the same without the clock spin-up; the same with --measure-traffic
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
PARTS=${PARTS:-123} bash $R/tools/gpu_profile_round.sh r03
for spec in "pmc_14x256_f32 3 14 256 f32" "pmc_28x128_f32 3 28 128 f32" "pmc_7x512_f32 3 7 512 f32" "pmc_56x64_f32 3 56 64 f32" \
            "pmc_pw_56x64_i8 1 56 64 i8" "pmc_pw_28x128_i8 1 28 128 i8" "pmc_pw_14x256_i8 1 14 256 i8" "pmc_pw_7x512_i8 1 7 512 i8" \
            "pmc_pw_28x128x256_s2_i8 1 28 128x256 i8 2" "pmc_l0_i8 3 56 256 i8" "pmc_l0_bp 3 56 256 bp"; do
  set -- $spec
  LCE_K=$2 LCE_STRIDE=${6:-1} bash $R/tools/gpu_pmc_one.sh r03/$1 $3 $4 $5 auto auto 20 > $OUT/$1.log 2>&1
  tail -2 $OUT/$1.log | cut -c1-200
done
if [ -f $R/build_exp/lib_sph.so ]; then
  for a in "56 256x256 f32" "56 256x256 i8" "56 256x256 bp" "14 256x256 f32"; do
    LCE_HIP_LIBRARY=$R/build_exp/lib_sph.so timeout 120 python $R/tools/stream_phases.py $a 2>&1 | grep -v amdgpu.ids >> $OUT/stream_phases.txt
    echo >> $OUT/stream_phases.txt
  done
fi
( echo "== AGPR accumulators (the empty-asm pin forces copies: ignore)"; timeout 200 $R/tools/probes/mfma_gap_a; echo "== VGPR accumulators (-amdgpu-mfma-vgpr-form), as the streaming kernel is built"; timeout 200 $R/tools/probes/mfma_gap_v ) > $OUT/probe_mfma_gap.txt 2>&1
cd $R && timeout 900 python bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
# the same command without the untimed clock spin-up (README: its share of the headline) and with the traffic measured in the run
timeout 600 python bench.py --spinup-ms 0 --no-extra --no-cpu-baseline > $OUT/bench_no_spinup.json 2> $OUT/bench_no_spinup.err
timeout 900 python bench.py --measure-traffic --no-extra --no-cpu-baseline > $OUT/bench_measured_traffic.json 2> $OUT/bench_measured_traffic.err
timeout 300 python tools/stream_check.py 2>&1 | grep -v amdgpu.ids > $OUT/stream_vs_block_gemm.txt
du -sh $OUT
