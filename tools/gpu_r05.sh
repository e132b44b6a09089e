#!/bin/bash
# Round 5's measurement calls, one function per call: bash tools/gpu_r05.sh <part>   (run on the GPU box by gpurun; output under gpurun_out/r05_<part>/)
cd $GRAFT_REPO_ROOT
PART=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_$PART
mkdir -p $OUT
IL4="engine=stream,stream_rows=4,stream_interleave=1"
IL8="engine=stream,stream_rows=8,stream_interleave=1"
case $PART in
window)
  # (1) the store stream alone, by write window (probe); (2) the same question IN the kernel: interleaved runs of 4- / 8- / 28-row segments vs one image per block
  { ./tools/probes/store_overlap | grep stream_pattern; } > $OUT/probe.txt 2>&1
  {
    python tools/ab_opts.py 56 256x256 f32 256 6 20 base rows4il:$IL4 rows8il:$IL8 rows28il:engine=stream,stream_rows=28,stream_interleave=1 rows4:engine=stream,stream_rows=4 rows8:engine=stream,stream_rows=8
    python tools/ab_opts.py 56 256x256 i8 256 4 20 base rows4il:$IL4 rows8il:$IL8
    python tools/ab_opts.py 56 256x256 bp 256 4 20 base rows4il:$IL4 rows8il:$IL8
    python tools/ab_opts.py 56 64x64 f32 256 5 40 base stream56:engine=stream rows4il:$IL4 rows8il:$IL8 rows4:engine=stream,stream_rows=4
    python tools/ab_opts.py 224 256x256 f32 16 5 20 base il:engine=stream,stream_interleave=1 rows28il:engine=stream,stream_rows=28,stream_interleave=1 rows8il:$IL8
    python tools/ab_opts.py 28 128x128 f32 256 5 60 base rows4il:$IL4 rows14il:engine=stream,stream_rows=14,stream_interleave=1
    python tools/ab_opts.py 14 256x256 f32 256 5 100 base
  } > $OUT/ab.txt 2>&1
  ;;
lowk)
  # where does 28x28x128's gain with interleaved 4-row segments come from -- the segments or the interleave?  and the other low-K layers
  R4="engine=stream,stream_rows=4"; R7="engine=stream,stream_rows=7"; R14="engine=stream,stream_rows=14"; R2="engine=stream,stream_rows=2"
  {
    python tools/ab_opts.py 28 128x128 f32 256 6 60 base rows4:$R4 rows4il:$R4,stream_interleave=1 rows7il:$R7,stream_interleave=1 rows14:$R14 rows14il:$R14,stream_interleave=1 rows2il:$R2,stream_interleave=1
    python tools/ab_opts.py 28 128x128 i8 256 4 60 base rows4:$R4 rows4il:$R4,stream_interleave=1 rows14il:$R14,stream_interleave=1
    python tools/ab_opts.py 28 128x128 bp 256 4 60 base rows4:$R4 rows4il:$R4,stream_interleave=1 rows14il:$R14,stream_interleave=1
    python tools/ab_opts.py 56 64x64 f32 256 4 40 base stream56:engine=stream rows8:engine=stream,stream_rows=8 rows14:$R14 rows14il:$R14,stream_interleave=1 rows28il:engine=stream,stream_rows=28,stream_interleave=1
    python tools/ab_opts.py 56 64x64 i8 256 4 40 base rows8il:engine=stream,stream_rows=8,stream_interleave=1 rows14il:$R14,stream_interleave=1
    python tools/ab_opts.py 14 256x256 f32 256 4 100 base rows7:$R7 rows7il:$R7,stream_interleave=1 rows2il:$R2,stream_interleave=1
    python tools/ab_opts.py 56 64x128s2 f32 256 4 60 base rows4il:$R4,stream_interleave=1 rows14il:$R14,stream_interleave=1
    python tools/ab_opts.py 28 128x256s2 f32 256 4 100 base rows7il:$R7,stream_interleave=1 rows2il:$R2,stream_interleave=1
    python tools/ab_opts.py 56 256x256 f32 256 3 20 base rows4il:engine=stream,stream_rows=4,stream_interleave=1 rows8il:engine=stream,stream_rows=8,stream_interleave=1
    ./tools/probes/store_overlap | grep stream_pattern
  } > $OUT/ab.txt 2>&1
  ;;
phases)
  # block timelines (LCE_STREAM_PHASES build: bash tools/build_exp.sh sph:"-DLCE_STREAM_PHASES") of the low-K float layer with whole images, 4-row
  # segments, and interleaved 4-row segments
  export LCE_HIP_LIBRARY=$PWD/build_exp/lib_sph.so
  {
    for o in "" "stream_rows=4" "stream_rows=4,stream_interleave=1"; do
      echo "## 28x28x128 f32 [$o]"; LCE_OPTS=$o python tools/stream_phases.py 28 128x128 f32
    done
    for o in "" "stream_rows=14,stream_interleave=1"; do
      echo "## 56x56x64 f32 [$o]"; LCE_OPTS=$o python tools/stream_phases.py 56 64x64 f32
    done
    echo "## 14x14x256 f32"; python tools/stream_phases.py 14 256x256 f32
  } > $OUT/phases.txt 2>&1
  ;;
ws1)
  # the weight-streaming kernel against the planner's choice (byte comparison first, then interleaved timings)
  {
    for d in f32 i8 bp; do
      python tools/ab_opts.py 14 256x256 $d 256 5 100 base wstream:engine=wstream
      python tools/ab_opts.py 7 512x512 $d 256 5 100 base wstream:engine=wstream
    done
    python tools/ab_opts.py 14 256x512s2 f32 256 4 100 base wstream:engine=wstream
    python tools/ab_opts.py 7 512x512s2 f32 256 4 100 base wstream:engine=wstream
    python tools/ab_opts.py 28 128x256s2 f32 256 4 100 base wstream:engine=wstream
    python tools/ab_opts.py 28 128x128 f32 256 4 60 base wstream:engine=wstream rows4il:engine=stream,stream_rows=4,stream_interleave=1
    python tools/ab_opts.py 14 256x256 f32 64 4 100 base wstream:engine=wstream
    python tools/ab_opts.py 14 256x256 f32 512 4 100 base wstream:engine=wstream
  } > $OUT/ab.txt 2>&1
  ;;
ws2)
  # the weight-streaming kernel with its schedule pinned: parts of at most 4 / 3 / 2 pixel blocks (when do the stores start?), groups of 1 / 2 / 4 images
  W="engine=wstream"
  {
    python tools/ab_opts.py 14 256x256 f32 256 5 100 base ws4:$W ws3:$W,wstream_blocks=3 ws2:$W,wstream_blocks=2 ws1:$W,wstream_blocks=1
    python tools/ab_opts.py 14 256x256 i8 256 5 100 base ws4:$W ws3:$W,wstream_blocks=3 ws2:$W,wstream_blocks=2
    python tools/ab_opts.py 14 256x256 bp 256 5 100 base ws4:$W ws3:$W,wstream_blocks=3 ws2:$W,wstream_blocks=2
    python tools/ab_opts.py 7 512x512 f32 256 5 100 base ws:$W i1b2:$W,wstream_images=1 i2b4:$W,wstream_images=2 i4b4:$W,wstream_images=4 i4b3:$W,wstream_images=4,wstream_blocks=3 i2b2:$W,wstream_images=2,wstream_blocks=2
    python tools/ab_opts.py 7 512x512 i8 256 5 100 base ws:$W i2b4:$W,wstream_images=2 i4b4:$W,wstream_images=4 i2b2:$W,wstream_images=2,wstream_blocks=2
    python tools/ab_opts.py 7 512x512 bp 256 5 100 base ws:$W i2b4:$W,wstream_images=2 i4b4:$W,wstream_images=4
    python tools/ab_opts.py 14 256x512s2 f32 256 4 100 base ws:$W i2:$W,wstream_images=2
  } > $OUT/ab.txt 2>&1
  ;;
paced)
  # the bank's loads paced inside the first block step (in-tree) vs the previous build (build_exp/lib_prev.so), same box, alternating processes
  {
    for spec in "14 256x256 f32 256 3 100" "14 256x256 i8 256 3 100" "14 256x256 bp 256 3 100" "7 512x512 f32 256 3 100" "7 512x512 i8 256 3 100" \
                "28 128x128 f32 256 3 60" "56 256x256 f32 256 3 20" "56 256x256 bp 256 3 20" "28 128x256s2 f32 256 3 100" "224 256x256 f32 16 3 20"; do
      bash tools/ab_libs.sh 3 "$spec base" build_exp/lib_prev.so base
    done
  } > $OUT/ab.txt 2>&1
  ;;
phases2)
  export LCE_HIP_LIBRARY=$PWD/build_exp/lib_sph.so
  {
    echo "## 14x14x256 f32 stream (paced bank)"; python tools/stream_phases.py 14 256x256 f32
    echo "## 7x7x512 f32 stream (paced bank)"; python tools/stream_phases.py 7 512x512 f32
    echo "## 14x14x256 f32 wstream"; LCE_OPTS=engine=wstream python tools/stream_phases.py 14 256x256 f32
    echo "## 14x14x256 i8 wstream"; LCE_OPTS=engine=wstream python tools/stream_phases.py 14 256x256 i8
    echo "## 14x14x256 bp wstream"; LCE_OPTS=engine=wstream python tools/stream_phases.py 14 256x256 bp
    echo "## 56x56x256 f32 stream"; python tools/stream_phases.py 56 256x256 f32
  } > $OUT/phases.txt 2>&1
  ;;
ws3)
  # weight-streaming kernel with the tile-major weight image (in-tree) vs the K-major one (build_exp/lib_prev.so), same box
  W="engine=wstream"
  {
    for spec in "14 256x256 f32 256 3 100" "14 256x256 i8 256 3 100" "14 256x256 bp 256 3 100" "7 512x512 f32 256 3 100" "7 512x512 i8 256 3 100" "7 512x512 bp 256 3 100" "14 256x512s2 f32 256 3 100"; do
      bash tools/ab_libs.sh 2 "$spec base ws:$W ws3:$W,wstream_blocks=3 ws2:$W,wstream_blocks=2" build_exp/lib_prev.so base
    done
  } > $OUT/ab.txt 2>&1
  ;;
sweep)
  # every candidate kernel on the planner's calibration grid (tools/engine_sweep.py)
  rm -f $OUT/engine_sweep.jsonl
  python tools/engine_sweep.py $OUT/engine_sweep.jsonl > $OUT/log.txt 2>&1
  ;;
sweepq)
  rm -f $OUT/engine_sweep.jsonl
  python tools/engine_sweep.py $OUT/engine_sweep.jsonl --quick 14x256x256 7x512x512 > $OUT/log.txt 2>&1
  ;;
evidence)
  # the round's rocprofv3 evidence in one call (as tools/sessions/gpu_evidence_r04.sh): PARTS=1234 selects
  R=$GRAFT_REPO_ROOT
  mkdir -p $OUT/layers
  export TMPDIR=/tmp
  cd /tmp
  PARTS=${PARTS:-1234}
  if [[ $PARTS == *1* ]]; then
    PARTS=1 bash $R/tools/gpu_profile_round.sh r05_evidence > $OUT/part1.log 2>&1; tail -3 $OUT/part1.log
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
  one wstream_14x256_i8 3 1 256 200 14 256 i8  auto auto
  one wstream_14x256_bp 3 1 256 200 14 256 bp  auto auto
  one wstream_14x256x512_s2_i8 3 2 256 200 14 256x512 i8 auto auto
  one ksplit_7x512_i8   3 1 256 200 7  512 i8  auto auto
  one strips_224x256    3 1 16  40  224 256 f32 auto auto
  one small_batch_14x256_b16 3 1 16 200 14 256 f32 auto auto
  fi
  if [[ $PARTS == *3* ]]; then
  for spec in "pmc_l0_f32 3 1 56 256 f32 256" "pmc_28x128_f32 3 1 28 128 f32 256" "pmc_14x256_i8 3 1 14 256 i8 256" "pmc_7x512_f32 3 1 7 512 f32 256" "pmc_14x256x512_s2_i8 3 2 14 256x512 i8 256"; do
    set -- $spec
    LCE_K=$2 LCE_STRIDE=$3 bash $R/tools/gpu_pmc_one.sh r05_evidence/$1 $4 $5 $6 auto auto 20 $7 > $OUT/$1.log 2>&1
    tail -2 $OUT/$1.log | cut -c1-160
  done
  fi
  if [[ $PARTS == *4* ]]; then
  cd $R && timeout 900 python bench.py --extra-json $OUT/bench_extra.json > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
  timeout 600 python bench.py --spinup-ms 0 --no-extra --no-cpu-baseline > $OUT/bench_no_spinup.json 2> $OUT/bench_no_spinup.err
  timeout 900 python bench.py --measure-traffic --no-extra --no-cpu-baseline > $OUT/bench_measured_traffic.json 2> $OUT/bench_measured_traffic.err
  fi
  du -sh $OUT
  ;;
box)
  # the headline on this box (bench_box_spread.txt: one call per box)
  python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
  python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('ms_per_step %.4f  roofline.frac %.4f  p10/p50/p90 %s  kernel %s' % (d['ms_per_step'], d['roofline']['frac'], d.get('ms_per_step_p10_p50_p90'), d['kernel']))" | tee -a $OUT/line.txt
  ;;
stacks)
  # the three stacks of bench.py with and without the weight-streaming kernel among the planner's candidates (LCE_PLAN_NO_WSTREAM=1), alternating, one box
  for r in 1 2 3; do
    for v in "" 1; do
      if [ -z "$v" ]; then tag=with_wstream; unset LCE_PLAN_NO_WSTREAM; else tag=without_wstream; export LCE_PLAN_NO_WSTREAM=1; fi
      python bench.py --no-cpu-baseline --extra-json "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('$tag', 'L0 %.4f' % d['ms_per_step'], {k: e[k] for k in e if 'layers' in k or k.startswith('quicknet_1') or k.startswith('quicknet_7')})" | tee -a $OUT/stacks.txt
    done
  done
  ;;
i8floor)
  # int8 epilogues with floor(y + 0.5) rounding (one v_cvt_rpi_i32_f32; planner-proven per plan) vs the round-half-away sequence: tests, per-layer A/B, stacks A/B
  python -m pytest tests/test_gpu_parity.py -q -x -k "int8 or i8 or ties" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
  E="int8_rounding=exact"
  {
    for spec in "56 64x64 i8 256 5 40" "28 128x128 i8 256 5 60" "14 256x256 i8 256 5 100" "7 512x512 i8 256 5 100" "56 256x256 i8 256 4 20" "56 64x128s2 i8 256 5 60" "28 128x256s2 i8 256 5 100" "14 256x512s2 i8 256 5 100"; do
      LCE_PLAN_DEBUG=1 python tools/ab_opts.py $spec base exact:$E 2>&1 | grep -v "us$\|estimate"
    done
    for spec in "56 64x64 i8 256 5 40" "56 256x256 i8 256 5 20" "28 128x128 i8 256 5 60" "14 256x256 i8 256 5 100"; do
      LCE_K=1 LCE_PLAN_DEBUG=1 python tools/ab_opts.py $spec base exact:$E 2>&1 | grep -v "us$\|estimate"
    done
  } > $OUT/ab.txt 2>&1
  for r in 1 2 3; do
    for v in "" 1; do
      if [ -z "$v" ]; then tag=floor; unset LCE_PLAN_INT8_EXACT; else tag=exact; export LCE_PLAN_INT8_EXACT=1; fi
      python bench.py --no-cpu-baseline --extra-json "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('$tag', 'L0 %.4f' % d['ms_per_step'], {k: e[k] for k in e if 'layers' in k or k.startswith('quicknet_1') or k.startswith('quicknet_7')})" | tee -a $OUT/stacks.txt
    done
  done
  ;;
occ2)
  # two blocks of the streaming kernel per CU on 64-input-channel layers (launch bounds (256, 2): <= 256 registers per wave): in-tree vs build_exp/lib_base.so (one block per CU)
  {
    for spec in "56 64x64 i8 256 3 40" "56 64x64 f32 256 3 40" "56 64x64 bp 256 3 40" "56 64x128s2 i8 256 3 60" "56 64x128s2 f32 256 3 60" "112 64x64 i8 64 3 20"; do
      bash tools/ab_libs.sh 2 "$spec one:engine=stream two:engine=stream,compute_units=512 three:engine=stream,compute_units=768" build_exp/lib_base.so base
    done
  } > $OUT/ab.txt 2>&1
  ;;
pkf32)
  # the packed-f32 output transform (v_pk_mul_f32 / v_pk_add_f32, -DLCE_STREAM_PK_F32) on the LOW-K streaming instances, where the epilogue is
  # not in an MFMA's shadow (round-4 review, item 3): build_exp/lib_pkf32.so vs the same single-translation-unit build without the flag
  {
    for spec in "56 64x64 f32 256 3 40" "28 128x128 f32 256 3 60" "56 64x64 i8 256 3 40" "56 64x128s2 f32 256 3 60" "56 256x256 f32 256 3 20"; do
      bash tools/ab_libs.sh 2 "$spec base stream:engine=stream" build_exp/lib_unity.so build_exp/lib_pkf32.so
    done
  } > $OUT/ab.txt 2>&1
  ;;
sweepi8)
  # the int8 rows of the calibration grid again, after the one-instruction int8 forms changed the int8 epilogues' cost
  python tools/engine_sweep.py $OUT/engine_sweep_i8.jsonl --dst=i8 > $OUT/log.txt 2>&1; tail -2 $OUT/log.txt | cut -c1-200
  ;;
*) echo "unknown part $PART"; exit 2;;
esac
