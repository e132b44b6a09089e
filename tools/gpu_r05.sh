#!/bin/bash
# Round 5's measurement calls, one function per call: bash tools/gpu_r05.sh <part>   (run on the GPU box by gpurun; output under gpurun_out/r05_<part>/)
cd $GRAFT_REPO_ROOT
PART=$1
OUT=gpurun_out/r05_$PART
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
*) echo "unknown part $PART"; exit 2;;
esac
