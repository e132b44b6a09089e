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
*) echo "unknown part $PART"; exit 2;;
esac
