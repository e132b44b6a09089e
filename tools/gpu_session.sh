#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprof kernel trace, microbench, variant sweep.
# Usage (from the repo root on the GPU box):  bash tools/gpu_session.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu --maxfail=15 -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
echo "== valu peak"
timeout 120 ./tools/valu_peak > $OUT/valu_peak.jsonl 2>&1; cat $OUT/valu_peak.jsonl
echo "== bench"
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== sweep"
timeout 900 python tools/sweep.py > $OUT/sweep.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"; cat $OUT/sweep.jsonl | head -80; tail -3 $OUT/sweep.err
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err ); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
find $OUT/prof -name "*kernel_trace*.csv" -size +2M -delete
du -sh $OUT
