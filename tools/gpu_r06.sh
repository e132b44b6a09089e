#!/bin/bash
# Round-6 GPU sessions (one gpurun call each):  bash tools/gpu_r06.sh <step> ...
#   new      the round's new parity tests alone, timed
#   tests    the whole -m gpu suite (what the driver runs at round end), timed
#   smoke    __graft_entry__.smoke()
#   bench    bench.py as the driver runs it (--steps 20 --warmup 5) -> gpurun_out/r06/bench_default_run.json (+ bench_extra.json)
#   prof     rocprofv3 --kernel-trace --stats over the same command (no extras) -> gpurun_out/r06/prof
OUT=gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
  echo "== $step"
  case $step in
    new)   ( time timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -x -q --durations=8 ) > $OUT/pytest_new.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest_new.log ;;
    tests) ( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest_gpu.log ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?"; tail -4 $OUT/smoke.log ;;
    bench) ( time timeout 1200 python bench.py --steps 20 --warmup 5 --extra-json $OUT/bench_extra.json ) > $OUT/bench_default_run.json 2> $OUT/bench.err; echo "rc=$?"; wc -c $OUT/bench_default_run.json; cat $OUT/bench_default_run.json; tail -5 $OUT/bench.err ;;
    bench1) ( time timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --one-operand-set ) > $OUT/bench_one_operand_set.json 2>> $OUT/bench.err; cat $OUT/bench_one_operand_set.json ;;
    sweep) ( time timeout 2400 python tools/planner_regret.py --remeasure $OUT/engine_sweep_r06_box${BOX:-1}.jsonl ) > $OUT/planner_regret_box${BOX:-1}.txt 2>&1; echo "rc=$?"; tail -15 $OUT/planner_regret_box${BOX:-1}.txt ;;
    abl)   # single-round float layers: what each part of the weight-stationary kernel costs (ablation builds of tools/build_abl.sh; the
           # ablated variants' results are wrong by construction, only the clock is read), interleaved on this box
           for spec in "14 256 f32" "7 512 f32" "14 256 i8" "7 512 i8"; do
             bash tools/abn.sh 3 "$spec stream auto 300" base build_exp/nofrag/liblce_hip.so build_exp/noepi/liblce_hip.so build_exp/noprod/liblce_hip.so build_exp/nobank/liblce_hip.so
           done > $OUT/single_round_ablations.txt 2>&1; cat $OUT/single_round_ablations.txt ;;
    traffic) ( cd /tmp && timeout 900 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --measure-traffic > $GRAFT_REPO_ROOT/$OUT/bench_measured_traffic.json 2>> $GRAFT_REPO_ROOT/$OUT/bench.err ); echo "rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_measured_traffic.json'));print(json.dumps(d['roofline'],indent=1))" ;;
    dual)  python tools/dual_cost.py all > $OUT/dual_cost.txt 2>&1; cat $OUT/dual_cost.txt ;;
    prof)  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err ); echo "rc=$?"
           for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/bench_kernel_stats.csv; done
           for f in $(find $OUT/prof -name "*kernel_trace*.csv" | head -1); do python tools/steady_stats.py $f 20 > $OUT/layer_steady_stats.txt 2>&1; cat $OUT/layer_steady_stats.txt | tail -12; done
           find $OUT/prof -name "*kernel_trace*.csv" -size +2M -delete ;;
    *) echo "unknown step $step" ;;
  esac
done
du -sh $OUT
