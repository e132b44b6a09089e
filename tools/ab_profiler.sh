BENCH_ARGS=${BENCH_ARGS:---steps 20 --warmup 5}
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  a=$(python $R/bench.py $BENCH_ARGS --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p10_p50_p90'])")
  rm -rf /tmp/abp; b=$(rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o t -- python $R/bench.py $BENCH_ARGS --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_p10_p50_p90'])")
  c=$(python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/abp/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'bconv2d_mfma' in r['Name']: print('rocprof avg %.1f min %.1f max %.1f us calls %s' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Calls']))
PY
)
  echo "unprofiled: $a | profiled (bench own events): $b | $c"
done
