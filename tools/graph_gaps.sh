# eager launches vs HIP-graph replay of QuickNet's 16-layer chain under rocprofv3 --kernel-trace (see tools/graph_gaps.py)
# usage: gpurun -- 'bash tools/graph_gaps.sh > gpurun_out/graph_gaps.txt'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in eager graph; do
  rm -rf /tmp/gg_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gg_$mode -o t -- python $R/tools/graph_gaps.py run $mode 50 2>/dev/null | grep "per chain"
  python $R/tools/graph_gaps.py report $(find /tmp/gg_$mode -name "t_kernel_trace.csv" | head -1) 50 16
done
echo "--- without the profiler"
for mode in eager graph; do python $R/tools/graph_gaps.py run $mode 200 2>/dev/null | grep "per chain"; done
