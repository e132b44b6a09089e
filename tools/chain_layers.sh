# per-layer kernel durations INSIDE the device-resident chains (rocprofv3 --kernel-trace; tools/graph_gaps.py report)
# usage: gpurun -- 'bash tools/chain_layers.sh > gpurun_out/chain_layers.txt'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in "quicknet 16" "birealnet 12"; do
  set -- $st
  rm -rf /tmp/cl_$1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/cl_$1 -o t -- python $R/tools/graph_gaps.py run eager 50 $1 2>/dev/null | grep "per chain"
  python $R/tools/graph_gaps.py report $(find /tmp/cl_$1 -name "t_kernel_trace.csv" | head -1) 50 $2
done
