#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command; prints the per-kernel stats.
# Usage: bash tools/gpu_trace.sh <tag> <python args...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python "$@" > $OUT/stdout.log 2> $OUT/stderr.log
echo "rc=$?"
python3 - "$OUT/t_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].split("(")[0].replace("void ","")
    if "at::native" in n or "rocclr" in n: continue
    print("%-44s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (n[:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -f $OUT/t_kernel_trace.csv
