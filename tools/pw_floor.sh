cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kv in "" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do
  rm -rf /tmp/lf; env $kv rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lf -o t -- $R/tools/probes/launch_floor > /dev/null 2>&1
  echo "== launch_floor $kv"
  python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('/tmp/lf/**/t_kernel_trace.csv', recursive=True):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in sorted(d):
        v=sorted(d[k]); print("%-22s grid %8d  median %6.2f us  min %6.2f" % (k[0], k[1], v[len(v)//2]/1e3, v[0]/1e3))
PY
done
one() {
  rm -rf /tmp/pwt
  env $6 LCE_K=1 LCE_STRIDE=$4 LCE_OPTS=$5 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pwt -o t -- python $R/tools/run_one.py $1 $2 $3 auto auto 200 > /dev/null 2>&1
  python3 - "$1 $2 $3 s$4 $5 $6" <<'PY'
import csv,sys,glob
for f in glob.glob('/tmp/pwt/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "bconv2d" in n:
            print("%-100s avg %7.2f us min %7.2f" % (sys.argv[1][-100:], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
}
for cfg in "14 256x256" "7 512x512"; do
  set -- $cfg
  one $1 $2 i8 1 engine=pointwise,pointwise_channels=64 A=1
  one $1 $2 i8 1 engine=pointwise,pointwise_channels=64 HIP_FORCE_DEV_KERNARG=0
  one $1 $2 i8 1 engine=pointwise,pointwise_channels=64 HIP_FORCE_DEV_KERNARG=1
  one $1 $2 i8 1 engine=pointwise,pointwise_channels=64 LCE_HIP_LIBRARY=$R/build_exp/lib_pwns.so
done
