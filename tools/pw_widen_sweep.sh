# rocprofv3 kernel durations of the small / deep 1x1 layers: block GEMM vs the pointwise kernel at 32 / 64 / 128 channels per
# block (round 3: 512 input channels, strides, the small-launch rule).  usage: gpurun -- 'bash tools/pw_widen_sweep.sh > gpurun_out/pw_widen.txt'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() {   # <label> <H> <CinxCout> <dst> <stride> <opts>
  rm -rf /tmp/pwt
  LCE_K=1 LCE_STRIDE=$5 LCE_OPTS=$6 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pwt -o t -- python $R/tools/run_one.py $2 $3 $4 auto auto 200 > /dev/null 2>&1
  python3 - "$1 $2 $3 $4 s$5 $6" <<'PY'
import csv,sys,glob
for f in glob.glob('/tmp/pwt/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "bconv2d" in n:
            print("%-44s %-44s calls %4s avg %7.2f us min %7.2f" % (sys.argv[1], n.split("(")[0].replace("void lce::","")[:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
}
for d in ${PW_DST:-i8 f32 bp}; do
  for cfg in "28 128x128 1" "14 256x256 1" "7 512x512 1" "56 64x64 1" "56 64x128 2" "28 128x256 2" "14 256x512 2" "56 256x256 1"; do
    set -- $cfg
    one gemm $1 $2 $d $3 engine=direct
    for ch in 32 64 128; do
      case "$2" in 512x*|*x32) [ $ch = 128 ] && continue;; esac
      one pw$ch $1 $2 $d $3 engine=pointwise,pointwise_channels=$ch
    done
    one auto $1 $2 $d $3 engine=auto
  done
done
