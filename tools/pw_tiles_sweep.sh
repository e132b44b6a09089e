cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() {   # <H> <CinxCout> <dst> <stride> <opts>
  rm -rf /tmp/pwt
  LCE_K=1 LCE_STRIDE=$4 LCE_OPTS=$5 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pwt -o t -- python $R/tools/run_one.py $1 $2 $3 auto auto 200 > /dev/null 2>&1
  python3 - "$1 $2 $3 s$4 $5" <<'PY'
import csv,sys,glob
for f in glob.glob('/tmp/pwt/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "bconv2d" in n:
            print("%-70s %-36s avg %7.2f us min %7.2f" % (sys.argv[1], n.split("(")[0].replace("void lce::","")[:36], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
}
for cfg in "28 128x128" "14 256x256" "7 512x512"; do
  set -- $cfg
  for ch in 64 128; do
    [ $2 = 512x512 ] && [ $ch = 128 ] && continue
    for k in 1 2 3 4 6; do
      one $1 $2 i8 1 engine=pointwise,pointwise_channels=$ch,pointwise_tiles=$k
    done
  done
done
