# rocprofv3 kernel durations of the 1x1 layers: block GEMM (auto before the pointwise rule) vs engine=pointwise
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "56 64" "28 128" "14 256" "56 256" "56 64x128" "7 512"; do
  for d in i8 f32 bp; do
    for e in direct pointwise; do
      rm -rf /tmp/pwt
      LCE_K=1 LCE_OPTS=engine=$e timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pwt -o t -- python $R/tools/run_one.py $cfg $d auto auto 50 > /dev/null 2>&1
      python3 - "$cfg $d $e" <<'PY'
import csv,sys,glob
for f in glob.glob('/tmp/pwt/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "bconv2d" in n:
            print("%-22s %-40s calls %3s avg %7.2f us min %7.2f" % (sys.argv[1], n.split("(")[0].replace("void lce::","")[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
    done
  done
done
