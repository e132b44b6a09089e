cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" $R/build_exp/lib_noskew.so; do
  for cfg in "56 256" "14 256"; do
    rm -rf /tmp/pl
    LCE_HIP_LIBRARY=$lib timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pl -o p -- python $R/tools/run_one.py $cfg f32 auto auto 20 > /dev/null 2>&1
    python3 - "$cfg ${lib:-base}" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob('/tmp/pl/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if "bconv2d_stream" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c,v in acc.items():
    print(sys.argv[1][-40:], c, "%.0f per launch" % (sum(v.values())/len(v)))
PY
  done
done
