cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pointwise" 2>&1 | tail -3
for cfg in "56 64" "28 128" "14 256" "7 512" "56 256" "56 64x128"; do
  for d in i8 f32 bp; do
    a=$(LCE_K=1 python tools/run_one.py $cfg $d auto auto 100 2>/dev/null | tail -1)
    b=$(LCE_K=1 LCE_OPTS=engine=pointwise python tools/run_one.py $cfg $d auto auto 100 2>/dev/null | tail -1)
    echo "$cfg $d | $a | $b"
  done
done
