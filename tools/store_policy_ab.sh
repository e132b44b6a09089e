# A/B of store cache policies (library variants from tools/build_exp.sh), interleaved on one box; see profiles/r03/store_cache_policy.txt
# words: base = bitpacked / sign words written through (sc1), wplain = ordinary write-back stores; sw16 = the streaming kernel's word stores with sc1
bash tools/abn.sh 3 "56 64 bp auto auto 100" base build_exp/lib_wplain.so
bash tools/abn.sh 3 "28 128 bp auto auto 100" base build_exp/lib_wplain.so
LCE_K=1 bash tools/abn.sh 3 "56 64 bp auto auto 100" base build_exp/lib_wplain.so
LCE_K=1 bash tools/abn.sh 3 "28 128 bp auto auto 100" base build_exp/lib_wplain.so
bash tools/abn.sh 3 "56 256 bp auto auto 60" base build_exp/lib_sw16.so
bash tools/abn.sh 3 "14 256 bp auto auto 100" base build_exp/lib_sw16.so
for r in 1 2 3; do
  for lib in "" build_exp/lib_wplain.so build_exp/lib_sw16.so; do
    echo "chain lib=${lib:-base} $(LCE_HIP_LIBRARY=${lib:+$PWD/$lib} python tools/graph_gaps.py run eager 100 2>/dev/null | grep 'per chain')"
  done
done
