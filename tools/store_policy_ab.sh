# A/B of store cache policies (library variants from tools/build_exp.sh), interleaved on one box; see profiles/r03/store_cache_policy.txt
LCE_K=1 bash tools/abn.sh 3 "56 64 i8 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
LCE_K=1 bash tools/abn.sh 3 "28 128 i8 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
LCE_K=1 bash tools/abn.sh 3 "14 256 i8 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
LCE_K=1 bash tools/abn.sh 3 "28 128 f32 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
bash tools/abn.sh 3 "56 64 f32 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
bash tools/abn.sh 3 "28 128 f32 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
bash tools/abn.sh 3 "28 128 i8 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
bash tools/abn.sh 2 "7 512 f32 auto auto 100" base build_exp/lib_s8_18.so build_exp/lib_s8_17.so
