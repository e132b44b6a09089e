for spec in "14 256 f32" "7 512 f32" "14 256 i8" "28 128 f32" "56 256 f32" "14 256 bp" "7 512 bp"; do
  bash tools/abn.sh 4 "$spec stream auto 300" base build_exp/ctx0/liblce_hip.so
done
