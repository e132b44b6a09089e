export TMPDIR=/tmp
mkdir -p gpurun_out/r04e
for a in "7 512x512 f32" "7 512x512 bp" "14 256x256 f32" "14 512x512 f32"; do
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_phases.so timeout 300 python tools/stream_phases.py $a 2>&1 | grep -v amdgpu.ids; echo
done | tee gpurun_out/r04e/stream_phases.txt
