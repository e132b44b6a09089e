# Store cache policies on CHAINS of layers with their own buffers (QuickNet: 16 float layers, 1.5 GB of outputs; config 5: 12 int8
# layers) -- the repeated launches of ONE layer that tools/abn.sh times rewrite one buffer, which the 256 MB Infinity Cache absorbs
# when it allocates on write (sc1) and does not when it does not (nt): the isolated gain of sc1 float rows was that artefact.
#   new = float block-GEMM rows sc1, int8 rows sc1, pointwise float rows sc1;  f32nt = float block-GEMM rows nt, rest as new;
#   i8old = int8 rows plain, rest as new;  oldpol = round-2 policy (float nt, int8 plain)
for r in 1 2 3; do
  for stack in quicknet birealnet; do
    for lib in new f32nt i8old oldpol; do
      echo "lib=$lib $(LCE_HIP_LIBRARY=$PWD/build_exp/lib_$lib.so python tools/graph_gaps.py run eager 100 $stack 2>/dev/null | grep 'per chain')"
    done
  done
done
