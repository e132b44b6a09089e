# Round 6: the float rows' store cache policy (product: non-temporal) on SUB-stacks of QuickNet with their own buffers -- the eight
# single-round layers (14x14x256 x4, 7x7x512 x4: 308 MB of outputs), the eight large ones, and the whole stack -- interleaved, one box.
for r in 1 2 3; do
  for part in "8 16" "0 8" "0 16"; do
    for lib in product plain sc1; do
      if [ "$lib" = product ]; then L=""; else L=$PWD/build_exp/$lib/liblce_hip.so; fi
      echo "lib=$lib $(LCE_HIP_LIBRARY=$L python tools/substack_ab.py quicknet $part 150 2>/dev/null | tail -1)"
    done
  done
done
