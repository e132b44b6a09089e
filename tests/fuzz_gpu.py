import importlib, sys, os
"""Extended randomized parity run on the GPU (not collected by pytest; `python tests/fuzz_gpu.py`): 360 specs -- random
geometry, wide images, 1x1 layers -- every output type and the second output against the oracle."""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, synth
import test_gpu_parity as T
amd = importlib.import_module("compute-engine_amd")
n = 0; kernels = {}
for seed in range(100, 160):
    g = synth.rng(seed)
    for k in range(6):
        spec = T._random_spec(g)
        if spec.out_h <= 0 or spec.out_w <= 0: continue
        # widen some images so that 2-D tiles and the pointwise kernel come up
        if k % 3 == 0:
            spec = O.ConvSpec(spec.batch, spec.in_h, 32 * int(g.integers(2, 5)), 64 * int(g.integers(1, 3)), spec.filter_h, spec.filter_w,
                              32 * int(g.integers(1, 5)), 1, spec.stride_h, spec.stride_w, 1, 1, spec.padding, spec.pad_values, spec.activation, spec.semantics)
        if k % 3 == 1:
            spec = O.ConvSpec(spec.batch + 2, spec.in_h + 8, spec.in_w + 8, 32 * int(g.integers(1, 9)), 1, 1, 32 * int(g.integers(1, 6)), 1, 1, 1, 1, 1,
                              spec.padding, 1, spec.activation, O.SEM_REFERENCE)
        if spec.out_h <= 0 or spec.out_w <= 0: continue
        try:
            names = T._check_all_dst(spec, seed * 10 + k, engine="auto")
        except AssertionError as e:
            print("MISMATCH", spec, e); raise
        for nm in names: kernels[nm.split(",")[0] + ("/2d" if nm.endswith("/2d") else "")] = kernels.get(nm.split(",")[0] + ("/2d" if nm.endswith("/2d") else ""), 0) + 1
        # dual output where legal
        for dst in (amd.F32, amd.I8):
            x, w, mul, bias = synth.conv_inputs(spec, seed + k, negative_mul_fraction=0.3)
            zero_pad = spec.padding == O.PADDING_SAME and spec.pad_values == 0
            if zero_pad and spec.semantics == O.SEM_OPTIMIZED and (dst == amd.I8 or spec.activation != O.ACT_NONE): continue
            zp = int(g.integers(-20, 20)); sc = float(spec.filter_h * spec.filter_w * spec.channels_in) / 40.0
            plan = amd.Bconv2dPlan(T._params(spec, dst, out_scale=sc, out_zero_point=zp)); plan.set_weights(w, mul, bias)
            xd = torch.from_numpy(x).to("cuda:0")
            y = plan.run(xd); y2, bits = plan.run_dual(xd); torch.cuda.synchronize()
            assert torch.equal(y.view(torch.uint8), y2.view(torch.uint8)), (spec, plan.kernel_name())
            assert torch.equal(bits, amd.bitpack(y, zp if dst == amd.I8 else 0)), (spec, plan.kernel_name())
        n += 1
print("fuzzed", n, "specs;", kernels)
