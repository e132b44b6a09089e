#!/usr/bin/env python
"""run vs run_dual (the second, sign-word output from the same epilogue) of one layer, per engine: HIP-event timings.
usage: dual_check.py HW CIN[xCOUT] f32|i8 [engine ...]      env LCE_STRIDE"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import synthetic_layers as SL
amd = importlib.import_module("compute-engine_amd")
hw = int(sys.argv[1]); cin, cout = (int(v) for v in sys.argv[2].split("x")) if "x" in sys.argv[2] else (int(sys.argv[2]),) * 2
dst = {"f32": amd.F32, "i8": amd.I8}[sys.argv[3]]
engines = sys.argv[4:] or ["auto", "direct"]
st = int(os.environ.get("LCE_STRIDE", "1"))
layer = SL.Layer(256, hw, hw, cin, 3, 3, cout, stride=st, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
def timed(fn, n=300):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        for _ in range(16): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for eng in engines:
    plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3)); plan.set_weights(w, mul, bias, thr); plan.set_option("engine", eng)
    y = plan.run(x); y2, bits = plan.run_dual(x)
    a = timed(lambda: plan.run(x, y)); b = timed(lambda: plan.run_dual(x, y2, bits))
    print("%-8s %-46s run %.2f us   run_dual %.2f us" % (eng, plan.kernel_name(), a, b), flush=True)
