#!/usr/bin/env python
"""Strided 3x3 layers with BITPACKED output, streaming kernel vs block GEMM (HIP events, batch 256): a measurement for the auto rule
(lce_plan.cpp, stream_candidate), which leaves these on the block GEMM.  usage: strided_bp_check.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import synthetic_layers as SL
amd = importlib.import_module("compute-engine_amd")


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


for hw, cin, cout in ((56, 64, 128), (28, 128, 256), (14, 256, 512)):
    layer = SL.Layer(256, hw, hw, cin, 3, 3, cout, stride=2, padding=SL.PADDING_SAME, pad_values=1)
    w, mul, bias, thr = SL.weights(layer, 3)
    x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
    for rnd in range(2):
        for eng in ("stream", "direct"):
            plan = amd.Bconv2dPlan(layer.params(amd, amd.BITPACKED, 1.0, 0)); plan.set_weights(w, mul, bias, thr); plan.set_option("engine", eng)
            y = plan.run(x)
            print(f"stride 2 {hw}x{hw} {cin}->{cout} bitpacked  {eng:7s} {plan.kernel_name():48s} {timed(lambda: plan.run(x, y)):.2f} us")
