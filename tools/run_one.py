#!/usr/bin/env python
"""Run ONE bconv2d configuration a few times (for rocprofv3 counter passes).
usage: run_one.py <H=W> <C> <f32|i8|bp> <valu|mfma> <tile|auto> [steps] [batch]"""
import importlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
hw, dname, engine, tile = int(sys.argv[1]), sys.argv[3], sys.argv[4], sys.argv[5]
cin, cout = (int(v) for v in sys.argv[2].split("x")) if "x" in sys.argv[2] else (int(sys.argv[2]), int(sys.argv[2]))
c = cin
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
B = int(sys.argv[7]) if len(sys.argv) > 7 else 256
dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
K = int(os.environ.get("LCE_K", "3"))   # filter height = width
STRIDE = int(os.environ.get("LCE_STRIDE", "1"))
layer = SL.Layer(B, hw, hw, c, K, K, cout, stride=STRIDE, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
if os.environ.get("LCE_ZERO"):   # constant operands: how much of the time is the power budget?
    x.zero_()
    w = np.zeros_like(w)
plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
plan.set_weights(w, mul, bias, thr)
plan.set_option("engine", engine)
for kv in filter(None, os.environ.get("LCE_OPTS", "").split(",")):   # e.g. LCE_OPTS=epilogue=direct
    plan.set_option(*kv.split("="))
if tile != "auto":
    if engine == "valu":
        plan.set_option("kernel", "tiled")
    plan.set_option("tile", tile)
out = plan.run(x)
torch.cuda.synchronize()
# untimed clock spin-up (as bench.py: the first ~50 launches of a process run at the governor's idle clock)
import time  # noqa: E402
_t = time.perf_counter()
while (time.perf_counter() - _t) * 1e3 < float(os.environ.get("LCE_SPINUP_MS", "40")):
    for _ in range(16):
        plan.run(x, out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    plan.run(x, out)
e1.record()
torch.cuda.synchronize()
print(plan.kernel_name(), e0.elapsed_time(e1) / steps, "ms")
