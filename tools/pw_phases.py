#!/usr/bin/env python
"""Timeline of ONE launch of the pointwise (1x1) kernel (needs a library built with -DLCE_PW_PHASES:
  bash tools/build_exp.sh pwph:"-DLCE_PW_PHASES";  LCE_HIP_LIBRARY=$PWD/build_exp/lib_pwph.so python tools/pw_phases.py 14 256x256 i8
Stamps (wave 0 of every block, shader cycle counter): 0 entry, 1 filter bank + first words resident, 2 last tile's stores
issued, 3 stores acknowledged (vmcnt 0).  Prints when the blocks enter / leave relative to the first entry, the phases'
lengths, and the launch's event time: what part of a small launch is dispatch, latency, and the drain of its stores."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
hw, dname = int(sys.argv[1]), sys.argv[3]
cin, cout = (int(v) for v in sys.argv[2].split("x"))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
layer = SL.Layer(B, hw, hw, cin, 1, 1, cout, stride=int(os.environ.get("LCE_STRIDE", "1")), padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
plan.set_weights(w, mul, bias, thr)
plan.set_option("engine", "pointwise")
for kv in filter(None, os.environ.get("LCE_OPTS", "").split(",")):
    plan.set_option(*kv.split("="))
o = plan.run(x)
torch.cuda.synchronize()
t = time.perf_counter()
while (time.perf_counter() - t) * 1e3 < float(os.environ.get("LCE_SPINUP_MS", "40")):
    for _ in range(16):
        plan.run(x, o)
    torch.cuda.synchronize()
lib = amd.lib()
assert lib.lce_hip_debug_clear_pw_tl() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.run(x, o)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = np.zeros((8192, 4), np.uint64)
assert lib.lce_hip_debug_read_pw_tl(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
live = buf[buf[:, 0] != 0].astype(np.int64)
t0 = live[:, 0].min()
print(plan.kernel_name(), "blocks stamped", len(live), "launch (events around it) %.2f us" % (ms * 1e3))
span = int(live[:, 3].max() - t0)
print("first entry -> last acknowledged store: %d cycles (%.2f us at 2.4 GHz)" % (span, span / 2400))
q = lambda v: "min %6d  median %6d  p90 %6d  max %6d" % (v.min(), np.median(v), np.percentile(v, 90), v.max())
print("entry after the first block's     :", q(live[:, 0] - t0))
print("entry -> operands resident        :", q(live[:, 1] - live[:, 0]))
print("operands -> last stores issued    :", q(live[:, 2] - live[:, 1]))
print("last stores issued -> acknowledged:", q(live[:, 3] - live[:, 2]))
print("block life                        :", q(live[:, 3] - live[:, 0]))
print("exit after the first block's entry:", q(live[:, 3] - t0))
