#!/usr/bin/env python
"""Per-block tile-step timeline of the streaming kernel (needs a library built with -DLCE_STREAM_PHASES:
  bash tools/build_exp.sh sph:"-DLCE_STREAM_PHASES";  LCE_HIP_LIBRARY=$PWD/build_exp/lib_sph.so python tools/stream_phases.py 56 256x256 f32
Stamps (wave 0 of every block, s_memtime): 0 entry, 1 filter bank + first rows resident, 2+T after tile step T, 63 exit.
Prints cycles per phase / per tile step (mean over blocks), the launch's wall time and the clock they imply."""
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
layer = SL.Layer(B, hw, hw, cin, 3, 3, cout, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
plan.set_weights(w, mul, bias, thr)
plan.set_option("engine", "stream")
for kv in filter(None, os.environ.get("LCE_OPTS", "").split(",")):
    plan.set_option(*kv.split("="))
o = plan.run(x)
torch.cuda.synchronize()
t = time.perf_counter()
while (time.perf_counter() - t) * 1e3 < float(os.environ.get("LCE_SPINUP_MS", "40")):
    for _ in range(16):
        plan.run(x, o)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.run(x, o)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lib = amd.lib()
buf = np.zeros((512, 64), np.uint64)
assert lib.lce_hip_debug_read_stream_tl(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
live = buf[buf[:, 0] != 0]
nt = int((live[0, 2:58] != 0).sum())
t0 = live[:, 0].astype(np.int64)
print(plan.kernel_name(), "blocks", len(live), "tile steps", nt, "launch %.4f ms" % ms)
span = int(live[:, 63].max() - live[:, 0].min())
print("first entry -> last exit: %d cycles = %.3f GHz if that is the launch" % (span, span / ms / 1e6))
print("entry skew (cycles): max %d" % int(t0.max() - t0.min()))
pro = (live[:, 1].astype(np.int64) - t0)
print("prologue (filter bank, ring, first rows): mean %d  min %d  max %d" % (pro.mean(), pro.min(), pro.max()))
if live[:, 58].any():
    names = ["bank's first run + first rows' loads issued", "(stamp)", "constants, ring padding + barrier", "first rows expanded between the bank's pieces", "tile-0 quota issued, bank resident + barrier"]
    marks = [58, 59, 60, 61, 1]
    prev = t0
    for nm, mk in zip(names, marks):
        cur_ = live[:, mk].astype(np.int64)
        print("   %-48s +%d" % (nm, (cur_ - prev).mean()))
        prev = cur_
d = np.diff(live[:, 1:2 + nt].astype(np.int64), axis=1)
print("tile steps: mean %d cycles  (first %d, median %d, last %d); per block step %d; per MFMA of a wave %.1f" % (
    d.mean(), d[:, 0].mean(), np.median(d), d[:, -1].mean(), d.mean() / 4, np.median(d) / 4 / (9 * (cin // 64 if cin >= 64 else 1) * 2)))
tail = live[:, 63].astype(np.int64) - live[:, 1 + nt].astype(np.int64)
print("drain: mean %d" % tail.mean())
tot = live[:, 63].astype(np.int64) - t0
print("block life: mean %d  min %d  max %d" % (tot.mean(), tot.min(), tot.max()))
print("per tile step (mean over blocks):", " ".join(str(int(v)) for v in d.mean(axis=0)))
