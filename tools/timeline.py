#!/usr/bin/env python
"""K-loop timeline of the matrix-core kernel from s_memtime stamps.

Needs a library built with -DLCE_TIMELINE (see the macro in lce_kernels_mfma.h):
  hipcc -DLCE_TIMELINE <usual flags> -shared -o build_exp/lib_tl.so lce_hip_api.hip lce_plan.cpp lce_prepare.cpp
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_tl.so python tools/timeline.py 56 256x256 bp direct 256x128
Stamps per K-step and wave: 0 = arrives at the step's wait+barrier, 1 = released, 2 = DMA and
fragment reads issued, 3 = all MFMAs of the step issued.  Printed: mean cycles of each segment."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_one.py")] + sys.argv[1:] + ["3", "256"],
                     capture_output=True, text=True)
print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
# run_one ran in a child; repeat one launch here to read the symbol from THIS process
sys.argv = [sys.argv[0]] + sys.argv[1:]
amd = importlib.import_module("compute-engine_amd")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

hw, dname, engine, tile = int(sys.argv[1]), sys.argv[3], sys.argv[4], sys.argv[5]
cin, cout = (int(v) for v in sys.argv[2].split("x"))
dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
B = 256
layer = SL.Layer(B, hw, hw, cin, 3, 3, cout, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
plan = amd.Bconv2dPlan(layer.params(amd, dst))
plan.set_weights(w, mul, bias, thr)
plan.set_option("engine", engine)
if tile != "auto":
    plan.set_option("tile", tile)
o = plan.run(x)
for _ in range(5):
    plan.run(x, o)
torch.cuda.synchronize()
buf = np.zeros((2, 8, 80, 4), np.uint64)
rc = amd.lib().lce_hip_debug_read_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
assert rc == 0, rc
print(plan.kernel_name())
for blk in range(2):
    t = buf[blk].astype(np.int64)
    live = t[:, :, 0] != 0
    waves = [w_ for w_ in range(8) if live[w_].any()]
    if not waves:
        continue
    steps = [k for k in range(80) if live[waves[0], k]]
    ks = steps[2:-2]                      # steady part
    seg = {"wait+barrier (0->1)": t[waves][:, ks, 1] - t[waves][:, ks, 0],
           "issue DMA+ds_read (1->2)": t[waves][:, ks, 2] - t[waves][:, ks, 1],
           "issue MFMAs (2->3)": t[waves][:, ks, 3] - t[waves][:, ks, 2],
           "to next step (3->0')": t[waves][:, [k + 1 for k in ks], 0] - t[waves][:, ks, 3],
           "whole K-step": t[waves][:, [k + 1 for k in ks], 0] - t[waves][:, ks, 0]}
    print("block %d: %d waves, K-steps %d..%d" % (blk, len(waves), ks[0], ks[-1]))
    for k, v in seg.items():
        print("   %-26s mean %7.1f   p10 %6.0f  p50 %6.0f  p90 %6.0f   per wave: %s" % (
            k, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90),
            " ".join("%5.0f" % x for x in v.mean(axis=1))))
