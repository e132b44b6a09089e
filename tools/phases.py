#!/usr/bin/env python
"""Per-block phase timeline of the matrix-core kernel (direct or workspace variant).

Needs a library built with -DLCE_PHASES (macro in lce_kernels_mfma.h):
  hipcc -DLCE_PHASES <usual flags> -shared -o build_exp/lib_ph.so lce_hip_api.hip lce_plan.cpp lce_prepare.cpp
  LCE_HIP_LIBRARY=$PWD/build_exp/lib_ph.so python tools/phases.py 56 256x256 f32 direct auto
Stamps (wave 0 of every block, s_memtime): 0 entry, 1 input halo expanded into LDS, 2 K loop entered,
3 K loop done, 4 epilogue issued, 5 stores acknowledged; plus the hardware id of the CU.
Prints the mean duration of each phase and, per CU, how much of the launch had 0 / 1 / 2+ blocks inside
their K loop (the matrix pipe is only fed at its rate with two)."""
import ctypes as C
import importlib
import json
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
cin, cout = (int(v) for v in sys.argv[2].split("x"))
K = int(os.environ.get("LCE_K", "3"))
dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 256
layer = SL.Layer(B, hw, hw, cin, K, K, cout, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
plan.set_weights(w, mul, bias, thr)
plan.set_option("engine", engine)
for kv in filter(None, os.environ.get("LCE_OPTS", "").split(",")):   # e.g. LCE_OPTS=epilogue=direct
    plan.set_option(*kv.split("="))
if tile != "auto":
    plan.set_option("tile", tile)
o = plan.run(x)
for _ in range(5):
    plan.run(x, o)
torch.cuda.synchronize()
lib = amd.lib()
# untimed clock spin-up (as bench.py): the stamped launch below runs at the warmed-up clock unless LCE_SPINUP_MS=0
import time  # noqa: E402
_t = time.perf_counter()
while (time.perf_counter() - _t) * 1e3 < float(os.environ.get("LCE_SPINUP_MS", "40")):
    for _ in range(16):
        plan.run(x, o)
    torch.cuda.synchronize()
assert lib.lce_hip_debug_clear_phases() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.run(x, o)
e1.record()
torch.cuda.synchronize()
buf = np.zeros((16384, 16), np.uint64)
rc = lib.lce_hip_debug_read_phases(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
assert rc == 0, rc
t = buf[buf[:, 0] != 0]
n = len(t)
ms = e0.elapsed_time(e1)
print(plan.kernel_name(), "%.4f ms" % ms, n, "blocks stamped")
st = t[:, :6].astype(np.int64)
t0 = st[:, 0].min()
span = st[:, 5].max() - t0
print("launch span %d ticks -> %.1f MHz tick rate if span == event time" % (span, span / (ms * 1e3)))
names = ["prologue (entry -> halo in LDS)", "pipeline start (-> K loop)", "K loop", "epilogue issue", "store drain", "whole block"]
durs = [st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3], st[:, 5] - st[:, 4], st[:, 5] - st[:, 0]]
res = {"kernel": plan.kernel_name(), "ms": ms, "blocks": n, "span_ticks": int(span)}
for nm, d in zip(names, durs):
    print("  %-34s mean %8.0f  p10 %7.0f p50 %7.0f p90 %7.0f  (%.1f %% of a block's life)" % (
        nm, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), 100.0 * d.mean() / durs[-1].mean()))
    res[nm] = float(d.mean())
if (t[:, 8] != 0).all():
    fine = t[:, [0, 8, 9, 10, 11, 1]].astype(np.int64)
    lab = ["entry -> halo block (kernel args, tile indices)", "index math + halo loads issued", "weight-ring fills issued",
           "first item's data arrived + expanded", "rest of the expansion + LDS writes"]
    for i, nm in enumerate(lab):
        d = fine[:, i + 1] - fine[:, i]
        print("    prologue: %-48s mean %7.0f  p10 %6.0f p50 %6.0f p90 %6.0f" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
# per CU: xcc (upper word), and HW_ID fields cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout)
hwid = t[:, 7]
cu_key = ((hwid >> 32) << 16) | (hwid & 0xFF00)
cus = np.unique(cu_key)
print("  distinct CUs seen:", len(cus))
occ = np.zeros(4)
blocks_per_cu = []
for c in cus:
    sel = st[cu_key == c]
    blocks_per_cu.append(len(sel))
    ev = [(a, 1) for a in sel[:, 2]] + [(b, -1) for b in sel[:, 3]]
    ev.sort()
    lo, hi = sel[:, 0].min(), sel[:, 5].max()
    cur, last = 0, lo
    for tt, d in ev:
        occ[min(cur, 3)] += tt - last
        last = tt
        cur += d
    occ[min(cur, 3)] += hi - last
occ /= occ.sum()
print("  blocks per CU: min %d max %d mean %.1f" % (min(blocks_per_cu), max(blocks_per_cu), np.mean(blocks_per_cu)))
print("  share of CU time with k blocks inside their K loop: " + "  ".join("k=%d: %.1f %%" % (k, 100 * occ[k]) for k in range(4)))
res["k_loop_occupancy"] = [float(v) for v in occ]
print(json.dumps(res))
