#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes: LceQuantize (a stream with exactly known HBM
bytes, used to calibrate FETCH_SIZE / WRITE_SIZE on this box) followed by the L0
LceBconv2d layer with float / int8 / bitpacked output, 3 launches each."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
dev = torch.device("cuda:0")
B = 256
fx = torch.randn((B, 56, 56, 256), device=dev)
for _ in range(3):
    xw = amd.bitpack(fx)
torch.cuda.synchronize()
spec = SL.Layer(batch=B, padding=SL.PADDING_SAME, pad_values=1, **bench.L0)
for dst in (amd.F32, amd.I8, amd.BITPACKED):
    s, name, plan, x, out = bench.time_layer(amd, torch, spec, dst, 3, 0, 0, dev, 0.125, 3)
    print(name, s * 1e3, "ms")
for hw, c in bench.QUICKNET:
    sp = SL.Layer(batch=B, in_h=hw, in_w=hw, channels_in=c, filter_h=3, filter_w=3, channels_out=c,
                  padding=SL.PADDING_SAME, pad_values=1)
    s, name, *_ = bench.time_layer(amd, torch, sp, amd.F32, 3, 0, hw, dev)
    print(name, hw, c, s * 1e3, "ms")
