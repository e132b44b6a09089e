#!/usr/bin/env python
"""A part of a model stack as a chain of layers with their OWN buffers (tools/layer_chain.py), timed: for A/B runs of library variants
(LCE_HIP_LIBRARY) on sub-stacks -- e.g. the store cache policy on QuickNet's single-round layers only (round 6).
usage: substack_ab.py <quicknet|birealnet> <first layer> <last layer (exclusive)> [iters=200]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import layer_chain  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
which, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda:0")
layers = (SL.quicknet_layers(256) if which == "quicknet" else SL.birealnet_layers(256))[a:b]
ch = layer_chain.LayerChain(amd, torch, layers, dev, dst="f32" if which == "quicknet" else "i8", seed=4000 + a)
ch.run_convs()
ch.run_chain()
torch.cuda.synchronize()


def timed(fn, n):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out_mb = sum(y.numel() * y.element_size() for y in ch.y) / 1e6
print("%s[%d:%d] %d layers, %.0f MB of outputs: convolutions %.1f us, chain %.1f us  (%s)" % (
    which, a, b, len(layers), out_mb, timed(ch.run_convs, iters), timed(ch.run_chain, iters), sorted(set(ch.kernel_names()))[0]))
