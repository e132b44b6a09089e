#!/usr/bin/env python
"""A model stack as device-resident chains of HALF batches on two (or more) streams: images are independent, so one
half's kernel boundaries (launch, prologue, tail) can overlap the other half's K loops.  Compares, on one box:
  one stream, whole batch (what bench.py times)  vs  S streams, batch / S each
for the convolutions alone and for the fused chain.    usage: chain_split.py quicknet|quicknet_large|birealnet [streams=2] [iters=60]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import layer_chain  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "quicknet"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
B = 256
dev = torch.device("cuda:0")
mk = {"quicknet": lambda b: SL.quicknet_layers(b), "quicknet_large": lambda b: SL.quicknet_layers(b, (6, 8, 12, 6)),
      "birealnet": lambda b: SL.birealnet_layers(b)}[which]
dst = "i8" if which == "birealnet" else "f32"
whole = layer_chain.LayerChain(amd, torch, mk(B), dev, dst=dst, seed=4000)
parts = [layer_chain.LayerChain(amd, torch, mk(B // S), dev, dst=dst, seed=4000) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split(method):
    main = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(main)
    for s, c in zip(streams, parts):
        with torch.cuda.stream(s):
            getattr(c, method)()
    for s in streams:
        main.wait_stream(s)


def spin():
    import time
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        whole.run_convs()
        torch.cuda.synchronize()


for rnd in range(3):
    spin()
    a = timed(whole.run_convs, iters)
    b = timed(lambda: split("run_convs"), iters)
    c = timed(whole.run_chain, iters)
    d = timed(lambda: split("run_chain"), iters)
    print(f"[{which}] convolutions: 1 stream x {B} = {a:.1f} us   {S} streams x {B // S} = {b:.1f} us ({b / a - 1:+.1%})    "
          f"fused chain: {c:.1f} us  vs {d:.1f} us ({d / c - 1:+.1%})")
print("kernels (whole):", sorted(set(whole.kernel_names())))
print("kernels (part): ", sorted(set(parts[0].kernel_names())))
