#!/usr/bin/env python
"""What the second output (lce_hip_bconv2d_run_dual: the layer's LceQuantize from the same epilogue) costs, layer by layer, on the
three BASELINE stacks: us per launch of run() and of run_dual(), each from a captured HIP graph of the layer on >= 4 operand sets.
usage: dual_cost.py [quicknet|birealnet|all] [batch=256]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")


def graph_us(fn, launches, reps=12):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.04:          # clock spin-up (bench.py does the same)
        g.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / launches


def layer_row(L, dst, seed):
    w, mul, bias, thr = SL.weights(L, seed)
    scale, zp = SL.int8_quant(seed) if dst == "i8" else (1.0, 0)
    plan = amd.Bconv2dPlan(L.params(amd, amd.F32 if dst == "f32" else amd.I8, scale, zp))
    plan.set_weights(w, mul, bias)
    x0 = torch.from_numpy(SL.activations(L, seed)).to(dev)
    n = 6
    xs = [x0] + [x0.clone() for _ in range(n - 1)]
    ys = [torch.empty(plan.output_shape, dtype=torch.float32 if dst == "f32" else torch.int8, device=dev) for _ in range(n)]
    b, oh, ow, c = plan.output_shape
    bs = [torch.empty((b, oh, ow, (c + 31) // 32), dtype=torch.int32, device=dev) for _ in range(n)]
    plan.run(xs[0], ys[0])
    plan.run_dual(xs[0], ys[0], bs[0])
    torch.cuda.synchronize()
    plain = graph_us(lambda: [plan.run(xs[k % n], ys[k % n]) for k in range(4 * n)], 4 * n)
    dual = graph_us(lambda: [plan.run_dual(xs[k % n], ys[k % n], bs[k % n]) for k in range(4 * n)], 4 * n)
    return plain, dual, plan.kernel_name()


stacks = {"quicknet": (SL.quicknet_layers(B)[::4], "f32"), "birealnet": (SL.birealnet_layers(B), "i8")}
for name, (layers, dst) in stacks.items():
    if which not in ("all", name):
        continue
    tot_p = tot_d = 0.0
    for k, L in enumerate(layers):
        p, d, kn = layer_row(L, dst, 4000 + k)
        mult = 4 if name == "quicknet" else 1
        tot_p += p * mult
        tot_d += d * mult
        print("%-10s %dx%d %3dx%d s%d %4d->%-4d  run %7.2f us   run_dual %7.2f us  (%+5.1f %%)   %s" % (
            name, L.in_h, L.in_w, L.filter_h, L.filter_w, L.stride, L.channels_in, L.channels_out, p, d, (d / p - 1) * 100, kn), flush=True)
    print("%-10s sum over the stack: run %.1f us, run_dual %.1f us (%+.1f %%)" % (name, tot_p, tot_d, (tot_d / tot_p - 1) * 100), flush=True)
