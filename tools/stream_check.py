#!/usr/bin/env python
"""engine=stream against the block GEMM (engine=direct) on the same operands: bit equality of the outputs and HIP-event timings.
usage: stream_check.py [layer ...]   layer = HWxCINxCOUT[xBATCH]  (default: L0 and the QuickNet 3x3 layers)
env: LCE_OPTS=key=val,... (extra plan options for the stream plan), LCE_STEPS (timed launches, default 20)"""
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
layers = sys.argv[1:] or ["56x256x256", "56x64x64", "28x128x128", "14x256x256"]
steps = int(os.environ.get("LCE_STEPS", "20"))


def timed(plan, x, out):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < 40:      # clock spin-up, as bench.py
        for _ in range(16):
            plan.run(x, out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        plan.run(x, out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for spec in layers:
    stride = 1
    dims = spec
    if "s" in spec:                       # HWxCINxCOUT[xBATCH]s<stride>
        dims, st = spec.split("s")
        stride = int(st)
    f = [int(v) for v in dims.split("x")]
    hw, cin, cout = f[:3]
    B = f[3] if len(f) > 3 else 256
    layer = SL.Layer(B, hw, hw, cin, 3, 3, cout, stride=stride, padding=SL.PADDING_SAME, pad_values=1)
    w, mul, bias, thr = SL.weights(layer, 3)
    x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
    for dname, dst in (("f32", amd.F32), ("i8", amd.I8), ("bp", amd.BITPACKED)):
        res = {}
        for engine in ("direct", "stream"):
            plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
            plan.set_weights(w, mul, bias, thr)
            try:
                plan.set_option("engine", engine)
                plan.kernel_name()
            except Exception as exc:
                print("%-14s %-3s engine=%s refused: %s" % (spec, dname, engine, str(exc)[:120]), flush=True)
                break
            if engine == "stream":
                for kv in filter(None, os.environ.get("LCE_OPTS", "").split(",")):
                    plan.set_option(*kv.split("="))
            out = plan.run(x)
            torch.cuda.synchronize()
            ms = timed(plan, x, out)
            res[engine] = (plan.kernel_name(), ms, out.clone())
        if len(res) < 2:
            continue
        same = torch.equal(res["direct"][2].view(torch.uint8), res["stream"][2].view(torch.uint8))
        print("%-14s %-3s %-38s %.4f ms | %-40s %.4f ms | %s x%.2f" % (
            spec, dname, res["direct"][0], res["direct"][1], res["stream"][0], res["stream"][1],
            "EQUAL" if same else "MISMATCH", res["direct"][1] / res["stream"][1]), flush=True)
        if not same:
            a, b = res["direct"][2].view(torch.uint8), res["stream"][2].view(torch.uint8)
            bad = (a != b).nonzero()
            print("   first mismatches at", bad[:5].tolist(), "count", int((a != b).sum()))
