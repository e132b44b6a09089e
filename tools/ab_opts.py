#!/usr/bin/env python
"""Interleaved A/B of PLAN OPTIONS on one layer, in ONE process on one box (round 5).

usage: ab_opts.py HxW CINxCOUT[sSTRIDE] DST BATCH ROUNDS STEPS name[:key=val[,key=val...]] name2:... ...
  e.g. ab_opts.py 56 256x256 f32 256 5 20 base rows4il:engine=stream,stream_rows=4,stream_interleave=1

Every variant is its own plan over the same operands.  First every variant's output is compared with the FIRST
variant's, byte for byte (a variant that changes the bytes is reported and dropped); then ROUNDS rounds, each timing
every variant in turn (an untimed spin-up, then STEPS launches between two HIP events) -- clock drift and the box's
state hit all variants alike.  Every variant writes its own output buffer, the buffers rotate through more memory
than the Infinity Cache holds (as bench.py's do).  Prints one line per round and the per-variant median."""
import importlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")


def main():
    hw, chans, dname, batch, rounds, steps = sys.argv[1:7]
    variants = sys.argv[7:]
    h, w_ = (int(v) for v in hw.split("x")) if "x" in hw else (int(hw), int(hw))
    stride = 1
    if "s" in chans:
        chans, st = chans.split("s")
        stride = int(st)
    cin, cout = (int(v) for v in chans.split("x")) if "x" in chans else (int(chans), int(chans))
    batch, rounds, steps = int(batch), int(rounds), int(steps)
    dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
    K = int(os.environ.get("LCE_K", "3"))
    layer = SL.Layer(batch, h, w_, cin, K, K, cout, stride=stride, padding=SL.PADDING_SAME, pad_values=1)
    w, mul, bias, thr = SL.weights(layer, 3)
    x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
    plans, names, outs = [], [], []
    ref = None
    for v in variants:
        name, _, opts = v.partition(":")
        plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
        plan.set_weights(w, mul, bias, thr)
        try:
            for kv in filter(None, opts.split(",")):
                plan.set_option(*kv.split("="))
            out = plan.run(x)
        except amd.LceHipError as e:
            print("# %s: refused (%s)" % (name, e))
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        elif not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
            print("# %s: BYTES DIFFER from %s -- dropped" % (name, names[0]))
            continue
        print("# %s = %s" % (name, plan.kernel_name()))
        plans.append(plan)
        names.append(name)
        # rotate through > 256 MB of output per variant (Infinity Cache), at least two buffers
        nbuf = max(2, min(8, int(600e6 // max(1, out.numel() * out.element_size())) + 1))
        outs.append([out] + [torch.empty_like(out) for _ in range(nbuf - 1)])
    spin_ms = float(os.environ.get("LCE_SPINUP_MS", "40"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {n: [] for n in names}
    for r in range(rounds):
        line = []
        order = range(len(plans)) if r % 2 == 0 else range(len(plans) - 1, -1, -1)
        for i in order:
            plan, bufs = plans[i], outs[i]
            t = time.perf_counter()
            k = 0
            while (time.perf_counter() - t) * 1e3 < spin_ms:
                for _ in range(16):
                    plan.run(x, bufs[k % len(bufs)])
                    k += 1
                torch.cuda.synchronize()
            e0.record()
            for s in range(steps):
                plan.run(x, bufs[s % len(bufs)])
            e1.record()
            torch.cuda.synchronize()
            res[names[i]].append(e0.elapsed_time(e1) / steps)
        print("[%s %s %s b%d] " % (hw, sys.argv[2], dname, batch) + " ".join("%s=%.4f" % (n, res[n][-1]) for n in names), flush=True)
    print("[%s %s %s b%d] MEDIAN " % (hw, sys.argv[2], dname, batch) + " ".join("%s=%.4f" % (n, statistics.median(res[n])) for n in names), flush=True)


if __name__ == "__main__":
    main()
