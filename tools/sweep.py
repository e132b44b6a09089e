#!/usr/bin/env python
"""Times every tiled-kernel variant on the BASELINE layer shapes (GPU box only).
Prints one JSON line per (layer, dst, tile)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")
dev = torch.device("cuda:0")
B = int(os.environ.get("SWEEP_BATCH", "256"))
steps = int(os.environ.get("SWEEP_STEPS", "10"))
VALU_BMAC_PEAK = 8.1e14
K = int(os.environ.get("SWEEP_K", "3"))   # filter height = width (1: the pointwise layers of config 5)
layers = [("L0_56x56x256", 56, 256)] + [(f"qn_{hw}x{hw}x{c}", hw, c) for hw, c in SL.QUICKNET_STAGES]
dsts = [("f32", amd.F32, SL.DST_F32), ("i8", amd.I8, SL.DST_I8), ("bp", amd.BITPACKED, SL.DST_BITPACKED)]
tiles = ([] if os.environ.get("SWEEP_MATRIX_ONLY") else ["4x16", "2x32", "2x16", "1x32", "1x16"]) + [
         "m256x256", "m256x128", "m512x64", "m128x256", "m128x128", "m256x64", "m128x64",
         "d256x256", "d256x128", "d512x64", "d128x256", "d128x128", "d256x64", "d128x64"]
if os.environ.get("SWEEP_PREFIX"):   # e.g. "d": the direct variant's tiles only
    tiles = [t for t in tiles if t.startswith(os.environ["SWEEP_PREFIX"])]
only = set(sys.argv[1:])
for lname, hw, c in layers:
    spec = SL.Layer(batch=B, in_h=hw, in_w=hw, channels_in=c, filter_h=K, filter_w=K, channels_out=c,
                    padding=SL.PADDING_SAME, pad_values=1)
    for dname, dst, od in dsts:
        if only and dname not in only and lname not in only:
            continue
        for tile in tiles:
            direct = tile.startswith("d")
            mfma = tile.startswith("m") or direct
            if not mfma and dst == amd.BITPACKED and not tile.endswith("32"):
                continue
            os.environ["LCE_SWEEP_TILE"] = tile
            w, mul, bias, thr = SL.weights(spec, 3)
            x = torch.from_numpy(SL.activations(spec, 4)).to(dev)
            plan = amd.Bconv2dPlan(spec.params(amd, dst, 0.125, 3))
            plan.set_weights(w, mul, bias, thr)
            plan.set_option("engine", "direct" if direct else "mfma" if mfma else "valu")
            if not mfma:
                plan.set_option("kernel", "tiled")
            plan.set_option("tile", tile.lstrip("md"))
            try:
                if not plan.kernel_name():
                    continue
            except Exception:
                continue
            out = plan.run(x)
            torch.cuda.synchronize()
            _t = time.perf_counter()   # untimed clock spin-up, as bench.py
            while (time.perf_counter() - _t) * 1e3 < float(os.environ.get("LCE_SPINUP_MS", "40")):
                for _ in range(16):
                    plan.run(x, out)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                plan.run(x, out)
            e1.record()
            torch.cuda.synchronize()
            s = e0.elapsed_time(e1) / 1e3 / steps
            print(json.dumps({"layer": lname, "dst": dname, "tile": tile, "kernel": plan.kernel_name(),
                              "ms": round(s * 1e3, 4), "Tbmac_per_s": round(spec.binary_macs / s / 1e12, 2),
                              "alu_frac": round(spec.binary_macs / s / VALU_BMAC_PEAK, 3),
                              "GBps": round(spec.algorithmic_bytes(od) / s / 1e9, 1)}), flush=True)
            del plan, out, x
