#!/usr/bin/env python
"""Every candidate kernel of the planner on a grid of 3x3 layers x batch sizes x output types, on one box (round 5).

The table this writes (JSON lines: one per (layer, batch, dst), `us` per candidate) is what the planner's cost estimate
(csrc/lce_plan.cpp, estimate_*) is calibrated on and what tests/test_planner_choice.py holds the auto rule to: the kernel
`auto` picks must be within 5 % of the best candidate's time recorded here.  Timed from a captured HIP graph of >= 20
launches that cycle through > 256 MB of operand sets (bench.py's method for short layers).

usage: engine_sweep.py OUT.jsonl [--quick] [--dst=f32|i8|bp] [--only HWxCINxCOUT[sS]] ..."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
import kernel_hash  # noqa: E402
import synthetic_layers as SL  # noqa: E402

amd = importlib.import_module("compute-engine_amd")

LAYERS = [  # (H = W, Cin, Cout, stride)
    (56, 64, 64, 1), (28, 128, 128, 1), (14, 256, 256, 1), (7, 512, 512, 1),          # QuickNet
    (56, 256, 256, 1),                                                                # BASELINE L0
    (56, 64, 128, 2), (28, 128, 256, 2), (14, 256, 512, 2), (7, 512, 512, 2),         # config 5's strided layers
    (28, 256, 256, 1), (14, 512, 512, 1), (14, 128, 128, 1), (7, 256, 256, 1),        # off the bench's grid
    (40, 192, 192, 1), (20, 320, 320, 1), (112, 64, 64, 1),
    (14, 192, 192, 1), (7, 384, 384, 1), (28, 160, 160, 1),                           # between the streaming family's instances, small images
]
# never used to fit the cost estimate's constants (lce_plan_cost.cpp): the out-of-sample rows of tools/planner_regret.py (round 6)
HOLDOUT = [(28, 192, 192, 1), (10, 256, 256, 1), (56, 128, 128, 1), (20, 128, 256, 2)]
BATCHES = [1, 16, 64, 256]
DSTS = ["f32", "i8", "bp"]
CANDIDATES = {
    "auto": {},
    "direct": {"engine": "direct"},
    "mfma": {"engine": "mfma"},
    "stream": {"engine": "stream"},
    "stream_il4": {"engine": "stream", "stream_rows": "4", "stream_interleave": "1"},
    "stream_il8": {"engine": "stream", "stream_rows": "8", "stream_interleave": "1"},
    "stream_il2": {"engine": "stream", "stream_rows": "2", "stream_interleave": "1"},
    "stream_il7": {"engine": "stream", "stream_rows": "7", "stream_interleave": "1"},
    "stream_il14": {"engine": "stream", "stream_rows": "14", "stream_interleave": "1"},
    "wstream": {"engine": "wstream"},
    # one / two resident blocks per CU (two: the bitpacked-output instance of the 64-input-channel bank; refused elsewhere)
    "stream_x1": {"engine": "stream", "stream_blocks_per_cu": "1"},
    "stream_x2": {"engine": "stream", "stream_blocks_per_cu": "2"},
}


class Rotation:
    """bench.Rotation with a cap: tiny launches (batch 1 / 16) would need thousands of operand sets to leave the Infinity Cache;
    32 sets are kept for them (their ranking is what matters here, and every candidate sees the same cache)."""

    def __init__(self, x, out):
        per_set = x.numel() * x.element_size() + out.numel() * out.element_size()
        self.n = int(min(32, max(4, -(-bench.ROTATE_BYTES // max(1, per_set)))))
        self.x = [x] + [x.clone() for _ in range(self.n - 1)]
        self.out = [out] + [torch.empty_like(out) for _ in range(self.n - 1)]
        self.i = 0

    def run(self, plan):
        k = self.i
        self.i = (k + 1) % self.n
        plan.run(self.x[k], self.out[k])


def prepare(layer, dname, opts, dev):
    dst = {"f32": amd.F32, "i8": amd.I8, "bp": amd.BITPACKED}[dname]
    w, mul, bias, thr = SL.weights(layer, 3)
    x = torch.from_numpy(SL.activations(layer, 4)).to(dev)
    plan = amd.Bconv2dPlan(layer.params(amd, dst, 0.125, 3))
    plan.set_weights(w, mul, bias, thr)
    for k, v in opts.items():
        plan.set_option(k, v)
    out = plan.run(x)
    torch.cuda.synchronize(dev)
    return plan, Rotation(x, out), plan.kernel_name()


def time_once(plan, rot, dev):
    fn = lambda: rot.run(plan)
    bench.spin_up(torch, dev, fn, ms=10.0)
    s = bench._event_time(torch, dev, fn, 20)
    if s < 100e-6:
        g = bench.graph_time(torch, dev, fn, 20, max(20, rot.n))
        if g is not None:
            s = g
    return s * 1e6


def main():
    out_path = sys.argv[1]
    quick = "--quick" in sys.argv
    only = [a for a in sys.argv[2:] if not a.startswith("--")]
    dsts = [a.split("=")[1] for a in sys.argv[2:] if a.startswith("--dst=")] or DSTS
    dev = torch.device("cuda:0")
    layers = LAYERS + HOLDOUT
    if only:
        def key(l):
            return "%dx%dx%d" % l[:3] + ("s%d" % l[3] if l[3] > 1 else "")
        layers = [l for l in LAYERS + HOLDOUT if key(l) in only]
    fresh = not os.path.exists(out_path) or os.path.getsize(out_path) == 0
    with open(out_path, "a") as f:
        if fresh:
            # what the table was measured with: tests/test_planner_choice.py refuses a table whose kernels are not the tree's
            props = torch.cuda.get_device_properties(dev)
            meta = {"meta": {"kernel_sources_sha256": kernel_hash.kernel_sources_hash(), "device": props.name,
                             "compute_units": props.multi_processor_count, "library": amd.LIB_PATH if hasattr(amd, "LIB_PATH") else None,
                             "holdout": [list(l) for l in HOLDOUT], "quick": quick}}
            f.write(json.dumps(meta) + "\n")
            f.flush()
        for (hw, cin, cout, st) in layers:
            for b in ([256] if quick else BATCHES):
                for dname in dsts:
                    layer = SL.Layer(b, hw, hw, cin, 3, 3, cout, stride=st, padding=SL.PADDING_SAME, pad_values=1)
                    row = {"hw": hw, "cin": cin, "cout": cout, "stride": st, "batch": b, "dst": dname, "us": {}, "kernel": {}}
                    seen, cands = {}, []
                    for cname, opts in CANDIDATES.items():
                        try:
                            plan, rot, name = prepare(layer, dname, opts, dev)
                        except amd.LceHipError:
                            continue
                        except Exception as e:   # noqa: BLE001
                            row.setdefault("errors", {})[cname] = repr(e)[:120]
                            continue
                        row["kernel"][cname] = name
                        if name in seen:          # (auto, or a forced option that the planner resolves to a kernel already listed)
                            continue
                        seen[name] = cname
                        cands.append((cname, plan, rot))
                    # three rounds over all candidates, alternating direction; the median of each
                    times = {c[0]: [] for c in cands}
                    for r in range(3):
                        for cname, plan, rot in (cands if r % 2 == 0 else cands[::-1]):
                            times[cname].append(time_once(plan, rot, dev))
                    for cname, ts in times.items():
                        row["us"][cname] = round(sorted(ts)[1], 2)
                    # `auto` and the duplicates share the time of the kernel they resolve to
                    for cname, name in row["kernel"].items():
                        if cname not in row["us"]:
                            row["us"][cname] = row["us"][seen[name]]
                    del cands
                    f.write(json.dumps(row) + "\n")
                    f.flush()
                    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
