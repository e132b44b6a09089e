#!/usr/bin/env python
"""Where does a HIP-graph replay of a device-resident chain spend its time, next to eager launches of the same chain?
Two roles:
  python tools/graph_gaps.py run <eager|graph> [iters] [quicknet|birealnet]
                                                           -- runs QuickNet's 16-layer fused chain (tools/layer_chain.py) that
                                                              way; meant to be wrapped in `rocprofv3 --kernel-trace`
  python tools/graph_gaps.py report <kernel_trace.csv> <iters> <kernels per iteration>
                                                           -- per iteration of the LAST `iters`: sum of kernel durations, sum of
                                                              the gaps between consecutive kernels, first start -> last end
(tools/graph_gaps.sh runs both under rocprofv3 and prints the two reports side by side.)"""
import csv
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(mode, iters, stack="quicknet"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import layer_chain
    import synthetic_layers as SL
    amd = importlib.import_module("compute-engine_amd")
    dev = torch.device("cuda:0")
    if stack == "birealnet":     # BASELINE config 5: int8 outputs, 1x1 + 3x3, strides 1 and 2
        ch = layer_chain.LayerChain(amd, torch, SL.birealnet_layers(256), dev, dst="i8", seed=4000)
    else:
        ch = layer_chain.LayerChain(amd, torch, SL.quicknet_layers(256), dev, dst="f32", seed=4000)
    ch.run_chain()
    torch.cuda.synchronize(dev)
    if mode == "graph":
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            ch.run_chain()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ch.run_chain()
        fn = graph.replay
    else:
        fn = ch.run_chain
    for _ in range(60):          # clock spin-up, untimed
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    print("%s %s: %.4f ms per chain by events (%d iterations, %d kernels each)" % (stack, mode, e0.elapsed_time(e1) / iters, iters, len(ch.plans)))


def report(path, iters, per_iter):
    rows = []
    for r in csv.DictReader(open(path)):
        if "bconv2d" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    rows = rows[-iters * per_iter:]
    dur = sum(e - s for s, e, _ in rows)
    gaps = [rows[k + 1][0] - rows[k][1] for k in range(len(rows) - 1)]
    inner = [g for k, g in enumerate(gaps) if (k + 1) % per_iter]          # between kernels of one chain
    outer = [g for k, g in enumerate(gaps) if (k + 1) % per_iter == 0]     # between two chains
    wall = rows[-1][1] - rows[0][0]
    med = lambda v: sorted(v)[len(v) // 2] if v else 0
    print("  kernels per chain %d, chains %d" % (per_iter, iters))
    print("  sum of kernel durations per chain : %8.1f us" % (dur / iters / 1e3))
    print("  gaps between kernels of a chain   : %8.1f us per chain  (median gap %.2f us, max %.2f us)" % (sum(inner) / iters / 1e3, med(inner) / 1e3, max(inner) / 1e3))
    print("  gap between two chains            : median %.2f us" % (med(outer) / 1e3))
    print("  first start -> last end           : %8.1f us per chain" % (wall / iters / 1e3))
    print("  per position in the chain (us, mean over the chains):")
    for k in range(per_iter):
        d = [rows[i * per_iter + k][1] - rows[i * per_iter + k][0] for i in range(iters)]
        print("    %2d %-60s %7.2f" % (k, rows[k][2].replace("void lce::", "")[:60], sum(d) / len(d) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 50, sys.argv[4] if len(sys.argv) > 4 else "quicknet")
    else:
        report(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
