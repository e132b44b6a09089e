#!/usr/bin/env python
"""Benchmark of the MI355X-native LceBconv2d hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the LceBconv2d
layer of BASELINE.json configs[1] -- 3x3, 256->256 channels, 56x56, SAME padding with
pad_values=1, batch 256 PER GPU, float32 output through the fused output transform --
with the bitpacked input, packed weights and output resident in HBM.  For N > 1 (launched
by torch.distributed.run, one rank per GPU) the batch dimension is sharded: every rank
runs its own 256 images, there is no data-path collective, RCCL is used
only for the barrier and the max-over-ranks of the elapsed time.  `value` is whole-job
binary MACs per second.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (bconv2d_mfma, or bconv2d_tiled with engine=valu): algorithmic
                bytes per launch / mean launch duration (events on the launch stream, kernel
                timed alone) vs the 8 TB/s HBM peak -- the binding roofline of this layer;
  compute       the same kernel against the other roofline (FP4 MFMA peak, or the measured
                v_xor+v_bcnt pair ceiling for the xor-popcount engine);
  cpu_baseline  the CPU oracle (a port of the reference's portable C++ path) on the host
                cores, on a bounded sample of the same workload (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthetic_layers as SL  # noqa: E402  (NumPy-only layer descriptions + seeded operands)

HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_FP4_PEAK_TFLOPS = 10000.0                         # MI355X_MICROARCH.md: FP6/FP4 MFMA ~10 PF dense
VALU_BMAC_PEAK = 8.1e14                                # measured v_xor+v_bcnt ceiling (profiles/r01/valu_peak_microbench.jsonl)

L0 = dict(in_h=56, in_w=56, channels_in=256, filter_h=3, filter_w=3, channels_out=256)
QUICKNET = [(56, 64), (28, 128), (14, 256), (7, 512)]  # (H=W, C): 4 layers each in QuickNet


def algorithmic_bytes(layer, dst) -> int:
    """input words + weights + params + output, each counted once."""
    return layer.algorithmic_bytes(dst)


def _event_time(torch, dev, fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / 1e3 / steps


def time_layer(amd, torch, layer, dst, steps, warmup, seed, dev, scale=1.0, zp=0, engine="auto"):
    """Returns (mean seconds per step from stream events, kernel name, plan, x, out).
    Synthetic operands from tools/synthetic_layers.py -- the oracle is not involved."""
    w, mul, bias, thr = SL.weights(layer, seed)
    x = torch.from_numpy(SL.activations(layer, seed)).to(dev)
    plan = amd.Bconv2dPlan(layer.params(amd, dst, scale, zp))
    plan.set_weights(w, mul, bias, thr if dst == amd.BITPACKED else None)
    plan.set_option("engine", engine)
    dt = {amd.F32: torch.float32, amd.I8: torch.int8, amd.BITPACKED: torch.int32}[dst]
    out = torch.empty(plan.output_shape, dtype=dt, device=dev)
    for _ in range(max(1, warmup)):
        plan.run(x, out)
    torch.cuda.synchronize(dev)
    return _event_time(torch, dev, lambda: plan.run(x, out), steps), plan.kernel_name(), plan, x, out


def cpu_baseline(target_seconds=12.0):
    """Time the CPU oracle (port of the reference's portable path) on the L0 layer."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))   # the oracle's ctypes wrapper lives with the tests
    import oracle_lib as O
    import synth
    cores = os.cpu_count() or 1
    lib_note = "oracle/liblce_oracle.so (-O3 -march=x86-64-v3 -ffp-contract=off, OpenMP)"
    one = O.ConvSpec(batch=1, padding=O.PADDING_SAME, pad_values=1, **L0)
    x1, w, mul, bias = synth.conv_inputs(one, 1)
    t = time.perf_counter()
    O.bconv2d(one, O.DST_F32, x1, w, mul, bias, threads=1)
    t1 = time.perf_counter() - t
    per_image_parallel = t1 / max(1, cores) * 1.3
    n = int(max(cores, min(256, target_seconds / max(per_image_parallel, 1e-6))))
    spec = O.ConvSpec(batch=n, padding=O.PADDING_SAME, pad_values=1, **L0)
    x = synth.random_words(synth.rng(2), spec.input_shape(), spec.channels_in)
    t = time.perf_counter()
    O.bconv2d(spec, O.DST_F32, x, w, mul, bias, threads=cores)
    dt = time.perf_counter() - t
    return {"value": spec.binary_macs / dt, "unit": "binary-MAC/s", "cores": cores, "kind": "port",
            "sample": f"{n} of 256 images of the same layer, float output, {dt:.1f} s wall; "
                      f"1 image on 1 thread: {t1 * 1e3:.1f} ms; {lib_note}",
            "single_thread_value": one.binary_macs / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    import torch
    amd = importlib.import_module("compute-engine_amd")
    shard = importlib.import_module("compute-engine_amd.batch_shard")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    # weak scaling: the global batch grows with N, every rank owns a contiguous slab of it
    global_batch = args.batch * world
    _, my_batch = shard.shard_range(global_batch, world, rank)
    spec = SL.Layer(batch=my_batch, padding=SL.PADDING_SAME, pad_values=1, **L0)
    # warm up + per-step time from stream events (a step = every kernel of one LceBconv2d call)
    step_sec, kname, plan, x, out = time_layer(amd, torch, spec, amd.F32, args.steps, args.warmup, 0, dev)

    # the contract's timed region: barrier + sync, exactly K steps, sync + barrier, MAX over ranks
    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[dev.index])
        torch.cuda.synchronize(dev)

    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()                    # same stream the plan launches on (torch's current stream)
    for i in range(args.steps):
        plan.run(x, out)
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    region_event_sec = evs[0].elapsed_time(evs[-1]) / 1e3 / args.steps
    per_step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step_ms[min(len(per_step_ms) - 1, int(q * len(per_step_ms)))]
    elapsed = shard.max_over_ranks(elapsed, dist, dev)

    total_bmacs = spec.binary_macs * args.steps * world
    value = total_bmacs / elapsed
    abytes = algorithmic_bytes(spec, SL.DST_F32)
    mfma = kname.startswith("bconv2d_mfma")
    direct = kname.startswith("bconv2d_mfma_direct")   # single kernel: the block expands its own input halo

    result = {
        "metric": "binary-MACs/sec (LceBconv2d 3x3 256->256, 56x56, batch 256/GPU, f32 out)",
        "value": value, "unit": "binary-MAC/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": ("fp4-e2m1 (exact +-1) matrix-core dot product, fp32 accumulate, f32 epilogue" if mfma
                  else "u32 xor+popcount, int32 accumulate, f32 epilogue"),
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: LceBconv2d 3x3 s1 SAME(pad_values=1) 56x56x256->256, "
                               "float32 output transform, device-resident bitpacked input",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"batch-shard x{world} (no data-path collective)"},
        "layer_latency_ms": step_sec * 1e3,
        # spread of the K timed steps (the chip's power management moves the clock during a run)
        "ms_per_step_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)],
        "per_gpu_value": value / world,
        "kernel": kname + ("+expand_fp4" if mfma and not direct else ""),
    }
    if rank == 0:
        # dominant kernel alone (the GEMM of the matrix-core engine, or the single VALU kernel)
        if mfma and not direct:
            plan.set_option("phase", "gemm")
            k_sec = _event_time(torch, dev, lambda: plan.run(x, out), args.steps)
            plan.set_option("phase", "expand")
            e_sec = _event_time(torch, dev, lambda: plan.run(x, out), args.steps)
            plan.set_option("phase", "all")
        else:
            # one kernel per step: its average duration IS the event time of the timed region / K
            k_sec, e_sec = region_event_sec, 0.0
        ach = abytes / k_sec / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                              "algorithmic_bytes_per_launch": abytes, "kernel": kname,
                              "kernel_ms": k_sec * 1e3, "expand_fp4_ms": e_sec * 1e3,
                              "kernel_ms_source": ("phase=gemm re-run, HIP events" if mfma and not direct else
                                                   "HIP events around the timed region / steps (one launch per step)"),
                              "note": "HBM is the binding roofline of this layer at spec peaks "
                                      "(0.106 ms vs 0.095 ms of FP4 MFMA); see `compute` for the other one"}
        if mfma:
            tf = 2.0 * spec.binary_macs / k_sec / 1e12
            result["compute"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_FP4_PEAK_TFLOPS,
                                 "model": "v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4), 2 FLOP per binary MAC"}
        else:
            result["compute"] = {"bound": "valu", "achieved": spec.binary_macs / k_sec, "peak": VALU_BMAC_PEAK,
                                 "unit": "binary-MAC/s", "frac": spec.binary_macs / k_sec / VALU_BMAC_PEAK,
                                 "model": "v_xor_b32 + v_bcnt_u32_b32 per 32 bMAC, measured pair ceiling"}
        if not args.no_extra and world == 1:
            extra = {}
            st, wu = max(5, args.steps // 5), 3
            sc, zp = 0.125, 3
            for nm, dst, od in (("l0_int8_out", amd.I8, SL.DST_I8), ("l0_bitpacked_out", amd.BITPACKED, SL.DST_BITPACKED)):
                s_, kn, *_ = time_layer(amd, torch, spec, dst, st, wu, 1, dev, sc, zp)
                extra[nm] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn,
                             "GBps_algorithmic": algorithmic_bytes(spec, od) / s_ / 1e9}
            # the other matrix-core variant (FP4 workspace + GEMM whose tiles span images)
            s_, kn, *_ = time_layer(amd, torch, spec, amd.F32, st, wu, 0, dev, engine="mfma")
            extra["l0_f32_workspace_gemm"] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn + "+expand_fp4"}
            # the xor-popcount engine on the same layer (the north star's literal formulation)
            s_, kn, *_ = time_layer(amd, torch, spec, amd.F32, st, wu, 0, dev, engine="valu")
            extra["l0_f32_valu_engine"] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn,
                                           "valu_pair_frac": spec.binary_macs / s_ / VALU_BMAC_PEAK}
            tot = 0.0
            for hw, c in QUICKNET:
                sp = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=3, filter_w=3,
                              channels_out=c, padding=SL.PADDING_SAME, pad_values=1)
                s_, kn, *_ = time_layer(amd, torch, sp, amd.F32, st, wu, hw, dev)
                tot += 4 * s_
                extra[f"quicknet_{hw}x{hw}x{c}_f32"] = {
                    "ms": s_ * 1e3, "bmac_per_s": sp.binary_macs / s_, "kernel": kn,
                    "GBps_algorithmic": algorithmic_bytes(sp, SL.DST_F32) / s_ / 1e9,
                    "hbm_frac": algorithmic_bytes(sp, SL.DST_F32) / s_ / 1e9 / HBM_PEAK_GBS}
            extra["quicknet_16_layers_ms"] = tot * 1e3
            # BASELINE config 4: QuickNetLarge = blocks (6, 8, 12, 6) of the same four layer shapes
            extra["quicknet_large_32_layers_ms"] = sum(
                n * extra[f"quicknet_{hw}x{hw}x{c}_f32"]["ms"] for n, (hw, c) in zip((6, 8, 12, 6), QUICKNET))
            # the same 16 layers as ONE device-resident chain, each fed by LceQuantize of the previous
            # float output (what a converted QuickNet does between its binary convolutions), replayed
            # from a captured HIP graph: launch gaps and the LceQuantize passes included
            try:
                chain = []
                for hw, c in QUICKNET:
                    sp = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=3, filter_w=3,
                                  channels_out=c, padding=SL.PADDING_SAME, pad_values=1)
                    _, _, pl, xq, yo = time_layer(amd, torch, sp, amd.F32, 1, 1, hw, dev)
                    chain.append((pl, xq, yo))

                def run_chain():
                    for pl, xq, yo in chain:
                        for _ in range(4):
                            pl.run(xq, yo)
                            amd.bitpack(yo, out=xq)
                run_chain()
                torch.cuda.synchronize(dev)
                eager = _event_time(torch, dev, run_chain, st)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    run_chain()
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    run_chain()
                graph.replay()
                torch.cuda.synchronize(dev)
                extra["quicknet_16_layers_with_lcequantize_chain_ms"] = {
                    "eager": eager * 1e3, "hip_graph_replay": _event_time(torch, dev, graph.replay, st) * 1e3}
                del chain, graph
            except Exception as e:   # a report line must not take the bench down
                extra["quicknet_16_layers_with_lcequantize_chain_ms"] = {"error": repr(e)[:200]}
            # BASELINE config 5 flavour: 1x1 int8-output layers with a RELU clamp (the HBM-bound cases)
            for hw, c in QUICKNET:
                sp = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=1, filter_w=1,
                              channels_out=c, activation=SL.ACT_RELU)
                s_, kn, *_ = time_layer(amd, torch, sp, amd.I8, st, wu, hw + 1, dev, sc, zp)
                ab = algorithmic_bytes(sp, SL.DST_I8)
                extra[f"pointwise_{hw}x{hw}x{c}_int8_relu"] = {
                    "ms": s_ * 1e3, "bmac_per_s": sp.binary_macs / s_, "kernel": kn,
                    "GBps_algorithmic": ab / s_ / 1e9, "hbm_frac": ab / s_ / 1e9 / HBM_PEAK_GBS}
            # LceQuantize stream: float32 56x56x256 feature map, batch 256
            fx = torch.randn((args.batch, 56, 56, 256), device=dev)
            ow = amd.bitpack(fx)
            torch.cuda.synchronize(dev)
            s_ = _event_time(torch, dev, lambda: amd.bitpack(fx, out=ow), st)
            qb = fx.numel() * 4 + ow.numel() * 4
            extra["lcequantize_f32_256x56x56x256"] = {"ms": s_ * 1e3, "GBps_algorithmic": qb / s_ / 1e9,
                                                      "hbm_frac": qb / s_ / 1e9 / HBM_PEAK_GBS}
            # LceDequantize (bits -> float) and LceBMaxPool2d (2x2 stride 2) on the same feature map
            fo = amd.unpack(ow, 256, torch.float32)
            torch.cuda.synchronize(dev)
            s_ = _event_time(torch, dev, lambda: amd.unpack(ow, 256, torch.float32), st)
            extra["lcedequantize_f32_256x56x56x256"] = {"ms": s_ * 1e3, "GBps_algorithmic": qb / s_ / 1e9,
                                                        "hbm_frac": qb / s_ / 1e9 / HBM_PEAK_GBS}
            po = amd.bmaxpool(ow, 2, 2, 2, 2, amd.PADDING_VALID)
            torch.cuda.synchronize(dev)
            s_ = _event_time(torch, dev, lambda: amd.bmaxpool(ow, 2, 2, 2, 2, amd.PADDING_VALID), st)
            pb = ow.numel() * 4 + po.numel() * 4
            extra["lcebmaxpool_2x2s2_256x56x56x256"] = {"ms": s_ * 1e3, "GBps_algorithmic": pb / s_ / 1e9,
                                                        "hbm_frac": pb / s_ / 1e9 / HBM_PEAK_GBS}
            del fx, ow, fo, po
            result["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
