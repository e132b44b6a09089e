#!/usr/bin/env python
"""Benchmark of the MI355X-native LceBconv2d hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the LceBconv2d
layer of BASELINE.json configs[1] -- 3x3, 256->256 channels, 56x56, SAME padding with
pad_values=1, batch 256 PER GPU, float32 output through the fused output transform --
with the bitpacked input, packed weights and output resident in HBM.

N > 1: one rank per GPU over RCCL.  Under `torch.distributed.run` the ranks are already there
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); started as a plain `python bench.py
--gpus N` this process launches them itself (torch.distributed.run, 127.0.0.1) and passes their output
through.  The batch dimension is sharded: every rank runs its own 256 images, there is no data-path
collective; RCCL carries the barrier, the max-over-ranks of the elapsed time and the
gather of the per-rank times.  `value` is whole-job binary MACs per second.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration (HIP events on the
                launch stream around the timed region; one launch per step) vs the 8 TB/s HBM peak;
  compute       the same kernel against the other roofline (FP4 MFMA peak, or the measured
                v_xor+v_bcnt pair ceiling for the xor-popcount engine);
  extra         the other BASELINE configs: L0 with int8 / bitpacked output, the four QuickNet layer
                shapes, QuickNet (16 layers), QuickNetLarge (32 layers, config 4's per-GPU shard) and the
                Bi-RealNet-style int8 stack (12 layers, config 5) as REAL device-resident chains, the
                streaming ops;
  cpu_baseline  the CPU oracle (a port of the reference's two portable C++ formulations) on the host
                cores, on a bounded sample of the same workload (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
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


def roofline_fields(parts, sec):
    """the BINDING roofline of a measurement and both individual fractions.  `parts` = one (algorithmic
    bytes, binary MACs) pair per launch (a stack: one per layer): a launch cannot finish before max(bytes / 8 TB/s,
    2 * bMAC / 10 PFLOP/s), a stack not before the sum of that over its layers.  `frac` is that bound / the measured
    time; `bound` names the roofline that binds (for a stack: the one binding most of the bound's time)."""
    t_h = [b / (HBM_PEAK_GBS * 1e9) for b, _ in parts]
    t_m = [2.0 * m / (MFMA_FP4_PEAK_TFLOPS * 1e12) for _, m in parts]
    t_bound = sum(max(a, b) for a, b in zip(t_h, t_m))
    by_hbm = sum(a for a, b in zip(t_h, t_m) if a >= b)
    return {"bound": "hbm" if by_hbm >= t_bound - by_hbm else "mfma", "frac": t_bound / sec,
            "hbm_frac": sum(t_h) / sec, "mfma_frac": sum(t_m) / sec,
            "GBps_algorithmic": sum(b for b, _ in parts) / sec / 1e9}


def _event_time(torch, dev, fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / 1e3 / steps


SPINUP_MS = 40.0   # set from --spinup-ms


def spin_up(torch, dev, fn, ms=None):
    """Untimed load before a measurement: the clock governor needs ~10 ms of work to leave its idle state, and the
    host-side setup between two measurements (plans, operands) is long enough for it to fall back.  Returns the launches."""
    ms = SPINUP_MS if ms is None else ms
    t, n = time.perf_counter(), 0
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(16):
            fn()
        n += 16
        torch.cuda.synchronize(dev)
    return n


INFINITY_CACHE_BYTES = 256 << 20     # MI355X_MICROARCH.md: 256 MB Infinity Cache in front of HBM
ROTATE_BYTES = 320 << 20             # a rotation of operand sets is at least this large (and at least 4 sets)


class Rotation:
    """Several (input, output) buffer sets of one layer, used round-robin: K back-to-back launches of ONE set re-read an
    input and rewrite an output that fit the 256 MB Infinity Cache (every layer but L0's float output does), so such a
    timing is a cache number, not an HBM number (round-3 review).  With >= 4 sets of > 256 MB in total every launch's
    operands have been evicted by the time their turn comes again."""

    def __init__(self, torch, x, out):
        per_set = x.numel() * x.element_size() + out.numel() * out.element_size()
        self.n = int(max(4, -(-ROTATE_BYTES // max(1, per_set))))
        self.x = [x] + [x.clone() for _ in range(self.n - 1)]
        self.out = [out] + [torch.empty_like(out) for _ in range(self.n - 1)]
        self.i = 0
        self.total_bytes = per_set * self.n

    def run(self, plan):
        k = self.i
        self.i = (k + 1) % self.n
        plan.run(self.x[k], self.out[k])

    def describe(self):
        return "%d operand sets used round-robin, %.0f MB in total (> the 256 MB Infinity Cache)" % (self.n, self.total_bytes / 1e6)


def time_layer(amd, torch, layer, dst, steps, warmup, seed, dev, scale=1.0, zp=0, engine="auto", rotate=False):
    """Returns (mean seconds per step from stream events, kernel name, plan, x, out) -- with rotate=True a sixth element,
    the Rotation whose buffer sets the timed launches cycled through.
    Synthetic operands from tools/synthetic_layers.py -- the oracle is not involved."""
    w, mul, bias, thr = SL.weights(layer, seed)
    x = torch.from_numpy(SL.activations(layer, seed)).to(dev)
    plan = amd.Bconv2dPlan(layer.params(amd, dst, scale, zp))
    plan.set_weights(w, mul, bias, thr if dst == amd.BITPACKED else None)
    plan.set_option("engine", engine)
    dt = {amd.F32: torch.float32, amd.I8: torch.int8, amd.BITPACKED: torch.int32}[dst]
    out = torch.empty(plan.output_shape, dtype=dt, device=dev)
    plan.run(x, out)
    rot = Rotation(torch, x, out) if rotate else None
    fn = (lambda: rot.run(plan)) if rotate else (lambda: plan.run(x, out))
    spin_up(torch, dev, fn)
    for _ in range(max(1, warmup)):
        fn()
    torch.cuda.synchronize(dev)
    res = (_event_time(torch, dev, fn, steps), plan.kernel_name(), plan, x, out)
    return res + (rot,) if rotate else res


def graph_time(torch, dev, fn, steps, launches=20):
    """Seconds per launch of `fn` from a captured HIP graph of `launches` launches: a 10-20 us kernel launched from Python
    through ctypes is timed at the HOST's launch rate otherwise (rocprofv3 says 9.3 us where events around eager launches
    say 13.5, profiles/r03/layer_kernel_stats.txt).  None when the capture fails (a profiler attached to the process)."""
    def many():
        for _ in range(launches):
            fn()
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            many()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            many()
        graph.replay()
        torch.cuda.synchronize(dev)
        spin_up(torch, dev, graph.replay, ms=10.0)
        return _event_time(torch, dev, graph.replay, max(5, steps // 2)) / launches
    except Exception:
        torch.cuda.synchronize(dev)
        return None


def small_layer_entry(torch, dev, s_eager, plan, x, out, steps, rot=None):
    """(seconds per launch, how it was timed) for an `extra` layer: from a HIP graph when the eager figure is short enough to
    be the host's launch rate.  With a Rotation the launches (eager and captured alike) cycle through its operand sets, and
    the entry also carries the round-3 figure -- one operand set, served by the Infinity Cache -- as `ms_one_operand_set`."""
    how = {"timed_from": "back_to_back_launches"}
    s_ = s_eager
    if s_eager < 60e-6:
        launches = 20 if rot is None else max(20, rot.n)
        g = graph_time(torch, dev, (lambda: plan.run(x, out)) if rot is None else (lambda: rot.run(plan)), steps, launches)
        if g is not None:
            s_, how = g, {"timed_from": "hip_graph_x%d" % launches, "ms_eager_launches": s_eager * 1e3}
    if rot is not None:
        how["operands"] = rot.describe()
        one = graph_time(torch, dev, lambda: plan.run(x, out), steps) if s_eager < 60e-6 else None
        if one is None:
            one = _event_time(torch, dev, lambda: plan.run(x, out), steps)
        how["ms_one_operand_set"] = one * 1e3
    return s_, how


def cpu_baseline(target_seconds=12.0, reps=5):
    """Time the CPU oracle -- BOTH portable formulations of the reference (the direct
    loop of core/bconv2d/reference.h and the indirect BGEMM of core/indirect_bgemm/kernel_4x2_portable.h) --
    on the L0 layer: one image on one thread each (as the reference runs them), then the faster one with
    OpenMP over all host cores on a bounded number of images."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))   # the oracle's ctypes wrapper lives with the tests
    import oracle_lib as O
    import synth
    cores = os.cpu_count() or 1
    lib_note = "oracle/liblce_oracle.so (-O3 -march=x86-64-v3 -ffp-contract=off, OpenMP)"
    one = O.ConvSpec(batch=1, padding=O.PADDING_SAME, pad_values=1, **L0)
    x1, w, mul, bias = synth.conv_inputs(one, 1)
    single = {}
    def median_time(fn, *a, **kw):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn(*a, **kw)
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2], ts
    for name, fn in (("BConv2DReference-shaped (direct loop)", O.bconv2d), ("Kernel4x2Portable-shaped (indirect BGEMM)", O.bconv2d_indirect)):
        fn(one, O.DST_F32, x1, w, mul, bias, threads=1)             # (untimed first call)
        single[name], _ = median_time(fn, one, O.DST_F32, x1, w, mul, bias, threads=1)
    best_name = min(single, key=single.get)
    best = O.bconv2d if best_name.startswith("BConv2DReference") else O.bconv2d_indirect
    # a bounded sample: the whole batch where one repetition stays under target / reps seconds, fewer images otherwise
    per_image_parallel = single[best_name] / max(1, cores) * 1.3
    n = int(max(1, min(256, (target_seconds / reps) / max(per_image_parallel, 1e-6))))
    n = min(256, max(n, min(cores, 256)))
    spec = O.ConvSpec(batch=n, padding=O.PADDING_SAME, pad_values=1, **L0)
    x = synth.random_words(synth.rng(2), spec.input_shape(), spec.channels_in)
    best(spec, O.DST_F32, x, w, mul, bias, threads=cores)           # (untimed: thread pool start, page faults of the output)
    dt, all_dt = median_time(best, spec, O.DST_F32, x, w, mul, bias, threads=cores)
    return {"value": spec.binary_macs / dt, "unit": "binary-MAC/s", "cores": cores, "kind": "port",
            "sample": f"{n} of 256 images of the same layer, float output, median of {reps} repetitions of {dt:.2f} s wall "
                      f"(min {min(all_dt):.2f}, max {max(all_dt):.2f}), {best_name}; {lib_note}",
            "repetitions": reps,
            "single_thread_one_image": {k: {"ms": v * 1e3, "bmac_per_s": one.binary_macs / v, "median_of": reps} for k, v in single.items()}}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def launch_selftest(args):
    """No GPU work: checks the launcher + process-group plumbing of this file with gloo on the CPU
    (tests/test_batch_shard_gloo.py drives it at world size 2)."""
    import torch
    import torch.distributed as dist
    shard = importlib.import_module("compute-engine_amd.batch_shard")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    assert (dist.get_world_size() if world > 1 else 1) == args.gpus, "world size != --gpus"
    start, count = shard.shard_range(args.batch * world, world, rank)
    elapsed = shard.max_over_ranks(0.001 * (rank + 1), dist if world > 1 else None)
    per_rank = shard.gather_over_ranks(0.001 * (rank + 1), dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "n_gpus": world, "backend": "gloo", "max_ms": elapsed * 1e3,
                          "per_rank_ms": [v * 1e3 for v in per_rank], "shard_of_rank0": [start, count]}))
    if world > 1:
        dist.destroy_process_group()


def verify_sharding(amd, torch, shard, dist, dev, world, rank, share_gpu):
    """Untimed, before the barrier-bracketed region of an N > 1 run: proves that the line the run is about to print comes
    from N ranks on N distinct devices whose shards are right.  (i) every rank reports its device (UUID, name, compute
    units) and the RCCL version; without --share-gpu the N UUIDs must differ.  (ii) a small batch-sharded LceBconv2d
    (QuickNet's 14x14x256 layer, 8 N + 3 images: ragged shards) runs slab by slab on the ranks' GPUs, all_gather_batch
    reassembles it over the collective backend, and EVERY rank compares the gathered tensor bit for bit with its own
    full-batch run of the same seeded operands.  A mismatch ends the run: no line is better than a wrong one."""
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device_index": dev.index, "uuid": str(getattr(props, "uuid", "")), "name": props.name,
            "compute_units": props.multi_processor_count, "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    uuids = [d["uuid"] for d in everyone]
    if not share_gpu and (len(set(uuids)) != world or "" in uuids):
        raise SystemExit(f"bench.py: {world} ranks but the devices are not {world} distinct GPUs: {everyone}")
    gb = 8 * world + 3
    layer = SL.Layer(batch=gb, in_h=14, in_w=14, channels_in=256, filter_h=3, filter_w=3, channels_out=256,
                     padding=SL.PADDING_SAME, pad_values=1)
    w, mul, bias, thr = SL.weights(layer, 77)
    x = SL.activations(layer, 78)                                        # the same seeded operands on every rank

    def run(images):
        L = SL.Layer(**{**layer.__dict__, "batch": images.shape[0]})
        plan = amd.Bconv2dPlan(L.params(amd, amd.F32, 1.0, 0))
        plan.set_weights(w, mul, bias, None)
        y = plan.run(torch.from_numpy(np.ascontiguousarray(images)).to(dev))
        torch.cuda.synchronize(dev)
        return y

    start, count = shard.shard_range(gb, world, rank)
    local = run(x[start:start + count])
    gathered = shard.all_gather_batch(local.cpu() if share_gpu else local, gb, dist)
    whole = run(x)
    same = bool(torch.equal(gathered.to(dev).view(torch.int32), whole.view(torch.int32)))
    verdicts = [None] * world
    dist.all_gather_object(verdicts, same)
    if not all(verdicts):
        raise SystemExit(f"bench.py: the batch-sharded run differs from the single-GPU run on ranks "
                         f"{[r for r, v in enumerate(verdicts) if not v]}: refusing to report a throughput")
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None
    return {"devices": everyone, "distinct_devices": len(set(uuids)), "rccl_version": rccl,
            "sharded_equals_single_gpu_on_every_rank": True,
            "check": f"LceBconv2d 3x3 14x14x256->256, {gb} images in {world} ragged slabs, all_gather_batch over the run's "
                     f"collective backend, float output compared bit for bit on every rank"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (weak scaling: the global batch is N times this)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total images over all GPUs (strong scaling: BASELINE config 4 is --gpus 8 --global-batch 2048); "
                         "overrides --batch")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="measure roofline.traffic in this run: two extra rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) "
                         "over the same layer (rank 0, N == 1)")
    ap.add_argument("--spinup-ms", type=float, default=40.0, help="untimed clock spin-up before the warmup steps")
    ap.add_argument("--one-operand-set", action="store_true",
                    help="time the headline on ONE input / output set (rounds 1-5) instead of a rotation of >= 4 sets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--extra-json", default=os.path.join(ROOT, "gpurun_out", "bench_extra.json"),
                    help="where the full `extra` entries go (the JSON line carries a compact map of them); '' = nowhere")
    ap.add_argument("--launch-selftest", action="store_true", help="CPU-only check of the multi-rank launch path (gloo)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid for boxes with ONE GPU: every rank uses cuda:0 and the ranks meet over gloo, so the N > 1 "
                         "code path (shards, barriers, gathers, the config-4 chain) can run there; not a measurement")
    args = ap.parse_args()
    global SPINUP_MS
    SPINUP_MS = args.spinup_ms

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.launch_selftest:
        return launch_selftest(args)

    import torch
    amd = importlib.import_module("compute-engine_amd")
    shard = importlib.import_module("compute-engine_amd.batch_shard")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() < (local_rank + 1 if world > 1 and not args.share_gpu else 1):
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, the node has {torch.cuda.device_count()}")
    dist = None
    # LCE_BENCH_RCCL_WITH_ONE_RANK=1 (under the launcher, WORLD_SIZE=1): the N > 1 code path -- RCCL process group, self-check,
    # gathers, the sharded config-4 chain -- with a single rank, so that every collective call of this file has run against
    # the real RCCL on a one-GPU box (tests/test_gpu_multi.py); the line it prints is a one-GPU line all the same
    one_rank_rccl = world == 1 and os.environ.get("LCE_BENCH_RCCL_WITH_ONE_RANK") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or one_rank_rccl:
        import torch.distributed as dist
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank if world > 1 and not args.share_gpu else 0)
    cdev = None if args.share_gpu else dev          # where the collectives' scalars live
    torch.cuda.set_device(dev)

    verified = verify_sharding(amd, torch, shard, dist, dev, world, rank, args.share_gpu) if dist is not None else None

    # weak scaling (default): the global batch grows with N; --global-batch fixes it (strong scaling).  Either way
    # every rank owns a contiguous slab of it
    strong = args.global_batch > 0
    global_batch = args.global_batch if strong else args.batch * world
    _, my_batch = shard.shard_range(global_batch, world, rank)
    spec = SL.Layer(batch=my_batch, padding=SL.PADDING_SAME, pad_values=1, **L0)
    # warm up + per-step time from stream events (a step = every kernel of one LceBconv2d call)
    step_sec, kname, plan, x, out, rot = time_layer(amd, torch, spec, amd.F32, args.steps, args.warmup, 0, dev, rotate=True)
    # The headline's launches cycle through >= 4 operand sets (3.4 GB) like every `extra` layer's (round-5 review: with ONE set the
    # 25.7 MB input of launch k + 1 is still in the 256 MB Infinity Cache from launch k -- 3 % of the bytes): every launch reads its
    # input from HBM.  `--one-operand-set` restores the round-5 loop; the line says which it was.
    if args.one_operand_set:
        rot = None
    step = (lambda: plan.run(x, out)) if rot is None else (lambda: rot.run(plan))

    # the contract's timed region: barrier + sync, exactly K steps, sync + barrier, MAX over ranks
    def barrier():
        if dist is not None:
            if args.share_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev.index])
        torch.cuda.synchronize(dev)

    # Clock spin-up (untimed, before the W warmup steps): the chip's clock governor needs some 10 ms of load to
    # leave its idle state -- the first ~50 launches of a process run 10-35 % slower than the steady state
    # (profiles/r02/l0_clock_ramp.txt: 20 timed steps right after 5 warmups 0.274 ms, after 50 warmups 0.241 ms,
    # same box, same binary).  A fixed stretch of the same layer brings every run, short or long, to the state a
    # serving process is in; the timed region below is still exactly W warmup + K timed steps.
    spin_launches = spin_up(torch, dev, step)
    for _ in range(args.warmup):
        step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()                    # same stream the plan launches on (torch's current stream)
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    elapsed_local = time.perf_counter() - t0
    barrier()
    region_event_sec = evs[0].elapsed_time(evs[-1]) / 1e3 / args.steps
    per_step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step_ms[min(len(per_step_ms) - 1, int(q * len(per_step_ms)))]
    elapsed = shard.max_over_ranks(elapsed_local, dist, cdev)
    per_rank = shard.gather_over_ranks(elapsed_local, dist, cdev)

    # BASELINE config 4 at its size when N > 1 (never part of `value`): every rank runs ITS shard of a QuickNetLarge batch --
    # the 32 binary convolutions at batch 256 as one device-resident chain -- in the same barrier-bracketed way; the
    # global batch is 256 x N (2048 on 8 GPUs).  Every rank takes part in the barriers and the gather whatever happens to it.
    config4 = None
    if dist is not None and not args.no_extra:
        chain4, local4 = None, float("nan")
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import layer_chain
            # (this rank's share of the batch: 256 under weak scaling and for config 4 as stated, --gpus 8 --global-batch 2048)
            chain4 = layer_chain.LayerChain(amd, torch, SL.quicknet_layers(my_batch, (6, 8, 12, 6)), dev, dst="f32", seed=4000)
            chain4.run_chain()
            torch.cuda.synchronize(dev)
            spin_up(torch, dev, chain4.run_chain)
        except Exception:
            chain4 = None
        barrier()
        try:
            if chain4 is not None:
                t4 = time.perf_counter()
                for _ in range(args.steps):
                    chain4.run_chain()
                torch.cuda.synchronize(dev)
                local4 = time.perf_counter() - t4
        except Exception:
            local4 = float("nan")
        barrier()
        per_rank4 = shard.gather_over_ranks(local4, dist, cdev)
        if all(v == v for v in per_rank4):
            worst = max(per_rank4)
            shares = [shard.shard_range(global_batch, world, r)[1] for r in range(world)]
            config4 = {"workload": "BASELINE configs[3]: QuickNetLarge's 32 LceBconv2d layers, float outputs + fused sign words, "
                                   "device-resident chain, batch %s per GPU" % (shares[0] if len(set(shares)) == 1 else shares),
                       "global_batch": global_batch, "chain_ms": worst / args.steps * 1e3,
                       "images_per_s": global_batch * args.steps / worst,
                       "per_rank_chain_ms": [v / args.steps * 1e3 for v in per_rank4],
                       "per_rank_images_per_s": [n * args.steps / v for n, v in zip(shares, per_rank4)]}
        else:
            config4 = {"error": "a rank could not run the chain", "per_rank_seconds": per_rank4}
        del chain4

    per_image_bmacs = SL.Layer(batch=1, padding=SL.PADDING_SAME, pad_values=1, **L0).binary_macs
    total_bmacs = per_image_bmacs * global_batch * args.steps
    value = total_bmacs / elapsed
    abytes = spec.algorithmic_bytes(SL.DST_F32)
    mfma = kname.startswith(("bconv2d_mfma", "bconv2d_stream", "bconv2d_pointwise"))   # the matrix-core engine's kernels
    direct = not kname.startswith("bconv2d_mfma<")     # single kernel: the block expands its own input halo

    result = {
        "metric": "binary-MACs/sec (LceBconv2d 3x3 256->256, 56x56, batch 256/GPU, f32 out)",
        "value": value, "unit": "binary-MAC/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": ("fp4-e2m1 (exact +-1) matrix-core dot product, fp32 accumulate, f32 epilogue" if mfma
                  else "u32 xor+popcount, int32 accumulate, f32 epilogue"),
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: LceBconv2d 3x3 s1 SAME(pad_values=1) 56x56x256->256, "
                               "float32 output transform, device-resident bitpacked input",
                   "per_gpu_batch": my_batch if strong else args.batch, "global_batch": global_batch,
                   "parallelism": f"batch-shard x{world} (no data-path collective)"},
        "layer_latency_ms": step_sec * 1e3,
        "clock_spin_up": {"ms": args.spinup_ms, "launches": spin_launches, "note": "untimed, before the W warmup steps"},
        # spread of the K timed steps (the chip's power management moves the clock during a run)
        "ms_per_step_p10_p50_p90": [pct(0.1), pct(0.5), pct(0.9)],
        "per_gpu_value": value / world,
        "rccl_world_size": dist.get_world_size() if dist is not None else 1,
        "collective_backend": ("gloo (--share-gpu test aid)" if args.share_gpu else "nccl (RCCL)") if dist is not None else None,
        "per_rank_ms_per_step": [v / args.steps * 1e3 for v in per_rank],
        # every rank's OWN rate (its images' bMAC over its own elapsed time) and their sum: what the N ranks did side by side, next to
        # `value` = the whole job over the SLOWEST rank's time.  (Scaling efficiency is the driver's to compute from the per-N lines.)
        "per_rank_value": [per_image_bmacs * shard.shard_range(global_batch, world, r)[1] * args.steps / v for r, v in enumerate(per_rank)],
        "kernel": kname + ("+expand_fp4" if mfma and not direct else ""),
    }
    if verified is not None:
        result["multi_gpu_self_check"] = verified
    if config4 is not None:
        result["config4_quicknet_large_sharded"] = config4
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
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.measure_traffic and world == 1:
            traffic, traffic_src = measure_traffic(kname, my_batch)
        if traffic is None and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname, {}).get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this kernel, not measured in this run)"
            except Exception:
                traffic = None
        result["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                              "algorithmic_bytes_per_launch": abytes, "kernel": kname,
                              "operands": (rot.describe() if rot is not None else
                                           "one set; the 25.7 MB input of each launch is served by the Infinity Cache"),
                              "mfma_frac": 2.0 * spec.binary_macs / (MFMA_FP4_PEAK_TFLOPS * 1e12) / k_sec,
                              "kernel_ms": k_sec * 1e3, "expand_fp4_ms": e_sec * 1e3,
                              "kernel_ms_source": ("phase=gemm re-run, HIP events" if mfma and not direct else
                                                   "HIP events around the timed region / steps (one launch per step)"),
                              "note": "HBM is the binding roofline of this layer at spec peaks (0.106 ms vs 0.095 ms of FP4 "
                                      "MFMA); the output stream alone needs 0.137 ms at the 6.0 TB/s this chip writes "
                                      "(tools/probes/store_overlap.hip); see `compute` for the other roofline"}
        if mfma:
            tf = 2.0 * spec.binary_macs / k_sec / 1e12
            result["compute"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_FP4_PEAK_TFLOPS,
                                 "model": "v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4; the unscaled encoding in the streaming / pointwise kernels, "
                                          "v_mfma_scale_... in the block GEMM), 2 FLOP per binary MAC"}
        else:
            result["compute"] = {"bound": "valu", "achieved": spec.binary_macs / k_sec, "peak": VALU_BMAC_PEAK,
                                 "unit": "binary-MAC/s", "frac": spec.binary_macs / k_sec / VALU_BMAC_PEAK,
                                 "model": "v_xor_b32 + v_bcnt_u32_b32 per 32 bMAC, measured pair ceiling"}
        full_extra = None
        if not args.no_extra and world == 1 and not one_rank_rccl:
            full_extra = extra_measurements(amd, torch, spec, args, dev)
            # The driver keeps the last 8 KB of stdout: the line carries a COMPACT map of every extra measurement (round 4's
            # 13 KB line lost five layers from the driver's record); the full entries (kernel names, operand rotation, chain
            # variants) go to a file beside it.
            result["extra"] = compact_extra(full_extra)
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as exc:   # noqa: BLE001  (the oracle library missing on the box: the GPU line still prints)
                result["cpu_baseline"] = {"error": repr(exc)[:300]}
        if full_extra is not None and args.extra_json:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.extra_json)), exist_ok=True)
                with open(args.extra_json, "w") as f:
                    json.dump({**{k: v for k, v in result.items() if k != "extra"}, "extra": full_extra}, f, indent=1)
                result["extra_detail_file"] = os.path.relpath(os.path.abspath(args.extra_json), ROOT)
            except OSError as exc:
                result["extra_detail_file"] = "not written: %r" % (exc,)
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def measure_traffic(kname, batch):
    """HBM bytes per launch of the bench kernel from two rocprofv3 --pmc passes over tools/run_one.py (the same layer, plan
    and operands; counters collected in their own runs, --kernel-trace only), corrected as MI355X_MICROARCH.md prescribes:
    both counters are in KiB, and on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read (x2;
    calibrated in profiles/pmc_traffic.json on the LceQuantize stream, whose byte count is exact)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found: traffic not measured"
    tmp = tempfile.mkdtemp(prefix="lce_pmc_", dir="/tmp")
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                                sys.executable, os.path.join(ROOT, "tools", "run_one.py"), "56", "256", "f32", "auto", "auto", "10",
                                str(batch)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): traffic not measured"
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and row["Kernel_Name"].split("(")[0].find("bconv2d") >= 0:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, f"no {ctr} rows for the bench kernel: traffic not measured"
            got[ctr] = sum(per.values()) / len(per) * 1024.0, len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, writes = got["FETCH_SIZE"][0] * 2.0, got["WRITE_SIZE"][0]
    return int(fetch + writes), (f"measured in this run: rocprofv3 --pmc FETCH_SIZE ({got['FETCH_SIZE'][1]} launches, raw "
                                 f"{got['FETCH_SIZE'][0]:.0f} B x2 = {fetch:.0f} B) + --pmc WRITE_SIZE ({got['WRITE_SIZE'][1]} "
                                 f"launches, {writes:.0f} B), separate passes over tools/run_one.py, kernel {kname}")


def compact_extra(extra):
    """{name: [us, frac of the BINDING roofline, "hbm" | "mfma", hbm_frac, mfma_frac]} for single launches; stacks:
    [us of the convolutions alone, frac, us of the device-resident chain, bound, hbm_frac, mfma_frac].  `_format` says so in the line."""
    out = {"_format": "name: [us, frac, bound, hbm_frac, mfma_frac]; stacks: [convolutions_only us, frac, device_resident_chain us, bound, "
                      "hbm_frac, mfma_frac]; frac = max(bytes / 8 TB/s, 2 bMAC / 10 PF) summed over the launches / time; every "
                      "entry >= 20 launches; full entries in extra_detail_file"}
    r3 = lambda v: None if v is None else round(float(v), 3)
    for name, e in extra.items():
        if isinstance(e, str) and name.startswith("error_in_"):
            out[name] = e[:120]          # (a section of the extras failed: say so in the line)
        if not isinstance(e, dict):
            continue
        tail = [e.get("bound"), r3(e.get("hbm_frac")), r3(e.get("mfma_frac"))]
        if "error" in e:
            out[name] = "error"
        elif "convolutions_only_ms" in e:
            out[name] = [r3(e["convolutions_only_ms"] * 1e3), r3(e.get("frac")), r3(e.get("device_resident_chain_ms", 0.0) * 1e3)] + tail
        elif "ms" in e:
            out[name] = [r3(e["ms"] * 1e3), r3(e.get("frac"))] + (tail if e.get("bound") else [])
        elif "ms_per_invoke_device_resident" in e:
            out[name] = [r3(e["ms_per_invoke_device_resident"] * 1e3), None]
    return out


def extra_measurements(amd, torch, spec, args, dev):
    """Everything else BASELINE.json names, on the same GPU (N == 1 only): never part of `value`."""
    import layer_chain
    extra = {}
    # (every extra entry is the mean of >= 20 launches: the driver's --steps 20 used to leave them 5, round-5 review)
    st, wu = max(20, args.steps), 3
    sc, zp = 0.125, 3
    hbm = lambda b, s, m=0: roofline_fields([(b, m)], s)     # bound / frac / hbm_frac / mfma_frac of one launch
    def section_l0_other_outputs():
        for nm, dst, od in (("l0_int8_out", amd.I8, SL.DST_I8), ("l0_bitpacked_out", amd.BITPACKED, SL.DST_BITPACKED)):
            s_, kn, _p, _x, _o, rot_ = time_layer(amd, torch, spec, dst, st, wu, 1, dev, sc, zp, rotate=True)
            extra[nm] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn, **hbm(spec.algorithmic_bytes(od), s_, spec.binary_macs),
                         "operands": rot_.describe()}
            del _p, _x, _o, rot_

    def section_l0_other_engines():
        # the other matrix-core variant (FP4 workspace + GEMM whose tiles span images) and the xor-popcount
        # engine (the north star's literal formulation) on the same layer
        s_, kn, *_ = time_layer(amd, torch, spec, amd.F32, st, wu, 0, dev, engine="mfma")
        extra["l0_f32_workspace_gemm"] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn + "+expand_fp4"}
        s_, kn, *_ = time_layer(amd, torch, spec, amd.F32, st, wu, 0, dev, engine="valu")
        extra["l0_f32_valu_engine"] = {"ms": s_ * 1e3, "bmac_per_s": spec.binary_macs / s_, "kernel": kn,
                                       "valu_pair_frac": spec.binary_macs / s_ / VALU_BMAC_PEAK}

    def section_single_layers():
        # BASELINE config 3: the four QuickNet layer shapes, one at a time ...
        for hw, c in SL.QUICKNET_STAGES:
            sp = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=3, filter_w=3,
                          channels_out=c, padding=SL.PADDING_SAME, pad_values=1)
            s_, kn, pl_, x_, o_, rot_ = time_layer(amd, torch, sp, amd.F32, st, wu, hw, dev, rotate=True)
            s_, how = small_layer_entry(torch, dev, s_, pl_, x_, o_, st, rot_)
            del rot_
            extra[f"quicknet_{hw}x{hw}x{c}_f32"] = {"ms": s_ * 1e3, "bmac_per_s": sp.binary_macs / s_, "kernel": kn,
                                                    **hbm(sp.algorithmic_bytes(SL.DST_F32), s_, sp.binary_macs), **how}
            if c == 64:
                # ... the 64-channel layer with bitpacked output: the one instance that runs two resident blocks per CU
                s_, kn, pl_, x_, o_, rot_ = time_layer(amd, torch, sp, amd.BITPACKED, st, wu, hw + 3, dev, rotate=True)
                s_, how = small_layer_entry(torch, dev, s_, pl_, x_, o_, st, rot_)
                del rot_
                extra[f"conv_{hw}x{hw}x{c}_bitpacked_out"] = {"ms": s_ * 1e3, "bmac_per_s": sp.binary_macs / s_, "kernel": kn,
                                                              **hbm(sp.algorithmic_bytes(SL.DST_BITPACKED), s_, sp.binary_macs), **how}
            # ... and the 1x1 int8 + RELU layers of config 5's flavour (the HBM-bound cases)
            sp1 = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=1, filter_w=1,
                           channels_out=c, activation=SL.ACT_RELU)
            s_, kn, pl_, x_, o_, rot_ = time_layer(amd, torch, sp1, amd.I8, st, wu, hw + 1, dev, sc, zp, rotate=True)
            s_, how = small_layer_entry(torch, dev, s_, pl_, x_, o_, st, rot_)
            del rot_
            extra[f"pointwise_{hw}x{hw}x{c}_int8_relu"] = {"ms": s_ * 1e3, "bmac_per_s": sp1.binary_macs / s_, "kernel": kn,
                                                           **hbm(sp1.algorithmic_bytes(SL.DST_I8), s_, sp1.binary_macs), **how}
            del pl_, x_, o_


    def section_strided_pointwise():
        # ... and the strided 1x1 shortcut convolutions of ResNet-style binary nets (round 3: on the pointwise kernel)
        for hw, c in SL.QUICKNET_STAGES[:3]:
            sps = SL.Layer(batch=args.batch, in_h=hw, in_w=hw, channels_in=c, filter_h=1, filter_w=1, channels_out=2 * c, stride=2,
                           activation=SL.ACT_RELU)
            s_, kn, pl_, x_, o_, rot_ = time_layer(amd, torch, sps, amd.I8, st, wu, hw + 2, dev, sc, zp, rotate=True)
            s_, how = small_layer_entry(torch, dev, s_, pl_, x_, o_, st, rot_)
            del rot_
            # (a stride-2 1x1 layer reads only a quarter of its input pixels: count those)
            pix_out = args.batch * sps.out_h * sps.out_w
            by = pix_out * sps.in_words * 4 + sps.channels_out * sps.in_words * 4 + sps.channels_out * 8 + pix_out * sps.channels_out
            extra[f"pointwise_stride2_{hw}x{hw}x{c}_to_{2 * c}_int8_relu"] = {"ms": s_ * 1e3, "bmac_per_s": sps.binary_macs / s_, "kernel": kn,
                                                                             **hbm(by, s_, sps.binary_macs), **how}
            del pl_, x_, o_


    def section_feature_maps():
        # the north star's synthetic 224x224xC feature maps (3x3, C -> C, float output), 16 images = the pixel count of the
        # 56x56 layers at batch 256
        for c in (64, 128, 256):
            spf = SL.Layer(batch=max(1, args.batch // 16), in_h=224, in_w=224, channels_in=c, filter_h=3, filter_w=3,
                           channels_out=c, padding=SL.PADDING_SAME, pad_values=1)
            s_, kn, _p, _x, _o, rot_ = time_layer(amd, torch, spf, amd.F32, st, wu, 224 + c, dev, rotate=True)
            extra[f"feature_map_224x224x{c}_f32_batch{spf.batch}"] = {"ms": s_ * 1e3, "bmac_per_s": spf.binary_macs / s_, "kernel": kn,
                                                                     **hbm(spf.algorithmic_bytes(SL.DST_F32), s_, spf.binary_macs),
                                                                     "operands": rot_.describe()}
            del _p, _x, _o, rot_


    def section_stacks():
        # configs 3, 4 (one GPU's shard) and 5 as REAL stacks: every layer has its own plan, weights and
        # buffers, run back to back on one stream -- (a) the convolutions alone, each on its own input;
        # (b) the device-resident chain, each output quantized into the next layer's input by the same
        # epilogue (lce_hip_bconv2d_run_dual); (c) the same chain with a separate LceQuantize pass per layer
        def stack(name, layers, dst):
            try:
                ch = layer_chain.LayerChain(amd, torch, layers, dev, dst=dst, seed=4000)
                ch.run_chain()
                ch.run_convs()
                torch.cuda.synchronize(dev)
                spin_up(torch, dev, ch.run_convs)
                convs = _event_time(torch, dev, ch.run_convs, st)
                ch.run_chain(fused=True)    # (untimed: a plan's first call of each kind selects and uploads)
                fused = _event_time(torch, dev, lambda: ch.run_chain(fused=True), st)
                entry = {"layers": len(layers), "batch": args.batch, "convolutions_only_ms": convs * 1e3,
                         "bmac_per_s": ch.binary_macs / convs,
                         **roofline_fields([(L.algorithmic_bytes(SL.DST_F32 if dst == "f32" else SL.DST_I8), L.binary_macs) for L in layers], convs),
                         "device_resident_chain_ms": fused * 1e3}
                ch.run_chain(fused=False)
                entry["chain_with_separate_lcequantize_ms"] = _event_time(torch, dev, lambda: ch.run_chain(fused=False), st) * 1e3
                # launch gaps: the chain replayed from a captured HIP graph
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    ch.run_chain()
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    ch.run_chain()
                graph.replay()
                torch.cuda.synchronize(dev)
                # (the first replays after instantiation are slow; profiles/r03/graph_gaps.txt: inside a replay the kernels run
                # back to back exactly as eager launches do, between two replays the GPU idles ~9 us)
                spin_up(torch, dev, graph.replay)
                entry["device_resident_chain_hip_graph_ms"] = _event_time(torch, dev, graph.replay, st) * 1e3
                entry["launches_timed"] = st
                entry["kernels"] = sorted(set(ch.kernel_names()))
                entry["kernels_device_resident_chain"] = sorted(set(ch.kernel_names(fused=True)))
                extra[name] = entry
                del ch, graph
            except Exception as e:   # a report line must not take the bench down
                extra[name] = {"error": repr(e)[:300]}

        stack("quicknet_16_layers", SL.quicknet_layers(args.batch), "f32")
        stack("quicknet_large_32_layers_per_gpu_shard", SL.quicknet_layers(args.batch, (6, 8, 12, 6)), "f32")
        stack("birealnet_style_12_layers_int8_relu", SL.birealnet_layers(args.batch), "i8")


    def section_streams():
        # LceQuantize stream: float32 56x56x256 feature map, batch 256
        fx = torch.randn((args.batch, 56, 56, 256), device=dev)
        ow = amd.bitpack(fx)
        torch.cuda.synchronize(dev)
        spin_up(torch, dev, lambda: amd.bitpack(fx, out=ow))
        s_ = _event_time(torch, dev, lambda: amd.bitpack(fx, out=ow), st)
        qb = fx.numel() * 4 + ow.numel() * 4
        extra["lcequantize_f32_256x56x56x256"] = {"ms": s_ * 1e3, **hbm(qb, s_)}
        # LceDequantize (bits -> float) and LceBMaxPool2d (2x2 stride 2) on the same feature map
        fo = amd.unpack(ow, 256, torch.float32)
        torch.cuda.synchronize(dev)
        spin_up(torch, dev, lambda: amd.unpack(ow, 256, torch.float32, out=fo))
        s_ = _event_time(torch, dev, lambda: amd.unpack(ow, 256, torch.float32, out=fo), st)
        extra["lcedequantize_f32_256x56x56x256"] = {"ms": s_ * 1e3, **hbm(qb, s_)}
        po = amd.bmaxpool(ow, 2, 2, 2, 2, amd.PADDING_VALID)
        torch.cuda.synchronize(dev)
        # (a 5 us kernel: timed from a captured HIP graph, or the host's launch rate is what is measured; the launches cycle through
        # enough input / output sets to exceed the Infinity Cache -- with one set the 26 MB input is a cache read)
        n_sets = int(max(4, -(-ROTATE_BYTES // ((ow.numel() + po.numel()) * 4))))
        ows = [ow] + [ow.clone() for _ in range(n_sets - 1)]
        pos = [po] + [torch.empty_like(po) for _ in range(n_sets - 1)]
        pool_all = lambda: [amd.bmaxpool(ows[k], 2, 2, 2, 2, amd.PADDING_VALID, out=pos[k]) for k in range(n_sets)]
        timed_from = "hip_graph_x%d" % n_sets
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                pool_all()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                pool_all()
            graph.replay()
            torch.cuda.synchronize(dev)
            s_ = _event_time(torch, dev, graph.replay, st) / n_sets
        except Exception:   # (a profiler attached to the process can invalidate the capture)
            torch.cuda.synchronize(dev)
            timed_from, s_ = "back_to_back_launches", _event_time(torch, dev, pool_all, st) / n_sets
        pb = ow.numel() * 4 + po.numel() * 4
        extra["lcebmaxpool_2x2s2_256x56x56x256"] = {"ms": s_ * 1e3, **hbm(pb, s_),
                                                    "timed_from": timed_from,
                                                    "operands": "%d input / output sets used round-robin, %.0f MB in total" % (n_sets, n_sets * pb / 1e6)}
        del ows, pos
        del fx, ow, fo, po

    # A report line must not take the bench down: a section that fails (out of memory on a shared box, a kernel refusing a shape)
    # leaves an `error_in_<section>` entry and the others still run; the headline above never depends on any of them.
    for section in (section_l0_other_outputs, section_l0_other_engines, section_single_layers, section_strided_pointwise, section_feature_maps, section_stacks, section_streams):
        try:
            section()
        except Exception as exc:   # noqa: BLE001
            extra["error_in_" + section.__name__[len("section_"):]] = repr(exc)[:300]
            torch.cuda.synchronize(dev)
    try:
        extra["tflite_ops_chain_host_tensors"] = tflite_chain_timing()
    except Exception as exc:   # the op glue is optional for the headline
        extra["tflite_ops_chain_host_tensors"] = {"error": repr(exc)}
    return extra


def tflite_chain_timing(batch=64):
    """PCIe-INCLUSIVE (never part of `value`): LceQuantize -> LceBconv2d -> LceBMaxPool2d -> LceBconv2d through the registered
    TFLite ops with HOST tensors (the interpreter's arena), as a converted model's binary section runs them, with and without
    the ops' device residency (csrc/tflite/lce_ops.cc: once the host has declared its graph, tensors that only LCE ops read
    stay in HBM).  Synthetic operands; the
    chain driver stands in for the interpreter (csrc/tflite/single_op_driver.cc)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import flexbuf               # flexbuffer writer for the ops' custom options (test tooling, not the oracle)
    import tflite_driver as T    # ctypes front-end of the chain driver
    g = np.random.default_rng(11)
    h = w_ = 56
    c = 64
    x = g.standard_normal((batch, h, w_, c)).astype(np.float32)
    w1 = g.integers(-2**31, 2**31, (c, 3, 3, c // 32), dtype=np.int64).astype(np.int32)
    w2 = g.integers(-2**31, 2**31, (c, 3, 3, c // 32), dtype=np.int64).astype(np.int32)
    thr = np.full(c, 9 * c // 2, np.int32)
    mul, bias = np.ones(c, np.float32), np.zeros(c, np.float32)
    out = {}
    for label, on in (("resident_declared_graph", True), ("every_op_stages_its_tensors", False)):
        m = T.ChainModel()
        t_x = m.add_tensor(T.FLOAT32, x.shape, x)
        t_q = m.add_tensor(T.INT32, (0,) * 4)
        t_w1 = m.add_tensor(T.INT32, w1.shape, w1, allocation=T.MMAP_RO)
        t_t = m.add_tensor(T.INT32, (c,), thr, allocation=T.MMAP_RO)
        t_c = m.add_tensor(T.INT32, (0,) * 4)
        t_p = m.add_tensor(T.INT32, (0,) * 4)
        t_w2 = m.add_tensor(T.INT32, w2.shape, w2, allocation=T.MMAP_RO)
        t_m = m.add_tensor(T.FLOAT32, (c,), mul, allocation=T.MMAP_RO)
        t_b = m.add_tensor(T.FLOAT32, (c,), bias, allocation=T.MMAP_RO)
        t_y = m.add_tensor(T.FLOAT32, (0,) * 4)
        m.add_node("LceQuantize", [t_x], [t_q])
        m.add_node("LceBconv2d", [t_q, t_w1, -1, -1, t_t], [t_c], flexbuf.bconv2d_options(c, 1, 1, 1, 1, 0, 1, 0))
        m.add_node("LceBMaxPool2d", [t_c], [t_p], flexbuf.bmaxpool_options(2, 2, 2, 2, 1))
        m.add_node("LceBconv2d", [t_p, t_w2, t_m, t_b, -1], [t_y], flexbuf.bconv2d_options(c, 1, 1, 1, 1, 0, 1, 0))
        T.set_residency(on)
        if on:
            m.declare_graph([t_y])     # what an application does once (lce_ops_register.h, DeclareGraphForDeviceResidency);
                                       # without it every op leaves its output in the arena, as the reference does
        try:
            if m.prepare() != 0 or m.invoke() != 0:
                raise RuntimeError(m.log)
            T.transfer_counts(reset=True)
            t = time.perf_counter()
            n = 5
            for _ in range(n):
                if m.invoke() != 0:
                    raise RuntimeError(m.log)
            dt = (time.perf_counter() - t) / n
            up, down, upb, downb = T.transfer_counts()
            out[label] = {"ms_per_invoke": dt * 1e3, "uploads_per_invoke": up / n, "downloads_per_invoke": down / n,
                          "bytes_up_per_invoke": upb / n, "bytes_down_per_invoke": downb / n}
        finally:
            T.set_residency(True)
            m.close()
    out["workload"] = (f"{batch} x {h}x{w_}x{c} float -> LceQuantize -> LceBconv2d 3x3 {c}->{c} (bitpacked) -> LceBMaxPool2d 2x2 -> "
                       f"LceBconv2d 3x3 {c}->{c} (float): {x.nbytes / 1e6:.1f} MB up, {batch * 28 * 28 * c * 4 / 1e6:.1f} MB down, host wall time")
    return out


if __name__ == "__main__":
    main()
