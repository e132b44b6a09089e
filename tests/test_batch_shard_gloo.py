"""Multi-process (gloo, world_size 2, 3, 4 and 8) test of the batch-sharded path: shards are a
partition of the batch, every rank's slab equals the same rows of the full-batch result,
the all-gather reassembles it, and the timing reduction takes the max."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
shard = importlib.import_module("compute-engine_amd.batch_shard")


def test_shard_range_is_a_partition():
    for b in (0, 1, 7, 256, 2048, 2049):
        for w in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(b, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == b
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard.shard_range(2048, 8, 3) == (768, 256)       # BASELINE config 4
    with pytest.raises(ValueError):
        shard.shard_range(8, 2, 2)


def _worker(rank, world, port, global_batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth
    kw = dict(in_h=6, in_w=5, channels_in=64, filter_h=3, filter_w=3, channels_out=40,
              padding=O.PADDING_SAME, pad_values=1)
    full = O.ConvSpec(batch=global_batch, **kw)
    x, w, mul, bias = synth.conv_inputs(full, 77)                 # same seed on every rank
    start, count = shard.shard_range(global_batch, world, rank)
    local = O.bconv2d(O.ConvSpec(batch=count, **kw), O.DST_F32, x[start:start + count], w, mul, bias) \
        if count else np.zeros((0,) + full.output_shape(O.DST_F32)[1:], np.float32)
    gathered = shard.all_gather_batch(torch.from_numpy(local), global_batch, dist).numpy()
    want = O.bconv2d(full, O.DST_F32, x, w, mul, bias)
    t = shard.max_over_ranks(0.001 * (rank + 1), dist)
    dist.barrier()
    q.put((rank, bool(np.array_equal(gathered, want)), t))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,global_batch", [(2, 6), (2, 5), (3, 4), (4, 7), (8, 11), (8, 5)])      # (8 ranks = the driver's largest run; 5 images: three ranks with empty shards)
def test_sharded_run_equals_full_batch(world, global_batch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in results) == list(range(world))
    assert all(ok for _, ok, _ in results)
    assert all(abs(t - 0.001 * world) < 1e-9 for _, _, t in results)      # MAX over ranks


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (torch.distributed.run on
    127.0.0.1) -- the path the driver's scaling runs take.  --launch-selftest swaps the GPU work for the
    process-group plumbing over gloo, so the launch, the world-size check, the max-over-ranks and the
    per-rank gather are exercised on the CPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                          "--launch-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE line
    r = lines[0]
    assert r["n_gpus"] == world and r["launch_selftest"] and len(r["per_rank_ms"]) == world
    assert abs(r["max_ms"] - world) < 1e-6 and r["per_rank_ms"] == pytest.approx([k + 1.0 for k in range(world)])
    assert r["shard_of_rank0"] == [0, 256]      # weak scaling: 256 images per rank


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "world size != --gpus" in (out.stderr + out.stdout)
