"""Multi-GPU leg of bench.py on real devices: only runs where the box has more than one GPU (the driver's
8-GPU node); on a 1-GPU box it is skipped and the launcher is covered by the gloo tests."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs")
def test_bench_on_every_gpu_of_the_node_over_rccl():
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2",
                          "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == n and r["rccl_world_size"] == n and len(r["per_rank_ms_per_step"]) == n
    assert r["config"]["global_batch"] == 256 * n and r["scaling"] == "weak"
    assert r["value"] > 0.5 * n * r["per_gpu_value"] / 1.0001       # whole-job value = sum over ranks


def test_bench_single_gpu_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-extra",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][-1]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 5 and r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    assert r["kernel"].startswith("bconv2d_mfma_direct<f32")
