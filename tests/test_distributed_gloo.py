"""N>1 path on CPU: world_size-2 gloo run of the env-shard helpers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from go2_sim2real_locomotion_rl_amd.distributed import global_mean_std, normalize_advantages, shard_envs, shard_seed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.random.default_rng(0).standard_normal(total).astype(np.float32))
        start, count = shard_envs(total, world, rank)
        local = full[start:start + count]
        mean, std = global_mean_std(local)
        norm = normalize_advantages(local)
        torch.save({"mean": mean, "std": std, "norm": norm, "start": start, "count": count, "seed": shard_seed(7, rank)}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_shard_helpers():
    assert shard_envs(32768, 8, 3) == (3 * 4096, 4096)
    parts = [shard_envs(10, 4, r) for r in range(4)]
    assert parts == [(0, 3), (3, 3), (6, 2), (8, 2)] and sum(c for _, c in parts) == 10
    assert shard_seed(1, 5) == 6


def test_global_advantage_stats_world2(tmp_path):
    total, world = 1001, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    full = np.random.default_rng(0).standard_normal(total).astype(np.float32)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for o in outs:
        assert float(o["mean"]) == pytest.approx(float(full.mean()), abs=1e-6)
        assert float(o["std"]) == pytest.approx(float(full.std(ddof=1)), rel=1e-5)          # torch.std: unbiased
    norm = torch.cat([o["norm"] for o in outs]).numpy()
    ft = torch.from_numpy(full)
    assert np.allclose(norm, ((ft - ft.mean()) / (ft.std() + 1e-8)).numpy(), atol=1e-5)   # rsl_rl 2.2.4 PPO formula
    assert [o["seed"] for o in outs] == [7, 8]


def test_single_process_fallback():
    x = torch.arange(10, dtype=torch.float32)
    mean, std = global_mean_std(x)
    assert float(mean) == pytest.approx(4.5) and float(std) == pytest.approx(float(x.std()), rel=1e-6)
