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


# ---- one batch sharded over two ranks keeps ONE curriculum level and ONE set of global DR scalars ----
def _shard_cfgs():
    from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs

    cfgs = get_walk_cfgs()
    cfgs[0]["episode_length_s"] = 0.3                                   # 15-step episodes: resets (and time-outs) inside a short run
    cfgs[0]["curriculum"].update({"update_every_episodes": 20, "global_dr_update_interval": 10, "ready_streak": 1, "cooldown_updates": 0,
                                  "ready_timeout_rate": 0.0, "ready_tracking": -1.0, "ready_fall_rate": 1.0, "hard_fall_rate": 2.0})   # always "ready": the level rises
    return cfgs


def _shard_worker(rank, world, port, n_local, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from go2_sim2real_locomotion_rl_amd import build
        from go2_sim2real_locomotion_rl_amd.capi import Go2Sim, load_cpu_oracle_lib
        from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg
        from go2_sim2real_locomotion_rl_amd.distributed import sync_env_globals
        from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

        sim = Go2Sim(load_cpu_oracle_lib(), pack_model(), n_local, 0, shard_seed(5, rank))
        f, i, _ = flatten_walk_cfg(n_local, *_shard_cfgs(), shared_globals=True)
        sim.env_configure(f, i)
        sync_env_globals(sim, initial=True)                             # what Go2Env / bench.make_sim do for a sharded batch, before the constructor's reset
        sim.env_reset()
        g0 = sim.env_globals()
        kp0 = np.zeros((n_local, 104), np.float32)
        obs = np.zeros((n_local, 49), np.float32); priv = np.zeros((n_local, 104), np.float32); rew = np.zeros(n_local, np.float32)
        rst = np.zeros(n_local, np.uint8); to = np.zeros(n_local, np.float32)
        rng = np.random.default_rng(100 + rank)
        log = [{"start": True, "level": g0.level, "t_sample": g0.t_sample, "friction": g0.friction, "mass_shift": g0.mass_shift, "com_shift": list(g0.com_shift),
                "leg_mass_shift": list(g0.leg_mass_shift), "sync_calls": g0.sync_calls, "throttle": g0.global_dr_reset_counter,
                "geom_friction": sim.get_field_np(C_("F_GEOM_FRICTION"))[:, 0].tolist()}]
        for s in range(steps):
            sim.env_step((0.3 * rng.standard_normal((n_local, 16))).astype(np.float32), obs, priv, rew, rst, to)
            if (s + 1) % 6 == 0:
                summed, dr = sync_env_globals(sim)
                g = sim.env_globals()
                log.append({"summed": summed.tolist(), "dr": dr.tolist(), "level": g.level, "t_sample": g.t_sample, "friction": g.friction,
                            "mass_shift": g.mass_shift, "curr_ep_total": g.curr_ep_total, "priv_friction": priv[:, 52].tolist(),
                            "geom_friction": sim.get_field_np(C_("F_GEOM_FRICTION"))[:, 0].tolist()})
        torch.save(log, os.path.join(out_dir, f"shard{rank}.pt"))
    finally:
        dist.destroy_process_group()


def C_(name):
    from go2_sim2real_locomotion_rl_amd.capi import C

    return C["GO2SIM_" + name]


def test_sharded_env_keeps_one_curriculum_and_one_global_dr(tmp_path):
    from go2_sim2real_locomotion_rl_amd import build

    build.build_oracle(verbose=False)
    world, n_local, steps = 2, 6, 60
    mp.spawn(_shard_worker, args=(world, _free_port(), n_local, steps, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"shard{r}.pt") for r in range(world))
    # after construction every shard stands where the SINGLE-PROCESS env (same seed as rank 0, all envs in one handle) starts: t_sample at level_init and
    # the first global draws, applied to every env (without the initial sync the shards sampled the easy end until the first rollout ended)
    from go2_sim2real_locomotion_rl_amd.capi import Go2Sim, load_cpu_oracle_lib
    from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg
    from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

    one = Go2Sim(load_cpu_oracle_lib(), pack_model(), world * n_local, 0, shard_seed(5, 0))
    f, i, _ = flatten_walk_cfg(world * n_local, *_shard_cfgs())
    one.env_configure(f, i); one.env_reset()
    g1 = one.env_globals()
    for x in (a[0], b[0]):
        assert x["start"] and x["sync_calls"] == 1
        assert x["level"] == g1.level and x["t_sample"] == g1.t_sample and x["t_sample"] > 0.0
        assert x["friction"] == g1.friction and x["mass_shift"] == g1.mass_shift and x["com_shift"] == list(g1.com_shift) and x["leg_mass_shift"] == list(g1.leg_mass_shift)
        assert x["throttle"] == g1.global_dr_reset_counter
        assert set(x["geom_friction"]) == {g1.friction}
    a, b = a[1:], b[1:]
    assert len(a) == len(b) == steps // 6
    levels = []
    for x, y in zip(a, b):
        assert x["summed"] == y["summed"]                                                   # the all-reduced counters
        assert x["level"] == y["level"] and x["t_sample"] == y["t_sample"] and x["curr_ep_total"] == y["curr_ep_total"]
        assert x["dr"] == y["dr"] and x["friction"] == y["friction"] and x["mass_shift"] == y["mass_shift"]   # rank 0's draws everywhere
        assert set(x["geom_friction"]) == set(y["geom_friction"]) == {x["friction"]}        # applied to every env of both shards
        levels.append(x["level"])
    assert sum(s["summed"][0] for s in a) >= 2 * n_local * 3                                # every env was reset several times
    assert levels[-1] > levels[0] >= 0.10                                                   # the shared level moved (summed counters crossed 20 episodes)
    assert len({s["friction"] for s in a}) > 1                                              # the throttled friction draw happened more than once


@pytest.mark.gpu
def test_shared_globals_hip_matches_oracle(oracle_lib, hip_lib, blob):
    """GO2SIM_IC_SHARED_GLOBALS on the HIP library: counters, state machine, draws and their application equal the oracle's bit for bit."""
    from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg
    from go2_sim2real_locomotion_rl_amd.distributed import sync_env_globals
    from util import CpuEnv, GpuEnv, bits_equal

    n, steps = 48, 60
    mut = lambda env_cfg, *_: (env_cfg.update(_shard_cfgs()[0]))
    cpu = CpuEnv(oracle_lib, blob, n, seed=5, mutate=mut, shared_globals=True); gpu = GpuEnv(hip_lib, blob, n, seed=5, mutate=mut, shared_globals=True)
    cpu.reset(); gpu.reset()
    rng = np.random.default_rng(3)
    n_resets = 0
    for s in range(steps):
        a = (0.3 * rng.standard_normal((n, 16))).astype(np.float32)
        oc, pc, rc, dc, tc = cpu.step(a); og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg) and bits_equal(oc, og) and bits_equal(pc, pg) and bits_equal(rc, rg), f"step {s}"
        if (s + 1) % 6 == 0:
            sc, drc = sync_env_globals(cpu.sim); sg, drg = sync_env_globals(gpu.sim)
            assert np.array_equal(sc, sg) and np.array_equal(drc, drg)
            n_resets += sc[0]
            gc, gg = cpu.sim.env_globals(), gpu.sim.env_globals()
            assert gc.level == gg.level and gc.t_sample == gg.t_sample and gc.sync_calls == gg.sync_calls and gc.friction == gg.friction
            assert bits_equal(cpu.field("F_GEOM_FRICTION"), gpu.field("F_GEOM_FRICTION")) and bits_equal(cpu.field("F_MASS_SHIFT"), gpu.field("F_MASS_SHIFT"))
            assert bits_equal(cpu.field("F_LINK_POS"), gpu.field("F_LINK_POS"))
    assert gc.level > 0.10 and n_resets >= 2 * n


@pytest.mark.gpu
def test_device_array_sync_entry_points_equal_host_forms(hip_lib, blob):
    """go2sim_env_sync_counters_dev / _sync_apply_dev / _set_global_dr_dev (what the RCCL path of sync_env_globals calls: device float64 arrays, no host
    round trip) leave a handle in the state the host forms leave its twin in, bit for bit, including the initial sync before the constructor's reset."""
    from util import GpuEnv, bits_equal

    n, steps = 40, 36
    mut = lambda env_cfg, *_: (env_cfg.update(_shard_cfgs()[0]))
    dev = torch.device("cuda:0")
    envs = [GpuEnv(hip_lib, blob, n, seed=5, mutate=mut, shared_globals=True) for _ in range(2)]

    def sync(e, device_form, initial=False):
        if not device_form:
            c = e.sim.env_sync_counters()
            if initial:
                c[4] += n
            dr = e.sim.env_sync_apply(c)
            e.sim.env_set_global_dr(dr)
            return c, dr
        c = torch.empty(5, dtype=torch.float64, device=dev); dr = torch.empty(10, dtype=torch.float64, device=dev)
        e.sim.env_sync_counters_dev(c)
        if initial:
            c[4] += float(n)
        e.sim.env_sync_apply_dev(c, dr)
        e.sim.env_set_global_dr_dev(dr)
        torch.cuda.synchronize()
        return c.cpu().numpy(), dr.cpu().numpy()

    for e, form in zip(envs, (False, True)):
        sync(e, form, initial=True)
        e.reset()
    rng = np.random.default_rng(3)
    for s in range(steps):
        a = (0.3 * rng.standard_normal((n, 16))).astype(np.float32)
        outs = [e.step(a) for e in envs]
        assert all(bits_equal(x, y) for x, y in zip(outs[0], outs[1])), f"step {s}"
        if (s + 1) % 6 == 0:
            (c0, d0), (c1, d1) = sync(envs[0], False), sync(envs[1], True)
            assert np.array_equal(c0, c1) and np.array_equal(d0, d1)
            g0, g1 = envs[0].sim.env_globals(), envs[1].sim.env_globals()
            assert g0.as_dict() == g1.as_dict()
            assert bits_equal(envs[0].field("F_GEOM_FRICTION"), envs[1].field("F_GEOM_FRICTION")) and bits_equal(envs[0].field("F_LINK_POS"), envs[1].field("F_LINK_POS"))
    assert g0.sync_calls >= 3 and g0.level > 0.10          # (an apply without counted resets draws nothing and is not numbered)
