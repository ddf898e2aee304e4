#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MI355X-native Go2 walk environment (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one Go2Env.step over the whole batch (2 physics substeps + env logic).  N=1 runs
BASELINE.json configs[1]: Go2 walk, flat plane, 4096 envs, one MI355X.  N>1 (launched through
torch.distributed.run, one rank per GPU) shards envs across GPUs as independent batches of 4096
(weak scaling) and performs the rollout-statistics all-gather (3 floats/rank, RCCL) every 24 steps
(SURVEY.md section 8d config 4, section 8e).

Protocol: walk cfg of go2_train_walk.py:68-372, curriculum level frozen at level_init=0.10,
action set C = open-loop sine gait 0.3*sin(2*pi*1.5Hz*t + phase_leg) (+0 stiffness actions), inputs resident
in HBM before the timed region, synthetic data.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : dominant kernel, HIP-event timed in a separate profiling pass of the same workload
  cpu_baseline : the CPU oracle (oracle/libgo2sim_cpu.so, kind "port") timed on the host cores on a
                 bounded sample (rank 0, N=1 only).  The oracle is used here ONLY as the timed baseline.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from go2_sim2real_locomotion_rl_amd.capi import C, Go2Sim, load_hip_lib  # noqa: E402
from go2_sim2real_locomotion_rl_amd.configs import (build_stair_terrain, flatten_base_cfg, flatten_walk_cfg, get_jump_cfgs, get_stair_cfgs,  # noqa: E402
                                                    get_walk_cfgs, with_per_env_dr)
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model  # noqa: E402

ENVS_PER_GPU = 4096
ALGO_BYTES_WALK = 5701  # algorithmic HBM bytes per env-step, SURVEY.md section 8(d)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ROLLOUT_LEN = 24  # num_steps_per_env, go2_train_walk.py:60
NPRIV = {"walk": 104, "stairs": 182, "jump_dr": 45}
NOBS = {"walk": 49, "stairs": 49, "jump_dr": 45}
NACT = {"walk": 16, "stairs": 16, "jump_dr": 12}
WORKLOAD = "walk"
KERNEL_CLASSES = ["k_dynamics", "k_collide", "k_constraint_solve", "k_integrate_fk", "k_env_pre", "k_env_post(a+globals+b)", "misc", "env_step_total"]


def make_actions(n_steps, n_envs, device, dt=0.02):
    """Action set C of SURVEY.md 8d: trot-like open-loop sine on the 12 position actions."""
    t = torch.arange(n_steps, device=device, dtype=torch.float32)[:, None, None] * dt
    leg_phase = torch.tensor([0.0, math.pi, math.pi, 0.0], device=device)  # FR, FL, RR, RL
    joint_gain = torch.tensor([0.3, 1.0, 1.0], device=device)  # hip, thigh, calf
    phase = leg_phase[:, None].expand(4, 3).reshape(12)
    gain = joint_gain[None, :].expand(4, 3).reshape(12)
    env_phase = torch.linspace(0.0, 2 * math.pi, n_envs, device=device)[None, :, None]
    pos = 0.3 * gain * torch.sin(2 * math.pi * 1.5 * t + phase + env_phase)
    act = torch.zeros(n_steps, n_envs, NACT[WORKLOAD], device=device)
    act[:, :, :12] = pos
    return act.contiguous()


def pmc_traffic_bytes(kernel, n_envs):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json; PMC counters cannot be
    collected from inside the timed process).  FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, both in KB; only valid for 4096 envs."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if n_envs != ENVS_PER_GPU or not os.path.exists(path):
        return None
    rec = json.load(open(path)).get(kernel)
    return None if rec is None else int((2 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1000)


def make_sim(lib, n_envs, device_index, seed, workload):
    """Configured handle: walk (flat plane) or stairs (go2_train_stair.py terrain + cfg), curriculum frozen at its initial level."""
    sim = Go2Sim(lib, pack_model(), n_envs, device_index, seed)
    if workload == "jump_dr":       # BASELINE configs[4]: base env (go2_train_jump.py) + per-env friction / base-mass randomisation
        f, i, _ = flatten_base_cfg(n_envs, *with_per_env_dr(get_jump_cfgs()))
        sim.env_configure(f, i)
        sim.env_reset()
        return sim
    cfgs = get_stair_cfgs() if workload == "stairs" else get_walk_cfgs()
    if workload == "stairs":
        hf, info = build_stair_terrain(cfgs[0]["terrain"])
        sim.set_terrain(hf, info["horizontal_scale"], info["vertical_scale"], info["terrain_origin"])
    f, i, _ = flatten_walk_cfg(n_envs, *cfgs, freeze_curriculum=True)
    sim.env_configure(f, i)
    sim.env_reset()
    return sim


def cpu_baseline(n_envs, steps, warmup):
    from go2_sim2real_locomotion_rl_amd.capi import load_cpu_oracle_lib

    lib = load_cpu_oracle_lib()
    sim = make_sim(lib, n_envs, 0, 1, WORKLOAD)
    act = make_actions(steps + warmup, n_envs, torch.device("cpu")).numpy()
    obs = np.zeros((n_envs, NOBS[WORKLOAD]), np.float32); priv = np.zeros((n_envs, NPRIV[WORKLOAD]), np.float32)
    rew = np.zeros(n_envs, np.float32); rst = np.zeros(n_envs, np.uint8); to = np.zeros(n_envs, np.float32)
    for s in range(warmup):
        sim.env_step(act[s], obs, priv, rew, rst, to)
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        sim.env_step(act[s], obs, priv, rew, rst, to)
    dt = time.perf_counter() - t0
    return n_envs * steps / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--workload", choices=["walk", "stairs", "jump_dr"], default="walk",
                    help="walk = BASELINE configs[1] (the headline metric); stairs = configs[2] (heightfield terrain), jump_dr = configs[4] "
                         "(jump env + per-env mass / friction randomisation) -- both reported for information")
    ap.add_argument("--rollout", action="store_true",
                    help="closed rollout loop for information (NOT the headline metric): actions from ActorCritic.act (random-init weights), "
                         "RolloutStorage.add_transitions every step, compute_returns + global advantage statistics every 24 steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product has no CPU fallback)")
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on device 0 and a gloo process group
    rehearsal = os.environ.get("GO2SIM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=device)
        dist = dist_mod
    coll_device = torch.device("cpu") if rehearsal else device

    global WORKLOAD
    WORKLOAD = args.workload
    if args.rollout:                      # information-only mode: no per-kernel replay, no CPU baseline
        args.no_profile_pass = args.no_cpu_baseline = True
    B = args.envs_per_gpu
    sim = make_sim(load_hip_lib(), B, local_rank, 1 + rank, WORKLOAD)
    K, W = args.steps, args.warmup
    actions = make_actions(K + W, B, device)
    obs = torch.zeros(B, NOBS[WORKLOAD], device=device); priv = torch.zeros(B, NPRIV[WORKLOAD], device=device)
    rew = torch.zeros(B, device=device); rst = torch.zeros(B, dtype=torch.uint8, device=device); to = torch.zeros(B, device=device)
    stats = torch.zeros(3, device=device)
    gathered = torch.zeros(3 * world, device=coll_device) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    policy = storage = None
    if args.rollout:
        from go2_sim2real_locomotion_rl_amd import ActorCritic, RolloutStorage

        policy = ActorCritic(NOBS[WORKLOAD], NPRIV[WORKLOAD], NACT[WORKLOAD], [512, 256, 128], [512, 256, 128], activation="elu", init_noise_std=0.3,
                             device=device, seed=1 + rank)
        storage = RolloutStorage(ROLLOUT_LEN, B, device=device)

    def step(s):
        if policy is not None:                                           # closed loop: policy -> env -> storage (-> returns every 24 steps)
            a = policy.act(obs, priv)
            sim.env_step(a, obs, priv, rew, rst, to, stream)
            storage.add_transitions(s % ROLLOUT_LEN, rew, rst, policy.values, to, gamma=0.99)
            if (s + 1) % ROLLOUT_LEN == 0:
                storage.compute_returns(policy.evaluate(priv), 0.99, 0.95)      # all-gathers the advantage moments when world > 1
            return
        sim.env_step(actions[s], obs, priv, rew, rst, to, stream)
        if world > 1 and (s + 1) % ROLLOUT_LEN == 0:
            # rollout advantage-normalisation statistics: [sum, sum of squares, count] per rank, all-gathered over xGMI
            stats[0] = rew.sum(); stats[1] = (rew * rew).sum(); stats[2] = float(B)
            dist.all_gather_into_tensor(gathered, stats.to(coll_device))

    for s in range(W):
        step(s)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        step(s)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    errno = sim.check_errno()
    total_envs = B * world
    value = total_envs * K / elapsed

    # ---- HIP-event pass: the SAME workload replayed on a fresh handle (same seed => identical trajectories), with HIP events
    # recorded around every kernel launch on the launch stream.  Kept out of the timed region so `value` carries no event overhead.
    roofline = None
    if not args.no_profile_pass:
        sim2 = make_sim(load_hip_lib(), B, local_rank, 1 + rank, WORKLOAD)
        sim2.enable_timing(True)
        sim2.read_timing(reset=True)
        for s in range(W):
            sim2.env_step(actions[s], obs, priv, rew, rst, to, stream)
        torch.cuda.synchronize()
        ms_w, cnt_w = sim2.read_timing(reset=True)
        for s in range(W, W + K):
            sim2.env_step(actions[s], obs, priv, rew, rst, to, stream)
        torch.cuda.synchronize()
        ms, cnt = sim2.read_timing(reset=True)
        sim2.enable_timing(False)
        per_launch = [(ms[k] / cnt[k]) if cnt[k] else 0.0 for k in range(8)]
        per_launch_all = [((ms[k] + ms_w[k]) / (cnt[k] + cnt_w[k])) if cnt[k] + cnt_w[k] else 0.0 for k in range(8)]
        per_step = [ms[k] / K for k in range(8)]
        dom = int(np.argmax(per_step[:6]))
        # one launch of a substep kernel advances B envs by one substep = half an env-step (2 substeps/step);
        # env kernels run once per env-step
        units = B * (0.5 if dom < 4 else 1.0)
        achieved = ALGO_BYTES_WALK * units / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": KERNEL_CLASSES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(KERNEL_CLASSES[dom], B) if WORKLOAD == "walk" else None,
            "avg_launch_ms": round(per_launch[dom], 4), "avg_launch_ms_incl_warmup": round(per_launch_all[dom], 4),
            "launches_timed": cnt[dom], "algo_bytes_per_env_step": ALGO_BYTES_WALK, "units_per_launch_env_steps": units,
            "ms_per_step_by_kernel": {KERNEL_CLASSES[k]: round(per_step[k], 4) for k in range(8)},
            "whole_step_achieved_GBs": round(value * ALGO_BYTES_WALK / 1e9, 3),
        }
        del sim2

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_cpu_envs, n_cpu_steps = 4096, 1500   # ~15-20 s on the 256 host threads of the GPU box
        v, dt = cpu_baseline(n_cpu_envs, n_cpu_steps, 3)
        cpu = {"value": round(v, 1), "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
               "sample": f"CPU oracle (OpenMP over envs, all host cores), same walk cfg/action set, {n_cpu_envs} envs x {n_cpu_steps} steps after 3 warm-up steps ({dt:.1f} s)"}

    if rank == 0:
        out = {
            "metric": "env-steps/sec (all envs) Go2 walk 4096 envs; 1/2/4/8-GPU scaling", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("Go2 walk flat-plane, num_envs=4096 per GPU, 2 substeps x dt 0.01, action set C (open-loop sine gait), curriculum frozen at level 0.10"
                                    if WORKLOAD == "walk" else
                                    "Go2 stairs heightfield terrain (BASELINE configs[2], NOT the headline metric), num_envs per GPU as given, action set C, curriculum frozen at level 0.65"
                                    if WORKLOAD == "stairs" else
                                    "Go2 jump env + per-env mass / friction randomisation (BASELINE configs[4], NOT the headline metric), num_envs per GPU as given, action set C on 12 position actions"),
                       "envs_per_gpu": B, "total_envs": total_envs, "parallelism": f"env-shard x{world}", "errno": errno,
                       "loop": "closed rollout loop: ActorCritic.act + env step + RolloutStorage (information only)" if args.rollout else "env step, open-loop actions"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
