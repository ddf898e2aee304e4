#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MI355X-native Go2 walk environment (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one Go2Env.step over the whole batch (2 physics substeps + env logic).  N=1 runs
BASELINE.json configs[1]: Go2 walk, flat plane, 4096 envs, one MI355X.  N>1 (launched through
torch.distributed.run, one rank per GPU) shards envs across GPUs as independent batches of 4096
(weak scaling); the env writes its rewards / done flags straight into the rollout storage and every 24 steps the rollout's
GAE pass produces the advantage moments, which are all-gathered (3 float64 per rank, RCCL) and applied.

Headline protocol (`value`): walk cfg of go2_train_walk.py:68-372, curriculum frozen at level_init=0.10, action set C =
open-loop sine gait 0.3*sin(2*pi*1.5Hz*t + phase_leg) (+0 stiffness actions), inputs resident in HBM before the timed region, W warm-up
steps from the reset, then exactly K timed steps.  With a short W the timed window is the landing / contact-onset transient.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries (N=1 only, all measured in this process):
  roofline      : dominant kernel of the timed window, HIP-event timed in a replay of the same workload (the north-star HBM roofline)
  roofline_issue: the roofline that fits the path: vector issue-slot utilisation, lane occupancy and waves per SIMD of the same kernel (stamped SQ
                  counter file under profiles/), and the single-env latency floor (step time at 256 envs, measured live)
  rollouts      : closed collection loops (policy + env + storage): stairs = the counterpart of the reference's logged collection time
  steady_state  : the SAME handle continued: settle phase, then >= 1000 timed steps (+ per-kernel ms from 200 event-timed steps)
  action_sets   : SURVEY 8d sets A (zeros), B (0.5*N(0,1): falls / resets) and C (sine gait), each 200 warm-up + 1000 timed steps; C_staggered_resets = set C with
                  the episode counters spread over the episode, so that a few envs are reset on every step (what a training run looks like)
  curriculum_live : set C with the metric-gated curriculum running (not frozen)
  workloads     : BASELINE configs[2] (stairs) and configs[4] (jump + per-env mass / friction DR) at the same env count
  ref_protocol_fps : the reference's own `go2` benchmark protocol (tests/test_rigid_benchmarks.py:316-374) through go2sim_scene_step
  cpu_baseline  : the CPU oracle (oracle/libgo2sim_cpu.so, kind "port") on all host cores and on one thread, bounded samples.
                  The oracle is used here ONLY as the timed baseline.
"""
import argparse
import ctypes
import hashlib
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
# algorithmic HBM bytes per env-step, SURVEY.md section 8(d) table (walk flat / stairs / base env)
ALGO_BYTES = {"walk": 5701, "stairs": 7165, "jump_dr": 4300}
ALGO_BYTES_WALK = ALGO_BYTES["walk"]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ROLLOUT_LEN = 24  # num_steps_per_env, go2_train_walk.py:60
NPRIV = {"walk": 104, "stairs": 182, "jump_dr": 45}
NOBS = {"walk": 49, "stairs": 49, "jump_dr": 45}
NACT = {"walk": 16, "stairs": 16, "jump_dr": 12}
# timing classes of go2sim_enable_timing.  One env step on flat ground = 7 launches: k_pre_dynamics (pre-physics part + dynamics of substep 1), collide,
# k_solve_integrate<true> (solve + integrate + FK of substep 1 + dynamics of substep 2), collide, k_solve_integrate<false> (solve + integrate + FK), k_env_post_a,
# k_env_post_b.  On the heightfield (one env per solver wavefront) the solve, k_integrate_fk_dynamics and k_integrate_fk stay separate launches: 9.
# The fused launches are accounted with the class of their first half: "k_dynamics" = k_pre_dynamics, "k_constraint_solve" = the two k_solve_integrate launches on flat
# ground ("k_integrate_fk" is then empty); "k_env_pre" is only non-zero with GO2SIM_NO_FUSE=1.
KERNEL_CLASSES = ["k_dynamics", "k_collide", "k_constraint_solve", "k_integrate_fk", "k_env_pre", "k_env_post(a+globals+b)", "misc", "env_step_total"]
LAUNCHES_PER_ENV_STEP = {"walk": 7, "jump_dr": 7, "stairs": 9}
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
PMC_TRAFFIC_FILES = {"walk": PMC_TRAFFIC_FILE, "stairs": os.path.join(ROOT, "profiles", "r04_stairs_pmc_traffic.json")}   # per workload (tools/profile_round.sh)
PMC_SQ_FILES = {5: os.path.join(ROOT, "profiles", "r04_pmc_sq.json"), 300: os.path.join(ROOT, "profiles", "r04_steady_pmc_sq.json")}   # by warm-up: landing window / steady gait
KERNEL_OF_CLASS = {"k_dynamics": "k_pre_dynamics", "k_collide": "k_collide", "k_constraint_solve": "k_constraint_solve", "k_integrate_fk": "k_integrate_fk_dynamics",   # (flat ground: the stamped files map k_constraint_solve to k_solve_integrate_team)
                   "k_env_post(a+globals+b)": "k_env_post_a"}


def source_hash():
    """sha256 over the HIP sources + headers the library is built from: stamps profiles/*_pmc_traffic.json (tools/profile_round.sh) so that a
    counter file measured on other kernels is not reported for this build."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "go2_sim2real_locomotion_rl_amd", "csrc", f) for f in ("go2sim.hip", "go2sim_policy.hip", "go2sim_gjk_dev.h")]
    inc = os.path.join(ROOT, "include")
    files += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    for f in files:
        if os.path.exists(f):
            h.update(open(f, "rb").read())
    return h.hexdigest()


def make_actions(n_steps, n_envs, device, dt=0.02, workload="walk", kind="C", seed=0):
    """SURVEY.md 8d action sets: A zeros; B 0.5*N(0,1) (falls, resets, worst-case contacts); C trot-like open-loop sine on the 12 position actions."""
    na = NACT[workload]
    if kind == "A":
        return torch.zeros(n_steps, n_envs, na, device=device)
    if kind == "B":
        g = torch.Generator(device="cpu").manual_seed(1234 + seed)
        return (0.5 * torch.randn(n_steps, n_envs, na, generator=g)).to(device).contiguous()
    t = torch.arange(n_steps, device=device, dtype=torch.float32)[:, None, None] * dt
    leg_phase = torch.tensor([0.0, math.pi, math.pi, 0.0], device=device)  # FR, FL, RR, RL
    joint_gain = torch.tensor([0.3, 1.0, 1.0], device=device)  # hip, thigh, calf
    phase = leg_phase[:, None].expand(4, 3).reshape(12)
    gain = joint_gain[None, :].expand(4, 3).reshape(12)
    env_phase = torch.linspace(0.0, 2 * math.pi, n_envs, device=device)[None, :, None]
    pos = 0.3 * gain * torch.sin(2 * math.pi * 1.5 * t + phase + env_phase)
    act = torch.zeros(n_steps, n_envs, na, device=device)
    act[:, :, :12] = pos
    return act.contiguous()


def pmc_traffic_bytes(kernel, n_envs, workload="walk"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (PMC counters cannot be collected from inside the timed
    process).  FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, both in KB; only valid for 4096 envs AND for the sources the file was
    measured on: a file whose source hash or kernel list does not match this build is refused (returns None, with the reason)."""
    if n_envs != ENVS_PER_GPU:
        return None, "traffic file is for 4096 envs"
    path = PMC_TRAFFIC_FILES.get(workload)
    if path is None:
        return None, f"counters are not collected for the {workload} workload"
    if not os.path.exists(path):
        return None, "no traffic file for this round"
    doc = json.load(open(path))
    if doc.get("source_sha256") != source_hash():
        return None, "traffic file was measured on different sources (stale): refused"
    rec = doc.get("kernels", {}).get(kernel)
    if rec is None:
        return None, f"kernel {kernel} not in the traffic file"
    return int((2 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024), None


def roofline_issue_of(kernel_class, n_envs, warmup, workload, latency_floor):
    """The roofline that FITS this path: the kernels move 0.3 TB/s of 8 and run no matrix instructions -- what a launch costs is the
    instruction stream of its slowest wavefront.  Reported for the dominant kernel from the committed, source-stamped SQ counter pass
    (tools/profile_round.sh -> tools/make_pmc_sq.py): the share of the chip's vector issue slots the launch uses over its duration, the same while the
    mean wave is alive, lanes active per vector instruction, waves per SIMD; plus, measured live, the single-environment latency floor (the step time
    at 256 envs: what one env's chain through the nine launches costs when the chip is almost empty)."""
    out = {"bound": "valu-issue / slowest-wave latency", "kernel": KERNEL_OF_CLASS.get(kernel_class, kernel_class), "peak": 1.0, "unit": "fraction of vector issue slots (4 cycles per wave64 instruction, 1024 SIMDs)",
           "latency_floor": latency_floor}
    path = PMC_SQ_FILES.get(5 if warmup < 100 else 300)
    note = None
    if workload != "walk" or n_envs != ENVS_PER_GPU:
        note = "SQ counters are collected for the walk workload at 4096 envs"
    elif not os.path.exists(path):
        note = "no SQ counter file for this round"
    else:
        doc = json.load(open(path))
        rec = doc.get("kernels", {}).get(out["kernel"])
        if doc.get("source_sha256") != source_hash():
            note = "SQ counter file was measured on different sources (stale): refused"
        elif rec is None:
            note = f"kernel {out['kernel']} not in the SQ counter file"
        else:
            out.update({"achieved": rec.get("valu_issue_util"), "frac": rec.get("valu_issue_util"), "valu_busy_while_mean_wave_alive": rec.get("valu_busy_while_alive"),
                        "lane_occupancy": rec.get("lane_occupancy"), "waves_per_simd": rec.get("waves_per_simd"), "avg_us_under_pmc": rec.get("avg_us_under_pmc"),
                        "source": os.path.relpath(path, ROOT), "window": "landing window (steps 5..25)" if warmup < 100 else "steady gait (steps 300..700)"})
    if note:
        out.update({"achieved": None, "frac": None, "note": note})
    return out


def latency_floor(device, local_rank, stream, W, K, n_envs=256):
    """Step time of the same workload and window at 256 envs: every kernel then holds at most one wave on a quarter of the SIMDs, so the step time
    is the length of one environment's chain through the launches."""
    sim = make_sim(load_hip_lib(), n_envs, local_rank, 1, "walk")
    act = make_actions(W + K, n_envs, device)
    buf = Buffers(n_envs, "walk", device)
    for s in range(W):
        sim.env_step(act[s], buf.obs, buf.priv, buf.rew, buf.rst, buf.to, stream)
    dt = timed_run(sim, act, W, K, buf, stream)
    del sim
    return {"n_envs": n_envs, "ms_per_step": round(dt / K * 1e3, 4), "window": f"steps {W}..{W + K} after the reset"}


def rollout_run(B, device, local_rank, workload, n_warm_rollouts, n_rollouts, stream):
    """`--workload <w> --rollout` semantics as one number: policy inference (ActorCritic.act, random-init weights) + env step + RolloutStorage every step,
    GAE + advantage statistics every 24 steps -- what rsl_rl's collection phase does; the counterpart of the reference's logged collection time."""
    from go2_sim2real_locomotion_rl_amd import ActorCritic, RolloutStorage

    sim = make_sim(load_hip_lib(), B, local_rank, 1, workload)
    buf = Buffers(B, workload, device)
    storage = RolloutStorage(ROLLOUT_LEN, B, device=device)
    policy = ActorCritic(NOBS[workload], NPRIV[workload], NACT[workload], [512, 256, 128], [512, 256, 128], activation="elu", init_noise_std=0.3, device=device, seed=1)

    def rollout():
        for t in range(ROLLOUT_LEN):
            a = policy.act(buf.obs, buf.priv)
            sim.env_step(a, buf.obs, buf.priv, buf.rew, buf.rst, buf.to, stream)
            storage.add_transitions(t, buf.rew, buf.rst, policy.values, buf.to, gamma=0.99)
        storage.compute_returns(policy.evaluate(buf.priv), 0.99, 0.95)

    for _ in range(n_warm_rollouts):
        rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_rollouts):
        rollout()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": round(B * ROLLOUT_LEN * n_rollouts / dt, 1), "unit": "env-steps/s", "collection_time_s_per_rollout": round(dt / n_rollouts, 6), "n_envs": B,
           "rollouts_timed": n_rollouts, "rollouts_warmup": n_warm_rollouts, "errno": sim.check_errno(),
           "loop": "ActorCritic.act (fused MLP + sampling) -> go2sim_env_step -> RolloutStorage.add_transitions, x 24; then evaluate + compute_returns (GAE, advantage normalisation)"}
    del sim
    return out


def make_sim(lib, n_envs, device_index, seed, workload, freeze_curriculum=True, shared_globals=False):
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
    f, i, _ = flatten_walk_cfg(n_envs, *cfgs, freeze_curriculum=freeze_curriculum, shared_globals=shared_globals)
    sim.env_configure(f, i)
    if shared_globals:       # first sync of a sharded batch before the constructor's reset: every shard starts at rank 0's t_sample / global draws
        from go2_sim2real_locomotion_rl_amd.distributed import sync_env_globals

        sync_env_globals(sim, None, None, initial=True)
    sim.env_reset()
    return sim


class Buffers:
    def __init__(self, B, workload, device):
        self.obs = torch.zeros(B, NOBS[workload], device=device); self.priv = torch.zeros(B, NPRIV[workload], device=device)
        self.rew = torch.zeros(B, device=device); self.rst = torch.zeros(B, dtype=torch.uint8, device=device); self.to = torch.zeros(B, device=device)


def timed_run(sim, actions, first, n, buf, stream):
    """n un-instrumented steps (actions[first .. first+n)), bracketed by device synchronisation -> seconds."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(first, first + n):
        sim.env_step(actions[s % actions.shape[0]], buf.obs, buf.priv, buf.rew, buf.rst, buf.to, stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def kernel_ms(sim, actions, first, n, buf, stream):
    """per-kernel-class ms of n event-timed steps on `sim` (plain launches with HIP events on the launch stream) -> (ms[8], cnt[8])"""
    sim.enable_timing(True)
    sim.read_timing(reset=True)
    for s in range(first, first + n):
        sim.env_step(actions[s % actions.shape[0]], buf.obs, buf.priv, buf.rew, buf.rst, buf.to, stream)
    torch.cuda.synchronize()
    ms, cnt = sim.read_timing(reset=True)
    sim.enable_timing(False)
    return ms, cnt


def protocol_run(B, device, local_rank, seed, workload, kind, warm, steps, stream, freeze=True, stagger=False):
    """fresh handle, `warm` warm-up steps from the reset, `steps` timed steps -> dict.  stagger: the episode counters are spread over the episode length, so a
    few envs time out and are reset on EVERY step (what a training run looks like) instead of all together"""
    sim = make_sim(load_hip_lib(), B, local_rank, seed, workload, freeze_curriculum=freeze)
    if stagger:
        g_ = torch.Generator(device="cpu").manual_seed(7)
        sim.env_set_episode_length(torch.randint(0, 1000, (B,), generator=g_, dtype=torch.int32).to(device), stream)
    act = make_actions(min(warm + steps, 600), B, device, workload=workload, kind=kind)   # sets A / C are periodic, B is i.i.d.: the tape is cycled
    buf = Buffers(B, workload, device)
    resets = 0
    for s in range(warm):
        sim.env_step(act[s % act.shape[0]], buf.obs, buf.priv, buf.rew, buf.rst, buf.to, stream)
    dt = timed_run(sim, act, warm, steps, buf, stream)
    g = sim.env_globals(stream)
    resets = int(g.reset_calls)
    out = {"value": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "warmup": warm, "steps": steps, "errno": sim.check_errno(),
           "reset_calls": resets, "curriculum_level": round(float(g.level), 4)}
    del sim
    return out


def ref_protocol(B, device, local_rank, stream, warm=1000, steps=1000, robot="go2"):
    """The reference's own rigid benchmarks (tests/test_rigid_benchmarks.py): plane + robot, dt = 0.01 with ONE substep per scene.step, engine position
    control, FPS = steps * n_envs / elapsed.
      go2      (:316-374): default gains kp 100 / kv 10 (genesis/utils/geom.py:2042-2047) holding the standing pose, joint angles initialised uniformly
               inside their limits;
      anymal_c (:378-412): anymal_c.urdf at z = 0.8, kp 1000 on the 12 motors, position targets 0 -- the second robot of the benchmark set that the
               model compiler handles (tools/compile_go2_model.py --robot anymal_c; same kernels, 14 collision geoms padded to the compile-time 28).
    Differences, stated: the ground is the plane.urdf box of the Go2Env scene (the benchmarks use gs.morphs.Plane), and warm-up / record are counted in
    steps (10 s + 10 s of simulated time) instead of 45 s + 15 s of wall clock."""
    from go2_sim2real_locomotion_rl_amd.model_blob import MODEL_JSON, load_model_json

    model = load_model_json(MODEL_JSON if robot == "go2" else os.path.join(os.path.dirname(MODEL_JSON), f"{robot}_model.json"))
    sim = Go2Sim(load_hip_lib(), pack_model(model), B, local_rank, 1)
    lim = np.array([[d["limit"][0], d["limit"][1]] for d in model["dofs"]], np.float32)[6:]
    eff = [abs(d["force_range"][1]) for d in model["dofs"]][6:]
    kp = 100.0 if robot == "go2" else 1000.0
    for k in range(12):
        sim.set_dof_gains(6 + k, kp, 10.0, -eff[k], eff[k])
    rng = np.random.default_rng(0)
    qpos = np.tile(np.asarray(model["qpos0"], np.float32)[:, None], (1, B))
    ctrl = np.zeros((18, B), np.float32)
    if robot == "go2":
        qpos[7:] = lim[:, :1] + (lim[:, 1:] - lim[:, :1]) * rng.random((12, B), dtype=np.float32)
        ctrl[6:] = np.array([0.0, 0.0, 0.0, 0.0, 0.8, 0.8, 1.0, 1.0, -1.5, -1.5, -1.5, -1.5], np.float32)[:, None]
    mode = np.zeros((18, B), np.int32); mode[6:] = 2
    put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    sim.set_field(C["GO2SIM_F_QPOS"], put(qpos), stream); sim.set_field(C["GO2SIM_F_CTRL_POS"], put(ctrl), stream)
    sim.set_field(C["GO2SIM_I_CTRL_MODE"], put(mode), stream)
    sim.reset_caches(None, 0, stream); sim.forward_kinematics(stream)
    for _ in range(warm):
        sim.scene_step(1, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.scene_step(1, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nc = torch.zeros(1, B, dtype=torch.int32, device=device); sim.get_field(C["GO2SIM_I_N_CONSTRAINTS"], nc, stream)
    torch.cuda.synchronize()
    where = "316-374 (go2, Newton)" if robot == "go2" else "378-412 (anymal, Newton)"
    out = {"value": round(B * steps / dt, 1), "unit": "FPS = scene steps (dt 0.01, 1 substep) x n_envs / s", "realtime_factor": round(B * steps / dt * 0.01, 1),
           "n_envs": B, "warmup_steps": warm, "steps": steps, "errno": sim.check_errno(), "constraint_rows_mean": round(float(nc.float().mean()), 2),
           "constraint_rows_max": int(nc.max()), "protocol": f"tests/test_rigid_benchmarks.py:{where}; ground = plane.urdf box; step-counted warm-up"}
    del sim
    return out


def go2env_class_run(B, device, local_rank, seed, W, K):
    """The headline window through the Go2Env CLASS (go2_env.Go2Env.step: fresh observation tensors per step, extras, errno poll), i.e. what
    rsl_rl's OnPolicyRunner sees; same cfg, seed and action tape as `value`, with the per-step extras snapshot on and off."""
    from go2_sim2real_locomotion_rl_amd import go2_env

    go2_env.init(seed=seed, device_index=local_rank)
    out = {}
    for key, log in (("log_extras_on", True), ("log_extras_off", False)):
        env = go2_env.Go2Env(B, *get_walk_cfgs(), seed=seed, device=device, freeze_curriculum=True, log_extras=log)
        env.reset()
        act = make_actions(W + K, B, device)
        for s in range(W):
            env.step(act[s])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(W, W + K):
            env.step(act[s])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = {"value": round(B * K / dt, 1), "ms_per_step": round(dt / K * 1e3, 4)}
        del env
    out["note"] = f"Go2Env.step x {K} after {W} warm-up steps from the reset; `value` of the headline times the C ABI underneath"
    return out


def ref_logged():
    """The only throughput figures the reference tree holds for this path (logs/test1 TensorBoard scalars of a 4096-env stairs run, hardware
    unstated), extracted by tools/extract_ref_tb_scalars.py into tests/golden/ref_test1_tb_scalars.json."""
    p = os.path.join(ROOT, "tests", "golden", "ref_test1_tb_scalars.json")
    if not os.path.exists(p):
        return None
    d = json.load(open(p)).get("derived", {})
    rows = d.get("per_iteration", [])
    return {"source": "logs/test1/events.out.tfevents.* (go2_train_stair.py run, 4096 envs x 24 steps per iteration, policy inference included, hardware unstated)",
            "collection_env_steps_per_s": [round(r["collection_env_steps_per_s"], 1) for r in rows],
            "total_fps": [r["total_fps"] for r in rows], "compare_with": "rollouts.stairs (policy + env + storage, the like-for-like loop) and workloads.stairs (env step only)"}


def cpu_baseline(n_envs, steps, warmup, threads=None):
    from go2_sim2real_locomotion_rl_amd.capi import load_cpu_oracle_lib

    lib = load_cpu_oracle_lib()
    if threads is not None:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    sim = make_sim(lib, n_envs, 0, 1, "walk")
    act = make_actions(steps + warmup, n_envs, torch.device("cpu")).numpy()
    obs = np.zeros((n_envs, NOBS["walk"]), np.float32); priv = np.zeros((n_envs, NPRIV["walk"]), np.float32)
    rew = np.zeros(n_envs, np.float32); rst = np.zeros(n_envs, np.uint8); to = np.zeros(n_envs, np.float32)
    for s in range(warmup):
        sim.env_step(act[s], obs, priv, rew, rst, to)
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        sim.env_step(act[s], obs, priv, rew, rst, to)
    dt = time.perf_counter() - t0
    return n_envs * steps / dt, dt


def roofline_of(ms, cnt, K, B, workload, value):
    per_launch = [(ms[k] / cnt[k]) if cnt[k] else 0.0 for k in range(8)]
    per_step = [ms[k] / K for k in range(8)]
    dom = int(np.argmax(per_step[:6]))
    # one launch of a substep kernel advances B envs by one substep = half an env-step (2 substeps/step); env kernels run once per env-step
    units = B * (0.5 if dom < 4 else 1.0)
    algo = ALGO_BYTES[workload]
    achieved = algo * units / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
    traffic, note = pmc_traffic_bytes(KERNEL_CLASSES[dom], B, workload)
    r = {"bound": "hbm", "kernel": KERNEL_CLASSES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": round(per_launch[dom], 4), "launches_timed": cnt[dom],
         "algo_bytes_per_env_step": algo, "units_per_launch_env_steps": units,
         "ms_per_step_by_kernel": {KERNEL_CLASSES[k]: round(per_step[k], 4) for k in range(8)}, "launches_per_env_step": LAUNCHES_PER_ENV_STEP[workload],
         "whole_step_achieved_GBs": round(value * algo / 1e9, 3)}
    if note:
        r["traffic_note"] = note
    return r


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
    ap.add_argument("--no-extras", action="store_true", help="skip steady_state / action_sets / curriculum_live / workloads / ref_protocol_fps")
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

    WORKLOAD = args.workload
    if args.rollout:                      # information-only mode: no per-kernel replay, no CPU baseline
        args.no_profile_pass = args.no_cpu_baseline = args.no_extras = True
    B = args.envs_per_gpu
    # N > 1: one batch sharded over the ranks keeps ONE curriculum level and ONE set of global DR scalars: distributed.sync_env_globals
    sim = make_sim(load_hip_lib(), B, local_rank, 1 + rank, WORKLOAD, shared_globals=(world > 1 and WORKLOAD != "jump_dr"))
    K, W = args.steps, args.warmup
    SS_SETTLE, SS_STEPS, SS_KSTEPS = 150, 1000, 200
    n_tape = K + W + (max(0, SS_SETTLE - (K + W)) + SS_STEPS + SS_KSTEPS if world == 1 else 0)
    actions = make_actions(n_tape, B, device, workload=WORKLOAD)
    buf = Buffers(B, WORKLOAD, device)
    obs, priv, rew, rst, to = buf.obs, buf.priv, buf.rew, buf.rst, buf.to
    stream = torch.cuda.current_stream().cuda_stream

    policy = storage = None
    zeros_B = None
    if args.rollout or world > 1:
        from go2_sim2real_locomotion_rl_amd import RolloutStorage

        storage = RolloutStorage(ROLLOUT_LEN, B, device=device)
        zeros_B = torch.zeros(B, device=device)
        if rehearsal:                                                     # gloo moves host tensors
            import go2_sim2real_locomotion_rl_amd.rollout as _ro
            _ag = _ro.allgather_moments
            _ro.allgather_moments = lambda m, group=None: _ag(m.cpu(), group).to(m.device)
    if args.rollout:
        from go2_sim2real_locomotion_rl_amd import ActorCritic

        policy = ActorCritic(NOBS[WORKLOAD], NPRIV[WORKLOAD], NACT[WORKLOAD], [512, 256, 128], [512, 256, 128], activation="elu", init_noise_std=0.3,
                             device=device, seed=1 + rank)

    if world > 1:
        from go2_sim2real_locomotion_rl_amd import distributed as _d

        _sync_globals = _d.sync_env_globals                              # moves host tensors under gloo (rehearsal), device tensors under nccl

    def step(s):
        if policy is not None:                                           # closed loop: policy -> env -> storage (-> returns every 24 steps)
            a = policy.act(obs, priv)
            sim.env_step(a, obs, priv, rew, rst, to, stream)
            storage.add_transitions(s % ROLLOUT_LEN, rew, rst, policy.values, to, gamma=0.99)
            if (s + 1) % ROLLOUT_LEN == 0:
                storage.compute_returns(policy.evaluate(priv), 0.99, 0.95)      # all-gathers the advantage moments when world > 1
            return
        if storage is not None:
            # sharded job: the env writes rewards / done flags of transition t straight into the rollout storage (the step's output pointers are
            # per-step arguments), and every 24 steps the rollout's GAE pass (zero value estimates: no critic in the open loop) yields the
            # advantage moments [sum, sum of squares, count], which are all-gathered over RCCL / xGMI and applied -- the one exchange of the path
            t = s % ROLLOUT_LEN
            sim.env_step(actions[s], obs, priv, storage.rewards[t], storage.dones[t], to, stream)
            if t == ROLLOUT_LEN - 1:
                storage.compute_returns(zeros_B, 0.99, 0.95)
                if WORKLOAD != "jump_dr":                                    # all-reduce of the curriculum counters + broadcast of rank 0's global DR draws
                    _sync_globals(sim, None, stream)
            return
        sim.env_step(actions[s], obs, priv, rew, rst, to, stream)

    if dist is not None:
        # The three collectives of a rollout end (all-gather of the advantage moments, all-reduce of the curriculum counters, broadcast of the global DR
        # scalars) are issued once on dummy tensors before anything is timed: RCCL sets up channels lazily per collective type, and with the driver's
        # window (20 steps from step 5) the first rollout end falls inside the timed region.
        _w3 = torch.zeros(3, device=coll_device, dtype=torch.float64)
        _wg = torch.empty(3 * world, device=coll_device, dtype=torch.float64)
        dist.all_gather_into_tensor(_wg, _w3)
        _w5 = torch.zeros(5, device=coll_device, dtype=torch.float64)
        dist.all_reduce(_w5, op=dist.ReduceOp.SUM)
        _w10 = torch.zeros(10, device=coll_device, dtype=torch.float64)
        dist.broadcast(_w10, src=0)
        torch.cuda.synchronize()
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
    using_graph, graph_fallbacks = sim.graph_status()
    total_envs = B * world
    value = total_envs * K / elapsed
    extras_on = rank == 0 and world == 1 and not args.no_extras

    # ---- steady state: the same handle continues -- settle until 150 steps after the reset, then >= 1000 un-instrumented timed steps, then
    # 200 event-timed steps for the per-kernel split
    steady = None
    if extras_on:
        pos = W + K
        settle = max(0, SS_SETTLE - pos)
        for s in range(pos, pos + settle):
            sim.env_step(actions[s], obs, priv, rew, rst, to, stream)
        pos += settle
        dt_ss = timed_run(sim, actions, pos, SS_STEPS, buf, stream)
        pos += SS_STEPS
        v_ss = B * SS_STEPS / dt_ss
        ms, cnt = kernel_ms(sim, actions, pos, SS_KSTEPS, buf, stream)
        r = roofline_of(ms, cnt, SS_KSTEPS, B, WORKLOAD, v_ss)
        steady = {"value": round(v_ss, 1), "unit": "env-steps/s", "ms_per_step": round(dt_ss / SS_STEPS * 1e3, 4), "settle_steps_after_reset": pos - SS_STEPS,
                  "steps": SS_STEPS, "errno": sim.check_errno(), "ms_per_step_by_kernel": r["ms_per_step_by_kernel"],
                  "roofline": {k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "units_per_launch_env_steps")}}

    # ---- HIP-event pass of the headline window: the SAME workload replayed on a fresh handle (same seed => identical trajectories), with HIP
    # events recorded around every kernel launch on the launch stream.  Kept out of the timed region so `value` carries no event overhead.
    roofline = roofline_issue = None
    if not args.no_profile_pass and world == 1:
        sim2 = make_sim(load_hip_lib(), B, local_rank, 1 + rank, WORKLOAD)
        sim2.enable_timing(True)
        for s in range(W):
            sim2.env_step(actions[s], obs, priv, rew, rst, to, stream)
        torch.cuda.synchronize()
        sim2.read_timing(reset=True)
        for s in range(W, W + K):
            sim2.env_step(actions[s], obs, priv, rew, rst, to, stream)
        torch.cuda.synchronize()
        ms, cnt = sim2.read_timing(reset=True)
        sim2.enable_timing(False)
        roofline = roofline_of(ms, cnt, K, B, WORKLOAD, value)
        del sim2
        roofline_issue = roofline_issue_of(roofline["kernel"], B, W, WORKLOAD, latency_floor(device, local_rank, stream, W, K) if WORKLOAD == "walk" else None)

    action_sets = curriculum_live = workloads = refp = refp_anymal = envcls = rollouts = None
    if extras_on:
        del sim
        envcls = go2env_class_run(B, device, local_rank, 1 + rank, W, K)
        action_sets = {k: protocol_run(B, device, local_rank, 1, "walk", k, 200, 1000, stream) for k in ("A", "B", "C")}
        action_sets["note"] = "SURVEY 8d: A zeros (standing), B 0.5*N(0,1) (falls / resets), C open-loop sine gait; 200 warm-up + 1000 timed steps each, curriculum frozen at 0.10"
        action_sets["C_staggered_resets"] = protocol_run(B, device, local_rank, 1, "walk", "C", 200, 1000, stream, stagger=True)
        action_sets["C_staggered_resets"]["note"] = ("set C with the episode counters spread over the 1000-step episode: about four envs time out, are re-drawn and dropped on EVERY step, "
                                                     "as in a training run (reset path, full-batch FK refresh and a few landing robots in every launch)")
        curriculum_live = protocol_run(B, device, local_rank, 1, "walk", "C", 200, 1000, stream, freeze=False)
        workloads = {"stairs": protocol_run(B, device, local_rank, 1, "stairs", "C", 100, 300, stream),
                     "jump_dr": protocol_run(B, device, local_rank, 1, "jump_dr", "C", 100, 300, stream),
                     "note": "BASELINE configs[2] (stair heightfield, level 0.65) and configs[4] (jump env + per-env mass / friction DR), set C, env-steps/s, 100 warm-up + 300 timed steps"}
        rollouts = {"stairs": rollout_run(B, device, local_rank, "stairs", 5, 20, stream), "walk": rollout_run(B, device, local_rank, "walk", 5, 20, stream),
                    "note": "closed collection loop (policy inference + env step + rollout storage, 24-step rollouts, 4096 envs): `stairs` is the like-for-like counterpart of "
                            "ref_logged.collection_env_steps_per_s (137.6 k / 93.5 k env-steps/s in the reference's logs/test1, hardware unstated); vs_baseline stays null"}
        refp = ref_protocol(B, device, local_rank, stream)
        refp_anymal = ref_protocol(B, device, local_rank, stream, robot="anymal_c")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_cpu_envs, n_cpu_steps = 4096, 1000   # ~10-15 s on the 256 host threads of the GPU box
        v, dt = cpu_baseline(n_cpu_envs, n_cpu_steps, 3)
        v1, dt1 = cpu_baseline(64, 150, 3, threads=1)   # the reference's CPU default is one thread (genesis/__init__.py:252-263)
        cpu = {"value": round(v, 1), "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
               "sample": f"CPU oracle (OpenMP over envs, all host cores), same walk cfg/action set C from the reset, {n_cpu_envs} envs x {n_cpu_steps} steps after 3 warm-up steps ({dt:.1f} s)",
               "single_thread": {"value": round(v1, 1), "cores": 1, "sample": f"same oracle on ONE thread, 64 envs x 150 steps after 3 warm-up steps ({dt1:.1f} s)"}}

    if rank == 0:
        what = {"walk": f"Go2 walk flat-plane, num_envs={B} per GPU, 2 substeps x dt 0.01, action set C (open-loop sine gait), curriculum frozen at level 0.10",
                "stairs": f"Go2 stairs heightfield terrain (BASELINE configs[2], NOT the headline metric), num_envs={B} per GPU, action set C, curriculum frozen at level 0.65",
                "jump_dr": f"Go2 jump env + per-env mass / friction randomisation (BASELINE configs[4], NOT the headline metric), num_envs={B} per GPU, action set C on 12 position actions"}[WORKLOAD]
        out = {
            "metric": "env-steps/sec (all envs) Go2 walk 4096 envs; 1/2/4/8-GPU scaling", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "envs_per_gpu": B, "total_envs": total_envs, "parallelism": f"env-shard x{world}", "errno": errno,
                       "window": f"steps {W}..{W + K} after the reset" + (" (landing / contact-onset transient)" if W + K < 100 else ""),
                       "step_graph": bool(using_graph), "graph_fallbacks": graph_fallbacks,
                       "loop": ("closed rollout loop: ActorCritic.act + env step + RolloutStorage (information only)" if args.rollout else
                                "env step, open-loop actions" + ("; rewards / dones land in the rollout storage; every 24 steps: GAE + RCCL all-gather of the advantage moments, all-reduce of the curriculum counters and broadcast of rank 0's global DR scalars (one curriculum level, one set of global draws over all shards)"
                                                                 if world > 1 else ""))},
            "roofline": roofline, "roofline_issue": roofline_issue, "cpu_baseline": cpu, "steady_state": steady, "rollouts": rollouts, "action_sets": action_sets, "curriculum_live": curriculum_live,
            "workloads": workloads, "ref_protocol_fps": refp, "ref_protocol_fps_anymal_c": refp_anymal, "go2env_class": envcls, "ref_logged": ref_logged(),
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
