"""Golden fixtures from the reference's OWN Go2Env files (build container only; nothing of /root/reference travels).

    python tools/make_ref_env_fixtures.py            # writes tests/golden/ref_env_<case>.npz

The reference's env classes (examples/locomotion/final/go2_env_{base,walk,stair}.py) import only torch, math, numpy, `genesis` and
`genesis.utils.geom`.  The repo's `genesis/` alias package resolves those two imports to go2_sim2real_locomotion_rl_amd.genesis_shim, which
this script points at the CPU oracle's C ABI (the test-only switch tests/util.gs_on_oracle).  The files are loaded with
importlib.util.spec_from_file_location *where they lie*, so every line of Go2Env.__init__ / step / reset_idx / reward / observation code that
runs is the reference's; the rigid-body physics underneath is the oracle's (Genesis itself is not importable: quadrants / mujoco / trimesh are
absent).  What these fixtures pin is therefore the Go2Env layer of the fused restatement
(go2sim_cpu_env_* / go2sim_env_*) against the reference's code; the physics rows stay "parity unpinned".

RNG: the reference draws from torch's global generator, the C ABI from a counter-based Philox stream.  Each case therefore uses configurations
whose random ranges are degenerate ([v, v]) -- the draw is v whatever the generator returns -- with *different* easy / hard values, so that the
curriculum interpolation (`_lerp_range(easy, hard, level)`) is still exercised.  Episodes are made short (0.6-1.0 s) and the action tapes harsh so
that every case passes through time-out resets and fall resets; the stairs case locks the terrain rows on both sides
(`_lock_terrain_rows`, go2_env_stair.py:399,1513) because the reference assigns them with torch.randperm.

The `*_rng` cases pin the lines that DO draw numbers (gs_rand_float / gs_rand_int with lower != upper, sample_level's mix branch, _apply_push,
observation / action noise, the per-env delay, init height / tilt, the single-axis command branch, _assign_terrain_rows): the loaded module's
global `torch` is replaced by `ScheduledTorch`, which forwards everything except rand / randn_like / randint / randperm; those return the
four-entry constant schedule of include/go2sim_detmath.h (GO2SIM_RNG_CONST), keyed like the C ABI's Philox stream (the env step for per-step
draws, the reset-call number inside reset_idx).  The counterpart on the C-ABI side is the diagnostic -DGO2SIM_RNG_CONST build of the oracle and
of the HIP library (build.ORACLE_VARIANTS / HIP_VARIANTS); the product build is untouched.  These cases keep the reference's shipped ranges.
"""
import copy
import importlib.util
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_DIR = "/root/reference/examples/locomotion/final"
OUT_DIR = os.path.join(ROOT, "tests", "golden")


def pinned_cfgs(case):
    """The reference's configuration of the task with every random range collapsed to one value (see the module docstring)."""
    from go2_sim2real_locomotion_rl_amd.configs import get_crouch_cfgs, get_jump_cfgs, get_stair_cfgs, get_walk_cfgs

    if case in ("base_jump", "base_crouch"):
        env_cfg, obs_cfg, reward_cfg, command_cfg = copy.deepcopy(get_jump_cfgs() if case == "base_jump" else get_crouch_cfgs())
        env_cfg["episode_length_s"] = 0.8 if case == "base_jump" else 0.6          # time-out resets inside the tape
        if case == "base_crouch":                                                   # loosen the crouch task's 0.05 m/s y-velocity cut so that episodes last
            env_cfg["termination_if_y_vel_greater_than"] = 0.5
            env_cfg["termination_if_z_vel_greater_than"] = 1.5
        return env_cfg, obs_cfg, reward_cfg, command_cfg
    env_cfg, obs_cfg, reward_cfg, command_cfg = copy.deepcopy(get_stair_cfgs() if case == "stairs" else get_walk_cfgs())
    cur = env_cfg["curriculum"]
    one = lambda v: [v, v]
    env_cfg.update({"friction_range": one(0.9), "kp_factor_range": one(1.1), "kd_factor_range": one(0.9), "kp_range": one(65.0), "kd_range": one(3.0),
                    "mass_shift_range": one(1.5), "com_shift_range": one(0.02), "leg_mass_shift_range": one(0.3), "gravity_offset_range": one(0.5),
                    "motor_strength_range": one(1.05), "obs_noise_level": 0.0, "action_noise_std": 0.0, "push_force_range": one(0.0),
                    "init_pos_z_range": one(0.40), "init_euler_range": one(3.0), "min_delay_steps": 0, "max_delay_steps": 1,
                    "episode_length_s": 0.9, "termination_if_roll_greater_than": 25, "termination_if_pitch_greater_than": 25})   # falls inside 45 steps
    cur.update({"friction_easy": one(0.7), "kp_factor_easy": one(0.97), "kd_factor_easy": one(1.03), "kp_easy": one(55.0), "kd_easy": one(1.5),
                "mass_shift_easy": one(0.2), "com_shift_easy": one(-0.004), "leg_mass_shift_easy": one(-0.05), "gravity_offset_easy": one(-0.1),
                "motor_strength_easy": one(0.98), "delay_easy_max_steps": 0, "mix_prob_current": 1.0, "global_dr_update_interval": 4,
                "update_every_episodes": 6})
    if case == "walk_delay1":
        env_cfg["min_delay_steps"] = 1
        cur["delay_easy_max_steps"] = 1
    if case == "walk_delay2":                                                       # a 3-deep action ring (the env's own default max_delay_steps, :374)
        env_cfg["min_delay_steps"] = env_cfg["max_delay_steps"] = 2
        cur["delay_easy_max_steps"] = 2
    command_cfg.update({"lin_vel_x_range": one(0.6), "lin_vel_y_range": one(-0.2), "ang_vel_range": one(0.4), "rel_standing_envs": 0.0})
    if case == "stairs":
        command_cfg.update({"lin_vel_x_range": one(0.5), "lin_vel_y_range": one(0.0), "ang_vel_range": one(0.0)})
    return env_cfg, obs_cfg, reward_cfg, command_cfg


RNG_U = (0.25, 0.75, 0.0625, 0.5)      # dm_rng_const_u / dm_rng_const_z, include/go2sim_detmath.h
RNG_Z = (0.5, -1.0, 0.25, -0.5)


class ScheduledTorch:
    """Stands in for the global `torch` of a loaded reference env module: rand / randn_like / randint / randperm return the constant schedule,
    every other attribute is torch's."""

    def __init__(self):
        self.mode, self.step_key, self.reset_key, self.calls = "step", 0, 0, {"rand": 0, "randn_like": 0, "randint": 0, "randperm": 0}

    def __getattr__(self, name):
        return getattr(torch, name)

    def _key(self):
        return (self.reset_key if self.mode == "reset" else self.step_key) & 3

    def rand(self, *size, **kw):
        self.calls["rand"] += 1
        size = kw.get("size", size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size)
        return torch.full(tuple(size), RNG_U[self._key()], dtype=torch.float32)

    def randn_like(self, t):
        self.calls["randn_like"] += 1
        return torch.full_like(t, RNG_Z[self._key()])

    def randint(self, low, high, size, device=None):
        self.calls["randint"] += 1
        return torch.full(tuple(size), low + int(RNG_U[self._key()] * (high - low)), dtype=torch.int64)

    def randperm(self, n, device=None):
        self.calls["randperm"] += 1
        return torch.arange(n)


def rng_cfgs(case):
    """The reference's configuration of the task with its SHIPPED random ranges; only time scales are shortened so that a 96-step tape passes through
    every kind of draw (resets, command resampling, pushes, curriculum updates, friction re-draws)."""
    from go2_sim2real_locomotion_rl_amd.configs import get_crouch_cfgs, get_jump_cfgs, get_stair_cfgs, get_walk_cfgs

    if case in ("base_jump_rng", "base_crouch_rng"):
        env_cfg, obs_cfg, reward_cfg, command_cfg = pinned_cfgs(case[:-4])
        command_cfg.update({"lin_vel_x_range": [-0.4, 0.8], "lin_vel_y_range": [-0.3, 0.1], "ang_vel_range": [-0.5, 0.25]})   # the shipped ranges are [0, 0]
        env_cfg["resampling_time_s"] = 0.3
        return env_cfg, obs_cfg, reward_cfg, command_cfg
    env_cfg, obs_cfg, reward_cfg, command_cfg = copy.deepcopy(get_stair_cfgs() if case.startswith("stairs") else get_walk_cfgs())
    cur = env_cfg["curriculum"]
    env_cfg.update({"episode_length_s": 0.9, "resampling_time_s": 0.3, "termination_if_roll_greater_than": 25, "termination_if_pitch_greater_than": 25,
                    "push_interval_s": 0.5, "min_delay_steps": 0, "max_delay_steps": 2})
    cur.update({"update_every_episodes": 6, "global_dr_update_interval": 4, "push_interval_easy_s": 0.6, "mix_prob_current": 0.6, "level_init": 0.30,
                "delay_easy_max_steps": 1})
    if case == "walk_rng_axis":                                                 # `choice = torch.randint(0, 3, (n,))`, go2_env_walk.py:948-958
        command_cfg["compound_commands"] = False
    if case == "stairs_rng":
        cur["level_init"] = 0.65                                                # as shipped: max_row = int(0.65 * 12) = 7, DR level 0.405 >= push_start
    return env_cfg, obs_cfg, reward_cfg, command_cfg


RNG_CASES = ["walk_rng", "walk_rng_axis", "stairs_rng", "base_jump_rng"]


def action_tape(T, B, n_act, seed):
    """Per-env action styles: standing, gentle / rough noise, a constant tipping bias, an open-loop trot."""
    rng = np.random.default_rng(seed)
    a = np.zeros((T, B, n_act), np.float32)
    t = np.arange(T, dtype=np.float32)[:, None]
    for b in range(B):
        style = b % 8
        noise = rng.standard_normal((T, n_act)).astype(np.float32)
        if style == 0:
            a[:, b] = 0.0
        elif style in (1, 2, 3):
            a[:, b] = (0.3, 0.5, 1.0)[style - 1] * noise
        elif style == 4:                                                    # left legs extend, right legs fold: the robot rolls over
            bias = np.zeros(n_act, np.float32)
            bias[[1, 7]] = 2.5; bias[[4, 10]] = -2.5; bias[[2, 8]] = 2.0; bias[[5, 11]] = -2.0
            a[:, b] = bias + 0.1 * noise
        elif style == 5:
            a[:, b] = 2.0 * noise
        elif style == 6:
            phase = np.array([0.0, 0.0, 0.0, np.pi, np.pi, np.pi, np.pi, np.pi, np.pi, 0.0, 0.0, 0.0], np.float32)
            a[:, b, :12] = 0.6 * np.sin(2.0 * np.pi * 1.5 * 0.02 * t + phase[None, :])
        else:
            a[:, b] = 0.5 * noise
            a[T // 2:, b] = 3.0 * noise[T // 2:]
    if n_act > 12:
        # Per-leg stiffness actions on a 0.5 grid in [-2, 2]: kp = 40 + 20 a is then one of 10, 20, ..., 70 (both clamps are reached), values for
        # which torch's CPU sqrt (MKL VML here, 1 ulp off for ~0.7 % of arbitrary inputs -- a host-library artefact, the reference's GPU sqrt is
        # correctly rounded) returns the correctly rounded root.  kd = 0.2 sqrt(kp) feeds the torque, so this keeps the tape free of that artefact.
        a[:, :, 12:] = np.clip(np.round(a[:, :, 12:] * 2.0) / 2.0, -2.0, 2.0)
    return a


def load_reference_env_module(stem):
    spec = importlib.util.spec_from_file_location("ref_" + stem, os.path.join(REF_DIR, stem + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fixture_path(case, physics):
    return os.path.join(OUT_DIR, f"ref_env_{case}.npz" if physics == "strict" else f"ref_env_{case}_{physics}.npz")


def run_case(case, B=8, T=96, seed=11, n_run=None, physics="strict"):
    """Runs the reference env of `case` over the first `n_run` (default: all) steps of the T-step action tape."""
    import go2_sim2real_locomotion_rl_amd.genesis_shim as gs
    from go2_sim2real_locomotion_rl_amd.capi import load_cpu_oracle_lib
    from util import gs_on_oracle

    # physics underneath the reference's env code: the strict oracle (reference CPU summation order) or its FAST ORDER build (the arithmetic of the
    # HIP product); the env layer that is being pinned is the same reference code either way
    rng_case = case.endswith("_rng") or "_rng_" in case
    if rng_case:
        from go2_sim2real_locomotion_rl_amd import build
        from go2_sim2real_locomotion_rl_amd.capi import Go2SimLib

        name = "rng_const_fast" if physics == "fast" else "rng_const"
        lib = Go2SimLib(build.build_oracle_variant(name, build.ORACLE_VARIANTS[name], verbose=False), "go2sim_cpu_")
    else:
        lib = load_cpu_oracle_lib(fast=(physics == "fast"))
    gs_on_oracle(gs, lib, seed=seed)
    stem = "go2_env_base" if case.startswith("base") else "go2_env_stair" if case.startswith("stairs") else "go2_env_walk"
    cfgs = rng_cfgs(case) if rng_case else pinned_cfgs(case)
    cfg_json = json.dumps(cfgs)                                           # before the env multiplies the reward scales by dt in place
    mod = load_reference_env_module(stem)
    sched = None
    if rng_case:
        sched = mod.torch = ScheduledTorch()
    torch.manual_seed(seed)
    log = io.StringIO()
    with redirect_stdout(log):
        env = mod.Go2Env(B, *copy.deepcopy(cfgs))
    if rng_case:
        inner = env.reset_idx

        def reset_idx(envs_idx):                                          # the reset-call number keys the draws of one reset_idx call (n > 0)
            if len(envs_idx) == 0:
                return inner(envs_idx)
            sched.mode = "reset"
            try:
                return inner(envs_idx)
            finally:
                sched.mode = "step"
                sched.reset_key += 1

        env.reset_idx = reset_idx
    rows = None
    if case == "stairs":
        rows = np.array([(3 * b + 1) % 13 for b in range(B)], np.int64)
        env._lock_terrain_rows = True
        env._env_terrain_row[:] = torch.from_numpy(rows)
    n_act = env.num_actions
    names = list(env.reward_functions.keys())
    terms = {}

    def wrap(name, fn):
        def call():
            out = fn()
            terms[name] = (out * env.reward_scales[name]).clone()
            return out
        return call

    for name in names:
        env.reward_functions[name] = wrap(name, env.reward_functions[name])
    with redirect_stdout(log):
        env.reset()
    acts = action_tape(T, B, n_act, seed)
    motors = torch.as_tensor(env.motors_dof_idx)
    rec = {k: [] for k in ("obs", "priv", "rew", "rew_terms", "done", "time_outs", "ctrl_pos", "ctrl_force", "base_pos", "commands", "episode_length", "level",
                           "delay_steps", "push_force", "terrain_row")}
    has_priv = getattr(env, "num_privileged_obs", None) is not None
    n_run = T if n_run is None else n_run
    for s in range(n_run):
        if sched is not None:
            sched.step_key = s
        with redirect_stdout(log):
            obs, rew, done, extras = env.step(torch.from_numpy(acts[s]))
        rec["obs"].append(obs.numpy().copy()); rec["rew"].append(rew.numpy().copy()); rec["done"].append(done.numpy().astype(np.uint8))
        rec["time_outs"].append(extras["time_outs"].numpy().copy())
        rec["priv"].append(env.privileged_obs_buf.numpy().copy() if has_priv else np.zeros((B, 0), np.float32))
        rec["rew_terms"].append(np.stack([terms[n].numpy() for n in names], axis=1))
        rec["ctrl_pos"].append(env.robot._get("F_CTRL_POS").t()[:, motors].numpy().copy())
        rec["ctrl_force"].append(env.robot._get("F_CTRL_FORCE").t()[:, motors].numpy().copy())
        rec["base_pos"].append(env.base_pos.numpy().copy()); rec["commands"].append(env.commands.numpy().copy())
        rec["episode_length"].append(env.episode_length_buf.numpy().astype(np.int32).copy())
        rec["level"].append(float(env.curriculum.level) if hasattr(env, "curriculum") else 0.0)
        rec["delay_steps"].append(env._delay_steps.numpy().astype(np.int32).copy() if hasattr(env, "_delay_steps") else np.zeros(B, np.int32))
        rec["push_force"].append(env._current_push_force.numpy().copy() if hasattr(env, "_current_push_force") else np.zeros((B, 3), np.float32))
        rec["terrain_row"].append(env._env_terrain_row.numpy().astype(np.int32).copy() if hasattr(env, "_env_terrain_row") else np.zeros(B, np.int32))
    out = {k: np.stack(v) if k != "level" else np.asarray(v, np.float64) for k, v in rec.items()}
    out["actions"] = acts[:n_run]
    done, to = out["done"].astype(bool), out["time_outs"] > 0
    meta = {"case": case, "reference_file": f"examples/locomotion/final/{stem}.py", "physics": physics, "n_envs": B, "steps": n_run, "seed": seed, "reward_names": names,
            "n_time_out_resets": int((done & to).sum()), "n_fall_resets": int((done & ~to).sum()),
            "terrain_rows": None if rows is None else rows.tolist(), "rng_calls": None if sched is None else dict(sched.calls),
            "reset_calls": None if sched is None else sched.reset_key}
    out["cfgs_json"] = np.array(cfg_json)
    out["meta_json"] = np.array(json.dumps(meta))
    return out, meta


def main():
    if not os.path.isdir(REF_DIR):
        raise SystemExit("the reference tree is not present: fixtures can only be regenerated in the build container")
    from go2_sim2real_locomotion_rl_amd import build

    build.build_oracle()
    os.makedirs(OUT_DIR, exist_ok=True)
    for case in sys.argv[1:] or ["base_jump", "base_crouch", "walk", "walk_delay1", "walk_delay2", "stairs"] + RNG_CASES:
        for physics in ("strict", "fast"):
            out, meta = run_case(case, B=16 if case in RNG_CASES else 8, physics=physics)
            path = fixture_path(case, physics)
            np.savez_compressed(path, **out)
            print(f"{path}: {meta['n_time_out_resets']} time-out resets, {meta['n_fall_resets']} fall resets, {os.path.getsize(path) // 1024} KiB"
                  + (f", generator calls {meta['rng_calls']}, {meta['reset_calls']} reset calls" if meta.get("rng_calls") else ""))


if __name__ == "__main__":
    main()
