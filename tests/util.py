"""Shared helpers for the parity tests."""
import numpy as np

from go2_sim2real_locomotion_rl_amd.capi import C, Go2Sim
from go2_sim2real_locomotion_rl_amd.configs import (build_stair_terrain, flatten_base_cfg, flatten_walk_cfg, get_crouch_cfgs, get_jump_cfgs,
                                                    get_stair_cfgs, get_walk_cfgs, with_per_env_dr)

NOBS, NPRIV, NACT = 49, 104, 16


def F(name):
    return C["GO2SIM_" + name]


def walk_cfg(n_envs, mutate=None, **kw):
    cfgs = get_walk_cfgs()
    if mutate is not None:
        mutate(*cfgs)
    return flatten_walk_cfg(n_envs, *cfgs, **kw)


def task_cfg(task, n_envs, mutate=None, **kw):
    """(fcfg, icfg, reward_names, n_obs, n_priv, n_act) of the walk / crouch / jump tasks."""
    if task == "walk":
        return walk_cfg(n_envs, mutate, **kw) + (NOBS, NPRIV, NACT)
    if task == "stairs":
        cfgs = get_stair_cfgs()
        if mutate is not None:
            mutate(*cfgs)
        return flatten_walk_cfg(n_envs, *cfgs, **kw) + (NOBS, 182, NACT)
    cfgs = get_crouch_cfgs() if task == "crouch" else get_jump_cfgs()
    if task.endswith("_dr"):                       # BASELINE.json configs[4]: base env + per-env friction / base-mass randomisation
        cfgs = with_per_env_dr(get_crouch_cfgs() if task.startswith("crouch") else get_jump_cfgs())
    if mutate is not None:
        mutate(*cfgs)
    return flatten_base_cfg(n_envs, *cfgs) + (45, 45, 12)


def install_stairs(sim):
    """gs.morphs.Terrain of go2_env_stair.py:424-433."""
    hf, info = build_stair_terrain(get_stair_cfgs()[0]["terrain"])
    sim.set_terrain(hf, info["horizontal_scale"], info["vertical_scale"], info["terrain_origin"])
    return hf, info


def make_actions(steps, n_envs, seed=0, kind="mixed", n_act=NACT):
    rng = np.random.default_rng(seed)
    a = np.zeros((steps, n_envs, n_act), np.float32)
    for s in range(steps):
        if kind == "zeros":
            scale = 0.0
        elif kind == "mixed":
            scale = 0.0 if s < steps // 3 else (0.5 if s < 2 * steps // 3 else 2.0)
        else:
            scale = float(kind)
        a[s] = (scale * rng.standard_normal((n_envs, n_act))).astype(np.float32)
    return a


class CpuEnv:
    """Go2Env on the CPU oracle with numpy buffers."""

    def __init__(self, lib, blob, n_envs, seed=1, task="walk", **cfg_kw):
        self.sim = Go2Sim(lib, blob, n_envs, 0, seed)
        f, i, self.reward_names, nobs, npriv, self.n_act = task_cfg(task, n_envs, **cfg_kw)
        self.fcfg, self.icfg = f, i
        if task == "stairs":
            install_stairs(self.sim)
        self.sim.env_configure(f, i)
        self.B = n_envs
        self.obs = np.zeros((n_envs, nobs), np.float32); self.priv = np.zeros((n_envs, npriv), np.float32)
        self.rew = np.zeros(n_envs, np.float32); self.rst = np.zeros(n_envs, np.uint8); self.to = np.zeros(n_envs, np.float32)

    def reset(self):
        self.sim.env_reset()

    def step(self, act):
        self.sim.env_step(np.ascontiguousarray(act, np.float32), self.obs, self.priv, self.rew, self.rst, self.to)
        return self.obs, self.priv, self.rew, self.rst, self.to

    def field(self, name):
        return self.sim.get_field_np(F(name))

    def env_buf(self, name, k, dtype=np.float32):
        out = np.zeros((self.B, k), dtype)
        self.sim.env_get(C["GO2SIM_EB_" + name], out)
        return out


class GpuEnv:
    """Go2Env on the HIP library with torch (ROCm) buffers; everything goes through the C ABI."""

    def __init__(self, lib, blob, n_envs, seed=1, task="walk", **cfg_kw):
        import torch

        self.torch = torch
        self.dev = torch.device("cuda:0")
        self.sim = Go2Sim(lib, blob, n_envs, 0, seed)
        f, i, self.reward_names, nobs, npriv, self.n_act = task_cfg(task, n_envs, **cfg_kw)
        if task == "stairs":
            install_stairs(self.sim)
        self.sim.env_configure(f, i)
        self.B = n_envs
        self.obs = torch.zeros(n_envs, nobs, device=self.dev); self.priv = torch.zeros(n_envs, npriv, device=self.dev)
        self.rew = torch.zeros(n_envs, device=self.dev); self.rst = torch.zeros(n_envs, dtype=torch.uint8, device=self.dev)
        self.to = torch.zeros(n_envs, device=self.dev)

    def reset(self):
        self.sim.env_reset()

    def step(self, act):
        a = self.torch.from_numpy(np.ascontiguousarray(act, np.float32)).to(self.dev)
        self.sim.env_step(a, self.obs, self.priv, self.rew, self.rst, self.to)
        self.torch.cuda.synchronize()
        return self.obs.cpu().numpy(), self.priv.cpu().numpy(), self.rew.cpu().numpy(), self.rst.cpu().numpy(), self.to.cpu().numpy()

    def field(self, name):
        torch = self.torch
        k, is_int = self.sim.field_size(F(name))
        t = torch.zeros(k, self.B, dtype=torch.int32 if is_int else torch.float32, device=self.dev)
        self.sim.get_field(F(name), t)
        torch.cuda.synchronize()
        return t.cpu().numpy()

    def set_field(self, name, arr):
        torch = self.torch
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)
        self.sim.set_field(F(name), t)
        torch.cuda.synchronize()

    def env_buf(self, name, k, dtype=np.float32):
        torch = self.torch
        t = torch.zeros(self.B, k, dtype=torch.int32 if dtype == np.int32 else torch.float32, device=self.dev)
        self.sim.env_get(C["GO2SIM_EB_" + name], t)
        torch.cuda.synchronize()
        return t.cpu().numpy()


def gs_on_oracle(gs, oracle_lib, seed=1):
    """TEST-ONLY: point the gs shim at the CPU twin of the C ABI (host tensors) so that scripts written against the Genesis surface can be
    checked here without a GPU.  The product module has no such switch (gs.init always loads the HIP library); this helper rebinds the
    module globals from outside."""
    import torch

    gs._lib, gs.device, gs._seed = oracle_lib, torch.device("cpu"), int(seed)


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.int32), b.view(np.int32))
    return np.array_equal(a, b)
