"""RolloutStorage -- the part of ``rsl_rl.storage.RolloutStorage`` (rsl-rl-lib==2.2.4) that sits between the env step and the PPO update:
``add_transitions`` (with PPO.process_env_step's time-out bootstrap) and ``compute_returns`` (GAE(lambda) + advantage normalisation), on the
device through the C ABI (include/go2sim_policy.h).  In a multi-GPU job the normalisation statistics are global: the local moments
[sum, sum of squares, count] are all-gathered over RCCL (one collective of 3 float64 per rank and rollout).

    storage = RolloutStorage(24, env.num_envs)
    for t in range(24):
        actions = policy.act(obs, critic_obs)
        obs, rew, dones, extras = env.step(actions)
        storage.add_transitions(t, rew, dones, policy.values, extras["time_outs"], gamma=0.99)
    storage.compute_returns(policy.evaluate(critic_obs), gamma=0.99, lam=0.95)
    storage.returns, storage.advantages          # [24, num_envs] device tensors (views of library memory)
"""
import ctypes

import torch

from .capi import C, Go2SimError, load_hip_lib
from .distributed import allgather_moments


def _p(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr() if isinstance(t, torch.Tensor) else t.ctypes.data)


class RolloutBuffers:
    """One go2sim_rollout handle (either library); pointers only, no torch arithmetic."""

    def __init__(self, lib, n_steps, n_envs, device=0):
        self.L, self.T, self.B = lib, int(n_steps), int(n_envs)
        h = ctypes.c_void_p()
        lib.check(lib.fn("rollout_create")(ctypes.c_int(device), ctypes.c_int(n_steps), ctypes.c_int(n_envs), ctypes.byref(h)), "rollout_create")
        self.h = h

    def add(self, t, rewards, dones, values, time_outs, gamma, stream=0):
        self.L.check(self.L.fn("rollout_add")(self.h, ctypes.c_int(t), _p(rewards), _p(dones), _p(values), _p(time_outs), ctypes.c_float(gamma),
                                              ctypes.c_void_p(stream)), "rollout_add")

    def compute_returns(self, last_values, gamma, lam, moments3, stream=0):
        self.L.check(self.L.fn("rollout_compute_returns")(self.h, _p(last_values), ctypes.c_float(gamma), ctypes.c_float(lam), _p(moments3),
                                                          ctypes.c_void_p(stream)), "rollout_compute_returns")

    def normalize(self, moments3, stream=0):
        self.L.check(self.L.fn("rollout_normalize")(self.h, _p(moments3), ctypes.c_void_p(stream)), "rollout_normalize")

    def ptr(self, which):
        p = ctypes.c_void_p()
        self.L.check(self.L.fn("rollout_ptr")(self.h, ctypes.c_int(C["GO2SIM_RB_" + which]), ctypes.byref(p)), "rollout_ptr")
        return p.value

    def close(self):
        if getattr(self, "h", None):
            self.L.fn("rollout_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RolloutStorage:
    def __init__(self, num_transitions_per_env, num_envs, device=None):
        if not torch.cuda.is_available():
            raise Go2SimError("no ROCm GPU visible: the go2sim product path has no CPU fallback")
        from .go2_env import _as_device_tensor

        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_transitions_per_env, self.num_envs = num_transitions_per_env, num_envs
        self._b = RolloutBuffers(load_hip_lib(), num_transitions_per_env, num_envs, self.device.index or 0)
        shape = (num_transitions_per_env, num_envs)
        view = lambda name, dt: _as_device_tensor(self._b.ptr(name), shape, dt, self.device)
        self.rewards, self.values = view("REWARDS", torch.float32), view("VALUES", torch.float32)
        self.returns, self.advantages = view("RETURNS", torch.float32), view("ADVANTAGES", torch.float32)
        self.dones = view("DONES", torch.uint8)
        self._moments = torch.zeros(3, dtype=torch.float64, device=self.device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def add_transitions(self, t, rewards, dones, values, time_outs=None, gamma=0.99):
        f = lambda x, dt: None if x is None else x.reshape(self.num_envs).to(device=self.device, dtype=dt).contiguous()
        self._b.add(t, f(rewards, torch.float32), f(dones, torch.uint8), f(values, torch.float32), f(time_outs, torch.float32), gamma, self._stream())

    def compute_returns(self, last_values, gamma, lam, group=None):
        lv = last_values.reshape(self.num_envs).to(device=self.device, dtype=torch.float32).contiguous()
        self._b.compute_returns(lv, gamma, lam, self._moments, self._stream())
        g = allgather_moments(self._moments, group)           # RCCL all-gather of 3 float64 per rank; identity for a single process
        self._b.normalize(g, self._stream())
        return self.returns, self.advantages
