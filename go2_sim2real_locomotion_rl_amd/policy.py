"""ActorCritic -- host-side mirror of ``rsl_rl.modules.ActorCritic`` (rsl-rl-lib==2.2.4, the version
``examples/locomotion/final/go2_train_walk.py:12-15`` pins) on top of the policy entry points of the C ABI
(include/go2sim_policy.h).  It covers the inference side that ``OnPolicyRunner`` / ``PPO.act`` call once per
environment step -- ``act``, ``evaluate``, ``get_actions_log_prob``, ``act_inference``, ``action_mean``,
``action_std`` -- with the reference's argument meaning; the PPO update itself (autograd) stays with the caller,
who pushes new parameters with :meth:`load_state_dict`.

    policy = ActorCritic(49, 104, 16, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu")
    policy.load_state_dict(torch.load("model_1000.pt", weights_only=True)["model_state_dict"])
    actions = policy.act(obs, critic_obs)        # one fused MLP kernel per network + one sampling kernel
    values, log_prob = policy.values, policy.actions_log_prob

State-dict keys are those of the reference class: ``actor.{0,2,4,6}.{weight,bias}``, ``critic.{0,2,4,6}.{weight,bias}``, ``std``.
Random numbers come from the library's counter-based Philox stream (seed, row, step), not from torch's global generator.
"""
import ctypes

import numpy as np
import torch

from .capi import Go2SimError, load_hip_lib


class Mlp:
    """One go2sim_mlp handle (works with either library: product or CPU oracle)."""

    def __init__(self, lib, dims, params, device=0):
        self.L, self.dims = lib, [int(d) for d in dims]
        params = np.ascontiguousarray(params, dtype=np.float32)
        self.n_params = params.size
        h = ctypes.c_void_p()
        d = (ctypes.c_int * len(self.dims))(*self.dims)
        rc = lib.fn("mlp_create")(ctypes.c_int(device), d, ctypes.c_int(len(self.dims) - 1), params.ctypes.data_as(ctypes.c_void_p),
                                  ctypes.c_size_t(params.size), ctypes.byref(h))
        lib.check(rc, "mlp_create")
        self.h = h

    def set_params(self, params, stream=0):
        params = np.ascontiguousarray(params, dtype=np.float32)
        self.L.check(self.L.fn("mlp_set_params")(self.h, params.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(params.size), ctypes.c_void_p(stream)),
                     "mlp_set_params")

    def forward(self, x, y, n_rows, stream=0):
        self.L.check(self.L.fn("mlp_forward")(self.h, _ptr(x), _ptr(y), ctypes.c_int(n_rows), ctypes.c_void_p(stream)), "mlp_forward")

    def close(self):
        if getattr(self, "h", None):
            self.L.fn("mlp_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(a):
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    return a.ctypes.data_as(ctypes.c_void_p)


def flatten_sequential(state_dict, prefix, n_layers):
    """[W0, b0, W1, b1, ...] of ``prefix.{0,2,4,...}`` as one float32 vector (the layout go2sim_mlp_create takes) + the layer sizes."""
    chunks, dims = [], []
    for l in range(n_layers):
        w = np.asarray(state_dict[f"{prefix}.{2 * l}.weight"].detach().cpu().numpy() if isinstance(state_dict[f"{prefix}.{2 * l}.weight"], torch.Tensor)
                       else state_dict[f"{prefix}.{2 * l}.weight"], dtype=np.float32)
        b = state_dict[f"{prefix}.{2 * l}.bias"]
        b = np.asarray(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b, dtype=np.float32)
        if l == 0:
            dims.append(w.shape[1])
        dims.append(w.shape[0])
        chunks += [w.reshape(-1), b.reshape(-1)]
    return np.concatenate(chunks), dims


def policy_act(lib, actor, critic, obs, critic_obs, std, n_rows, seed, step, deterministic, actions, mean, values, log_prob, stream=0):
    rc = lib.fn("policy_act")(actor.h, critic.h if critic is not None else ctypes.c_void_p(0), _ptr(obs), _ptr(critic_obs), _ptr(std),
                              ctypes.c_int(n_rows), ctypes.c_uint64(seed), ctypes.c_uint32(step), ctypes.c_int(int(deterministic)),
                              _ptr(actions), _ptr(mean), _ptr(values), _ptr(log_prob), ctypes.c_void_p(stream))
    lib.check(rc, "policy_act")


class ActorCritic:
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(512, 256, 128), critic_hidden_dims=(512, 256, 128),
                 activation="elu", init_noise_std=1.0, *, device=None, seed=1, **kwargs):
        if activation != "elu":
            raise Go2SimError("go2sim implements the reference's activation ('elu', go2_train_walk.py:42)")
        if not torch.cuda.is_available():
            raise Go2SimError("no ROCm GPU visible: the go2sim product path has no CPU fallback")
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_actions = num_actions
        self._L = load_hip_lib()
        self._adims = [num_actor_obs, *actor_hidden_dims, num_actions]
        self._cdims = [num_critic_obs, *critic_hidden_dims, 1]
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for prefix, dims in (("actor", self._adims), ("critic", self._cdims)):      # nn.Linear default init (kaiming_uniform(a=sqrt 5))
            for l in range(len(dims) - 1):
                bound = 1.0 / (dims[l] ** 0.5)
                sd[f"{prefix}.{2 * l}.weight"] = (torch.rand(dims[l + 1], dims[l], generator=g) * 2 - 1) * bound
                sd[f"{prefix}.{2 * l}.bias"] = (torch.rand(dims[l + 1], generator=g) * 2 - 1) * bound
        sd["std"] = init_noise_std * torch.ones(num_actions)
        self._actor = self._critic = None
        self.std = torch.ones(num_actions, device=self.device)
        self.load_state_dict(sd)
        self._seed, self._step = int(seed), 0
        self._bufs = {}

    # ---- parameters ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        pa, da = flatten_sequential(state_dict, "actor", len(self._adims) - 1)
        pc, dc = flatten_sequential(state_dict, "critic", len(self._cdims) - 1)
        if da != self._adims or dc != self._cdims:
            raise Go2SimError(f"state dict has layer sizes {da} / {dc}, expected {self._adims} / {self._cdims}")
        dev = self.device.index or 0
        if self._actor is None:
            self._actor, self._critic = Mlp(self._L, da, pa, dev), Mlp(self._L, dc, pc, dev)
        else:
            self._actor.set_params(pa); self._critic.set_params(pc)
        self.std = state_dict["std"].detach().to(device=self.device, dtype=torch.float32).contiguous().clone()
        self._state = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in state_dict.items()}
        return True

    def state_dict(self):
        return dict(self._state)

    def load_checkpoint(self, path, strict=False):
        """Load ``model_<it>.pt`` of a training run (rsl_rl 2.2.4 ``OnPolicyRunner.save`` layout), read with ``torch.load(weights_only=True)``.
        strict=False is the eval scripts' compatibility load (go2_eval_stairs.py:368-450): layers whose shapes differ from this model -- a critic
        trained with another privileged-observation width -- keep their current values, everything else (the actor in particular) is taken from the
        file.  -> (loaded keys, skipped {key: reason}, iteration)"""
        from .eval_io import compatible_state_dict, read_checkpoint

        ckpt = read_checkpoint(path)
        saved = ckpt["model_state_dict"]
        if strict:
            self.load_state_dict(saved)
            return sorted(saved), {}, int(ckpt.get("iter", 0) or 0)
        merged, loaded, skipped = compatible_state_dict(self.state_dict(), saved)
        self.load_state_dict(merged)
        return loaded, skipped, int(ckpt.get("iter", 0) or 0)

    # ---- rsl_rl ActorCritic surface (inference side) ---------------------------------------------------
    def _buf(self, name, shape):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(*shape, device=self.device, dtype=torch.float32)
            self._bufs[name] = t
        return t

    def _run(self, obs, critic_obs, deterministic):
        B = obs.shape[0]
        obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
        co = None if critic_obs is None else critic_obs.to(device=self.device, dtype=torch.float32).contiguous()
        self._actions, self._mean = self._buf("actions", (B, self.num_actions)), self._buf("mean", (B, self.num_actions))
        self._values = self._buf("values", (B,)) if co is not None else None
        self._logp = self._buf("logp", (B,))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        policy_act(self._L, self._actor, self._critic if co is not None else None, obs, co, self.std, B, self._seed, self._step, deterministic,
                   self._actions, self._mean, self._values, self._logp, stream)
        self._step += 1
        return self._actions

    def act(self, observations, critic_observations=None, **kwargs):
        """ActorCritic.act (+ evaluate when critic observations are given: PPO.act calls both on every step)."""
        return self._run(observations, critic_observations, False)

    def act_inference(self, observations):
        return self._run(observations, None, True)

    def evaluate(self, critic_observations, **kwargs):
        B = critic_observations.shape[0]
        co = critic_observations.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, 1, device=self.device)
        self._critic.forward(co, out, B, torch.cuda.current_stream(self.device).cuda_stream)
        return out

    def get_actions_log_prob(self, actions):
        """Normal(mean, std).log_prob(actions).sum(-1) for the distribution of the last act()."""
        if actions.data_ptr() == self._actions.data_ptr():
            return self._logp
        var = self.std * self.std
        return (-((actions - self._mean) ** 2) / (2 * var) - torch.log(self.std) - 0.9189385332046727).sum(-1)

    @property
    def action_mean(self):
        return self._mean

    @property
    def action_std(self):
        return self.std.expand_as(self._mean)

    @property
    def values(self):
        return None if self._values is None else self._values.unsqueeze(-1)

    @property
    def actions_log_prob(self):
        return self._logp

    def reset(self, dones=None):
        pass
