"""Go2Env -- host-side mirror of the reference's walk environment class on top of the go2sim C ABI.

Same constructor, methods, return values and attribute set as
``examples/locomotion/final/go2_env_walk.py`` (class Go2Env :154; step :985-1109; get_observations :1145;
get_privileged_observations :1151; reset :1242) so that ``rsl_rl.OnPolicyRunner`` and the reference's
train / eval scripts can use it unchanged:

    env = Go2Env(num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg)
    obs, extras = env.get_observations()
    obs, rew, reset, extras = env.step(actions)         # all torch tensors on env.device

Differences that are deliberate (DESIGN.md "boundary"):
  * the whole step runs inside libgo2sim.so (HIP, gfx950); there is no per-call torch arithmetic and no
    host synchronisation: the curriculum state machine and the "global" domain randomisation live on the device;
  * ``extras["episode"]`` / ``extras["curriculum"]`` / ``extras["domain_randomization"]`` values are 0-dim device tensors
    (views of one per-step snapshot of the device globals) instead of python floats (rsl_rl accepts both);
  * the errno poll of ``scene.step`` (simulator.py:267, every 10 substeps = 5 env steps) is asynchronous: the reduction is
    enqueued after the step and examined at the following steps (at the latest when the next poll is due), so the loop does not wait for the device;
  * random numbers come from the counter-based Philox stream of the C ABI (include/go2sim.h), not from
    torch's global generator.
"""
import ctypes
import math

import torch

from .capi import C, EnvGlobals, Go2Sim, Go2SimError, load_hip_lib
from .configs import build_stair_terrain, flatten_base_cfg, flatten_walk_cfg
from .model_blob import pack_model

_DEVICE = None
_SEED = 1


def init(backend=None, precision="32", logging_level=None, performance_mode=True, seed=None, device_index=0, **_):
    """Counterpart of ``gs.init`` (genesis/__init__.py:60): selects the ROCm device and the RNG seed."""
    global _DEVICE, _SEED
    if str(precision) != "32":
        raise Go2SimError("go2sim computes in fp32 only (the reference path runs gs.init(precision='32'))")
    if not torch.cuda.is_available():
        raise Go2SimError("no ROCm GPU visible: the go2sim product path has no CPU fallback")
    torch.cuda.set_device(device_index)
    _DEVICE = torch.device("cuda", device_index)
    if seed is not None:
        _SEED = int(seed)
    return _DEVICE


def _as_device_tensor(ptr, shape, dtype, device):
    """Zero-copy torch view of library-owned device memory (no ownership transfer, include/go2sim.h)."""
    itemsize = torch.empty((), dtype=dtype).element_size()
    n = 1
    for s in shape:
        n *= s

    class _Mem:
        __cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype],
            "data": (int(ptr), False), "version": 2, "strides": None,
        }

    assert n * itemsize > 0
    return torch.as_tensor(_Mem(), device=device)


class Go2Env:
    def __init__(self, num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg, show_viewer=False, *, seed=None, device=None,
                 freeze_curriculum=False, log_extras=True, errno_poll_every=5, shared_globals=False):
        if show_viewer:
            raise Go2SimError("the viewer is outside the accelerated path (SURVEY.md section 8f)")
        self.device = device if device is not None else (_DEVICE if _DEVICE is not None else init())
        self.num_envs = num_envs
        self.num_obs = obs_cfg["num_obs"]
        self.num_privileged_obs = obs_cfg.get("num_privileged_obs", None)
        self.num_actions = env_cfg["num_actions"]
        self.num_pos_actions = env_cfg.get("num_pos_actions", 12)
        self.num_commands = command_cfg["num_commands"]
        self.simulate_action_latency = env_cfg.get("simulate_action_latency", True)
        self.dt = 0.02
        self.max_episode_length = math.ceil(env_cfg["episode_length_s"] / self.dt)
        self.env_cfg, self.obs_cfg, self.reward_cfg, self.command_cfg = env_cfg, obs_cfg, reward_cfg, command_cfg
        self.obs_scales = obs_cfg["obs_scales"]
        self.reward_scales = dict(reward_cfg["reward_scales"])

        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._sim = Go2Sim(load_hip_lib(), pack_model(), num_envs, dev_index, _SEED if seed is None else int(seed))
        # the two env families of the reference: walk (go2_env_walk.py: 16 actions, 49 / 104 obs) and base (go2_env_base.py: 12 / 45)
        self.is_base_env = self.num_obs == 45 and self.num_actions == 12
        if self.is_base_env:
            fcfg, icfg, self._reward_names = flatten_base_cfg(num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg)
            self.num_privileged_obs = None
        else:
            fcfg, icfg, self._reward_names = flatten_walk_cfg(num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg,
                                                              freeze_curriculum=freeze_curriculum, shared_globals=shared_globals)
            if self.num_obs != 49 or self.num_privileged_obs not in (None, 104, 182) or self.num_actions != 16:
                raise Go2SimError("go2sim implements the walk layout (16 actions, 49 / 104 obs; go2_train_walk.py:300-320), the stair layout "
                                  "(49 / 182 obs; go2_train_stair.py:282-300) and the base layout (12 actions, 45 obs; go2_train_crouch.py / "
                                  "go2_train_jump.py)")
            terrain_cfg = env_cfg.get("terrain", None)
            if terrain_cfg is not None and terrain_cfg.get("enabled", False):     # go2_env_stair.py:352-433: gs.morphs.Terrain instead of the plane
                hf, info = build_stair_terrain(terrain_cfg)
                self._terrain_info = info
                self._sim.set_terrain(hf, info["horizontal_scale"], info["vertical_scale"], info["terrain_origin"])
        self._sim.env_configure(fcfg, icfg)

        B, dev = num_envs, self.device
        self.obs_buf = torch.zeros(B, self.num_obs, device=dev)
        self.privileged_obs_buf = torch.zeros(B, 45 if self.is_base_env else (self.num_privileged_obs or 104), device=dev)
        self.rew_buf = torch.zeros(B, device=dev)
        self.reset_buf = torch.zeros(B, dtype=torch.uint8, device=dev)
        self._time_outs = torch.zeros(B, device=dev)
        self._actions = torch.zeros(B, self.num_actions, device=dev)
        self.extras = {"observations": {}}
        # live device view of the curriculum / DR globals and the last reset's episode log
        gptr = self._sim.env_globals_ptr()
        self._glob_f32 = _as_device_tensor(gptr, (ctypes.sizeof(EnvGlobals) // 4,), torch.float32, dev)
        self._ep_off = EnvGlobals.last_episode_rew.offset // 4
        self._glob_f64 = self._glob_f32.view(torch.float64)     # the python-float state of the reference is kept in double (include/go2sim.h)
        self._level_off = EnvGlobals.level.offset // 8
        self._goff = {name: getattr(EnvGlobals, name).offset // (8 if ct is ctypes.c_double else 4) for name, ct in EnvGlobals._fields_}
        self._episode_keys = ["rew_" + n for n in self._reward_names]
        self._use_terrain = hasattr(self, "_terrain_info")
        self._terrain_rows_locked = False
        self.log_extras = bool(log_extras)            # per-step snapshot of the episode / curriculum / DR logs (one small device copy)
        self.errno_poll_every = int(errno_poll_every)  # env steps between errno polls (simulator.py:267: every 10 substeps)
        self._steps_since_poll = 0
        self._views = {}
        self._shared_globals = bool(shared_globals) and not self.is_base_env
        if self._shared_globals:
            # one batch sharded over ranks: the first sync (before the constructor's reset) puts every shard at the single-process starting point -- rank 0's
            # t_sample at level_init and first global draws; without it the shard's first episodes would sample the easy end of every DR range
            self.sync_globals(initial=True)
        self.reset()

    # ---- reference API -------------------------------------------------------------------------
    def step(self, actions):
        """go2_env_walk.py:985-1109.  ``actions`` [num_envs, 16] float32 on ``self.device``.

        Like the reference (``self.obs_buf = torch.cat(...)``, go2_env_walk.py:1084 / go2_env_base.py:175), every step returns NEW
        observation tensors: rsl_rl's PPO keeps ``transition.observations = obs`` by reference across ``env.step`` and copies it afterwards,
        so an observation buffer overwritten in place would pair o_{t+1} with a_t in the rollout storage."""
        if actions.shape != (self.num_envs, self.num_actions):
            raise Go2SimError(f"actions must have shape {(self.num_envs, self.num_actions)}, got {tuple(actions.shape)}")
        a = actions
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = self._actions.copy_(actions)
        self._raise_on_errno(self._sim.errno_poll_result())               # result of the poll enqueued after an earlier step
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.obs_buf = torch.empty_like(self.obs_buf)                      # caching allocator: host-side only, no kernel
        self.privileged_obs_buf = torch.empty_like(self.privileged_obs_buf)
        self._sim.env_step(a, self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self._time_outs, stream)
        self._steps_since_poll += 1
        poll = self._steps_since_poll >= self.errno_poll_every             # RATE_CHECK_ERRNO = 10 substeps, simulator.py:45,267
        if poll:
            self._steps_since_poll = 0
            self._raise_on_errno(self._sim.errno_poll_wait())              # the previous poll is a whole cadence old: this practically never blocks,
            self._sim.errno_poll_begin(stream)                             # and an error surfaces at most 2 x errno_poll_every steps after it happened
        self.extras["time_outs"] = self._time_outs
        self.extras["observations"]["critic"] = self.privileged_obs_buf if self.num_privileged_obs else self.obs_buf
        if self.log_extras:
            self._refresh_extras(full=poll)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _refresh_extras(self, full=True):
        """extras["episode"] (go2_env_walk.py:1228-1235) every step, extras["curriculum"] (:674-686) and extras["domain_randomization"]
        (:756,813) every ``errno_poll_every`` steps, from ONE snapshot of the device globals: a single small device copy, no synchronisation;
        the values are 0-dim views of that snapshot (the reference stores python floats obtained with ``.item()``)."""
        snap = self._glob_f32.clone()
        n = len(self._reward_names)
        self.extras["episode"] = dict(zip(self._episode_keys, snap[self._ep_off:self._ep_off + n].unbind(0)))
        if self._use_terrain:
            rows = snap.view(torch.int32)
            self.extras["episode"]["terrain_mean_row"] = rows[self._goff["terrain_row_sum"]].float() / rows[self._goff["last_reset_count"]].clamp(min=1).float()
        if self.is_base_env or not full:
            return
        f, i, d = snap, snap.view(torch.int32), snap.view(torch.float64)
        g = self._goff
        self.extras["curriculum"] = {
            "level": d[g["level"]], "timeout_rate_ema": d[g["timeout_rate_ema"]], "tracking_ema": d[g["tracking_ema"]],
            "fall_rate_ema": d[g["fall_rate_ema"]], "ready_streak": i[g["ready_streak"]], "hard_streak": i[g["hard_streak"]],
            "cooldown": i[g["cooldown"]], "obs_noise_level_cur": d[g["obs_noise_level_cur"]], "action_noise_std_cur": d[g["action_noise_std_cur"]],
            "push_enable": i[g["push_enable"]], "push_force_range_cur": d[g["push_force_lo"]:g["push_force_lo"] + 2],
            "push_interval_steps": i[g["push_interval"]], "delay_max_cur": i[g["delay_max_cur"]],
            "cmd_ranges": {"lin_vel_x_range": d[g["cmd_x_lo"]:g["cmd_x_lo"] + 2], "lin_vel_y_range": d[g["cmd_y_lo"]:g["cmd_y_lo"] + 2],
                           "ang_vel_range": d[g["cmd_yaw_lo"]:g["cmd_yaw_lo"] + 2]},
        }
        self.extras["domain_randomization"] = {"friction": f[g["friction"]], "mass_shift": f[g["mass_shift"]],
                                               "com_shift": f[g["com_shift"]:g["com_shift"] + 3],
                                               "leg_mass_shift": f[g["leg_mass_shift"]:g["leg_mass_shift"] + 4]}

    def _raise_on_errno(self, v):
        """rigid_solver.py:1189-1213: the exceptions scene.step raises from its errno poll."""
        if not v:
            return
        if v & C["GO2SIM_ERR_INVALID_FORCE_NAN"]:
            raise Go2SimError("Invalid constraint forces causing 'nan'. Some environments were not advanced.")
        if v & C["GO2SIM_ERR_INVALID_ACC_NAN"]:
            raise Go2SimError("Invalid accelerations causing 'nan'. Some environments were not advanced.")
        if v & (C["GO2SIM_ERR_OVERFLOW_CANDIDATE_CONTACTS"] | C["GO2SIM_ERR_OVERFLOW_COLLISION_PAIRS"]):
            raise Go2SimError("Exceeding max number of broad phase candidate contact pairs / contacts.")

    def get_observations(self):
        self.extras["observations"]["critic"] = self.privileged_obs_buf if self.num_privileged_obs else self.obs_buf
        return self.obs_buf, self.extras

    def get_privileged_observations(self):
        return self.privileged_obs_buf if self.num_privileged_obs is not None else None

    def reset(self):
        """go2_env_walk.py:1242-1245 (reset_idx over all envs; obs_buf is not recomputed, as in the reference)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._sim.env_reset(stream)
        self.reset_buf.fill_(1)
        if self.log_extras:
            self._refresh_extras()
        return self.obs_buf, None

    def reset_idx(self, envs_idx):
        """go2_env_walk.py:1156-1240: curriculum bookkeeping, "global" DR draws and per-env reset of ``envs_idx`` (index tensor / list)."""
        idx = torch.as_tensor(envs_idx, device=self.device).to(torch.int32).contiguous()
        if idx.numel() == 0:
            return
        self._sim.env_reset_idx(idx, idx.numel(), torch.cuda.current_stream(self.device).cuda_stream)
        self.reset_buf[idx.long()] = 1
        if self.log_extras:
            self._refresh_extras()

    # ---- eval / teleop surface (go2_eval_walk.py, go2_eval_stairs.py) ------------------------------
    @property
    def _lock_terrain_rows(self):
        return self._terrain_rows_locked

    @_lock_terrain_rows.setter
    def _lock_terrain_rows(self, value):
        """go2_env_stair.py:399,1513; go2_eval_stairs.py:657 sets it after the first reset."""
        self._terrain_rows_locked = bool(value)
        self._sim.env_lock_terrain_rows(self._terrain_rows_locked, torch.cuda.current_stream(self.device).cuda_stream)

    def set_terrain_rows(self, rows):
        """``env._env_terrain_row[:] = rows`` of the eval scripts."""
        r = torch.as_tensor(rows, device=self.device).to(torch.int32).expand(self.num_envs).contiguous()
        self._sim.env_set_terrain_rows(r, torch.cuda.current_stream(self.device).cuda_stream)

    def respawn(self, envs_idx, pos, quat=None, clear_buffers=True):
        """Teleport ``envs_idx`` to ``pos`` [n, 3] (``quat`` [n, 4] wxyz, default base_init_quat) in the default joint pose with zero velocity:
        respawn_at_start (go2_eval_stairs.py:314-361, clear_buffers=True) / respawn_on_tile (go2_eval_walk.py:399-480, clear_buffers=False)."""
        idx = torch.as_tensor(envs_idx, device=self.device).to(torch.int32).contiguous()
        p = torch.as_tensor(pos, device=self.device, dtype=torch.float32).reshape(idx.numel(), 3).contiguous()
        q = None if quat is None else torch.as_tensor(quat, device=self.device, dtype=torch.float32).reshape(idx.numel(), 4).contiguous()
        self._sim.env_respawn(idx, p, q, clear_buffers, idx.numel(), torch.cuda.current_stream(self.device).cuda_stream)

    def respawn_at_start(self, envs_idx=(0,)):
        """go2_eval_stairs.py:314-361: back to the spawn point of the env's terrain row (or base_init_pos on flat ground)."""
        idx = torch.as_tensor(envs_idx, device=self.device).long()
        if self._use_terrain:
            rows = self._env_view("TERRAIN_ROW", 1, torch.int32)[idx].long()
            centers = torch.tensor(self._terrain_info["row_centers"], device=self.device, dtype=torch.float32)
            pos = centers[rows].clone()
            pos[:, 2] += float(self.env_cfg["base_init_pos"][2])
        else:
            pos = torch.tensor(self.env_cfg["base_init_pos"], device=self.device, dtype=torch.float32).expand(idx.numel(), 3)
        self.respawn(idx, pos, None, True)

    # ---- state the reference exposes as attributes ----------------------------------------------
    def _env_view(self, name, k, dtype=torch.float32):
        key = (name, k)
        if key not in self._views:
            self._views[key] = torch.zeros(self.num_envs, k, dtype=dtype, device=self.device)
        buf = self._views[key]   # env_get delivers the reference's [n_envs, k] layout
        self._sim.env_get(C["GO2SIM_EB_" + name], buf, torch.cuda.current_stream(self.device).cuda_stream)
        return buf if k > 1 else buf[:, 0]

    @property
    def episode_length_buf(self):
        return self._env_view("EPISODE_LENGTH", 1, torch.int32)

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        """rsl_rl ``init_at_random_ep_len`` assigns a fresh tensor here (on_policy_runner)."""
        v = value.to(device=self.device, dtype=torch.int32).contiguous()
        self._sim.env_set_episode_length(v, torch.cuda.current_stream(self.device).cuda_stream)

    commands = property(lambda self: self._env_view("COMMANDS", 3))
    base_lin_vel = property(lambda self: self._env_view("BASE_LIN_VEL", 3))
    base_ang_vel = property(lambda self: self._env_view("BASE_ANG_VEL", 3))
    projected_gravity = property(lambda self: self._env_view("PROJECTED_GRAVITY", 3))
    dof_pos = property(lambda self: self._env_view("DOF_POS", 12))
    dof_vel = property(lambda self: self._env_view("DOF_VEL", 12))
    base_pos = property(lambda self: self._env_view("BASE_POS", 3))
    base_quat = property(lambda self: self._env_view("BASE_QUAT", 4))
    base_euler = property(lambda self: self._env_view("BASE_EULER", 3))

    @property
    def episode_sums(self):
        sums = self._env_view("EPISODE_SUMS", 32)
        return {n: sums[:, i] for i, n in enumerate(self._reward_names)}

    def set_commands(self, commands):
        """Eval scripts overwrite ``env.commands`` (go2_eval_walk.py); here through the setter of the C ABI."""
        v = commands.to(device=self.device, dtype=torch.float32).contiguous()
        self._sim.env_set_commands(v, torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def curriculum_level(self):
        return self._glob_f64[self._level_off]

    def curriculum_state(self):
        """extras["curriculum"] of the reference (go2_env_walk.py:674-690); this call synchronises the stream."""
        return self._sim.env_globals(torch.cuda.current_stream(self.device).cuda_stream).as_dict()

    def sync_globals(self, group=None, initial=False):
        """One batch sharded over ranks (``shared_globals=True``): all-reduce of the curriculum counters, the shared state machine, broadcast of
        rank 0's global DR draws (distributed.sync_env_globals).  Call on every rank once per rollout."""
        from .distributed import sync_env_globals

        return sync_env_globals(self._sim, group, torch.cuda.current_stream(self.device).cuda_stream, initial=initial)

    def check_errno(self):
        """rigid_solver.py:1208-1211: blocking form of the poll (``step`` runs the asynchronous one every ``errno_poll_every`` steps)."""
        v = self._sim.check_errno(torch.cuda.current_stream(self.device).cuda_stream)
        self._raise_on_errno(v)
        return v

    def graph_status(self):
        """(step graph in use, number of times the library fell back to plain launches)."""
        return self._sim.graph_status()
