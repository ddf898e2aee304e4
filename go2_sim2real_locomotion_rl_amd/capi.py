"""ctypes binding of the go2sim C ABI (include/go2sim.h).

One wrapper class serves both libraries that export the ABI:
  * csrc/libgo2sim.so        (prefix ``go2sim_``,     HIP/gfx950 product, device pointers)
  * oracle/libgo2sim_cpu.so  (prefix ``go2sim_cpu_``, CPU twin used ONLY by tests / smoke / cpu_baseline)

Enum values (fields, cfg indices, rewards, env buffers) are parsed from the header text so that the
Python side can never drift from the C side.
"""
import ctypes
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(REPO_ROOT, "include", "go2sim.h")
HIP_LIB = os.path.join(_HERE, "csrc", "libgo2sim.so")
CPU_LIB = os.path.join(REPO_ROOT, "oracle", "libgo2sim_cpu.so")
CPU_LIB_FAST = os.path.join(REPO_ROOT, "oracle", "libgo2sim_cpu_fast.so")


def _parse_header(path=HEADER):
    txt = open(path).read()
    txt_nc = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    consts = {}
    for m in re.finditer(r"#define\s+(GO2SIM_\w+)\s+(-?(?:0x[0-9A-Fa-f]+|\d+))\s*$", txt_nc, flags=re.M):
        consts[m.group(1)] = int(m.group(2), 0)
    for em in re.finditer(r"enum\s+\w+\s*\{(.*?)\}", txt_nc, flags=re.S):
        val = -1
        for item in em.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [s.strip() for s in item.split("=", 1)]
                val = int(eval(expr, {}, dict(consts)))
            else:
                name, val = item, val + 1
            consts[name] = val
    decls = re.findall(r"^int\s+(go2sim_\w+)\s*\(", txt_nc, flags=re.M)
    return consts, decls


C, DECLARED_FUNCS = _parse_header()
# the policy-inference entry points (include/go2sim_policy.h) live in the same two libraries
_C_POLICY, _DECL_POLICY = _parse_header(os.path.join(REPO_ROOT, "include", "go2sim_policy.h"))
C.update(_C_POLICY)
DECLARED_FUNCS = DECLARED_FUNCS + _DECL_POLICY


class EnvGlobals(ctypes.Structure):
    _fields_ = [
        ("level", ctypes.c_double), ("timeout_rate_ema", ctypes.c_double), ("tracking_ema", ctypes.c_double), ("fall_rate_ema", ctypes.c_double),
        ("curr_timeout_total", ctypes.c_double), ("curr_tracking_sum", ctypes.c_double),
        ("obs_noise_level_cur", ctypes.c_double), ("action_noise_std_cur", ctypes.c_double),
        ("push_force_lo", ctypes.c_double), ("push_force_hi", ctypes.c_double),
        ("cmd_x_lo", ctypes.c_double), ("cmd_x_hi", ctypes.c_double), ("cmd_y_lo", ctypes.c_double), ("cmd_y_hi", ctypes.c_double),
        ("cmd_yaw_lo", ctypes.c_double), ("cmd_yaw_hi", ctypes.c_double), ("t_sample", ctypes.c_double),
        ("ema_valid", ctypes.c_int), ("ready_streak", ctypes.c_int), ("hard_streak", ctypes.c_int), ("cooldown", ctypes.c_int),
        ("curr_ep_total", ctypes.c_int), ("curr_tracking_n", ctypes.c_int),
        ("push_enable", ctypes.c_int), ("push_interval", ctypes.c_int), ("push_counter", ctypes.c_int), ("delay_max_cur", ctypes.c_int),
        ("global_dr_reset_counter", ctypes.c_int),
        ("friction", ctypes.c_float), ("mass_shift", ctypes.c_float), ("com_shift", ctypes.c_float * 3), ("leg_mass_shift", ctypes.c_float * 4),
        ("action_write_idx", ctypes.c_int), ("step_count", ctypes.c_uint), ("reset_calls", ctypes.c_uint),
        ("last_reset_count", ctypes.c_int), ("last_episode_rew", ctypes.c_float * 32),
        ("n_reset_now", ctypes.c_int), ("ep_acc", ctypes.c_float * 32), ("terrain_mean_row", ctypes.c_float), ("terrain_row_sum", ctypes.c_int),
        ("lock_terrain_rows", ctypes.c_int), ("sync_calls", ctypes.c_int), ("shard_counters", ctypes.c_double * 5),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


def _ptr(x):
    """Accept raw addresses (int), None, numpy arrays, torch tensors or ctypes objects."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return ctypes.c_void_p(x.data_ptr())
    return x


class Go2SimError(RuntimeError):
    pass


class Go2SimLib:
    """Thin, explicit binding: one Python method per exported C function."""

    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise Go2SimError(
                f"go2sim shared library not found: {path}. Build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (the product has no CPU fallback)."
            )
        self.path, self.prefix = path, prefix
        self.lib = ctypes.CDLL(path)
        self.is_device = prefix == "go2sim_"
        for name in DECLARED_FUNCS:
            fn = getattr(self.lib, prefix + name[len("go2sim_"):])
            fn.restype = ctypes.c_int

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc, what):
        if rc != 0:
            raise Go2SimError(f"{self.prefix}{what} failed with status {rc}")


class Go2Sim:
    """One simulator handle (n_envs environments on one device)."""

    def __init__(self, lib: Go2SimLib, model_blob: bytes, n_envs: int, device: int = 0, seed: int = 1):
        self.L = lib
        self.n_envs = n_envs
        self._blob = model_blob
        h = ctypes.c_void_p()
        rc = lib.fn("create")(model_blob, ctypes.c_size_t(len(model_blob)), ctypes.c_int(n_envs), ctypes.c_int(device),
                              ctypes.c_uint64(seed), ctypes.byref(h))
        lib.check(rc, "create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene level ---------------------------------------------------------------------------
    def _call(self, name, *args):
        self.L.check(self.L.fn(name)(self.h, *args), name)

    def scene_reset(self, stream=None):
        self._call("scene_reset", _ptr(stream))

    def substep(self, stream=None):
        self._call("substep", _ptr(stream))

    def scene_step(self, substeps=2, stream=None):
        self._call("scene_step", ctypes.c_int(substeps), _ptr(stream))

    def forward_kinematics(self, stream=None):
        self._call("forward_kinematics", _ptr(stream))

    def field_size(self, field):
        k, is_int = ctypes.c_int(), ctypes.c_int()
        self.L.check(self.L.fn("field_size")(ctypes.c_int(field), ctypes.byref(k), ctypes.byref(is_int)), "field_size")
        return k.value, bool(is_int.value)

    def get_field(self, field, dst, stream=None):
        self._call("get_field", ctypes.c_int(field), _ptr(dst), _ptr(stream))

    def set_field(self, field, src, stream=None):
        self._call("set_field", ctypes.c_int(field), _ptr(src), _ptr(stream))

    def field_ptr(self, field):
        p = ctypes.c_void_p()
        self._call("field_ptr", ctypes.c_int(field), ctypes.byref(p))
        return p.value

    def reset_caches(self, envs_idx=None, n_sel=0, stream=None):
        self._call("reset_caches", _ptr(envs_idx), ctypes.c_int(n_sel), _ptr(stream))

    def set_friction(self, mu, stream=None):
        self._call("set_friction", ctypes.c_float(mu), _ptr(stream))

    def set_terrain(self, hf_int16, horizontal_scale, vertical_scale, origin, stream=None):
        hf = np.ascontiguousarray(hf_int16, dtype=np.int16)
        org = np.ascontiguousarray(origin, dtype=np.float32)
        self._terrain_keepalive = (hf, org)
        self._call("set_terrain", _ptr(hf), ctypes.c_int(hf.shape[0]), ctypes.c_int(hf.shape[1]), ctypes.c_float(horizontal_scale),
                   ctypes.c_float(vertical_scale), _ptr(org), _ptr(stream))

    def set_dof_gains(self, dof, kp, kv, flo, fhi):
        self._call("set_dof_gains", ctypes.c_int(dof), ctypes.c_float(kp), ctypes.c_float(kv), ctypes.c_float(flo), ctypes.c_float(fhi))

    def check_errno(self, stream=None):
        v = ctypes.c_int()
        self._call("check_errno", ctypes.byref(v), _ptr(stream))
        return v.value

    # ---- env level -----------------------------------------------------------------------------
    def env_configure(self, fcfg: np.ndarray, icfg: np.ndarray):
        fcfg = np.ascontiguousarray(fcfg, dtype=np.float64)
        icfg = np.ascontiguousarray(icfg, dtype=np.int32)
        self._call("env_configure", _ptr(fcfg), ctypes.c_int(fcfg.size), _ptr(icfg), ctypes.c_int(icfg.size))

    def env_step(self, actions, obs, priv, rew, reset, timeout, stream=None):
        self._call("env_step", _ptr(actions), _ptr(obs), _ptr(priv), _ptr(rew), _ptr(reset), _ptr(timeout), _ptr(stream))

    def env_reset(self, stream=None):
        self._call("env_reset", _ptr(stream))

    def env_reset_idx(self, envs_idx, n=None, stream=None):
        self._call("env_reset_idx", _ptr(envs_idx), ctypes.c_int(len(envs_idx) if n is None else n), _ptr(stream))

    def env_respawn(self, envs_idx, pos, quat=None, clear_buffers=True, n=None, stream=None):
        self._call("env_respawn", _ptr(envs_idx), ctypes.c_int(len(envs_idx) if n is None else n), _ptr(pos), _ptr(quat),
                   ctypes.c_int(int(clear_buffers)), _ptr(stream))

    def env_lock_terrain_rows(self, lock=True, stream=None):
        self._call("env_lock_terrain_rows", ctypes.c_int(int(lock)), _ptr(stream))

    def env_set_terrain_rows(self, rows, stream=None):
        self._call("env_set_terrain_rows", _ptr(rows), _ptr(stream))

    def errno_poll_begin(self, stream=None):
        self._call("errno_poll_begin", _ptr(stream))

    def errno_poll_result(self):
        v, ready = ctypes.c_int(), ctypes.c_int()
        self._call("errno_poll_result", ctypes.byref(v), ctypes.byref(ready))
        return (v.value if ready.value else None)

    def errno_poll_wait(self):
        v = ctypes.c_int()
        self._call("errno_poll_wait", ctypes.byref(v))
        return v.value

    def graph_status(self):
        using, nfb = ctypes.c_int(), ctypes.c_int()
        self._call("graph_status", ctypes.byref(using), ctypes.byref(nfb))
        return bool(using.value), nfb.value

    def env_get(self, buf, dst, stream=None):
        self._call("env_get", ctypes.c_int(buf), _ptr(dst), _ptr(stream))

    def env_set_episode_length(self, ep, stream=None):
        self._call("env_set_episode_length", _ptr(ep), _ptr(stream))

    def env_set_commands(self, cmd, stream=None):
        self._call("env_set_commands", _ptr(cmd), _ptr(stream))

    def env_globals(self, stream=None):
        g = EnvGlobals()
        self._call("env_globals", ctypes.byref(g), _ptr(stream))
        return g

    def env_globals_ptr(self):
        p = ctypes.c_void_p()
        self._call("env_globals_ptr", ctypes.byref(p))
        return p.value

    def env_sync_counters(self, stream=None):
        out = (ctypes.c_double * 5)()
        self._call("env_sync_counters", out, _ptr(stream))
        return np.array(out[:], np.float64)

    def env_sync_apply(self, summed_counters, stream=None):
        c = (ctypes.c_double * 5)(*[float(v) for v in summed_counters])
        dr = (ctypes.c_double * 10)()
        self._call("env_sync_apply", c, dr, _ptr(stream))
        return np.array(dr[:], np.float64)

    def env_set_global_dr(self, dr10, stream=None):
        self._call("env_set_global_dr", (ctypes.c_double * 10)(*[float(v) for v in dr10]), _ptr(stream))

    def env_sync_counters_dev(self, counters5, stream=None):
        """float64[5] tensor / array in the library's memory space (device for the HIP library); stream-ordered, no synchronisation"""
        self._call("env_sync_counters_dev", _ptr(counters5), _ptr(stream))

    def env_sync_apply_dev(self, summed5, dr_out10, stream=None):
        self._call("env_sync_apply_dev", _ptr(summed5), _ptr(dr_out10), _ptr(stream))

    def env_set_global_dr_dev(self, dr10, stream=None):
        self._call("env_set_global_dr_dev", _ptr(dr10), _ptr(stream))

    def env_set_level(self, level, stream=None):
        self._call("env_set_level", ctypes.c_double(level), _ptr(stream))

    def enable_timing(self, enable=True):
        self._call("enable_timing", ctypes.c_int(int(enable)))

    def read_timing(self, reset=True):
        ms = (ctypes.c_float * 8)()
        cnt = (ctypes.c_int * 8)()
        self._call("read_timing", ms, cnt, ctypes.c_int(int(reset)))
        return list(ms), list(cnt)

    # ---- numpy convenience (host library only; used by tests) ----------------------------------
    def get_field_np(self, field):
        assert not self.L.is_device
        k, is_int = self.field_size(field)
        out = np.zeros((k, self.n_envs), dtype=np.int32 if is_int else np.float32)
        self.get_field(field, out)
        return out

    def set_field_np(self, field, arr):
        assert not self.L.is_device
        k, is_int = self.field_size(field)
        arr = np.ascontiguousarray(arr, dtype=np.int32 if is_int else np.float32).reshape(k, self.n_envs)
        self.set_field(field, arr)


def load_hip_lib():
    """The product library.  Fails loudly when the HIP extension has not been built."""
    return Go2SimLib(HIP_LIB, "go2sim_")


def load_cpu_oracle_lib(fast=False):
    """The CPU oracle.  Test infrastructure only (tests/, smoke(), bench cpu_baseline).  fast=False: the strict build (the reference's CPU /
    serial summation order); fast=True: the -DGO2SIM_FAST_ORDER build that mirrors the HIP product's reduction order bit for bit."""
    return Go2SimLib(CPU_LIB_FAST if fast else CPU_LIB, "go2sim_cpu_")
