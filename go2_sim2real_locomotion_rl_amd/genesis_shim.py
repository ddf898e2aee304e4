"""gs -- the Genesis Python surface that the reference's Go2Env files use, on top of the go2sim C ABI.

    import genesis as gs                     # alias package at the repo root (genesis/__init__.py), or
    import go2_sim2real_locomotion_rl_amd.genesis_shim as gs
    gs.init(backend=gs.gpu, precision="32")
    scene = gs.Scene(sim_options=gs.options.SimOptions(dt=0.02, substeps=2), rigid_options=gs.options.RigidOptions(...))
    scene.add_entity(gs.morphs.URDF(file="urdf/plane/plane.urdf", fixed=True))
    robot = scene.add_entity(gs.morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=..., quat=...))
    scene.build(n_envs=4096); robot.control_dofs_position(q, dofs_idx); scene.step(); robot.get_pos() ...

This is the *general* (slow) level of the boundary: every accessor is one or two device copies through go2sim_get_field / set_field, exactly
as the reference's accessors are kernels over the SoA state (genesis/engine/entities/rigid_entity/rigid_entity.py, solvers/rigid/abd/accessor.py).
The fused fast path is go2_env.Go2Env.  Conventions follow the reference: tensors are [n_envs(sel), n_idx(, k)], envs_idx is an index tensor or
None, dofs_idx_local / links_idx_local are python lists or tensors, getters return new tensors on gs.device, quaternions are (w, x, y, z).
Only the compiled Go2 scene (plane or heightfield terrain + Go2 URDF) is available; anything else raises GenesisException."""
import types

import numpy as np
import torch

from .capi import C, Go2Sim, Go2SimError, load_hip_lib
from .model_blob import load_model_json, pack_model

gpu, cpu = "gpu", "cpu"
tc_float, tc_int = torch.float32, torch.int32
device = None
_lib = None
_seed = 1
EPS = float(np.finfo(np.float32).eps)     # gs.EPS = max(1e-15, eps of the float type) (genesis/__init__.py:62,229), precision "32"


class GenesisException(Exception):
    pass


def init(backend=gpu, precision="32", logging_level=None, performance_mode=True, seed=None, **_):
    """gs.init (genesis/__init__.py:60).  There is one backend: the HIP library on the current ROCm device."""
    global device, _lib, _seed
    if str(precision) != "32":
        raise GenesisException("go2sim computes in fp32 only")
    if backend == cpu:
        raise GenesisException("go2sim has no CPU backend (gs.cpu is defined for source compatibility only)")
    if not torch.cuda.is_available():
        raise GenesisException("no ROCm GPU visible: the go2sim product path has no CPU fallback")
    _lib, device = load_hip_lib(), torch.device("cuda", torch.cuda.current_device())
    if seed is not None:
        _seed = int(seed)


class _Opt:
    def __init__(self, **kw):
        self.__dict__.update(kw)


options = types.SimpleNamespace(SimOptions=_Opt, ViewerOptions=_Opt, VisOptions=_Opt, RigidOptions=_Opt)
constraint_solver = types.SimpleNamespace(Newton="Newton", CG="CG")
morphs = types.SimpleNamespace(URDF=lambda **kw: _Opt(kind="urdf", **kw), Terrain=lambda **kw: _Opt(kind="terrain", **kw))


# ---- genesis.utils.geom helpers used by Go2Env (torch branch of genesis/utils/geom.py; same operation order, so an env file that runs on
# the shim sees the values it would see on Genesis) --------------------------------------------------------------------------------------
def inv_quat(q):
    """Conjugate (geom.py `inv_quat`: the sign of the vector part flips; unit quaternions assumed)."""
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)


def transform_quat_by_quat(v, u):
    """quat_mul(u, v) followed by a normalisation (geom.py:989-1023, the 8-multiplication product of `_tc_quat_mul`)."""
    w1, x1, y1, z1 = u.unbind(-1)
    w2, x2, y2, z2 = v.unbind(-1)
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    out = torch.stack([qq - ww + (z1 - y1) * (y2 - z2), qq - xx + (x1 + w1) * (x2 + w2),
                       qq - yy + (w1 - x1) * (y2 + z2), qq - zz + (z1 + y1) * (w2 - x2)], dim=-1)
    return out / torch.linalg.vector_norm(out, ord=2, dim=-1, keepdim=True)


def _quat_products(q):
    w, x, y, z = q[..., :1], q[..., 1:2], q[..., 2:3], q[..., 3:]
    return w * w, w * x, w * y, w * z, x * x, x * y, x * z, y * y, y * z, z * z


def transform_by_quat(v, q):
    """Rotation-matrix form with the division by |q|^2 applied to v first (geom.py:1048-1070, `_tc_transform_by_quat`)."""
    ww, wx, wy, wz, xx, xy, xz, yy, yz, zz = _quat_products(q)
    vs = v / (ww + xx + yy + zz)
    vx, vy, vz = vs[..., :1], vs[..., 1:2], vs[..., 2:]
    return torch.cat([vx * (xx + ww - yy - zz) + vy * (2.0 * xy - 2.0 * wz) + vz * (2.0 * xz + 2.0 * wy),
                      vx * (2.0 * wz + 2.0 * xy) + vy * (ww - xx + yy - zz) + vz * (2.0 * yz - 2.0 * wx),
                      vx * (2.0 * xz - 2.0 * wy) + vy * (2.0 * wx + 2.0 * yz) + vz * (ww - xx - yy + zz)], dim=-1)


def quat_to_xyz(q, rpy=False, degrees=False):
    """Euler angles by atan2 of half-scaled products, with the cos(pitch) < EPS branch (geom.py:717-774, `_tc_quat_to_xyz`).  `rpy` defaults
    to False like the reference; the Go2Env files pass rpy=True."""
    ww, wx, wy, wz, xx, xy, xz, yy, yz, zz = _quat_products(q)
    if rpy:
        sinp, sinrcosp, sinycosp = wy - xz, wx + yz, wz + xy
    else:
        sinp, sinrcosp, sinycosp = xz + wy, wx - yz, wz - xy
    cosrcosp = (ww - xx - yy + zz) / 2
    cosycosp = (ww + xx - yy - zz) / 2
    cosp = torch.sqrt(cosycosp**2 + sinycosp**2)
    roll, pitch, yaw = torch.atan2(sinrcosp, cosrcosp), torch.atan2(sinp, cosp), torch.atan2(sinycosp, cosycosp)
    singular = cosp < EPS
    roll = roll.masked_fill(singular, 0.0)
    yaw = torch.where(singular, torch.atan2((wz - xy) if rpy else (wz + xy), (ww - xx + yy - zz) / 2), yaw)
    out = torch.cat([roll, pitch, yaw], dim=-1)
    return torch.rad2deg(out) if degrees else out


utils = types.SimpleNamespace(geom=types.SimpleNamespace(inv_quat=inv_quat, transform_quat_by_quat=transform_quat_by_quat,
                                                         transform_by_quat=transform_by_quat, quat_to_xyz=quat_to_xyz))


def _F(name):
    return C["GO2SIM_" + name]


def _cross(a, b):
    """a x b along the last axis as separate multiplies and subtractions (torch.cross may contract them into fused multiply-adds on the CPU; the
    C ABI and the reference's kernels round every product)."""
    ax, ay, az = a.unbind(-1)
    bx, by, bz = b.unbind(-1)
    return torch.stack([ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx], dim=-1)


def _broadcast(values, shape):
    """What the reference's setters do to their value argument before the accessor kernel runs (rigid_solver.py:1835-1871 ->
    genesis/utils/misc.py `broadcast_tensor`): a value with fewer dimensions than (n_envs(sel), n_idx(, k)) is matched right-aligned when
    that fits, otherwise its dimensions are assigned to the expected ones preferring the leading ones, and size-1 dimensions expand.
    So `set_mass_shift([s], [0])` and `set_COM_shift([[x, y, z]], [0])` with envs_idx=None reach every env (go2_env_walk.py:810,820,843),
    and a [B, 3] force for one link becomes [B, 1, 3]."""
    import itertools

    t = torch.as_tensor(values, dtype=torch.float32, device=device)
    shape = tuple(int(n) for n in shape)
    if t.ndim == 0:
        t = t[None]
    elif t.ndim > len(shape):
        raise GenesisException(f"Invalid input shape: {tuple(t.shape)}. Expecting at most {len(shape)}D tensor.")
    elif t.ndim < len(shape) and any(a != b for a, b in zip(t.shape, shape[-t.ndim:])):
        for keep in reversed(list(itertools.combinations(range(len(shape)), t.ndim))):
            dims, k = [], 0
            for i, n in enumerate(shape):
                if i in keep:
                    if t.shape[k] not in (n, 1):
                        break
                    dims.append(t.shape[k]); k += 1
                else:
                    dims.append(1)
            else:
                t = t.reshape(dims)
                break
    try:
        return t.expand(shape)
    except RuntimeError as e:
        raise GenesisException(f"Invalid input shape: {tuple(t.shape)}. Expected shape: {shape}.") from e


class _Named:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class RigidEntity:
    """The subset of genesis RigidEntity that Go2Env touches (rigid_entity.py; accessor semantics of abd/accessor.py:652-875)."""

    def __init__(self, scene, morph, link_start, link_end, dof_start, dof_end):
        self._scene, self.morph = scene, morph
        self._link_start, self._link_end, self._dof_start, self._dof_end = link_start, link_end, dof_start, dof_end
        m = scene._model
        self.links = [_Named(name=l["name"], idx=i, idx_local=i - link_start) for i, l in enumerate(m["links"]) if link_start <= i < link_end]
        self.joints = [_Named(name=j["name"], dof_start=j["dof_start"], q_start=j["q_start"], idx=k) for k, j in enumerate(m["joints"])
                       if link_start <= j["link"] < link_end]
        self.n_links, self.n_dofs = link_end - link_start, dof_end - dof_start

    # ---- lookup ----
    def get_joint(self, name):
        for j in self.joints:
            if j.name == name:
                return j
        raise GenesisException(f"Joint not found for name: {name}.")

    def get_link(self, name):
        for l in self.links:
            if l.name == name:
                return l
        raise GenesisException(f"Link not found for name: {name}.")

    # ---- helpers ----
    @property
    def _sim(self):
        if self._scene._sim is None:
            raise GenesisException("Scene is not built yet.")
        return self._scene._sim

    def _get(self, field, dtype=torch.float32):
        k, is_int = self._sim.field_size(_F(field))
        t = torch.zeros(k, self._sim.n_envs, dtype=torch.int32 if is_int else torch.float32, device=device)
        self._sim.get_field(_F(field), t)
        return t

    def _set(self, field, t):
        self._sim.set_field(_F(field), t.contiguous())

    def _envs(self, envs_idx):
        if envs_idx is None:
            return torch.arange(self._sim.n_envs, device=device)
        return torch.as_tensor(envs_idx, device=device).long().reshape(-1)

    def _dofs(self, dofs_idx_local):
        if dofs_idx_local is None:
            return torch.arange(self._dof_start, self._dof_end, device=device)
        return torch.as_tensor(dofs_idx_local, device=device).long().reshape(-1) + self._dof_start

    def _links(self, links_idx_local):
        if links_idx_local is None:
            return torch.arange(self._link_start, self._link_end, device=device)
        return torch.as_tensor(links_idx_local, device=device).long().reshape(-1) + self._link_start

    def _link_vec(self, field, k):
        return self._get(field).reshape(-1, k, self._sim.n_envs).permute(2, 0, 1)     # [B, n_links_total, k]

    def _root_com(self):
        return self._get("F_ROOT_COM").t()                                                   # [B, 3] (robot)

    # ---- getters (rigid_entity.py:2600-2900) ----
    def get_pos(self, envs_idx=None):
        return self._link_vec("F_LINK_POS", 3)[self._envs(envs_idx), self._link_start].clone()

    def get_quat(self, envs_idx=None):
        return self._link_vec("F_LINK_QUAT", 4)[self._envs(envs_idx), self._link_start].clone()

    def get_links_pos(self, links_idx_local=None, envs_idx=None):
        return self._link_vec("F_LINK_POS", 3)[self._envs(envs_idx)][:, self._links(links_idx_local)].clone()

    def get_links_vel(self, links_idx_local=None, envs_idx=None):
        """Linear velocity of the link origins (kernel_get_links_vel, ref = link_origin)."""
        pos, cdv, cda = self._link_vec("F_LINK_POS", 3), self._link_vec("F_LINK_CDVEL", 3), self._link_vec("F_LINK_CDANG", 3)
        vel = cdv + _cross(cda, pos - self._root_com()[:, None, :])
        return vel[self._envs(envs_idx)][:, self._links(links_idx_local)].clone()

    def get_vel(self, envs_idx=None):
        return self.get_links_vel([0], envs_idx)[:, 0]

    def get_ang(self, envs_idx=None):
        return self._link_vec("F_LINK_CDANG", 3)[self._envs(envs_idx), self._link_start].clone()

    def get_links_net_contact_force(self, envs_idx=None):
        return self._link_vec("F_CONTACT_FORCE", 3)[self._envs(envs_idx)][:, self._links(None)].clone()

    def get_dofs_position(self, dofs_idx_local=None, envs_idx=None):
        return self._get("F_DOF_POS").t()[self._envs(envs_idx)][:, self._dofs(dofs_idx_local)].clone()

    def get_dofs_velocity(self, dofs_idx_local=None, envs_idx=None):
        return self._get("F_VEL").t()[self._envs(envs_idx)][:, self._dofs(dofs_idx_local)].clone()

    def get_dofs_control_force(self, dofs_idx_local=None, envs_idx=None):
        """kernel_get_dofs_control_force, abd/accessor.py:848-875."""
        d = self._dofs(dofs_idx_local)
        mode = self._get("I_CTRL_MODE").t()[:, d]
        dofs = self._scene._model["dofs"]
        kp = torch.tensor([self._scene._gains[i][0] for i in d.tolist()], device=device)
        kv = torch.tensor([self._scene._gains[i][1] for i in d.tolist()], device=device)
        lo = torch.tensor([dofs[i]["force_range"][0] for i in d.tolist()], device=device)
        hi = torch.tensor([dofs[i]["force_range"][1] for i in d.tolist()], device=device)
        vel, pos = self._get("F_VEL").t()[:, d], self._get("F_DOF_POS").t()[:, d]
        cf, cp, cv = self._get("F_CTRL_FORCE").t()[:, d], self._get("F_CTRL_POS").t()[:, d], self._get("F_CTRL_VEL").t()[:, d]
        force = torch.where(mode == 0, cf, torch.where(mode == 1, kv * (cv - vel), kp * (cp - pos) + kv * (cv - vel)))
        return torch.minimum(torch.maximum(force, lo), hi)[self._envs(envs_idx)]

    # ---- control (rigid_entity.py:2450-2560) ----
    def _write_dofs(self, field, values, dofs_idx_local, envs_idx, mode=None):
        d, e = self._dofs(dofs_idx_local), self._envs(envs_idx)
        cur = self._get(field)
        cur[d[:, None], e[None, :]] = _broadcast(values, (len(e), len(d))).t()
        self._set(field, cur)
        if mode is not None:
            m = self._get("I_CTRL_MODE")
            m[d[:, None], e[None, :]] = mode
            self._set("I_CTRL_MODE", m)

    def control_dofs_force(self, force, dofs_idx_local=None, envs_idx=None):
        self._write_dofs("F_CTRL_FORCE", force, dofs_idx_local, envs_idx, mode=0)

    def control_dofs_position(self, position, dofs_idx_local=None, envs_idx=None):
        self._write_dofs("F_CTRL_POS", position, dofs_idx_local, envs_idx, mode=2)
        d, e = self._dofs(dofs_idx_local), self._envs(envs_idx)
        self._write_dofs("F_CTRL_VEL", torch.zeros(len(e), len(d), device=device), dofs_idx_local, envs_idx)

    def set_dofs_kp(self, kp, dofs_idx_local=None):
        self._set_gains(kp, dofs_idx_local, 0)

    def set_dofs_kv(self, kv, dofs_idx_local=None):
        self._set_gains(kv, dofs_idx_local, 1)

    def _set_gains(self, values, dofs_idx_local, which):
        dofs = self._scene._model["dofs"]
        for i, v in zip(self._dofs(dofs_idx_local).tolist(), list(np.asarray(values, dtype=np.float64).reshape(-1))):
            g = self._scene._gains[i]
            g[which] = float(v)
            self._sim.set_dof_gains(i, g[0], g[1], dofs[i]["force_range"][0], dofs[i]["force_range"][1])

    # ---- state setters (rigid_solver.py:1876-2029, 2385-2427): cache reset + full-batch FK like the reference ----
    def set_dofs_position(self, position, dofs_idx_local=None, zero_velocity=True, envs_idx=None):
        d, e = self._dofs(dofs_idx_local), self._envs(envs_idx)
        q = self._get("F_QPOS")
        qpos0 = torch.tensor(self._scene._model["qpos0"], device=device)
        qi = d + 1                                                                       # revolute dof d <-> qpos index d + 1 (free joint: 7 q, 6 dofs)
        vals = _broadcast(position, (len(e), len(d))).t()
        q[qi[:, None], e[None, :]] = qpos0[qi][:, None] + vals
        self._set("F_QPOS", q)
        if zero_velocity:
            v = self._get("F_VEL")
            v[d[:, None], e[None, :]] = 0.0
            self._set("F_VEL", v)
        self._after_state_write(e)

    def set_pos(self, pos, zero_velocity=True, envs_idx=None):
        self._set_base(pos, 0, 3, zero_velocity, envs_idx)

    def set_quat(self, quat, zero_velocity=True, envs_idx=None):
        self._set_base(quat, 3, 7, zero_velocity, envs_idx)

    def _set_base(self, val, lo, hi, zero_velocity, envs_idx):
        e = self._envs(envs_idx)
        q = self._get("F_QPOS")
        q[lo:hi, e] = _broadcast(val, (len(e), hi - lo)).t()
        self._set("F_QPOS", q)
        if zero_velocity:
            self.zero_all_dofs_velocity(envs_idx)
        else:
            self._after_state_write(e)

    def zero_all_dofs_velocity(self, envs_idx=None):
        e = self._envs(envs_idx)
        v = self._get("F_VEL")
        v[:, e] = 0.0
        self._set("F_VEL", v)
        self._after_state_write(e)

    def _after_state_write(self, e):
        self._sim.reset_caches(e.to(torch.int32).contiguous(), len(e))
        self._sim.forward_kinematics()

    # ---- domain randomisation hooks ----
    def set_friction(self, friction):
        self._sim.set_friction(float(friction))

    def set_mass_shift(self, mass_shift, links_idx_local=None, envs_idx=None):
        l, e = self._links(links_idx_local), self._envs(envs_idx)
        cur = self._get("F_MASS_SHIFT")
        cur[l[:, None], e[None, :]] = _broadcast(mass_shift, (len(e), len(l))).t()
        self._set("F_MASS_SHIFT", cur)

    def set_COM_shift(self, com_shift, links_idx_local=None, envs_idx=None):
        l, e = self._links(links_idx_local), self._envs(envs_idx)
        cur = self._get("F_COM_SHIFT").reshape(-1, 3, self._sim.n_envs)
        cur[l[:, None], :, e[None, :]] = _broadcast(com_shift, (len(e), len(l), 3)).permute(1, 0, 2)
        self._set("F_COM_SHIFT", cur.reshape(-1, self._sim.n_envs))


class _RigidSolver:
    def __init__(self, scene):
        self._scene = scene

    def apply_links_external_force(self, force, links_idx, envs_idx=None, ref="link_origin", local=False):
        """func_apply_link_external_force, abd/misc.py:695-715 (ref = link origin): applied for the next scene.step only."""
        scene = self._scene
        robot = scene._robot
        e = robot._envs(envs_idx)
        l = torch.as_tensor(links_idx, device=device).long().reshape(-1)
        f = _broadcast(force, (len(e), len(l), 3))
        ext = robot._get("F_EXT_FORCE").reshape(-1, 6, scene._sim.n_envs)                  # [link, (ang3, vel3), B]
        pos = robot._link_vec("F_LINK_POS", 3)[e][:, l]
        torque = _cross(pos - robot._root_com()[e][:, None, :], f)
        ext[l[:, None], 3:6, e[None, :]] -= f.permute(1, 0, 2)
        ext[l[:, None], 0:3, e[None, :]] -= torque.permute(1, 0, 2)
        robot._set("F_EXT_FORCE", ext.reshape(-1, scene._sim.n_envs))

    def check_errno(self):
        v = self._scene._sim.check_errno()
        if v & C["GO2SIM_ERR_INVALID_FORCE_NAN"]:
            raise GenesisException("Invalid constraint forces causing 'nan'. Some environments were not advanced.")
        if v & C["GO2SIM_ERR_INVALID_ACC_NAN"]:
            raise GenesisException("Invalid accelerations causing 'nan'. Some environments were not advanced.")


class Scene:
    """gs.Scene for the compiled Go2 scene: one ground entity (plane URDF or heightfield Terrain) and the Go2 URDF."""

    def __init__(self, sim_options=None, viewer_options=None, vis_options=None, rigid_options=None, show_viewer=False, **_):
        if show_viewer:
            raise GenesisException("the viewer is outside the accelerated path")
        if device is None:
            raise GenesisException("Genesis hasn't been initialized. Did you call `gs.init()`?")
        self._substeps = int(getattr(sim_options, "substeps", 2)) if sim_options is not None else 2
        self._model = load_model_json()
        self._gains = {i: [float(d.get("kp", 0.0)), float(d.get("kv", 0.0))] for i, d in enumerate(self._model["dofs"])}
        self._entities, self._sim, self._robot, self._ground_morph = [], None, None, None
        self.sim = types.SimpleNamespace(rigid_solver=_RigidSolver(self))
        self.rigid_solver = self.sim.rigid_solver

    def add_entity(self, morph, **_):
        ents = self._model["entities"]
        kind = getattr(morph, "kind", None)
        if kind == "terrain" or (kind == "urdf" and "plane" in str(getattr(morph, "file", ""))):
            if self._ground_morph is not None:
                raise GenesisException("the compiled scene has exactly one ground entity")
            self._ground_morph = morph
            ent = RigidEntity(self, morph, ents[0]["link_start"], ents[0]["link_end"], ents[0]["dof_start"], ents[0]["dof_end"])
        elif kind == "urdf" and "go2" in str(getattr(morph, "file", "")):
            ent = RigidEntity(self, morph, ents[1]["link_start"], ents[1]["link_end"], ents[1]["dof_start"], ents[1]["dof_end"])
            self._robot = ent
        else:
            raise GenesisException("go2sim ships the compiled plane / terrain + Go2 scene only (tools/compile_go2_model.py)")
        self._entities.append(ent)
        return ent

    def build(self, n_envs=1, **_):
        if self._robot is None or self._ground_morph is None:
            raise GenesisException("add the ground entity and the Go2 URDF before build()")
        self._sim = Go2Sim(_lib, pack_model(self._model), int(n_envs), device.index or 0, _seed)
        g = self._ground_morph
        if getattr(g, "kind", None) == "terrain":
            self._sim.set_terrain(np.asarray(g.height_field, np.int16), float(g.horizontal_scale), float(g.vertical_scale),
                                  list(getattr(g, "pos", (0.0, 0.0, 0.0))))
        m = self._robot.morph
        if getattr(m, "pos", None) is not None or getattr(m, "quat", None) is not None:     # initial base pose of the URDF morph
            q = self._robot._get("F_QPOS")
            if getattr(m, "pos", None) is not None:
                q[0:3] = torch.tensor(np.asarray(m.pos, np.float32), device=device)[:, None]
            if getattr(m, "quat", None) is not None:
                q[3:7] = torch.tensor(np.asarray(m.quat, np.float32), device=device)[:, None]
            self._robot._set("F_QPOS", q)
            self._sim.forward_kinematics()
        self.n_envs = int(n_envs)

    def step(self):
        self._sim.scene_step(self._substeps)

    def reset(self):
        self._sim.scene_reset()
