"""Parity tests proper: the HIP path vs the CPU oracle through the C ABI, same seeds / actions.

Bar (BASELINE.json north_star): observations and per-term rewards within fp32 tolerance, contact counts and done
masks bit-exact.  Because both sides evaluate identical IEEE binary32 operation sequences
(include/go2sim_detmath.h, -ffp-contract=off) the tests demand BIT-EXACT equality of everything (tolerance 0);
the only exception are the double-precision reset statistics (atomic accumulation order), compared to 1e-6."""
import numpy as np
import pytest

from util import CpuEnv, GpuEnv, F, bits_equal, gs_on_oracle, make_actions

pytestmark = pytest.mark.gpu

FIELDS = ["F_QPOS", "F_VEL", "F_ACC", "F_QACC_WS", "F_MASS_MAT", "F_FORCE", "F_ACC_SMOOTH", "I_N_BROAD", "I_N_CONTACTS", "I_CONTACT_GEOMS",
          "F_CONTACT_POS", "F_CONTACT_NORMAL", "F_CONTACT_PEN", "F_NORMAL_CACHE", "I_N_CONSTRAINTS", "I_SOLVER_ITERS", "F_EFC_FORCE",
          "F_QFRC_CONSTRAINT", "F_CONTACT_FORCE", "F_LINK_POS", "F_LINK_QUAT", "F_LINK_CDVEL", "F_LINK_CDANG", "F_SORT_VALUE", "I_SORT_IG",
          "I_ERRNO", "I_IS_WARMSTART", "F_MASS_SHIFT", "F_COM_SHIFT", "F_GEOM_FRICTION"]


def _compare_fields(cpu, gpu, tag):
    for fn in FIELDS:
        assert bits_equal(cpu.field(fn), gpu.field(fn)), f"{tag}: field {fn} differs"


def _compare_globals(cpu, gpu):
    a, b = cpu.sim.env_globals().as_dict(), gpu.sim.env_globals().as_dict()
    for k in a:
        if k in ("ep_acc", "n_reset_now"):
            continue
        if k in ("last_episode_rew", "curr_tracking_sum", "curr_timeout_total", "tracking_ema", "timeout_rate_ema", "fall_rate_ema"):
            assert np.allclose(a[k], b[k], rtol=1e-6, atol=1e-7), k  # double accumulation order (atomics) differs
        else:
            assert a[k] == b[k], (k, a[k], b[k])


@pytest.mark.parametrize("n_envs,steps,kind,seed", [(64, 150, "mixed", 7), (4, 80, "0.5", 1), (1, 40, "zeros", 3), (130, 60, "2.0", 11)])
def test_env_step_bit_exact(oracle_lib, hip_lib, blob, n_envs, steps, kind, seed):
    """Go2Env.step parity incl. resets, pushes, noise, DR; ragged batch sizes (1, 130) exercise the tail wavefront."""
    cpu, gpu = CpuEnv(oracle_lib, blob, n_envs, seed=seed), GpuEnv(hip_lib, blob, n_envs, seed=seed)
    cpu.reset(); gpu.reset()
    _compare_fields(cpu, gpu, "after reset")
    acts = make_actions(steps, n_envs, seed=seed, kind=kind)
    n_resets = 0
    for s, a in enumerate(acts):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg), f"done mask differs at step {s}"
        assert np.array_equal(cpu.field("I_N_CONTACTS"), gpu.field("I_N_CONTACTS")), f"contact counts differ at step {s}"
        assert bits_equal(oc, og) and bits_equal(pc, pg), f"observations differ at step {s}"
        assert bits_equal(rc, rg) and bits_equal(tc, tg), f"reward / time_out differ at step {s}"
        assert bits_equal(cpu.env_buf("REW_TERMS", 32), gpu.env_buf("REW_TERMS", 32)), f"per-term rewards differ at step {s}"
        n_resets += int(dc.sum())
        if s % 25 == 0:
            _compare_fields(cpu, gpu, f"step {s}")
    _compare_fields(cpu, gpu, "final")
    _compare_globals(cpu, gpu)
    assert gpu.sim.check_errno() == cpu.sim.check_errno() == 0
    if kind in ("mixed", "2.0"):
        assert n_resets > 0, "the action set was meant to provoke falls / resets"


def test_live_curriculum_long_run_bit_exact(oracle_lib, hip_lib, blob):
    """600 steps with the curriculum state machine updating every few episodes (level moves in both directions, DR ranges and command
    ranges follow it, pushes start once the level is up): the device-side CurriculumManager / global DR (run by the last workgroup of the
    post-physics kernel) against the oracle, bit for bit, including the level trajectory."""
    def mutate(env_cfg, *_):
        env_cfg["episode_length_s"] = 1.0                         # 50-step episodes: calm phases end in time-outs, violent ones in falls
        env_cfg["curriculum"].update({"update_every_episodes": 24, "ready_streak": 1, "hard_streak": 1, "cooldown_updates": 1, "step_up": 0.05,
                                      "step_down": 0.04, "ready_timeout_rate": 0.5, "ready_tracking": -1.0, "ready_fall_rate": 0.5,
                                      "hard_fall_rate": 0.6, "ema_alpha": 0.5, "global_dr_update_interval": 16})

    n_envs, steps = 96, 600
    cpu, gpu = CpuEnv(oracle_lib, blob, n_envs, seed=21, mutate=mutate), GpuEnv(hip_lib, blob, n_envs, seed=21, mutate=mutate)
    for e in (cpu, gpu):
        e.sim.env_set_level(0.3)
        e.reset()
    rng = np.random.default_rng(5)
    levels = []
    for s in range(steps):
        scale = 3.0 if (s // 150) % 2 else 0.02                   # calm and violent phases: the level goes up, down, up, down
        a = (scale * rng.standard_normal((n_envs, 16))).astype(np.float32)
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg) and bits_equal(oc, og) and bits_equal(pc, pg) and bits_equal(rc, rg), f"step {s}"
        if s % 20 == 0:
            gc_, gg_ = cpu.sim.env_globals(), gpu.sim.env_globals()
            assert gc_.level == gg_.level and gc_.push_enable == gg_.push_enable and gc_.friction == gg_.friction, f"globals differ at step {s}"
            levels.append(gc_.level)
    _compare_fields(cpu, gpu, "final")
    _compare_globals(cpu, gpu)
    assert max(levels) > 0.4 and min(levels) < 0.05, "the level was meant to rise and fall"


def test_graph_and_plain_launch_paths_agree(hip_lib, blob, monkeypatch):
    """go2sim_env_step launches its kernel sequence as one hipGraph (default) or kernel by kernel (GO2SIM_NO_GRAPH=1, timing mode): same bits.
    Also covers a change of the output buffers between steps (the graph is rebuilt) and of the action buffer (a node parameter)."""
    import torch

    n_envs, steps = 70, 60
    g = GpuEnv(hip_lib, blob, n_envs, seed=4)
    monkeypatch.setenv("GO2SIM_NO_GRAPH", "1")
    p = GpuEnv(hip_lib, blob, n_envs, seed=4)
    monkeypatch.delenv("GO2SIM_NO_GRAPH")
    g.reset(); p.reset()
    acts = make_actions(steps, n_envs, seed=9, kind="mixed")
    for s_, a in enumerate(acts):
        if s_ == 20:                                               # new output tensors -> new graph
            g.obs = torch.zeros_like(g.obs); g.priv = torch.zeros_like(g.priv)
        og, pg, rg, dg, tg = g.step(a)
        op, pp, rp, dp, tp = p.step(a)
        assert bits_equal(og, op) and bits_equal(pg, pp) and bits_equal(rg, rp) and np.array_equal(dg, dp) and bits_equal(tg, tp), f"step {s_}"
    _compare_fields(g, p, "final")


def test_scene_step_bit_exact_with_uploaded_state(oracle_lib, hip_lib, blob):
    """gs.Scene.step parity: random (qpos, vel, ctrl) uploaded through set_field on both sides."""
    B = 96
    cpu, gpu = CpuEnv(oracle_lib, blob, B, seed=2), GpuEnv(hip_lib, blob, B, seed=2)
    rng = np.random.default_rng(5)
    q = cpu.field("F_QPOS")
    q[2] = rng.uniform(0.25, 0.5, B); quat = rng.standard_normal((4, B)) * 0.15 + np.array([[1], [0], [0], [0]])
    q[3:7] = quat / np.linalg.norm(quat, axis=0)
    q[7:19] = np.array([0, 0, 0, 0, 0.8, 0.8, 1.0, 1.0, -1.5, -1.5, -1.5, -1.5])[:, None] + 0.2 * rng.standard_normal((12, B))
    v = rng.standard_normal((18, B)).astype(np.float32)
    ctrl = np.zeros((18, B), np.float32); ctrl[6:] = 8.0 * rng.standard_normal((12, B))
    for e in (cpu, gpu):
        setter = e.sim.set_field_np if e is cpu else None
        for name, arr in (("F_QPOS", q.astype(np.float32)), ("F_VEL", v), ("F_CTRL_FORCE", ctrl)):
            if e is cpu:
                e.sim.set_field_np(F(name), arr)
            else:
                e.set_field(name, arr)
        e.sim.reset_caches()
        e.sim.forward_kinematics()
    for s in range(30):
        cpu.sim.scene_step(2); gpu.sim.scene_step(2)
        _compare_fields(cpu, gpu, f"scene step {s}")
    assert (cpu.field("I_N_CONTACTS") > 0).any()


def test_full_size_short_run_bit_exact(oracle_lib, hip_lib, blob):
    """BASELINE configs[1] size (4096 envs): a short run compared bit for bit."""
    B = 4096
    cpu, gpu = CpuEnv(oracle_lib, blob, B, seed=21), GpuEnv(hip_lib, blob, B, seed=21)
    cpu.reset(); gpu.reset()
    acts = make_actions(12, B, seed=4, kind="0.5")
    for s, a in enumerate(acts):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg) and bits_equal(oc, og) and bits_equal(rc, rg), f"step {s}"
    assert np.array_equal(cpu.field("I_N_CONTACTS"), gpu.field("I_N_CONTACTS"))
    assert bits_equal(cpu.field("F_QPOS"), gpu.field("F_QPOS"))


def test_full_size_invariants(hip_lib, blob):
    """Size-independent properties at 4096 envs over a longer horizon: finite values, unit quaternions, contact padding,
    weight carried by the contacts, idempotent kinematics refresh, determinism of a re-run."""
    B, steps = 4096, 150
    finals = []
    for rep in range(2):
        gpu = GpuEnv(hip_lib, blob, B, seed=31, freeze_curriculum=True)
        gpu.reset()
        act = np.zeros((B, 16), np.float32)
        for s in range(steps):
            obs, priv, rew, rst, to = gpu.step(act)
        assert np.isfinite(obs).all() and np.isfinite(rew).all() and gpu.sim.check_errno() == 0
        q = gpu.field("F_QPOS")
        assert np.allclose(np.linalg.norm(q[3:7], axis=0), 1.0, atol=1e-5)
        nc = gpu.field("I_N_CONTACTS")[0]
        pen = gpu.field("F_CONTACT_PEN")
        assert (nc >= 0).all() and (nc <= 150).all()
        mask = np.arange(150)[:, None] >= nc[None, :]
        assert (pen[mask] == 0).all()
        cf = gpu.field("F_CONTACT_FORCE").reshape(14, 3, B)
        standing = (rst == 0) & (nc >= 4)
        g = gpu.sim.env_globals()
        weight = (15.019 + g.mass_shift + sum(g.leg_mass_shift)) * 9.81
        assert standing.mean() > 0.9
        assert abs(np.median(cf[1:, 2, standing].sum(0)) - weight) / weight < 0.05
        before = {f: gpu.field(f) for f in ("F_LINK_POS", "F_LINK_QUAT", "F_LINK_CDVEL")}
        gpu.sim.forward_kinematics()
        for f, v in before.items():
            assert bits_equal(v, gpu.field(f)), "forward_kinematics must be idempotent"
        finals.append((obs.copy(), q.copy()))
    assert bits_equal(finals[0][0], finals[1][0]) and bits_equal(finals[0][1], finals[1][1])


def test_env_get_and_setters(hip_lib, oracle_lib, blob):
    B = 70
    cpu, gpu = CpuEnv(oracle_lib, blob, B, seed=5), GpuEnv(hip_lib, blob, B, seed=5)
    cpu.reset(); gpu.reset()
    ep = np.arange(B, dtype=np.int32) * 10
    cmd = np.random.default_rng(1).uniform(-1, 1, (B, 3)).astype(np.float32)
    cpu.sim.env_set_episode_length(ep); cpu.sim.env_set_commands(cmd)
    gpu.sim.env_set_episode_length(gpu.torch.from_numpy(ep).to(gpu.dev)); gpu.sim.env_set_commands(gpu.torch.from_numpy(cmd).to(gpu.dev))
    a = make_actions(3, B, seed=2, kind="0.3")
    for x in a:
        cpu.step(x); gpu.step(x)
    for name, k, dt in (("COMMANDS", 3, np.float32), ("EPISODE_LENGTH", 1, np.int32), ("BASE_EULER", 3, np.float32), ("DOF_POS", 12, np.float32),
                        ("FOOT_CONTACT", 4, np.int32), ("FEET_AIR_TIME", 4, np.float32), ("EPISODE_SUMS", 32, np.float32), ("TORQUE", 12, np.float32)):
        assert bits_equal(cpu.env_buf(name, k, dt), gpu.env_buf(name, k, dt)), name
    assert np.array_equal(gpu.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0], ep + 3)


def test_graft_entry_smoke():
    import __graft_entry__

    __graft_entry__.smoke()


_STRICT_RUN = pytest.mark.skipif(__import__("os").environ.get("GO2SIM_TEST_STRICT") == "1", reason="loads the product library itself: not part of a GO2SIM_TEST_HIP_LIB / GO2SIM_TEST_STRICT run")


@_STRICT_RUN
def test_go2env_class_matches_c_abi(hip_lib, blob):
    """The reference-shaped Go2Env class (go2_env.py) is a zero-arithmetic wrapper: same numbers as the raw C-ABI run."""
    import torch

    from go2_sim2real_locomotion_rl_amd import Go2Env, get_walk_cfgs, init

    B = 48
    init(seed=9)
    env = Go2Env(B, *get_walk_cfgs())
    raw = GpuEnv(hip_lib, blob, B, seed=9)
    raw.reset()
    assert env.num_envs == B and env.num_obs == 49 and env.num_privileged_obs == 104 and env.num_actions == 16 and env.max_episode_length == 1000
    obs0, extras0 = env.get_observations()
    assert obs0.shape == (B, 49) and extras0["observations"]["critic"].shape == (B, 104)
    acts = make_actions(25, B, seed=3, kind="mixed")
    for a in acts:
        obs, rew, reset, extras = env.step(torch.from_numpy(a).to(env.device))
        o, p, r, d, t = raw.step(a)
        assert bits_equal(obs.cpu().numpy(), o) and bits_equal(rew.cpu().numpy(), r) and np.array_equal(reset.cpu().numpy(), d)
        assert bits_equal(extras["observations"]["critic"].cpu().numpy(), p) and bits_equal(extras["time_outs"].cpu().numpy(), t)
    assert set(extras["episode"]) == {"rew_" + n for n in env.reward_scales}
    assert np.array_equal(env.episode_length_buf.cpu().numpy(), raw.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0])
    assert bits_equal(env.dof_pos.cpu().numpy(), raw.env_buf("DOF_POS", 12))
    env.episode_length_buf = torch.randint(0, 1000, (B,), device=env.device)
    assert env.check_errno() == 0
    with pytest.raises(Exception):
        env.step(torch.zeros(B, 12, device=env.device))


@pytest.mark.parametrize("task,n_envs,steps,kind,seed", [("crouch", 70, 120, "0.5", 3), ("jump", 33, 160, "mixed", 8), ("jump_dr", 40, 140, "0.5", 9), ("crouch_dr", 48, 100, "mixed", 10),
                                                           ("jump_dr", 4096, 10, "0.5", 12), ("crouch_dr", 4096, 10, "0.5", 13)])
def test_base_env_bit_exact(oracle_lib, hip_lib, blob, task, n_envs, steps, kind, seed):
    """go2_env_base.py (crouch / jump): engine PD, reset before reward, 45 observations -- GPU vs oracle, tolerance 0."""
    cpu, gpu = CpuEnv(oracle_lib, blob, n_envs, seed=seed, task=task), GpuEnv(hip_lib, blob, n_envs, seed=seed, task=task)
    cpu.reset(); gpu.reset()
    acts = make_actions(steps, n_envs, seed=seed, kind=kind, n_act=12)
    n_resets = 0
    for s, a in enumerate(acts):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg), f"done mask differs at step {s}"
        assert bits_equal(oc, og) and bits_equal(rc, rg) and bits_equal(tc, tg), f"obs / reward differ at step {s}"
        assert bits_equal(cpu.env_buf("REW_TERMS", 32), gpu.env_buf("REW_TERMS", 32)), f"per-term rewards differ at step {s}"
        n_resets += int(dc.sum())
    _compare_fields(cpu, gpu, "final")
    _compare_globals(cpu, gpu)
    assert n_resets > 0 or steps < 100


def test_go2env_class_base_family(hip_lib, blob):
    import torch

    from go2_sim2real_locomotion_rl_amd import Go2Env, init
    from go2_sim2real_locomotion_rl_amd.configs import get_jump_cfgs

    init(seed=2)
    env = Go2Env(16, *get_jump_cfgs())
    assert env.num_obs == 45 and env.num_actions == 12 and env.num_privileged_obs is None and env.max_episode_length == 150
    obs, rew, reset, extras = env.step(torch.zeros(16, 12, device=env.device))
    assert obs.shape == (16, 45) and extras["observations"]["critic"] is obs and env.get_privileged_observations() is None
    assert set(extras["episode"]) == {"rew_" + n for n in env.reward_scales}


@_STRICT_RUN
def test_gs_surface_on_hip_backend(oracle_lib):
    """The gs shim drives the HIP library by default; its accessor results equal the oracle-backed shim on the same script."""
    import torch

    import go2_sim2real_locomotion_rl_amd.genesis_shim as gs
    from go2_sim2real_locomotion_rl_amd.configs import get_jump_cfgs

    env_cfg = get_jump_cfgs()[0]
    res = []
    for backend_lib in (None, oracle_lib):
        if backend_lib is None:
            gs.init(backend=gs.gpu, precision="32", seed=4)
        else:
            gs_on_oracle(gs, backend_lib, seed=4)
        scene = gs.Scene(sim_options=gs.options.SimOptions(dt=0.02, substeps=2))
        scene.add_entity(gs.morphs.URDF(file="urdf/plane/plane.urdf", fixed=True))
        robot = scene.add_entity(gs.morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=env_cfg["base_init_pos"], quat=env_cfg["base_init_quat"]))
        scene.build(n_envs=9)
        motors = [robot.get_joint(n).dof_start for n in env_cfg["joint_names"]]
        robot.set_dofs_kp([60.0] * 12, motors); robot.set_dofs_kv([2.0] * 12, motors)
        default = torch.tensor([env_cfg["default_joint_angles"][n] for n in env_cfg["joint_names"]], device=gs.device)
        robot.set_dofs_position(default.repeat(9, 1), motors, zero_velocity=True)
        for _ in range(20):
            robot.control_dofs_position(default.repeat(9, 1), motors)
            scene.step()
        assert robot.get_pos().device.type == ("cuda" if backend_lib is None else "cpu")
        res.append([t.cpu().numpy() for t in (robot.get_pos(), robot.get_quat(), robot.get_dofs_position(motors), robot.get_links_net_contact_force())])
    for a, b in zip(*res):
        assert bits_equal(a, b)


def test_go2env_step_returns_fresh_observation_tensors_and_logs(hip_lib, blob):
    """rsl_rl's PPO keeps `transition.observations = obs` by reference across env.step (storage.add_transitions copies it afterwards), so a step must
    not overwrite the tensor it returned before: like the reference (`self.obs_buf = torch.cat(...)`, go2_env_walk.py:1084) every step hands out new
    observation tensors.  Also: the extras keys of the reference (go2_env_walk.py:674-686, 756, 813, 1228-1235)."""
    import torch

    from go2_sim2real_locomotion_rl_amd import Go2Env, init
    from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs

    init(seed=5)
    env = Go2Env(32, *get_walk_cfgs())
    g = torch.Generator(device="cpu").manual_seed(0)
    a = (0.3 * torch.randn(32, 16, generator=g)).to(env.device)
    o1, r1, d1, ex1 = env.step(a)
    c1 = ex1["observations"]["critic"]
    keep_o, keep_c = o1.clone(), c1.clone()
    o2, _, _, ex2 = env.step(a)
    torch.cuda.synchronize()
    assert o2.data_ptr() != o1.data_ptr() and ex2["observations"]["critic"].data_ptr() != c1.data_ptr()
    assert torch.equal(o1, keep_o) and torch.equal(c1, keep_c), "a later step overwrote an observation tensor it had returned"
    assert not torch.equal(o1, o2)
    assert set(ex2["episode"]) == {"rew_" + n for n in env.reward_scales}
    for _ in range(6):                                                     # curriculum / DR logs refresh at the errno-poll cadence (5 steps)
        _, _, _, ex = env.step(a)
    assert {"level", "timeout_rate_ema", "tracking_ema", "fall_rate_ema", "obs_noise_level_cur", "push_enable", "delay_max_cur", "cmd_ranges"} <= set(ex["curriculum"])
    assert {"friction", "mass_shift", "com_shift", "leg_mass_shift"} <= set(ex["domain_randomization"])
    assert float(ex["curriculum"]["level"]) == pytest.approx(0.10) and env.graph_status() == (True, 0)
    # reset_idx on a subset through the class
    env.reset_idx(torch.tensor([1, 5], device=env.device))
    assert int(env.episode_length_buf[1]) == 0 and int(env.episode_length_buf[5]) == 0 and int(env.episode_length_buf[0]) > 0


def test_go2env_errno_poll_raises_like_scene_step(hip_lib, blob):
    """simulator.py:267 polls errno every 10 substeps and scene.step raises (rigid_solver.py:1189-1213).  Go2Env.step enqueues the poll every 5 env
    steps and raises at the following step -- no env is silently frozen by the NaN guard."""
    import torch

    from go2_sim2real_locomotion_rl_amd import Go2Env, Go2SimError, init
    from go2_sim2real_locomotion_rl_amd.capi import C
    from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs

    init(seed=2)
    env = Go2Env(16, *get_walk_cfgs())
    a = torch.zeros(16, 16, device=env.device)
    for _ in range(7):
        env.step(a)                                                        # healthy: the polls come back clean
    vel = torch.zeros(18, 16, device=env.device)
    env._sim.get_field(C["GO2SIM_F_VEL"], vel)
    vel[3, 4] = float("nan")                                               # one env with a NaN velocity: its state commit is skipped, errno set
    env._sim.set_field(C["GO2SIM_F_VEL"], vel)
    with pytest.raises(Go2SimError, match="nan"):
        for _ in range(12):
            env.step(a)
    with pytest.raises(Go2SimError):
        env.check_errno()


def test_normal_cache_field_semantics(oracle_lib, hip_lib, blob):
    """The contact-normal cache (collider `contact_cache.normal`, narrowphase.py:689-760) is kept on the device as per-env records plus a valid-bit
    mask instead of explicit zero vectors.  Through the field API it must behave like the plain array of the reference: an uploaded array reads
    back unchanged (zeros included), continues the simulation exactly like the oracle given the same upload, and reset_caches zeroes it."""
    n = 32
    cpu, gpu = CpuEnv(oracle_lib, blob, n, seed=5), GpuEnv(hip_lib, blob, n, seed=5)
    cpu.reset(); gpu.reset()
    acts = make_actions(12, n, seed=5, kind="0.5")
    for a in acts[:6]:
        cpu.step(a); gpu.step(a)
    nc = cpu.field("F_NORMAL_CACHE")
    assert np.abs(nc).max() > 0 and bits_equal(nc, gpu.field("F_NORMAL_CACHE"))
    rng = np.random.default_rng(0)
    up = nc.copy()
    up[:, ::2] = 0.0                                             # wipe the guesses of every other env
    sel = rng.integers(0, up.shape[0] // 3, 20)
    for p in sel:                                                # and plant (non-unit) guesses for a few more pairs
        up[3 * p:3 * p + 3, 1::2] = rng.standard_normal((3, 1)).astype(np.float32)
    cpu.sim.set_field_np(F("F_NORMAL_CACHE"), up); gpu.set_field("F_NORMAL_CACHE", up)
    assert bits_equal(gpu.field("F_NORMAL_CACHE"), up)
    for s, a in enumerate(acts[6:]):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert bits_equal(oc, og) and bits_equal(rc, rg) and np.array_equal(dc, dg), s
        assert bits_equal(cpu.field("F_NORMAL_CACHE"), gpu.field("F_NORMAL_CACHE")), s
        assert bits_equal(cpu.field("F_CONTACT_PEN"), gpu.field("F_CONTACT_PEN")), s
    gpu.sim.reset_caches(None, 0); cpu.sim.reset_caches(None, 0)
    assert not gpu.field("F_NORMAL_CACHE").any() and not cpu.field("F_NORMAL_CACHE").any()


@pytest.mark.gpu
def test_mass_matrix_field_semantics(oracle_lib, hip_lib, blob):
    """The device record holds the mass matrix as a packed lower triangle (what k_dynamics hands to the solver); GO2SIM_F_MASS_MAT presents the
    reference's full symmetric [n_dofs, n_dofs] matrix (`rigid_solver.mass_mat`): equal to the oracle's, symmetric, and an uploaded matrix reads
    back unchanged."""
    n = 24
    cpu, gpu = CpuEnv(oracle_lib, blob, n, seed=3), GpuEnv(hip_lib, blob, n, seed=3)
    cpu.reset(); gpu.reset()
    for a in make_actions(5, n, seed=3, kind="0.5"):
        cpu.step(a); gpu.step(a)
    Mc, Mg = cpu.field("F_MASS_MAT"), gpu.field("F_MASS_MAT")
    assert Mg.shape == (18 * 18, n) and bits_equal(Mc, Mg)
    full = Mg.reshape(18, 18, n)
    assert np.array_equal(full, full.transpose(1, 0, 2)) and (np.abs(full[np.arange(18), np.arange(18)]) > 0).all()
    rng = np.random.default_rng(1)
    low = np.tril(rng.standard_normal((18, 18))).astype(np.float32)
    sym = (low + np.tril(low, -1).T)[:, :, None] * np.linspace(1.0, 2.0, n, dtype=np.float32)[None, None, :]
    gpu.set_field("F_MASS_MAT", sym.reshape(18 * 18, n).astype(np.float32))
    assert bits_equal(gpu.field("F_MASS_MAT"), sym.reshape(18 * 18, n).astype(np.float32))


def test_row_form_factorisation_bit_exact(oracle_lib, hip_lib, blob, monkeypatch):
    """GO2SIM_NO_ARROW=1 (read when a model is parsed, both libraries): the dense row-form factorisation and solves of the Newton Hessian, which the product
    takes only when a constraint row couples two legs, for every solve -- HIP against the fast oracle, bit for bit."""
    monkeypatch.setenv("GO2SIM_NO_ARROW", "1")
    n_envs, steps = 64, 60
    cpu, gpu = CpuEnv(oracle_lib, blob, n_envs, seed=5), GpuEnv(hip_lib, blob, n_envs, seed=5)
    monkeypatch.delenv("GO2SIM_NO_ARROW")
    cpu.reset(); gpu.reset()
    acts = make_actions(steps, n_envs, seed=5, kind="mixed")
    for s, a in enumerate(acts):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg) and bits_equal(oc, og) and bits_equal(pc, pg) and bits_equal(rc, rg), f"step {s}"
    _compare_fields(cpu, gpu, "final")
    ref = CpuEnv(oracle_lib, blob, n_envs, seed=5)                      # the arrow form gives other last bits: the switch did something
    ref.reset()
    for a in acts:
        o_ref = ref.step(a)[0]
    assert not np.array_equal(o_ref, oc)
