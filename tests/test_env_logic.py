"""Go2Env (walk) semantics on the CPU oracle: shapes, reward terms recomputed independently in numpy from the
env buffers, termination / time-out / reset behaviour (go2_env_walk.py:985-1240)."""
import numpy as np
import pytest

from go2_sim2real_locomotion_rl_amd.capi import C
from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs
from util import CpuEnv, F, make_actions


def test_shapes_and_first_reset(oracle_lib, blob):
    env = CpuEnv(oracle_lib, blob, 4, seed=1)
    env.reset()
    g = env.sim.env_globals()
    assert g.reset_calls == 1 and g.last_reset_count == 4
    ep = env.env_buf("EPISODE_LENGTH", 1, np.int32)
    assert (ep == 0).all()
    q = env.field("F_QPOS")
    assert ((q[2] >= 0.38) & (q[2] <= 0.45)).all()                     # init_pos_z_range
    assert np.allclose(q[7:11], 0.0) and np.allclose(q[11:13], 0.8) and np.allclose(q[13:15], 1.0) and np.allclose(q[15:19], -1.5)
    cmd = env.env_buf("COMMANDS", 3)
    assert (np.abs(cmd[:, 0]) <= 0.19 + 1e-6).all()                    # cmd curriculum at level 0.10: 10% + 90%*0.1 of the span
    obs, priv, rew, rst, to = env.step(np.zeros((4, 16), np.float32))
    assert obs.shape == (4, 49) and priv.shape == (4, 104) and rew.shape == (4,)
    assert np.array_equal(priv[:, :49], obs)
    assert (env.env_buf("EPISODE_LENGTH", 1, np.int32) == 1).all()


def test_reward_terms_recomputed_in_numpy(oracle_lib, blob):
    env_cfg, obs_cfg, reward_cfg, command_cfg = get_walk_cfgs()
    env = CpuEnv(oracle_lib, blob, 6, seed=4)
    env.reset()
    acts = make_actions(40, 6, seed=7, kind="0.4")
    prev_actions = np.zeros((6, 16), np.float32); prev_dof_vel = np.zeros((6, 12), np.float32)
    default = np.array([env_cfg["default_joint_angles"][n] for n in env_cfg["joint_names"]], np.float32)
    names = env.reward_names
    for s, a in enumerate(acts):
        obs, priv, rew, rst, to = env.step(a)
        terms = env.env_buf("REW_TERMS", 32)[:, :len(names)]
        assert np.allclose(terms.sum(1), rew, atol=1e-6)
        lin, ang = env.env_buf("BASE_LIN_VEL", 3), env.env_buf("BASE_ANG_VEL", 3)
        cmd, dof_pos, dof_vel = env.env_buf("COMMANDS", 3), env.env_buf("DOF_POS", 12), env.env_buf("DOF_VEL", 12)
        pg, base_pos = env.env_buf("PROJECTED_GRAVITY", 3), env.env_buf("BASE_POS", 3)
        keep = rst == 0  # reset envs had their buffers overwritten after the reward was computed
        sc = {n: reward_cfg["reward_scales"][n] * 0.02 for n in names}
        exp = {
            "tracking_lin_vel": np.exp(-((cmd[:, :2] - lin[:, :2]) ** 2).sum(1) / 0.25),
            "tracking_ang_vel": np.exp(-((cmd[:, 2] - ang[:, 2]) ** 2) / 0.25),
            "lin_vel_z": lin[:, 2] ** 2,
            "base_height": (base_pos[:, 2] - 0.3) ** 2,
            "action_rate": ((prev_actions - a) ** 2).sum(1),
            "similar_to_default": np.abs(dof_pos - default).sum(1),
            "orientation_penalty": (pg[:, :2] ** 2).sum(1),
            "dof_acc": (((dof_vel - prev_dof_vel) / 0.02) ** 2).sum(1),
            "dof_vel": (dof_vel ** 2).sum(1),
            "ang_vel_xy": (ang[:, :2] ** 2).sum(1),
        }
        for n, v in exp.items():
            k = names.index(n)
            assert np.allclose(terms[keep, k], (v * sc[n])[keep], rtol=2e-4, atol=1e-7), (s, n)
        prev_actions = a.copy(); prev_dof_vel = dof_vel.copy()
        prev_actions[rst != 0] = 0; prev_dof_vel[rst != 0] = 0


def test_termination_on_roll_and_fresh_state_after_reset(oracle_lib, blob):
    env = CpuEnv(oracle_lib, blob, 3, seed=6)
    env.reset()
    env.step(np.zeros((3, 16), np.float32))
    q = env.field("F_QPOS")
    q[3:7, 1] = np.array([np.cos(np.pi / 3), np.sin(np.pi / 3), 0, 0], np.float32)  # env 1 rolled by 120 deg
    env.sim.set_field_np(F("F_QPOS"), q)
    env.sim.forward_kinematics()
    obs, priv, rew, rst, to = env.step(np.zeros((3, 16), np.float32))
    assert rst.tolist() == [0, 1, 0] and to.tolist() == [0, 0, 0]
    ep = env.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0]
    assert ep.tolist() == [2, 0, 2]
    assert np.allclose(obs[1, 9:33], 0.0, atol=0.05)   # dof_pos - default and dof_vel of the reset env are zero (+ obs noise)
    assert np.array_equal(obs[1, 33:], np.zeros(16, np.float32))
    assert env.sim.env_globals().reset_calls == 2


def test_time_out_uses_strict_greater(oracle_lib, blob):
    """episode_length_buf > max_episode_length (go2_env_walk.py:1062): episodes last max_episode_length + 1 steps."""
    env = CpuEnv(oracle_lib, blob, 2, seed=3)
    env.reset()
    env.sim.env_set_episode_length(np.array([998, 999], np.int32))
    a = np.zeros((2, 16), np.float32)
    _, _, _, rst, to = env.step(a)     # lengths 999, 1000
    assert rst.tolist() == [0, 0]
    _, _, _, rst, to = env.step(a)     # lengths 1000, 1001
    assert rst.tolist() == [0, 1] and to.tolist() == [0.0, 1.0]


def test_action_latency_ring(oracle_lib, blob):
    """Per-env delay read from the 2-deep action ring; the observation carries the APPLIED action (:1091)."""
    env = CpuEnv(oracle_lib, blob, 64, seed=8)
    env.sim.env_set_level(1.0)  # delay_max_cur = 1 at level 1 (delay_easy_max_steps=0 .. max_delay_steps=1)
    env.reset()
    a0 = np.full((64, 16), 0.25, np.float32); a1 = np.full((64, 16), -0.5, np.float32)
    o0, *_ = env.step(a0)
    o1, _, _, rst, _ = env.step(a1)
    applied = o1[:, 33:]
    assert rst.sum() == 0
    delayed = np.isclose(applied[:, 0], 0.25); undelayed = np.isclose(applied[:, 0], -0.5)
    assert (delayed | undelayed).all() and delayed.any() and undelayed.any()


def test_global_dr_and_privileged_layout(oracle_lib, blob):
    env = CpuEnv(oracle_lib, blob, 4, seed=10)
    env.reset()
    obs, priv, *_ = env.step(np.zeros((4, 16), np.float32))
    g = env.sim.env_globals()
    i = 49 + 3
    assert np.allclose(priv[:, i], g.friction)
    assert np.allclose(priv[:, i + 37], g.mass_shift)                  # 1 + 12 + 12 + 12
    assert np.allclose(priv[:, i + 38:i + 41], np.array(list(g.com_shift)))
    assert np.allclose(priv[:, i + 41:i + 45], np.array(list(g.leg_mass_shift)))
    ms = env.field("F_MASS_SHIFT")
    assert np.allclose(ms[1], g.mass_shift) and np.allclose(ms[[3, 2, 5, 4]], np.array(list(g.leg_mass_shift))[:, None])
    kpf = priv[:, i + 1:i + 13]
    assert ((kpf >= 0.8) & (kpf <= 1.2)).all()


def test_curriculum_state_machine(oracle_lib, blob):
    """CurriculumManager.update (go2_env_walk.py:101-142): with every episode ending in a fall the level drops."""
    def mutate(env_cfg, *_):
        env_cfg["curriculum"]["update_every_episodes"] = 64

    env = CpuEnv(oracle_lib, blob, 64, seed=12, mutate=mutate)
    env.sim.env_set_level(0.5)
    env.reset()
    rng = np.random.default_rng(0)
    level0 = env.sim.env_globals().level
    for s in range(300):
        env.step((4.0 * rng.standard_normal((64, 16))).astype(np.float32))
    g = env.sim.env_globals()
    assert g.fall_rate_ema > 0.5 and g.level < level0
