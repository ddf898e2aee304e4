"""Per-subset reset and the eval / teleop surface: go2sim_env_reset_idx (go2_env_walk.py:1156-1240),
go2sim_env_respawn (respawn_at_start go2_eval_stairs.py:314-361, respawn_on_tile go2_eval_walk.py:399-480), go2sim_env_lock_terrain_rows
(go2_env_stair.py:399,1513) and the cfgs.pkl layout -- on the CPU oracle with independent numpy expectations, and GPU-vs-oracle parity."""
import numpy as np
import pytest

from go2_sim2real_locomotion_rl_amd.capi import C
from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs
from util import CpuEnv, GpuEnv, bits_equal, make_actions


def _run(env, acts):
    for a in acts:
        env.step(a)


def test_reset_idx_resets_only_the_listed_envs(oracle_lib, blob):
    B = 24
    env = CpuEnv(oracle_lib, blob, B, seed=11, task="walk")
    env.reset()
    _run(env, make_actions(30, B, seed=1, kind="0.3"))
    ep0 = env.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0].copy()
    qpos0, sums0 = env.field("F_QPOS").copy(), env.env_buf("EPISODE_SUMS", 32).copy()
    calls0 = env.sim.env_globals().reset_calls
    idx = np.array([3, 7, 7, 20], np.int32)                                       # an index may repeat: counted once
    env.sim.env_reset_idx(idx, len(idx))
    g = env.sim.env_globals()
    assert g.reset_calls == calls0 + 1 and g.last_reset_count == 3 and g.n_reset_now == 3
    ep1 = env.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0]
    listed = np.zeros(B, bool); listed[[3, 7, 20]] = True
    assert (ep1[listed] == 0).all() and (ep1[~listed] == ep0[~listed]).all()
    qpos1, sums1 = env.field("F_QPOS"), env.env_buf("EPISODE_SUMS", 32)
    assert bits_equal(qpos1[:, ~listed], qpos0[:, ~listed]) and bits_equal(sums1[~listed], sums0[~listed])
    assert (sums1[listed] == 0).all()
    # reset pose: default joint angles, z in [0.38, 0.45], zero velocity, cleared warm start
    env_cfg = get_walk_cfgs()[0]
    assert ((qpos1[2, listed] >= 0.38 - 1e-6) & (qpos1[2, listed] <= 0.45 + 1e-6)).all()
    assert (env.field("F_VEL")[:, listed] == 0).all() and (env.field("I_IS_WARMSTART")[0, listed] == 0).all()
    dof = env.env_buf("DOF_POS", 12)
    want = np.array([env_cfg["default_joint_angles"][n] for n in env_cfg["joint_names"]], np.float32)
    assert np.allclose(dof[listed], want)
    # episode log of the call = mean per-second reward of the listed envs (go2_env_walk.py:1228-1235)
    names = env.reward_names
    per_sec = sums0[listed][:, :len(names)] / (np.maximum(ep0[listed], 1).astype(np.float32) * np.float32(0.02))[:, None]
    assert np.allclose(np.array(g.last_episode_rew[:len(names)]), per_sec.mean(0), rtol=1e-5, atol=1e-6)
    # n == 0 is a no-op (`if len(envs_idx) == 0: return`)
    env.sim.env_reset_idx(np.zeros(1, np.int32), 0)
    assert env.sim.env_globals().reset_calls == calls0 + 1
    # stepping continues normally afterwards
    _run(env, make_actions(5, B, seed=2, kind="0.3"))
    assert env.sim.check_errno() == 0


def test_respawn_moves_only_the_listed_envs(oracle_lib, blob):
    B = 8
    env = CpuEnv(oracle_lib, blob, B, seed=4, task="walk")
    env.reset()
    _run(env, make_actions(25, B, seed=3, kind="0.5"))
    ep0 = env.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0].copy()
    qpos0 = env.field("F_QPOS").copy()
    idx = np.array([5, 1], np.int32)
    pos = np.array([[1.5, -2.0, 0.5], [0.25, 0.75, 0.44]], np.float32)
    quat = np.array([[1, 0, 0, 0], [0.9238795, 0, 0, 0.3826834]], np.float32)
    env.sim.env_respawn(idx, pos, quat, True, len(idx))
    qpos1 = env.field("F_QPOS")
    other = np.ones(B, bool); other[idx] = False
    assert bits_equal(qpos1[:, other], qpos0[:, other])
    assert np.array_equal(qpos1[:3, 5], pos[0]) and np.array_equal(qpos1[:3, 1], pos[1]) and np.array_equal(qpos1[3:7, 1], quat[1])
    assert (env.field("F_VEL")[:, idx] == 0).all()
    assert np.array_equal(env.env_buf("EPISODE_LENGTH", 1, np.int32)[:, 0], ep0)           # episode counters untouched (eval teleport, not a reset)
    lp = env.field("F_LINK_POS")                                                          # the kinematics were refreshed: base link at the new place
    assert np.allclose(lp[3:6, 5], pos[0], atol=1e-6)
    # default quaternion when none is given
    env.sim.env_respawn(np.array([2], np.int32), np.array([[0, 0, 0.42]], np.float32), None, False, 1)
    assert np.array_equal(env.field("F_QPOS")[3:7, 2], np.array(get_walk_cfgs()[0]["base_init_quat"], np.float32))
    _run(env, make_actions(5, B, seed=5, kind="0.3"))
    assert env.sim.check_errno() == 0


def test_lock_terrain_rows_keeps_rows_across_resets(oracle_lib, blob):
    B = 40
    env = CpuEnv(oracle_lib, blob, B, seed=9, task="stairs")
    env.reset()
    rows = np.full(B, 3, np.int32); rows[::2] = 6
    env.sim.env_set_terrain_rows(rows)
    env.sim.env_lock_terrain_rows(True)
    assert env.sim.env_globals().lock_terrain_rows == 1
    env.sim.env_reset_idx(np.arange(B, dtype=np.int32), B)
    assert np.array_equal(env.env_buf("TERRAIN_ROW", 1, np.int32)[:, 0], rows)
    bp = env.env_buf("BASE_POS", 3)
    from go2_sim2real_locomotion_rl_amd.configs import build_stair_terrain, get_stair_cfgs

    centers = np.asarray(build_stair_terrain(get_stair_cfgs()[0]["terrain"])[1]["row_centers"], np.float32)
    assert np.allclose(bp[:, :2], centers[rows, :2])                                      # spawned on the locked rows
    assert env.sim.env_globals().terrain_mean_row == pytest.approx(rows.mean(), abs=1e-5)
    env.sim.env_lock_terrain_rows(False)
    env.sim.env_reset_idx(np.arange(B, dtype=np.int32), B)
    assert not np.array_equal(env.env_buf("TERRAIN_ROW", 1, np.int32)[:, 0], rows)          # unlocked: rows are re-assigned


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["walk", "stairs"])
def test_reset_idx_respawn_lock_gpu_bit_exact(oracle_lib, hip_lib, blob, task):
    """Locked-row respawn + subset resets interleaved with steps: GPU == oracle bit for bit."""
    import torch

    B = 70
    cpu = CpuEnv(oracle_lib, blob, B, seed=6, task=task); gpu = GpuEnv(hip_lib, blob, B, seed=6, task=task)
    cpu.reset(); gpu.reset()
    acts = make_actions(45, B, seed=8, kind="mixed")
    dev = gpu.dev
    rng = np.random.default_rng(3)
    for s, a in enumerate(acts):
        if s == 10 and task == "stairs":
            rows = rng.integers(0, 13, B).astype(np.int32)
            cpu.sim.env_set_terrain_rows(rows); gpu.sim.env_set_terrain_rows(torch.from_numpy(rows).to(dev))
            cpu.sim.env_lock_terrain_rows(True); gpu.sim.env_lock_terrain_rows(True)
        if s in (12, 20, 33):
            idx = np.unique(rng.integers(0, B, 9)).astype(np.int32)
            cpu.sim.env_reset_idx(idx, len(idx)); gpu.sim.env_reset_idx(torch.from_numpy(idx).to(dev), len(idx))
        if s == 26:
            idx = np.array([0, 69, 13], np.int32)
            pos = np.array([[0.0, 0.0, 0.5], [2.0, 1.0, 0.6], [4.0, -1.0, 0.7]], np.float32)
            cpu.sim.env_respawn(idx, pos, None, True, 3)
            gpu.sim.env_respawn(torch.from_numpy(idx).to(dev), torch.from_numpy(pos).to(dev), None, True, 3)
        oc = cpu.step(a); og = gpu.step(a)
        for x, y, nm in zip(oc, og, ("obs", "priv", "rew", "reset", "timeout")):
            assert bits_equal(x, y), f"{task} step {s}: {nm} differs"
    for f in ("F_QPOS", "F_VEL", "F_CONTACT_FORCE", "I_N_CONTACTS", "I_N_CONSTRAINTS"):
        assert bits_equal(cpu.field(f), gpu.field(f)), f
    gc, gg = cpu.sim.env_globals(), gpu.sim.env_globals()
    assert gc.reset_calls == gg.reset_calls and gc.last_reset_count == gg.last_reset_count and gc.lock_terrain_rows == gg.lock_terrain_rows
    assert np.allclose(np.array(gc.last_episode_rew), np.array(gg.last_episode_rew), rtol=1e-6, atol=1e-7)
    assert bits_equal(cpu.env_buf("TERRAIN_ROW", 1, np.int32), gpu.env_buf("TERRAIN_ROW", 1, np.int32))
    assert C["GO2SIM_E_OK"] == 0


def test_cfgs_pkl_layout_round_trip(tmp_path):
    """cfgs.pkl = [env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg] (go2_train_walk.py:462-465): written and read back without executing
    anything from the file; a pickle that needs a global (i.e. could run code) is refused."""
    import pickle

    from go2_sim2real_locomotion_rl_amd.eval_io import load_cfgs, save_cfgs

    env_cfg, obs_cfg, reward_cfg, command_cfg = get_walk_cfgs()
    train_cfg = {"algorithm": {"class_name": "PPO", "gamma": 0.99}, "num_steps_per_env": 24, "policy": {"actor_hidden_dims": [512, 256, 128]}}
    path = tmp_path / "cfgs.pkl"
    save_cfgs(path, env_cfg, obs_cfg, reward_cfg, command_cfg, train_cfg)
    got = load_cfgs(path)
    assert len(got) == 5 and got[0] == env_cfg and got[2]["reward_scales"] == reward_cfg["reward_scales"] and got[4] == train_cfg
    assert list(got[2]["reward_scales"]) == list(reward_cfg["reward_scales"])         # insertion order = evaluation order of the reward terms
    assert pickle.load(open(path, "rb"))[3] == command_cfg                            # the reference's own `pickle.load` reads it too
    bad = tmp_path / "bad.pkl"
    pickle.dump([np.float32(1.0)] * 5, open(bad, "wb"))                               # numpy scalars pickle through a global
    with pytest.raises(pickle.UnpicklingError):
        load_cfgs(bad)


def test_rsl_rl_checkpoint_layout_and_compatibility_load(tmp_path):
    """model_<it>.pt in rsl_rl 2.2.4's layout, read with weights_only=True; the partial load of go2_eval_stairs.py:368-450 takes the actor and
    skips a critic trained with another privileged-observation width."""
    import torch

    from go2_sim2real_locomotion_rl_amd.eval_io import compatible_state_dict, critic_input_mismatch, read_checkpoint, save_checkpoint

    def sd(n_priv, seed):
        g = torch.Generator().manual_seed(seed)
        out = {}
        for prefix, dims in (("actor", [49, 512, 256, 128, 16]), ("critic", [n_priv, 512, 256, 128, 1])):
            for l in range(4):
                out[f"{prefix}.{2 * l}.weight"] = torch.randn(dims[l + 1], dims[l], generator=g)
                out[f"{prefix}.{2 * l}.bias"] = torch.randn(dims[l + 1], generator=g)
        out["std"] = torch.ones(16)
        return out

    walk, stairs = sd(104, 1), sd(182, 2)
    path = tmp_path / "model_300.pt"
    save_checkpoint(path, walk, {"state": {}, "param_groups": []}, it=300, infos=None)
    ck = read_checkpoint(path)
    assert ck["iter"] == 300 and set(ck) >= {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    assert all(torch.equal(ck["model_state_dict"][k], walk[k]) for k in walk)
    assert critic_input_mismatch(ck["model_state_dict"], 182) == 78 and critic_input_mismatch(ck["model_state_dict"], 104) == 0
    merged, loaded, skipped = compatible_state_dict(stairs, ck["model_state_dict"])
    assert list(skipped) == ["critic.0.weight"] and "saved=[512, 104]" in skipped["critic.0.weight"]
    assert torch.equal(merged["actor.0.weight"], walk["actor.0.weight"]) and torch.equal(merged["critic.0.weight"], stairs["critic.0.weight"])
    assert torch.equal(merged["critic.2.weight"], walk["critic.2.weight"]) and len(loaded) == len(walk) - 1
    bare = tmp_path / "bare.pt"
    torch.save(walk, bare)
    assert read_checkpoint(bare)["model_state_dict"].keys() == walk.keys()
