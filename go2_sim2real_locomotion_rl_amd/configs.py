"""Configuration dictionaries of the reference training scripts, transcribed as data, and their
flattening into the go2sim env-config arrays (include/go2sim.h enums go2sim_fcfg / go2sim_icfg).

get_walk_cfgs() mirrors examples/locomotion/final/go2_train_walk.py:68-372 (env/obs/reward/command
dicts consumed by Go2Env.__init__, go2_env_walk.py:155-525); the dict keys are the reference's own so a
user's customised cfg dicts drop in unchanged."""
import copy
import math

import numpy as np

from .capi import C
from .model_blob import load_model_json

REWARD_IDS = {
    "tracking_lin_vel": "GO2SIM_R_TRACKING_LIN_VEL", "tracking_ang_vel": "GO2SIM_R_TRACKING_ANG_VEL",
    "lin_vel_z": "GO2SIM_R_LIN_VEL_Z", "base_height": "GO2SIM_R_BASE_HEIGHT", "action_rate": "GO2SIM_R_ACTION_RATE",
    "similar_to_default": "GO2SIM_R_SIMILAR_TO_DEFAULT", "orientation_penalty": "GO2SIM_R_ORIENTATION_PENALTY",
    "dof_acc": "GO2SIM_R_DOF_ACC", "dof_vel": "GO2SIM_R_DOF_VEL", "ang_vel_xy": "GO2SIM_R_ANG_VEL_XY",
    "feet_air_time": "GO2SIM_R_FEET_AIR_TIME", "foot_slip": "GO2SIM_R_FOOT_SLIP", "foot_clearance": "GO2SIM_R_FOOT_CLEARANCE",
    "joint_tracking": "GO2SIM_R_JOINT_TRACKING", "energy": "GO2SIM_R_ENERGY", "torque_load": "GO2SIM_R_TORQUE_LOAD",
    "stand_still": "GO2SIM_R_STAND_STILL", "stand_still_vel": "GO2SIM_R_STAND_STILL_VEL", "feet_stance": "GO2SIM_R_FEET_STANCE",
    "orientation_roll_only": "GO2SIM_R_ORIENTATION_ROLL_ONLY", "forward_progress": "GO2SIM_R_FORWARD_PROGRESS",   # go2_env_stair.py
}


def get_walk_cfgs():
    """go2_train_walk.py:68-372 (values transcribed; comments there explain the choices)."""
    kp_nominal, kd_nominal = 60.0, 2.0
    curriculum_cfg = {
        "enabled": True, "level_init": 0.10, "level_min": 0.0, "level_max": 1.0, "ema_alpha": 0.03,
        "ready_timeout_rate": 0.80, "ready_tracking": 0.75, "ready_fall_rate": 0.15, "ready_streak": 4,
        "hard_fall_rate": 0.25, "hard_streak": 2, "step_up": 0.01, "step_down": 0.03, "cooldown_updates": 5,
        "update_every_episodes": 4096, "mix_prob_current": 0.80, "mix_level_low": 0.00, "mix_level_high": 0.50,
        "friction_easy": [0.6, 0.8], "kp_easy": [0.90 * kp_nominal, 1.10 * kp_nominal],
        "kd_easy": [0.75 * kd_nominal, 1.25 * kd_nominal], "kp_factor_easy": [0.95, 1.05], "kd_factor_easy": [0.95, 1.05],
        "mass_shift_easy": [-0.2, 0.5], "com_shift_easy": [-0.005, 0.005], "leg_mass_shift_easy": [-0.1, 0.1],
        "gravity_offset_easy": [-0.2, 0.2], "motor_strength_easy": [0.97, 1.03], "push_start": 0.0,
        "push_interval_easy_s": 10.0, "delay_easy_max_steps": 0, "global_dr_update_interval": 200,
    }
    env_cfg = {
        "num_actions": 16, "num_pos_actions": 12, "pls_enable": True, "pls_kp_range": [10.0, 70.0], "pls_kp_default": 40.0,
        "pls_kp_action_scale": 20.0, "kp": kp_nominal, "kd": kd_nominal, "torque_limits": [23.7, 23.7, 45.0] * 4,
        "simulate_action_latency": True, "foot_names": ["FR_calf", "FL_calf", "RR_calf", "RL_calf"],
        "foot_contact_threshold": 3.0,
        "default_joint_angles": {
            "FL_hip_joint": 0.0, "FR_hip_joint": 0.0, "RL_hip_joint": 0.0, "RR_hip_joint": 0.0,
            "FL_thigh_joint": 0.8, "FR_thigh_joint": 0.8, "RL_thigh_joint": 1.0, "RR_thigh_joint": 1.0,
            "FL_calf_joint": -1.5, "FR_calf_joint": -1.5, "RL_calf_joint": -1.5, "RR_calf_joint": -1.5,
        },
        "joint_names": ["FR_hip_joint", "FR_thigh_joint", "FR_calf_joint", "FL_hip_joint", "FL_thigh_joint", "FL_calf_joint",
                        "RR_hip_joint", "RR_thigh_joint", "RR_calf_joint", "RL_hip_joint", "RL_thigh_joint", "RL_calf_joint"],
        "termination_if_roll_greater_than": 45, "termination_if_pitch_greater_than": 45,
        "termination_if_z_vel_greater_than": 100.0, "termination_if_y_vel_greater_than": 100.0,
        "base_init_pos": [0.0, 0.0, 0.42], "base_init_quat": [1.0, 0.0, 0.0, 0.0], "episode_length_s": 20.0,
        "resampling_time_s": 5.0, "action_scale": 0.25, "clip_actions": 100.0, "curriculum": curriculum_cfg,
        "friction_range": [0.3, 1.25], "kp_factor_range": [0.8, 1.2], "kd_factor_range": [0.8, 1.2],
        "kp_range": [50.0, 70.0], "kd_range": [1.0, 5.0],
        "obs_noise": {"ang_vel": 0.2, "gravity": 0.05, "dof_pos": 0.01, "dof_vel": 1.5}, "obs_noise_level": 1.0,
        "action_noise_std": 0.1, "push_interval_s": 5.0, "push_force_range": [-150.0, 150.0], "push_duration_s": [0.05, 0.2],
        "init_pos_z_range": [0.38, 0.45], "init_euler_range": [-5.0, 5.0], "mass_shift_range": [-1.0, 3.0],
        "com_shift_range": [-0.03, 0.03], "leg_mass_shift_range": [-0.5, 0.5], "gravity_offset_range": [-1.0, 1.0],
        "motor_strength_range": [0.9, 1.1], "min_delay_steps": 0, "max_delay_steps": 1,
    }
    num_obs = 3 + 3 + 3 + 12 + 12 + 16
    obs_cfg = {"num_obs": num_obs, "num_privileged_obs": num_obs + 3 + 1 + 12 + 12 + 12 + 1 + 3 + 4 + 3 + 3 + 1,
               "obs_scales": {"lin_vel": 2.0, "ang_vel": 0.25, "dof_pos": 1.0, "dof_vel": 0.05}}
    reward_cfg = {
        "tracking_sigma": 0.25, "base_height_target": 0.3, "feet_height_target": 0.075, "feet_air_time_target": 0.1,
        "reward_scales": {
            "tracking_lin_vel": 1.5, "tracking_ang_vel": 0.8, "lin_vel_z": -2.0, "base_height": -0.6, "action_rate": -0.01,
            "similar_to_default": -0.1, "orientation_penalty": -5.0, "dof_acc": -2.5e-7, "dof_vel": -5e-4, "ang_vel_xy": -0.05,
            "feet_air_time": 0.2, "foot_slip": -0.1, "foot_clearance": -0.1, "joint_tracking": -0.1, "energy": 0.0,
            "torque_load": 0.0, "stand_still": -0.5, "stand_still_vel": -2.0, "feet_stance": -0.3,
        },
    }
    command_cfg = {"num_commands": 3, "lin_vel_x_range": [-1.0, 1.0], "lin_vel_y_range": [-0.3, 0.3], "ang_vel_range": [-1.0, 1.0],
                   "cmd_curriculum": True, "cmd_curriculum_start_frac": 0.1, "compound_commands": True, "rel_standing_envs": 0.1}
    return env_cfg, obs_cfg, reward_cfg, command_cfg


def _name_maps(model):
    links = {l["name"]: i for i, l in enumerate(model["links"])}
    joints = {j["name"]: j for j in model["joints"]}
    return links, joints


def flatten_walk_cfg(num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg, *, model=None, per_env_global_dr=False,
                     freeze_curriculum=False, shared_globals=False):
    """Go2Env.__init__ (go2_env_walk.py:155-525) as data: returns (fcfg float32[FC_COUNT], icfg int32[IC_COUNT],
    reward_names) for go2sim_env_configure."""
    model = load_model_json() if model is None else model
    links, joints = _name_maps(model)
    env_cfg = copy.deepcopy(env_cfg)
    f = np.zeros(C["GO2SIM_FC_COUNT"], dtype=np.float64)   # python floats of the cfg dicts, unrounded (include/go2sim.h: host scalars)
    i = np.zeros(C["GO2SIM_IC_COUNT"], dtype=np.int32)
    dt = 0.02
    F = lambda name: C["GO2SIM_FC_" + name]
    I = lambda name: C["GO2SIM_IC_" + name]
    curr = env_cfg.get("curriculum", {}) or {}

    f[F("DT")] = dt
    f[F("ACTION_SCALE")] = env_cfg["action_scale"]
    f[F("CLIP_ACTIONS")] = env_cfg["clip_actions"]
    f[F("KP")], f[F("KD")] = env_cfg["kp"], env_cfg["kd"]
    pls = bool(env_cfg.get("pls_enable", False))
    if pls:
        f[F("PLS_KP_MIN")], f[F("PLS_KP_MAX")] = env_cfg["pls_kp_range"]
        f[F("PLS_KP_DEFAULT")] = env_cfg["pls_kp_default"]
        f[F("PLS_KP_ACTION_SCALE")] = env_cfg["pls_kp_action_scale"]
    npos = env_cfg.get("num_pos_actions", 12)
    assert npos == 12
    tl = env_cfg.get("torque_limits", None) or [23.7] * 12
    f[F("TORQUE_LIMIT0"):F("TORQUE_LIMIT0") + 12] = tl
    f[F("DEFAULT_DOF_POS0"):F("DEFAULT_DOF_POS0") + 12] = [env_cfg["default_joint_angles"][n] for n in env_cfg["joint_names"]]
    f[F("TERM_PITCH_DEG")] = env_cfg["termination_if_pitch_greater_than"]
    f[F("TERM_ROLL_DEG")] = env_cfg["termination_if_roll_greater_than"]
    f[F("TERM_ZVEL")] = env_cfg["termination_if_z_vel_greater_than"]
    f[F("TERM_YVEL")] = env_cfg["termination_if_y_vel_greater_than"]
    f[F("BASE_INIT_POS0"):F("BASE_INIT_POS0") + 3] = env_cfg["base_init_pos"]
    f[F("BASE_INIT_QUAT0"):F("BASE_INIT_QUAT0") + 4] = env_cfg["base_init_quat"]
    if "init_pos_z_range" in env_cfg:
        i[I("HAS_INIT_Z")] = 1
        f[F("INIT_Z_LO")], f[F("INIT_Z_HI")] = env_cfg["init_pos_z_range"]
    if "init_euler_range" in env_cfg:
        i[I("HAS_INIT_EULER")] = 1
        f[F("INIT_EULER_LO_DEG")], f[F("INIT_EULER_HI_DEG")] = env_cfg["init_euler_range"]
    sc = obs_cfg["obs_scales"]
    f[F("OBS_SCALE_LIN_VEL")], f[F("OBS_SCALE_ANG_VEL")] = sc["lin_vel"], sc["ang_vel"]
    f[F("OBS_SCALE_DOF_POS")], f[F("OBS_SCALE_DOF_VEL")] = sc["dof_pos"], sc["dof_vel"]
    f[F("TRACKING_SIGMA")] = reward_cfg["tracking_sigma"]
    f[F("BASE_HEIGHT_TARGET")] = reward_cfg["base_height_target"]
    f[F("FEET_HEIGHT_TARGET")] = reward_cfg.get("feet_height_target", 0.075)
    f[F("FEET_AIR_TIME_TARGET")] = reward_cfg.get("feet_air_time_target", 0.1)
    f[F("FOOT_CONTACT_THRESHOLD")] = env_cfg.get("foot_contact_threshold", 1.0)
    names = list(reward_cfg["reward_scales"].keys())
    assert len(names) <= 32
    for k, name in enumerate(names):
        if name not in REWARD_IDS:
            raise AttributeError(f"Reward function '_reward_{name}' not found in Go2Env.")
        f[F("REWARD_SCALE0") + k] = reward_cfg["reward_scales"][name] * dt
        i[I("REWARD_ID0") + k] = C[REWARD_IDS[name]]
    i[I("N_REWARDS")] = len(names)
    f[F("CMD_X_LO")], f[F("CMD_X_HI")] = command_cfg["lin_vel_x_range"]
    f[F("CMD_Y_LO")], f[F("CMD_Y_HI")] = command_cfg["lin_vel_y_range"]
    f[F("CMD_YAW_LO")], f[F("CMD_YAW_HI")] = command_cfg["ang_vel_range"]
    f[F("CMD_START_FRAC")] = command_cfg.get("cmd_curriculum_start_frac", 0.1)
    i[I("CMD_CURRICULUM")] = int(bool(command_cfg.get("cmd_curriculum", False)))
    i[I("COMPOUND_COMMANDS")] = int(bool(command_cfg.get("compound_commands", True)))
    i[I("N_STANDING")] = int(float(command_cfg.get("rel_standing_envs", 0.0)) * num_envs)

    kp_nom, kd_nom = float(env_cfg.get("kp", 60.0)), float(env_cfg.get("kd", 2.0))

    def rng(flag, key_hard, easy_default, easy_key, base):
        easy = curr.get(easy_key, easy_default)
        hard = env_cfg.get(key_hard, easy)
        f[F(base + "_EASY_LO")], f[F(base + "_EASY_HI")] = easy
        f[F(base + "_HARD_LO")], f[F(base + "_HARD_HI")] = hard
        if flag:
            i[I(flag)] = int(key_hard in env_cfg)

    rng("HAS_FRICTION_DR", "friction_range", [0.6, 0.9], "friction_easy", "FRICTION")
    rng("HAS_KPF_DR", "kp_factor_range", [0.95, 1.05], "kp_factor_easy", "KPF")
    rng("HAS_KDF_DR", "kd_factor_range", [0.85, 1.15], "kd_factor_easy", "KDF")
    rng("HAS_KP_RANGE", "kp_range", [0.9 * kp_nom, 1.1 * kp_nom], "kp_easy", "KPR")
    rng(None, "kd_range", [0.75 * kd_nom, 1.25 * kd_nom], "kd_easy", "KDR")
    rng("HAS_MASS_DR", "mass_shift_range", [-0.2, 0.5], "mass_shift_easy", "MASS")
    rng("HAS_COM_DR", "com_shift_range", [-0.005, 0.005], "com_shift_easy", "COM")
    rng("HAS_LEGM_DR", "leg_mass_shift_range", [-0.1, 0.1], "leg_mass_shift_easy", "LEGM")
    rng("HAS_GOFF_DR", "gravity_offset_range", [-0.2, 0.2], "gravity_offset_easy", "GOFF")
    rng("HAS_MSTR_DR", "motor_strength_range", [0.97, 1.03], "motor_strength_easy", "MSTR")
    if not pls and "kp_range" in env_cfg:
        raise NotImplementedError("non-PLS per-env kp/kd ranges are not wired yet (reference configs use PLS)")

    on = env_cfg.get("obs_noise", None)
    i[I("HAS_OBS_NOISE")] = int(on is not None)
    if on is not None:
        f[F("OBS_NOISE_LEVEL_MAX")] = env_cfg.get("obs_noise_level", 0.0)
        f[F("OBS_NOISE_ANG_VEL")], f[F("OBS_NOISE_GRAVITY")] = on.get("ang_vel", 0.0), on.get("gravity", 0.0)
        f[F("OBS_NOISE_DOF_POS")], f[F("OBS_NOISE_DOF_VEL")] = on.get("dof_pos", 0.0), on.get("dof_vel", 0.0)
    f[F("ACTION_NOISE_STD_MAX")] = env_cfg.get("action_noise_std", 0.0)
    pfr = env_cfg.get("push_force_range", None)
    i[I("HAS_PUSH")] = int(pfr is not None)
    if pfr is not None:
        f[F("PUSH_FORCE_LO")], f[F("PUSH_FORCE_HI")] = pfr
        dur = env_cfg.get("push_duration_s", [0.05, 0.15])
        i[I("PUSH_DUR_LO")], i[I("PUSH_DUR_HI")] = max(1, int(dur[0] / dt)), max(1, int(dur[1] / dt))
    f[F("PUSH_INTERVAL_S_HARD")] = env_cfg.get("push_interval_s", 5.0)
    f[F("PUSH_INTERVAL_S_EASY")] = curr.get("push_interval_easy_s", 6.0)
    f[F("PUSH_START")] = curr.get("push_start", 0.30)
    i[I("MIN_DELAY")] = int(env_cfg.get("min_delay_steps", 0))
    i[I("MAX_DELAY")] = int(env_cfg.get("max_delay_steps", 2))
    i[I("DELAY_EASY_MAX")] = int(curr.get("delay_easy_max_steps", 1))
    if not 0 <= i[I("MAX_DELAY")] < C["GO2SIM_ACTION_RING_MAX"]:
        raise ValueError(f"max_delay_steps must be in [0, {C['GO2SIM_ACTION_RING_MAX'] - 1}] (the action ring is max_delay_steps + 1 deep, go2_env_walk.py:373-380)")

    i[I("CURR_ENABLED")] = int(bool(curr.get("enabled", False)))
    f[F("CURR_LEVEL_INIT")] = curr.get("level_init", 0.0)
    f[F("CURR_LEVEL_MIN")], f[F("CURR_LEVEL_MAX")] = curr.get("level_min", 0.0), curr.get("level_max", 1.0)
    f[F("CURR_EMA_ALPHA")] = curr.get("ema_alpha", 0.05)
    f[F("CURR_READY_TIMEOUT_RATE")] = curr.get("ready_timeout_rate", 0.7)
    f[F("CURR_READY_TRACKING")] = curr.get("ready_tracking", 0.6)
    f[F("CURR_READY_FALL_RATE")] = curr.get("ready_fall_rate", 0.30)
    f[F("CURR_HARD_FALL_RATE")] = curr.get("hard_fall_rate", 0.55)
    f[F("CURR_STEP_UP")], f[F("CURR_STEP_DOWN")] = curr.get("step_up", 0.02), curr.get("step_down", 0.01)
    f[F("CURR_MIX_PROB_CURRENT")] = curr.get("mix_prob_current", 0.80)
    f[F("CURR_MIX_LEVEL_LOW")], f[F("CURR_MIX_LEVEL_HIGH")] = curr.get("mix_level_low", 0.0), curr.get("mix_level_high", 0.6)
    i[I("CURR_READY_STREAK")], i[I("CURR_HARD_STREAK")] = curr.get("ready_streak", 3), curr.get("hard_streak", 2)
    i[I("CURR_COOLDOWN")] = curr.get("cooldown_updates", 1)
    i[I("CURR_UPDATE_EVERY")] = curr.get("update_every_episodes", 2048)
    i[I("GLOBAL_DR_INTERVAL")] = curr.get("global_dr_update_interval", 200)

    i[I("ENV_KIND")] = 0
    i[I("NUM_ACTIONS")], i[I("NUM_POS_ACTIONS")] = env_cfg["num_actions"], npos
    i[I("NUM_OBS")], i[I("NUM_PRIV_OBS")] = obs_cfg["num_obs"], obs_cfg.get("num_privileged_obs") or obs_cfg["num_obs"]
    i[I("PLS_ENABLE")] = int(pls)
    i[I("MANUAL_PD")] = int(pls or ("kp_factor_range" in env_cfg))
    if not i[I("MANUAL_PD")]:
        raise NotImplementedError("engine-PD control path (control_dofs_position) is the base env; not wired in the walk env yet")
    i[I("SUBSTEPS")] = 2
    i[I("MAX_EPISODE_LENGTH")] = math.ceil(env_cfg["episode_length_s"] / dt)
    i[I("RESAMPLE_STEPS")] = int(env_cfg["resampling_time_s"] / dt)
    for k, jn in enumerate(env_cfg["joint_names"]):
        i[I("MOTOR_DOF0") + k] = joints[jn]["dof_start"]
    for k, ln in enumerate(env_cfg.get("foot_names", [])):
        i[I("FOOT_LINK0") + k] = links[ln]
    for k, ln in enumerate(["FR_hip", "FL_hip", "RR_hip", "RL_hip"]):
        i[I("HIP_LINK0") + k] = links[ln]
    robot_links = [idx for idx, l in enumerate(model["links"]) if l["entity"] == 1]
    i[I("BASE_LINK")] = robot_links[0]
    i[I("PUSH_LINK")] = robot_links[1]  # `self.robot.links[1].idx` (go2_env_walk.py:339-342): first depth-1 link, not the base
    i[I("PER_ENV_GLOBAL_DR")] = int(per_env_global_dr)
    i[I("FREEZE_CURRICULUM")] = int(freeze_curriculum)
    i[I("SHARED_GLOBALS")] = int(shared_globals)       # one shard of a larger batch: distributed.sync_env_globals combines the shards

    # ---- stair env extras (go2_env_stair.py:352-398, 972-988, 1615-1626) ----
    f[F("LIN_VEL_Z_DEADZONE")] = float(reward_cfg.get("lin_vel_z_deadzone", 0.0))
    dr_sched = env_cfg.get("dr_schedule", None)
    i[I("DR_SCHEDULE")] = int(dr_sched is not None)
    if dr_sched is not None:
        f[F("DR_PHASE1_LEVEL")], f[F("DR_TERRAIN_GATE")] = dr_sched.get("phase1_level", 0.15), dr_sched.get("terrain_gate", 0.50)
    tcfg = env_cfg.get("terrain", None)
    if tcfg is not None and tcfg.get("enabled", False):
        import torch

        _, info = build_stair_terrain(tcfg)
        n_rows = info["num_difficulty_rows"]
        if n_rows > 16:
            raise NotImplementedError("at most 16 difficulty rows")
        i[I("USE_TERRAIN")], i[I("N_TERRAIN_ROWS")] = 1, n_rows
        f[F("TERRAIN_ORIGIN_X")], f[F("TERRAIN_ORIGIN_Y")] = info["terrain_origin"][0], info["terrain_origin"][1]
        f[F("TERRAIN_H_SCALE")] = info["horizontal_scale"]
        f[F("ROW_CENTER0"):F("ROW_CENTER0") + 3 * n_rows] = np.asarray(info["row_centers"], np.float32).reshape(-1)
        hs = tcfg.get("height_scan", {})
        nx, ny = int(hs.get("num_x", 11)), int(hs.get("num_y", 7))
        if nx * ny > 80:
            raise NotImplementedError("height scan grids of at most 80 points")
        xr, yr = hs.get("x_range", [-0.5, 0.5]), hs.get("y_range", [-0.3, 0.3])
        gx, gy = torch.meshgrid(torch.linspace(float(xr[0]), float(xr[1]), nx), torch.linspace(float(yr[0]), float(yr[1]), ny), indexing="ij")
        i[I("SCAN_N")] = nx * ny
        f[F("SCAN_X0"):F("SCAN_X0") + nx * ny] = gx.reshape(-1).numpy()
        f[F("SCAN_Y0"):F("SCAN_Y0") + nx * ny] = gy.reshape(-1).numpy()
    return f, i, names


# ---------------------------------------------------------------------------------------------
# base env (crouch / jump): examples/locomotion/final/go2_env_base.py + go2_train_crouch.py / go2_train_jump.py
# ---------------------------------------------------------------------------------------------
BASE_REWARD_IDS = {
    "tracking_lin_vel": "GO2SIM_R_TRACKING_LIN_VEL", "tracking_ang_vel": "GO2SIM_R_TRACKING_ANG_VEL", "lin_vel_z": "GO2SIM_R_LIN_VEL_Z",
    "action_rate": "GO2SIM_R_ACTION_RATE", "similar_to_default": "GO2SIM_R_SIMILAR_TO_DEFAULT", "base_height": "GO2SIM_R_BASE_HEIGHT",
    "jump_impulse": "GO2SIM_R_JUMP_IMPULSE", "jump_apex": "GO2SIM_R_JUMP_APEX", "xy_stability": "GO2SIM_R_XY_STABILITY",
    "orientation": "GO2SIM_R_ORIENTATION", "no_shake": "GO2SIM_R_NO_SHAKE", "crouch": "GO2SIM_R_CROUCH", "crouch_2": "GO2SIM_R_CROUCH_2",
    "ground_penalty": "GO2SIM_R_GROUND_PENALTY", "crouch_target": "GO2SIM_R_CROUCH_TARGET", "no_fall": "GO2SIM_R_NO_FALL",
    "y_stability": "GO2SIM_R_Y_STABILITY", "torque_load": "GO2SIM_R_TORQUE_LOAD_BASE", "crouch_progress": "GO2SIM_R_CROUCH_PROGRESS",
    "crouch_speed": "GO2SIM_R_CROUCH_SPEED",
}

_BASE_JOINTS = {
    "default_joint_angles": {
        "FL_hip_joint": 0.0, "FR_hip_joint": 0.0, "RL_hip_joint": 0.0, "RR_hip_joint": 0.0,
        "FL_thigh_joint": 0.8, "FR_thigh_joint": 0.8, "RL_thigh_joint": 1.0, "RR_thigh_joint": 1.0,
        "FL_calf_joint": -1.5, "FR_calf_joint": -1.5, "RL_calf_joint": -1.5, "RR_calf_joint": -1.5,
    },
    "joint_names": ["FR_hip_joint", "FR_thigh_joint", "FR_calf_joint", "FL_hip_joint", "FL_thigh_joint", "FL_calf_joint",
                    "RR_hip_joint", "RR_thigh_joint", "RR_calf_joint", "RL_hip_joint", "RL_thigh_joint", "RL_calf_joint"],
}
_BASE_OBS = {"num_obs": 45, "obs_scales": {"lin_vel": 2.0, "ang_vel": 0.25, "dof_pos": 1.0, "dof_vel": 0.05}}
_ZERO_CMD = {"num_commands": 3, "lin_vel_x_range": [0, 0], "lin_vel_y_range": [0, 0], "ang_vel_range": [0, 0]}


def get_crouch_cfgs():
    """go2_train_crouch.py:10-91 (values transcribed)."""
    env_cfg = dict(_BASE_JOINTS, num_actions=12, kp=60.0, kd=2.0, termination_if_roll_greater_than=10, termination_if_pitch_greater_than=10,
                   termination_if_z_vel_greater_than=0.7, termination_if_y_vel_greater_than=0.05, base_init_pos=[0.0, 0.0, 0.35],
                   base_init_quat=[0.0, 0.0, 0.0, 1.0], episode_length_s=10.0, resampling_time_s=2.0, action_scale=0.65,
                   simulate_action_latency=True, clip_actions=100.0, crouch_speed=5.0)
    reward_cfg = {"reward_scales": {"crouch_target": 50.0, "ground_penalty": 10.0, "orientation": 30.0, "no_shake": 0.0, "xy_stability": 0.0,
                                    "action_rate": -0.05, "similar_to_default": 1.0, "no_fall": 0.0, "torque_load": 0.0, "crouch_progress": 50.0}}
    return env_cfg, copy.deepcopy(_BASE_OBS), reward_cfg, copy.deepcopy(_ZERO_CMD)


def with_per_env_dr(cfgs, friction_range=(0.4, 0.9), mass_shift_range=(-1.0, 3.0)):
    """BASELINE.json configs[4]: a base-env configuration plus per-env friction / base-mass randomisation at reset
    (ranges: go2_train_jump.py:58 `friction_range`, go2_train_walk.py:135 `mass_shift_range`)."""
    env_cfg, obs_cfg, reward_cfg, command_cfg = cfgs
    env_cfg = dict(env_cfg, per_env_dr={"friction_range": tuple(friction_range), "mass_shift_range": tuple(mass_shift_range)})
    return env_cfg, obs_cfg, reward_cfg, command_cfg


def get_jump_cfgs():
    """go2_train_jump.py:10-101 (values transcribed; the DR / push keys of that file are not read by go2_env_base.py)."""
    env_cfg = dict(_BASE_JOINTS, num_actions=12, kp=60.0, kd=2.0, termination_if_roll_greater_than=25, termination_if_pitch_greater_than=25,
                   termination_if_z_vel_greater_than=100.0, termination_if_y_vel_greater_than=100.0, base_init_pos=[0.0, 0.0, 0.42],
                   base_init_quat=[0.0, 0.0, 0.0, 1.0], episode_length_s=3.0, resampling_time_s=2.0, action_scale=0.65,
                   simulate_action_latency=True, clip_actions=100.0, crouch_speed=5.0, friction_range=(0.4, 0.9), kp_scale_range=(0.4, 1.5),
                   kd_scale_range=(0.25, 2.0), push_enable=True, push_interval_s=1.0, push_prob=1.0, push_force_range=(0.0, 0.0),
                   push_z_scale=0.0, push_duration_s=0.15, push_direction_mode="random")
    reward_cfg = {"jump_apex_height": 0.55, "jump_apex_sigma": 0.06,
                  "reward_scales": {"jump_impulse": 6.0, "jump_apex": 20.0, "xy_stability": 12.0, "orientation": 3.0, "no_shake": 1.0, "crouch": 6.0}}
    return env_cfg, copy.deepcopy(_BASE_OBS), reward_cfg, copy.deepcopy(_ZERO_CMD)


def flatten_base_cfg(num_envs, env_cfg, obs_cfg, reward_cfg, command_cfg, *, model=None):
    """Go2Env.__init__ of go2_env_base.py:12-121 as data (ENV_KIND = 1): engine PD (`control_dofs_position`), one-step action
    latency, no domain randomisation / noise / pushes / curriculum, reset before the reward, 45 observations."""
    model = load_model_json() if model is None else model
    links, joints = _name_maps(model)
    f = np.zeros(C["GO2SIM_FC_COUNT"], dtype=np.float64)   # python floats of the cfg dicts, unrounded (include/go2sim.h: host scalars)
    i = np.zeros(C["GO2SIM_IC_COUNT"], dtype=np.int32)
    dt = 0.02
    F = lambda name: C["GO2SIM_FC_" + name]
    I = lambda name: C["GO2SIM_IC_" + name]
    if env_cfg["num_actions"] != 12 or obs_cfg["num_obs"] != 45:
        raise ValueError("the base env has 12 actions and 45 observations (go2_env_base.py:176-187)")
    f[F("DT")] = dt
    f[F("ACTION_SCALE")], f[F("CLIP_ACTIONS")] = env_cfg["action_scale"], env_cfg["clip_actions"]
    f[F("KP")], f[F("KD")] = env_cfg["kp"], env_cfg["kd"]
    f[F("TORQUE_LIMIT0"):F("TORQUE_LIMIT0") + 12] = 0.0
    f[F("DEFAULT_DOF_POS0"):F("DEFAULT_DOF_POS0") + 12] = [env_cfg["default_joint_angles"][n] for n in env_cfg["joint_names"]]
    f[F("TERM_PITCH_DEG")], f[F("TERM_ROLL_DEG")] = env_cfg["termination_if_pitch_greater_than"], env_cfg["termination_if_roll_greater_than"]
    f[F("TERM_ZVEL")], f[F("TERM_YVEL")] = env_cfg["termination_if_z_vel_greater_than"], env_cfg["termination_if_y_vel_greater_than"]
    f[F("BASE_INIT_POS0"):F("BASE_INIT_POS0") + 3] = env_cfg["base_init_pos"]
    f[F("BASE_INIT_QUAT0"):F("BASE_INIT_QUAT0") + 4] = env_cfg["base_init_quat"]
    sc = obs_cfg["obs_scales"]
    f[F("OBS_SCALE_LIN_VEL")], f[F("OBS_SCALE_ANG_VEL")] = sc["lin_vel"], sc["ang_vel"]
    f[F("OBS_SCALE_DOF_POS")], f[F("OBS_SCALE_DOF_VEL")] = sc["dof_pos"], sc["dof_vel"]
    f[F("TRACKING_SIGMA")] = reward_cfg.get("tracking_sigma", 0.25)
    f[F("BASE_HEIGHT_TARGET")] = reward_cfg.get("base_height_target", 0.3)
    f[F("JUMP_APEX_HEIGHT")] = reward_cfg.get("jump_apex_height", 0.0)
    f[F("JUMP_APEX_SIGMA")] = reward_cfg.get("jump_apex_sigma", 0.05)
    f[F("EPISODE_LENGTH_S")] = env_cfg["episode_length_s"]
    names = list(reward_cfg["reward_scales"].keys())
    assert len(names) <= 32
    for k, name in enumerate(names):
        if name not in BASE_REWARD_IDS:
            raise AttributeError(f"'Go2Env' object has no attribute '_reward_{name}'")
        f[F("REWARD_SCALE0") + k] = reward_cfg["reward_scales"][name] * dt
        i[I("REWARD_ID0") + k] = C[BASE_REWARD_IDS[name]]
    i[I("N_REWARDS")] = len(names)
    f[F("CMD_X_LO")], f[F("CMD_X_HI")] = command_cfg["lin_vel_x_range"]
    f[F("CMD_Y_LO")], f[F("CMD_Y_HI")] = command_cfg["lin_vel_y_range"]
    f[F("CMD_YAW_LO")], f[F("CMD_YAW_HI")] = command_cfg["ang_vel_range"]
    i[I("COMPOUND_COMMANDS")] = 1
    i[I("ENV_KIND")] = 1
    i[I("NUM_ACTIONS")], i[I("NUM_POS_ACTIONS")] = 12, 12
    i[I("NUM_OBS")], i[I("NUM_PRIV_OBS")] = 45, 45
    i[I("MANUAL_PD")] = 0
    i[I("SUBSTEPS")] = 2
    i[I("MAX_EPISODE_LENGTH")] = math.ceil(env_cfg["episode_length_s"] / dt)
    i[I("RESAMPLE_STEPS")] = int(env_cfg["resampling_time_s"] / dt)
    latency = 1 if env_cfg.get("simulate_action_latency", True) else 0
    i[I("MIN_DELAY")] = i[I("MAX_DELAY")] = i[I("DELAY_EASY_MAX")] = latency     # exec_actions = last_actions (go2_env_base.py:124)
    for k, jn in enumerate(env_cfg["joint_names"]):
        i[I("MOTOR_DOF0") + k] = joints[jn]["dof_start"]
    for k, ln in enumerate(["FR_calf", "FL_calf", "RR_calf", "RL_calf"]):
        i[I("FOOT_LINK0") + k] = links[ln]
    for k, ln in enumerate(["FR_hip", "FL_hip", "RR_hip", "RL_hip"]):
        i[I("HIP_LINK0") + k] = links[ln]
    robot_links = [idx for idx, l in enumerate(model["links"]) if l["entity"] == 1]
    i[I("BASE_LINK")], i[I("PUSH_LINK")] = robot_links[0], robot_links[1]
    i[I("CURR_UPDATE_EVERY")] = 1 << 30
    i[I("GLOBAL_DR_INTERVAL")] = 1 << 30
    i[I("FREEZE_CURRICULUM")] = 1
    # Extension for BASELINE.json configs[4] ("crouch+jump with per-env mass/friction domain randomisation"); go2_env_base.py itself
    # reads none of the DR keys of its train scripts.  env_cfg["per_env_dr"] = {"friction_range": (lo, hi), "mass_shift_range": (lo, hi)}
    # draws, at every reset of an env, one friction coefficient for all its geoms and one mass shift of its base link.
    dr = env_cfg.get("per_env_dr")
    if dr:
        i[I("PER_ENV_GLOBAL_DR")] = 1
        if dr.get("friction_range") is not None:
            lo, hi = dr["friction_range"]
            i[I("HAS_FRICTION_DR")] = 1
            f[F("FRICTION_EASY_LO"):F("FRICTION_EASY_LO") + 4] = [lo, hi, lo, hi]
        if dr.get("mass_shift_range") is not None:
            lo, hi = dr["mass_shift_range"]
            i[I("HAS_MASS_DR")] = 1
            f[F("MASS_EASY_LO"):F("MASS_EASY_LO") + 4] = [lo, hi, lo, hi]
    return f, i, names


# ---------------------------------------------------------------------------------------------
# stair terrain (go2_env_stair.py:47-186): difficulty rows along y, repeated up / down flights along x
# ---------------------------------------------------------------------------------------------
def get_stair_terrain_cfg():
    """`terrain` block of go2_train_stair.py:97-119."""
    return {"enabled": True, "horizontal_scale": 0.05, "vertical_scale": 0.005, "num_difficulty_rows": 13, "row_width_m": 6.0,
            "step_depth_m": 0.39, "num_steps": 6, "num_flights": 4, "flat_before_m": 2.0, "flat_top_m": 1.5, "flat_gap_m": 1.5,
            "flat_after_m": 2.0, "step_height_min": 0.02, "step_height_max": 0.15}


def build_stair_terrain(terrain_cfg):
    """Heightfield (int16 [x cells, y cells]) and placement metadata, as go2_env_stair.py:47-186 builds them."""
    g = terrain_cfg.get
    h_scale, v_scale = g("horizontal_scale", 0.05), g("vertical_scale", 0.005)
    num_rows, row_width_m = g("num_difficulty_rows", 10), g("row_width_m", 6.0)
    num_steps, num_flights = g("num_steps", 6), g("num_flights", 4)
    cells = lambda metres: max(1, int(round(metres / h_scale)))
    step_depth, flat_before, flat_top = cells(g("step_depth_m", 0.30)), cells(g("flat_before_m", 2.0)), cells(g("flat_top_m", 1.5))
    flat_gap, flat_after, row_width = cells(g("flat_gap_m", 1.5)), cells(g("flat_after_m", 2.0)), cells(row_width_m)
    stair_section = num_steps * step_depth
    total_x = flat_before + num_flights * (2 * stair_section + flat_top + flat_gap) + flat_after
    total_y = num_rows * row_width
    step_heights_m = np.linspace(g("step_height_min", 0.02), g("step_height_max", 0.15), num_rows)
    step_units = np.round(step_heights_m / v_scale).astype(np.int16)
    hf = np.zeros((total_x, total_y), dtype=np.int16)
    centers = []
    for row in range(num_rows):
        y0, y1, sh = row * row_width, (row + 1) * row_width, step_units[row]
        x = flat_before
        for _ in range(num_flights):
            for s in range(num_steps):
                hf[x:x + step_depth, y0:y1] = (s + 1) * sh
                x += step_depth
            top = num_steps * sh
            hf[x:x + flat_top, y0:y1] = top
            x += flat_top
            for s in range(num_steps):
                hf[x:x + step_depth, y0:y1] = top - (s + 1) * sh
                x += step_depth
            x += flat_gap
        centers.append((flat_before * h_scale * 0.5, (y0 + row_width / 2.0) * h_scale, 0.0))
    origin = (0.0, -total_y * h_scale / 2.0, 0.0)
    info = {"heightfield": hf, "horizontal_scale": h_scale, "vertical_scale": v_scale, "terrain_origin": origin,
            "num_difficulty_rows": num_rows, "row_centers": [(cx + origin[0], cy + origin[1], cz + origin[2]) for cx, cy, cz in centers],
            "step_heights_m": step_heights_m.tolist(), "total_x_m": total_x * h_scale, "total_y_m": total_y * h_scale,
            "num_flights": num_flights, "num_steps": num_steps, "step_depth_m": g("step_depth_m", 0.30)}
    return hf, info


def get_stair_cfgs():
    """go2_train_stair.py:84-372 (values transcribed): the walk configuration plus the stair terrain, the terrain-relative rewards, the
    two-phase DR schedule and the 182-wide privileged observation (49 + 55 + terrain row + 77 height-scan points)."""
    env_cfg, obs_cfg, reward_cfg, command_cfg = get_walk_cfgs()
    terrain = get_stair_terrain_cfg()
    terrain["height_scan"] = {"num_x": 11, "num_y": 7, "x_range": [-0.5, 0.5], "y_range": [-0.3, 0.3]}
    env_cfg["curriculum"].update({"level_init": 0.65, "ready_timeout_rate": 0.60, "ready_tracking": 0.45, "ready_fall_rate": 0.35, "ready_streak": 5,
                                  "hard_fall_rate": 0.40, "hard_streak": 2, "step_up": 0.01, "step_down": 0.03, "cooldown_updates": 5,
                                  "push_start": 0.3})
    env_cfg.update({"episode_length_s": 25.0, "terrain": terrain, "dr_schedule": {"phase1_level": 0.15, "terrain_gate": 0.50}})
    obs_cfg["num_privileged_obs"] = obs_cfg["num_obs"] + 55 + 1 + 77
    reward_cfg.update({"feet_height_target": 0.17, "lin_vel_z_deadzone": 0.15})
    reward_cfg["reward_scales"] = {
        "tracking_lin_vel": 1.5, "tracking_ang_vel": 0.8, "forward_progress": 0.4, "lin_vel_z": -1.0, "base_height": -0.1, "action_rate": -0.01,
        "similar_to_default": -0.05, "orientation_roll_only": -5.0, "dof_acc": -2.5e-7, "dof_vel": -5e-4, "ang_vel_xy": -0.05,
        "feet_air_time": 0.2, "foot_slip": -0.15, "foot_clearance": -0.5, "joint_tracking": -0.1, "energy": 0.0, "torque_load": 0.0,
        "stand_still": -0.5, "stand_still_vel": -2.0, "feet_stance": -0.3}
    command_cfg.update({"lin_vel_x_range": [0.3, 0.8], "lin_vel_y_range": [0.0, 0.0], "ang_vel_range": [0.0, 0.0], "cmd_curriculum": False,
                        "rel_standing_envs": 0.05})
    command_cfg.pop("cmd_curriculum_start_frac", None)   # not a key of go2_train_stair.py's command_cfg (the env defaults it to 0.1, go2_env_stair.py:584)
    return env_cfg, obs_cfg, reward_cfg, command_cfg
