/* go2sim.h -- C ABI of the MI355X-native vectorised Go2 locomotion environment (libgo2sim.so)
 * and of its CPU twin used as test oracle (oracle/libgo2sim_cpu.so, prefix go2sim_cpu_).
 *
 * The reference (saifahmadgit/go2-sim2real-locomotion-rl, a Genesis v0.4.0 fork) has no FFI on this
 * path: the seam is the Python API consumed by Go2Env.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain C, no torch types; every pointer argument named *_dev is a DEVICE pointer for go2sim_*
 *    (hipMalloc / torch ROCm storage) and a HOST pointer for go2sim_cpu_*.  Caller owns all buffers
 *    it passes; the library owns its state until go2sim_destroy.
 *  - all calls are stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *    never synchronise, never allocate after create; not thread-safe per handle.
 *  - return value: 0 = ok, negative = GO2SIM_E_* ; never throws.
 *  - batch layout at the boundary follows the reference getters: row-major [n_envs][k]
 *    (entities/rigid_entity/rigid_entity.py getters, "tensors are [n_envs(sel), n_idx(, k)]").
 *    Internally all per-env state is SoA [feature][n_envs] (genesis/utils/array_class.py layout).
 *  - quaternions are (w,x,y,z).
 */
#ifndef GO2SIM_H
#define GO2SIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO2SIM_E_OK 0
#define GO2SIM_E_BADARG -1
#define GO2SIM_E_BADMODEL -2
#define GO2SIM_E_NOMEM -3
#define GO2SIM_E_HIP -4
#define GO2SIM_E_NODEVICE -5

/* ---- model blob ("GO2M" v1), produced by go2_sim2real_locomotion_rl_amd/model_blob.py ----------
 * int32  header[32] : 0 magic 0x4D324F47, 1 version, 2 n_links, 3 n_joints, 4 n_dofs, 5 n_qs,
 *                     6 n_geoms, 7 n_entities, 8 n_possible_pairs, 9 max_collision_pairs,
 *                     10 max_contact_pairs, 11 max_broad_pairs, 12 n_contacts_per_pair, 13 iterations,
 *                     14 ls_iterations, 15 ccd_iterations, 16 support_res, 17 cyl_sections,
 *                     18 n_floats, 19 n_ints
 * float32 F[n_floats]:
 *   globals[16]  : substep_dt, gravity xyz, eps, tolerance, ls_tolerance, meaninertia, mc_perturbation,
 *                  mc_tolerance, mpr_to_gjk_overlap_ratio, ccd_eps, ccd_tolerance, 0, 0, 0
 *   links [n][26]: pos3 quat4 inertial_pos3 inertial_quat4 inertial_i9 mass1 invweight2
 *   joints[n][10]: pos3 sol_params7
 *   dofs  [n][17]: motion_ang3 motion_vel3 limit2 invweight armature damping stiffness frictionloss
 *                  kp kv force_range2
 *   qpos0 [n_qs]
 *   geoms [n][113]: pos3 quat4 data7 friction1 sol_params7 center3 init_aabb24 rim64(cylinder ring xy)
 *   mass_parent_mask [n_dofs*n_dofs]
 * int32 I[n_ints]:
 *   links [n][13]: parent root entity is_fixed joint_start joint_end dof_start dof_end q_start q_end
 *                  n_dofs geom_start geom_end
 *   joints[n][5] : type link q_start dof_start dof_end
 *   geoms [n][3] : type link is_convex
 *   entities[n][6]: link_start link_end dof_start dof_end geom_start geom_end
 *   collision_pair_idx [n_geoms*n_geoms]
 *   support_theta_to_ring [support_res]
 */
#define GO2SIM_MODEL_MAGIC 0x4D324F47
#define GO2SIM_MODEL_VERSION 1

/* compile-time shape of the Go2+plane model the kernels are specialised for */
/* (a SHAPE VARIANT of the two libraries is the same sources compiled with other counts, e.g. -DGO2SIM_NL=3 -DGO2SIM_ND=1 ... for a fixed-base
 *  pendulum: build.SHAPES / build.build_shape_variant; test infrastructure for the reference's analytic known answers, tests/test_analytic_shapes.py) */
#ifndef GO2SIM_NL
#define GO2SIM_NL 14   /* links   (plane + 13)            */
#define GO2SIM_ND 18   /* dofs    (6 free + 12 revolute)  */
#define GO2SIM_NQ 19   /* qpos                            */
#define GO2SIM_NG 28   /* geoms   (ground box + 27)       */
#define GO2SIM_NJ 13   /* joints                          */
#endif
#ifndef GO2SIM_NPAIR_MAX
#define GO2SIM_NPAIR_MAX 320
#endif
#define GO2SIM_MAX_CONTACTS 150
#define GO2SIM_MAX_BROAD 240
#define GO2SIM_NMOTOR 12
#define GO2SIM_NFEET 4

/* ---- state fields for go2sim_get_field / go2sim_set_field ------------------------------------
 * `k` = floats (or ints) per env.  Raw field transfers use the INTERNAL SoA layout [k][n_envs]
 * (this is the parity-test / checkpoint interface, not the Go2Env one). */
enum go2sim_field {
  GO2SIM_F_QPOS = 0,          /* f32 k=19  rigid_global_info.qpos                       */
  GO2SIM_F_VEL = 1,           /* f32 k=18  dofs_state.vel                               */
  GO2SIM_F_ACC = 2,           /* f32 k=18  dofs_state.acc                               */
  GO2SIM_F_QACC_WS = 3,       /* f32 k=18  constraint_state.qacc_ws                     */
  GO2SIM_F_CTRL_FORCE = 4,    /* f32 k=18  dofs_state.ctrl_force                        */
  GO2SIM_F_EXT_FORCE = 5,     /* f32 k=84  links cfrc_applied (ang3,vel3) x 14          */
  GO2SIM_F_MASS_SHIFT = 6,    /* f32 k=14  links_state.mass_shift                       */
  GO2SIM_F_COM_SHIFT = 7,     /* f32 k=42  links_state.i_pos_shift                      */
  GO2SIM_F_FRICTION_RATIO = 8,/* f32 k=28  geoms_state.friction_ratio                   */
  GO2SIM_F_LINK_POS = 9,      /* f32 k=42  links_state.pos                              */
  GO2SIM_F_LINK_QUAT = 10,    /* f32 k=56  links_state.quat                             */
  GO2SIM_F_LINK_CDVEL = 11,   /* f32 k=42  links_state.cd_vel                           */
  GO2SIM_F_LINK_CDANG = 12,   /* f32 k=42  links_state.cd_ang                           */
  GO2SIM_F_ROOT_COM = 13,     /* f32 k=3   links_state.root_COM of the robot            */
  GO2SIM_F_CONTACT_FORCE = 14,/* f32 k=42  links_state.contact_force                    */
  GO2SIM_F_MASS_MAT = 15,     /* f32 k=324 rigid_global_info.mass_mat                   */
  GO2SIM_F_FORCE = 16,        /* f32 k=18  dofs_state.force (pre-constraint: qf_smooth) */
  GO2SIM_F_ACC_SMOOTH = 17,   /* f32 k=18  dofs_state.acc_smooth                        */
  GO2SIM_F_CONTACT_POS = 18,  /* f32 k=450 contact_data.pos                             */
  GO2SIM_F_CONTACT_NORMAL = 19,/* f32 k=450 contact_data.normal                         */
  GO2SIM_F_CONTACT_PEN = 20,  /* f32 k=150 contact_data.penetration                     */
  GO2SIM_F_NORMAL_CACHE = 21, /* f32 k=960 contact_cache.normal (320 pairs x 3)         */
  GO2SIM_F_SORT_VALUE = 22,   /* f32 k=56  sort_buffer.value                            */
  GO2SIM_F_GEOM_FRICTION = 23,/* f32 k=28  geoms_info.friction (global in the reference)*/
  GO2SIM_F_EFC_FORCE = 24,    /* f32 k=GO2SIM_MAX_ROWS constraint_state.efc_force       */
  GO2SIM_F_QFRC_CONSTRAINT = 25, /* f32 k=18                                            */
  GO2SIM_F_CTRL_POS = 26,     /* f32 k=18  dofs_state.ctrl_pos (control_dofs_position)    */
  GO2SIM_F_CTRL_VEL = 27,     /* f32 k=18  dofs_state.ctrl_vel                          */
  GO2SIM_F_DOF_POS = 28,      /* f32 k=18  dofs_state.pos (qpos - qpos0 per dof; FK output) */
  /* int32 fields */
  GO2SIM_I_N_CONTACTS = 64,   /* i32 k=1   collider_state.n_contacts                    */
  GO2SIM_I_CONTACT_GEOMS = 65,/* i32 k=300 contact_data.geom_a, geom_b (150 + 150)      */
  GO2SIM_I_N_CONSTRAINTS = 66,/* i32 k=1                                                */
  GO2SIM_I_ERRNO = 67,        /* i32 k=1   errno bitmask                                */
  GO2SIM_I_IS_WARMSTART = 68, /* i32 k=1                                                */
  GO2SIM_I_FIRST_TIME = 69,   /* i32 k=1   collider_state.first_time                    */
  GO2SIM_I_SORT_IG = 70,      /* i32 k=56  sort_buffer.i_g | (is_max << 8)              */
  GO2SIM_I_N_BROAD = 71,      /* i32 k=1                                                */
  GO2SIM_I_SOLVER_ITERS = 72, /* i32 k=1   Newton iterations used in the last solve     */
  GO2SIM_I_CTRL_MODE = 73     /* i32 k=18  dofs_state.ctrl_mode (0 force,1 vel,2 pos)   */
};
#define GO2SIM_MAX_ROWS 624 /* 4*150 contact rows + 12 joint limits + 12 spare */

/* errno bits (genesis/utils/array_class.py ErrorCode) */
#define GO2SIM_ERR_OVERFLOW_CANDIDATE_CONTACTS 1
#define GO2SIM_ERR_OVERFLOW_COLLISION_PAIRS 2
#define GO2SIM_ERR_INVALID_FORCE_NAN 4
#define GO2SIM_ERR_INVALID_ACC_NAN 8

/* ---- Go2Env (walk) configuration: flat float / int arrays indexed by these enums --------------
 * Mirrors the cfg dicts of examples/locomotion/final/go2_train_walk.py:68-372 as consumed by
 * examples/locomotion/final/go2_env_walk.py:155-525. */
enum go2sim_fcfg {
  /* ---- host scalars: values that Go2Env keeps as python floats and combines in float64 before anything reaches a float32 tensor (the
     curriculum state machine, `_lerp_range(easy, hard, level)`, `_apply_curriculum_level`, `math.radians(init_euler_range)`, the obs-noise vector).
     The library keeps entries [0, GO2SIM_FC_N_HOST) in double precision as well and does that arithmetic in double. ---- */
  GO2SIM_FC_DT = 0,
  GO2SIM_FC_INIT_Z_LO, GO2SIM_FC_INIT_Z_HI, GO2SIM_FC_INIT_EULER_LO_DEG, GO2SIM_FC_INIT_EULER_HI_DEG,
  GO2SIM_FC_OBS_SCALE_LIN_VEL, GO2SIM_FC_OBS_SCALE_ANG_VEL, GO2SIM_FC_OBS_SCALE_DOF_POS, GO2SIM_FC_OBS_SCALE_DOF_VEL,
  GO2SIM_FC_CMD_X_LO, GO2SIM_FC_CMD_X_HI, GO2SIM_FC_CMD_Y_LO, GO2SIM_FC_CMD_Y_HI,
  GO2SIM_FC_CMD_YAW_LO, GO2SIM_FC_CMD_YAW_HI, GO2SIM_FC_CMD_START_FRAC,
  GO2SIM_FC_FRICTION_EASY_LO, GO2SIM_FC_FRICTION_EASY_HI, GO2SIM_FC_FRICTION_HARD_LO, GO2SIM_FC_FRICTION_HARD_HI,
  GO2SIM_FC_KPF_EASY_LO, GO2SIM_FC_KPF_EASY_HI, GO2SIM_FC_KPF_HARD_LO, GO2SIM_FC_KPF_HARD_HI,
  GO2SIM_FC_KDF_EASY_LO, GO2SIM_FC_KDF_EASY_HI, GO2SIM_FC_KDF_HARD_LO, GO2SIM_FC_KDF_HARD_HI,
  GO2SIM_FC_KPR_EASY_LO, GO2SIM_FC_KPR_EASY_HI, GO2SIM_FC_KPR_HARD_LO, GO2SIM_FC_KPR_HARD_HI,
  GO2SIM_FC_KDR_EASY_LO, GO2SIM_FC_KDR_EASY_HI, GO2SIM_FC_KDR_HARD_LO, GO2SIM_FC_KDR_HARD_HI,
  GO2SIM_FC_MASS_EASY_LO, GO2SIM_FC_MASS_EASY_HI, GO2SIM_FC_MASS_HARD_LO, GO2SIM_FC_MASS_HARD_HI,
  GO2SIM_FC_COM_EASY_LO, GO2SIM_FC_COM_EASY_HI, GO2SIM_FC_COM_HARD_LO, GO2SIM_FC_COM_HARD_HI,
  GO2SIM_FC_LEGM_EASY_LO, GO2SIM_FC_LEGM_EASY_HI, GO2SIM_FC_LEGM_HARD_LO, GO2SIM_FC_LEGM_HARD_HI,
  GO2SIM_FC_GOFF_EASY_LO, GO2SIM_FC_GOFF_EASY_HI, GO2SIM_FC_GOFF_HARD_LO, GO2SIM_FC_GOFF_HARD_HI,
  GO2SIM_FC_MSTR_EASY_LO, GO2SIM_FC_MSTR_EASY_HI, GO2SIM_FC_MSTR_HARD_LO, GO2SIM_FC_MSTR_HARD_HI,
  GO2SIM_FC_OBS_NOISE_LEVEL_MAX, GO2SIM_FC_OBS_NOISE_ANG_VEL, GO2SIM_FC_OBS_NOISE_GRAVITY,
  GO2SIM_FC_OBS_NOISE_DOF_POS, GO2SIM_FC_OBS_NOISE_DOF_VEL, GO2SIM_FC_ACTION_NOISE_STD_MAX,
  GO2SIM_FC_PUSH_FORCE_LO, GO2SIM_FC_PUSH_FORCE_HI, GO2SIM_FC_PUSH_INTERVAL_S_HARD,
  GO2SIM_FC_PUSH_INTERVAL_S_EASY, GO2SIM_FC_PUSH_START,
  GO2SIM_FC_CURR_LEVEL_INIT, GO2SIM_FC_CURR_LEVEL_MIN, GO2SIM_FC_CURR_LEVEL_MAX, GO2SIM_FC_CURR_EMA_ALPHA,
  GO2SIM_FC_CURR_READY_TIMEOUT_RATE, GO2SIM_FC_CURR_READY_TRACKING, GO2SIM_FC_CURR_READY_FALL_RATE,
  GO2SIM_FC_CURR_HARD_FALL_RATE, GO2SIM_FC_CURR_STEP_UP, GO2SIM_FC_CURR_STEP_DOWN,
  GO2SIM_FC_CURR_MIX_PROB_CURRENT, GO2SIM_FC_CURR_MIX_LEVEL_LOW, GO2SIM_FC_CURR_MIX_LEVEL_HIGH,
  GO2SIM_FC_EPISODE_LENGTH_S,                               /* base env (go2_env_base.py:232-236): episode-log normalisation */
  GO2SIM_FC_DR_PHASE1_LEVEL, GO2SIM_FC_DR_TERRAIN_GATE,     /* stair env: two-phase DR schedule (go2_env_stair.py:972-988) */
  GO2SIM_FC_N_HOST,
  /* ---- constants of the per-env float32 arithmetic (python floats that meet a float32 tensor directly) ---- */
  GO2SIM_FC_ACTION_SCALE = GO2SIM_FC_N_HOST, GO2SIM_FC_CLIP_ACTIONS,
  GO2SIM_FC_KP, GO2SIM_FC_KD,
  GO2SIM_FC_PLS_KP_MIN, GO2SIM_FC_PLS_KP_MAX, GO2SIM_FC_PLS_KP_DEFAULT, GO2SIM_FC_PLS_KP_ACTION_SCALE,
  GO2SIM_FC_TORQUE_LIMIT0, /* 12 values, env joint order */
  GO2SIM_FC_DEFAULT_DOF_POS0 = GO2SIM_FC_TORQUE_LIMIT0 + 12, /* 12 values */
  GO2SIM_FC_TERM_PITCH_DEG = GO2SIM_FC_DEFAULT_DOF_POS0 + 12, GO2SIM_FC_TERM_ROLL_DEG,
  GO2SIM_FC_TERM_ZVEL, GO2SIM_FC_TERM_YVEL,
  GO2SIM_FC_BASE_INIT_POS0, GO2SIM_FC_BASE_INIT_QUAT0 = GO2SIM_FC_BASE_INIT_POS0 + 3,
  GO2SIM_FC_TRACKING_SIGMA = GO2SIM_FC_BASE_INIT_QUAT0 + 4, GO2SIM_FC_BASE_HEIGHT_TARGET, GO2SIM_FC_FEET_HEIGHT_TARGET,
  GO2SIM_FC_FEET_AIR_TIME_TARGET, GO2SIM_FC_FOOT_CONTACT_THRESHOLD,
  GO2SIM_FC_REWARD_SCALE0, /* 32 values: reward_scales[name] * dt (the float64 product) in evaluation order */
  /* base env (go2_env_base.py): jump rewards */
  GO2SIM_FC_JUMP_APEX_HEIGHT = GO2SIM_FC_REWARD_SCALE0 + 32, GO2SIM_FC_JUMP_APEX_SIGMA,
  /* stair env (go2_env_stair.py): terrain-relative rewards, spawn rows, height scan */
  GO2SIM_FC_LIN_VEL_Z_DEADZONE,
  GO2SIM_FC_TERRAIN_ORIGIN_X, GO2SIM_FC_TERRAIN_ORIGIN_Y, GO2SIM_FC_TERRAIN_H_SCALE,
  GO2SIM_FC_ROW_CENTER0,                                   /* 16 x (x, y, z) spawn centres of the difficulty rows */
  GO2SIM_FC_SCAN_X0 = GO2SIM_FC_ROW_CENTER0 + 48,          /* 80 body-frame x offsets of the height-scan grid (row-major nx x ny) */
  GO2SIM_FC_SCAN_Y0 = GO2SIM_FC_SCAN_X0 + 80,              /* 80 body-frame y offsets */
  GO2SIM_FC_COUNT = GO2SIM_FC_SCAN_Y0 + 80
};
enum go2sim_icfg {
  GO2SIM_IC_ENV_KIND = 0, /* 0 = walk (go2_env_walk.py), 1 = base (go2_env_base.py: crouch / jump; engine PD, reset before reward, 45 obs) */
  GO2SIM_IC_NUM_ACTIONS, GO2SIM_IC_NUM_POS_ACTIONS, GO2SIM_IC_NUM_OBS, GO2SIM_IC_NUM_PRIV_OBS,
  GO2SIM_IC_PLS_ENABLE, GO2SIM_IC_MANUAL_PD, GO2SIM_IC_SUBSTEPS,
  GO2SIM_IC_MAX_EPISODE_LENGTH, GO2SIM_IC_RESAMPLE_STEPS,
  GO2SIM_IC_MOTOR_DOF0, /* 12 values: global dof index of env joint i */
  GO2SIM_IC_FOOT_LINK0 = GO2SIM_IC_MOTOR_DOF0 + 12, /* 4 global link indices */
  GO2SIM_IC_HIP_LINK0 = GO2SIM_IC_FOOT_LINK0 + 4,   /* 4 global link indices (leg mass DR) */
  GO2SIM_IC_PUSH_LINK = GO2SIM_IC_HIP_LINK0 + 4, GO2SIM_IC_BASE_LINK,
  GO2SIM_IC_HAS_INIT_Z, GO2SIM_IC_HAS_INIT_EULER,
  GO2SIM_IC_N_REWARDS, GO2SIM_IC_REWARD_ID0, /* 32 values: enum go2sim_reward in evaluation order */
  GO2SIM_IC_CMD_CURRICULUM = GO2SIM_IC_REWARD_ID0 + 32, GO2SIM_IC_COMPOUND_COMMANDS, GO2SIM_IC_N_STANDING,
  GO2SIM_IC_HAS_FRICTION_DR, GO2SIM_IC_HAS_KPF_DR, GO2SIM_IC_HAS_KDF_DR, GO2SIM_IC_HAS_KP_RANGE,
  GO2SIM_IC_HAS_MASS_DR, GO2SIM_IC_HAS_COM_DR, GO2SIM_IC_HAS_LEGM_DR, GO2SIM_IC_HAS_GOFF_DR, GO2SIM_IC_HAS_MSTR_DR,
  GO2SIM_IC_HAS_OBS_NOISE, GO2SIM_IC_HAS_PUSH, GO2SIM_IC_PUSH_DUR_LO, GO2SIM_IC_PUSH_DUR_HI,
  GO2SIM_IC_MIN_DELAY, GO2SIM_IC_MAX_DELAY, GO2SIM_IC_DELAY_EASY_MAX,
  GO2SIM_IC_CURR_ENABLED, GO2SIM_IC_CURR_READY_STREAK, GO2SIM_IC_CURR_HARD_STREAK, GO2SIM_IC_CURR_COOLDOWN,
  GO2SIM_IC_CURR_UPDATE_EVERY, GO2SIM_IC_GLOBAL_DR_INTERVAL,
  GO2SIM_IC_PER_ENV_GLOBAL_DR, /* 0 = reference behaviour (one value for all envs), 1 = per-env draws */
  GO2SIM_IC_FREEZE_CURRICULUM, /* 1 = keep the level fixed (bench protocol) */
  /* stair env (go2_env_stair.py) */
  GO2SIM_IC_USE_TERRAIN,       /* terrain-relative rewards, spawn rows, terrain_row + height scan in the privileged obs */
  GO2SIM_IC_DR_SCHEDULE,       /* 1 = two-phase DR level (_get_dr_level :972-988) and t_sample = dr level (:1507) */
  GO2SIM_IC_N_TERRAIN_ROWS, GO2SIM_IC_SCAN_N,
  GO2SIM_IC_SHARED_GLOBALS,    /* 1 = this handle is one shard of a larger batch (one process per GPU): the curriculum counters are only
                                  accumulated, and neither the curriculum update nor the "global" DR draws run inside env_step; the host combines the
                                  shards with go2sim_env_sync_counters / _sync_apply / _set_global_dr (distributed.sync_env_globals) */
  GO2SIM_IC_COUNT
};
/* reward terms of go2_env_walk.py:1251-1366 */
enum go2sim_reward {
  GO2SIM_R_TRACKING_LIN_VEL = 0, GO2SIM_R_TRACKING_ANG_VEL, GO2SIM_R_LIN_VEL_Z, GO2SIM_R_BASE_HEIGHT,
  GO2SIM_R_ACTION_RATE, GO2SIM_R_SIMILAR_TO_DEFAULT, GO2SIM_R_ORIENTATION_PENALTY, GO2SIM_R_DOF_ACC,
  GO2SIM_R_DOF_VEL, GO2SIM_R_ANG_VEL_XY, GO2SIM_R_FEET_AIR_TIME, GO2SIM_R_FOOT_SLIP, GO2SIM_R_FOOT_CLEARANCE,
  GO2SIM_R_JOINT_TRACKING, GO2SIM_R_ENERGY, GO2SIM_R_TORQUE_LOAD, GO2SIM_R_STAND_STILL, GO2SIM_R_STAND_STILL_VEL,
  GO2SIM_R_FEET_STANCE,
  /* reward terms of go2_env_base.py:246-390 (crouch / jump tasks) */
  GO2SIM_R_JUMP_IMPULSE, GO2SIM_R_JUMP_APEX, GO2SIM_R_XY_STABILITY, GO2SIM_R_ORIENTATION, GO2SIM_R_NO_SHAKE, GO2SIM_R_CROUCH,
  GO2SIM_R_CROUCH_2, GO2SIM_R_GROUND_PENALTY, GO2SIM_R_CROUCH_TARGET, GO2SIM_R_NO_FALL, GO2SIM_R_Y_STABILITY, GO2SIM_R_TORQUE_LOAD_BASE,
  GO2SIM_R_CROUCH_PROGRESS, GO2SIM_R_CROUCH_SPEED,
  /* additional terms of go2_env_stair.py:1659-1771 */
  GO2SIM_R_ORIENTATION_ROLL_ONLY, GO2SIM_R_FORWARD_PROGRESS,
  GO2SIM_R_COUNT
};

/* env-level buffers readable with go2sim_env_get (row-major [n_envs][k], device pointers) */
enum go2sim_env_buf {
  GO2SIM_EB_COMMANDS = 0,     /* f32 k=3  */
  GO2SIM_EB_EPISODE_LENGTH,   /* i32 k=1  */
  GO2SIM_EB_BASE_LIN_VEL,     /* f32 k=3  */
  GO2SIM_EB_BASE_ANG_VEL,     /* f32 k=3  */
  GO2SIM_EB_PROJECTED_GRAVITY,/* f32 k=3  */
  GO2SIM_EB_DOF_POS,          /* f32 k=12 */
  GO2SIM_EB_DOF_VEL,          /* f32 k=12 */
  GO2SIM_EB_BASE_POS,         /* f32 k=3  */
  GO2SIM_EB_BASE_QUAT,        /* f32 k=4  */
  GO2SIM_EB_BASE_EULER,       /* f32 k=3  degrees */
  GO2SIM_EB_EPISODE_SUMS,     /* f32 k=32 */
  GO2SIM_EB_FOOT_CONTACT,     /* i32 k=4  */
  GO2SIM_EB_FEET_AIR_TIME,    /* f32 k=4  */
  GO2SIM_EB_REW_TERMS,        /* f32 k=32 per-term reward of the last step (already x scale) */
  GO2SIM_EB_TORQUE,           /* f32 k=12 last commanded torque */
  GO2SIM_EB_TERRAIN_ROW,      /* i32 k=1  difficulty row of the env (go2_env_stair.py:_env_terrain_row) */
  GO2SIM_EB_COUNT
};

/* Depth limit of the per-env action ring of the walk / stair envs: `_action_history[B, max_delay_steps + 1, A]`
   (go2_env_walk.py:373-380, 916-923).  env_cfg["max_delay_steps"] <= GO2SIM_ACTION_RING_MAX - 1; the reference's walk cfg uses 1, the
   env's own default is 2. */
#define GO2SIM_ACTION_RING_MAX 4

/* device-side env globals (curriculum + "global" DR scalars), host-readable snapshot */
typedef struct go2sim_env_globals {
  /* python-float (float64) state of Go2Env / CurriculumManager, kept in double like the reference keeps it */
  double level;                /* CurriculumManager.level (go2_env_walk.py:54) */
  double timeout_rate_ema, tracking_ema, fall_rate_ema;
  double curr_timeout_total, curr_tracking_sum;            /* _curr_timeout_total, _curr_tracking_sum (:461-462) */
  double obs_noise_level_cur, action_noise_std_cur;         /* :634-635 */
  double push_force_lo, push_force_hi;                      /* _push_force_range_cur (:651) */
  double cmd_x_lo, cmd_x_hi, cmd_y_lo, cmd_y_hi, cmd_yaw_lo, cmd_yaw_hi;   /* _cmd_cur_ranges (:662-672) */
  double t_sample;                                          /* level handed to the DR helpers by reset_idx (:1162) */
  int   ema_valid, ready_streak, hard_streak, cooldown;
  int   curr_ep_total, curr_tracking_n;
  int   push_enable, push_interval, push_counter;
  int   delay_max_cur;
  int   global_dr_reset_counter;
  float friction, mass_shift, com_shift[3], leg_mass_shift[4];   /* float(tensor.item()) of float32 draws (:750, :809, :817-819, :841) */
  int   action_write_idx;
  unsigned int step_count, reset_calls;
  /* extras["episode"]: mean per-second reward of the envs reset in the last reset call */
  int   last_reset_count; float last_episode_rew[32];
  int   n_reset_now; float ep_acc[32]; /* scratch accumulators */
  float terrain_mean_row;      /* extras["episode"]["terrain_mean_row"] of the last reset call (go2_env_stair.py:1588-1590);
                                  filled by go2sim_env_globals() from terrain_row_sum / last_reset_count */
  int   terrain_row_sum;
  int   lock_terrain_rows;     /* env._lock_terrain_rows (go2_env_stair.py:399,1513; set by go2_eval_stairs.py:657) */
  int   sync_calls;            /* GO2SIM_IC_SHARED_GLOBALS: number of go2sim_env_sync_apply calls (keys the global-DR draws) */
  double shard_counters[5];    /* GO2SIM_IC_SHARED_GLOBALS: this shard's increments since the last go2sim_env_sync_counters:
                                  episodes, time-outs, tracking sum, tracking n (go2_env_walk.py:460-463, 712-715), resets counted for the friction throttle (:744) */
} go2sim_env_globals_t;

typedef struct go2sim go2sim_t;

/* gs.init + gs.Scene(...).add_entity(plane).add_entity(go2).build(n_envs)  (genesis/__init__.py:55,
 * engine/scene.py:803; rigid_solver.py:337-523).  `device` = HIP device ordinal. */
int go2sim_create(const void* model_blob, size_t nbytes, int n_envs, int device, uint64_t seed, go2sim_t** out);
int go2sim_destroy(go2sim_t* h);
int go2sim_n_envs(const go2sim_t* h);

/* Scene._reset(): restore qpos0 / zero velocity, clear collider + warm start (scene.py:936,
 * rigid_solver.py:1730-1779). */
int go2sim_scene_reset(go2sim_t* h, void* stream);
/* one RigidSolver.substep (rigid_solver.py:1116-1184) */
int go2sim_substep(go2sim_t* h, void* stream);
/* Scene.step(): `substeps` substeps + clear_external_force (scene.py:979, simulator.py:262-286) */
int go2sim_scene_step(go2sim_t* h, int substeps, void* stream);
/* kernel_forward_kinematics_links_geoms over the full batch (abd/forward_kinematics.py:28-66) */
int go2sim_forward_kinematics(go2sim_t* h, void* stream);

/* raw SoA field transfer (device<->device for go2sim_*, host<->host for go2sim_cpu_*), parity /
 * checkpoint interface: buffer is [k][n_envs] of f32 or i32. */
int go2sim_field_size(int field, int* k_out, int* is_int_out);
int go2sim_get_field(go2sim_t* h, int field, void* dst_dev, void* stream);
int go2sim_set_field(go2sim_t* h, int field, const void* src_dev, void* stream);

/* Zero-copy access to the internal SoA state (the reference's qd_to_torch(..., copy=False) views,
 * genesis/utils/misc.py:596): returns the DEVICE address of field `field` laid out [k][n_envs].  The
 * Python shim wraps it as a torch tensor and implements the RigidEntity setters/getters Go2Env uses
 * (rigid_entity.py:2173-2238,2482-2537,2707-2742,3032,3189-3240; rigid_solver.py:1314-1361) on top. */
int go2sim_field_ptr(go2sim_t* h, int field, void** ptr_out);
/* collider.reset(envs_idx) + constraint_solver.reset(envs_idx) + errno[envs_idx] = 0
 * (rigid_solver.py:2403-2410): clears the contact-normal cache and the solver warm start of the given
 * envs (int32 device index array, or NULL = all envs). */
int go2sim_reset_caches(go2sim_t* h, const int* envs_idx_dev, int n_sel, void* stream);
/* RigidEntity.set_friction on ground + robot: geoms_info.friction = mu for every geom
 * (rigid_entity.py:3189, rigid_geom.py:327-336). */
int go2sim_set_friction(go2sim_t* h, float mu, void* stream);
/* gs.morphs.Terrain(height_field=hf, horizontal_scale=, vertical_scale=, pos=origin) in place of the plane entity
 * (go2_env_stair.py:424-433; rigid_entity.py:505-552; collider.py:374-394): geom 0 becomes a GEOM_TYPE.TERRAIN heightfield whose cells are
 * collided as 6-vertex prisms (narrowphase.py:345-512).  `hf_host` is a HOST pointer to int16[rows][cols] (heights = hf * vertical_scale);
 * origin = world position of cell (0, 0).  Resets the collision caches and refreshes the kinematics. */
int go2sim_set_terrain(go2sim_t* h, const int16_t* hf_host, int rows, int cols, float horizontal_scale, float vertical_scale,
                       const float* origin3_host, void* stream);
/* dofs_info.kp / kv / force_range for one dof (rigid_solver.py:2270-2292); host values. */
int go2sim_set_dof_gains(go2sim_t* h, int dof_idx, float kp, float kv, float force_lo, float force_hi);
/* RigidSolver.check_errno (rigid_solver.py:1189-1213): OR-reduction of errno over envs, written to a
 * host int.  This call synchronises the stream. */
int go2sim_check_errno(go2sim_t* h, int* errno_host, void* stream);

/* Non-blocking form of the same poll (Simulator.step polls every 10 substeps, simulator.py:267): _begin enqueues the reduction and a copy to
 * pinned host memory on `stream`; _result sets *ready = 1 and *errno_host once that copy has completed (never waits). */
int go2sim_errno_poll_begin(go2sim_t* h, void* stream);
int go2sim_errno_poll_result(go2sim_t* h, int* errno_host, int* ready);
/* waits for the poll in flight (if any) and returns its value: bounds the staleness when the host runs ahead of the device -- the caller waits
 * on a poll that is a whole cadence old before starting the next one, which in practice never blocks */
int go2sim_errno_poll_wait(go2sim_t* h, int* errno_host);
/* whether go2sim_env_step currently replays its hipGraph (1) or issues plain launches (0), and how often the graph path was abandoned */
int go2sim_graph_status(go2sim_t* h, int* using_graph, int* n_fallbacks);

/* ---- fused Go2Env fast path (examples/locomotion/final/go2_env_walk.py) ------------------------ */
/* Go2Env.__init__ buffers + cfg (go2_env_walk.py:155-525) */
/* fcfg_host: float64 [GO2SIM_FC_COUNT] (enum go2sim_fcfg; the cfg dicts hold python floats: entries below GO2SIM_FC_N_HOST keep their double value,
 * all entries are also kept rounded to float32 for the per-env arithmetic); icfg_host: int32 [GO2SIM_IC_COUNT] */
int go2sim_env_configure(go2sim_t* h, const double* fcfg_host, int n_f, const int* icfg_host, int n_i);
/* Go2Env.step (go2_env_walk.py:985-1109): actions [n_envs][num_actions] -> obs [n_envs][num_obs],
 * priv [n_envs][num_priv_obs], rew [n_envs], reset [n_envs] (u8 bool), time_outs [n_envs] f32. */
int go2sim_env_step(go2sim_t* h, const float* actions_dev, float* obs_dev, float* priv_dev, float* rew_dev,
                    uint8_t* reset_dev, float* timeout_dev, void* stream);
/* Go2Env.reset (go2_env_walk.py:1242-1245): reset_idx(all envs); obs buffers are left as they are. */
int go2sim_env_reset(go2sim_t* h, void* stream);
/* Go2Env.reset_idx(envs_idx) (go2_env_walk.py:1156-1240; go2_env_stair.py:1499-1600): curriculum bookkeeping + "global" DR draws + per-env
 * reset of the listed envs (int32 device index array, n >= 0; n == 0 is a no-op as in the reference).  reset_buf is left 1 on the listed envs
 * and 0 elsewhere (the reference leaves the other entries as they were; the next step rewrites the whole buffer). */
int go2sim_env_reset_idx(go2sim_t* h, const int* envs_idx_dev, int n, void* stream);
/* ---- eval / teleop surface (examples/locomotion/final/go2_eval_walk.py, go2_eval_stairs.py) ----
 * respawn_at_start (go2_eval_stairs.py:314-361) / respawn_on_tile (go2_eval_walk.py:399-480): robot.set_dofs_position(default, zero_velocity)
 * + set_pos + set_quat + zero_all_dofs_velocity on the listed envs; pos_dev [n][3], quat_dev [n][4] wxyz or NULL = base_init_quat;
 * clear_buffers != 0 also clears last_actions / action history / last_dof_vel / base velocities and sets _last_base_pos_x, as respawn_at_start does. */
int go2sim_env_respawn(go2sim_t* h, const int* envs_idx_dev, int n, const float* pos_dev, const float* quat_dev, int clear_buffers, void* stream);
/* env._lock_terrain_rows = lock (go2_env_stair.py:399; go2_eval_stairs.py:657): reset_idx then keeps every env on its terrain row */
int go2sim_env_lock_terrain_rows(go2sim_t* h, int lock, void* stream);
/* env._env_terrain_row[:] = rows (int32 [n_envs] device array; clamped to the configured rows) */
int go2sim_env_set_terrain_rows(go2sim_t* h, const int* rows_dev, void* stream);
/* env buffers (device pointer to [n_envs][k] row-major copy) and globals snapshot (synchronises) */
int go2sim_env_get(go2sim_t* h, int env_buf, void* dst_dev, void* stream);
int go2sim_env_set_episode_length(go2sim_t* h, const int* ep_len_dev, void* stream); /* rsl_rl init_at_random_ep_len */
int go2sim_env_set_commands(go2sim_t* h, const float* cmd_dev, void* stream);
int go2sim_env_globals(go2sim_t* h, go2sim_env_globals_t* out_host, void* stream);
int go2sim_env_set_level(go2sim_t* h, double level, void* stream);
/* ---- one batch sharded over several handles / processes (GO2SIM_IC_SHARED_GLOBALS) --------------------------------------
 * The reference has ONE CurriculumManager and one set of "global" DR scalars for all envs (go2_env_walk.py:458-463, 737-756, 803-848).
 * A shard accumulates its increments of the four curriculum counters and of the friction-throttle counter on the device;
 *   go2sim_env_sync_counters   copies them out ([5] doubles, host) and clears them                                        (synchronises)
 *   -- the caller sums them over the shards (one all-reduce of 5 doubles, RCCL) --
 *   go2sim_env_sync_apply      adds the summed increments to the handle's totals and runs what Go2Env.reset_idx runs once per call on them:
 *                              _maybe_update_curriculum_on_reset (:688-729), sample_level (:85-93) and the global DR draws (:737-756, 803-848);
 *                              the draws of this shard are returned in dr_out[10] = friction, mass_shift, com_shift[3], leg_mass_shift[4], t_sample
 *   -- the caller broadcasts the draws of shard 0 --
 *   go2sim_env_set_global_dr   stores the 10 scalars and applies the first nine to every env of the shard (set_friction / set_mass_shift / set_COM_shift of
 *                              :751-752, 810, 820, 843) followed by the full-batch kinematics refresh.
 * All shards then hold the same level, the same t_sample and the same global scalars.  Staleness against the single-process env: the reset calls
 * between two syncs use the level / scalars of the last sync. */
int go2sim_env_sync_counters(go2sim_t* h, double* counters5_host, void* stream);
int go2sim_env_sync_apply(go2sim_t* h, const double* summed_counters5_host, double* dr_out10_host, void* stream);
int go2sim_env_set_global_dr(go2sim_t* h, const double* dr10_host, void* stream);
/* The same three steps with DEVICE arrays (caller-owned, float64), stream-ordered and without any host synchronisation: the form the RCCL path uses
 * (distributed.sync_env_globals under backend "nccl": counters and scalars never leave the device; the host forms above serve gloo and tests).
 * The very first go2sim_env_sync_apply[_dev] of a handle draws the scalars even when no reset has been counted yet, so that a sync issued between
 * go2sim_env_configure and go2sim_env_reset puts every shard where the single-process env starts: t_sample at level_init and the first global
 * draws of the constructor's reset_idx (go2_env_walk.py:1243-1245).  For that initial sync the caller adds its env count to counters[4] (the
 * friction-throttle increment the constructor's reset would make: distributed.sync_env_globals(initial=True)); the reset that follows does not
 * count its envs for the throttle again. */
int go2sim_env_sync_counters_dev(go2sim_t* h, double* counters5_dev, void* stream);
int go2sim_env_sync_apply_dev(go2sim_t* h, const double* summed_counters5_dev, double* dr_out10_dev, void* stream);
int go2sim_env_set_global_dr_dev(go2sim_t* h, const double* dr10_dev, void* stream);
/* zero-copy address of the live go2sim_env_globals_t (device memory for the HIP library): lets the host shim expose
 * extras["episode"] / extras["curriculum"] (go2_env_Omni_walk_16output.py:674-690, 1229-1234) as device tensors
 * without a stream synchronisation. */
int go2sim_env_globals_ptr(go2sim_t* h, void** ptr_out);

/* diagnostics: ONE narrow-phase query on explicit world poses of geoms i_ga, i_gb (host pointers: pos[3], quat[4] wxyz; out8 = {is_col,
 * penetration, normal[3], pos[3]}), evaluated by the library's own narrow-phase code: which = 0 MPR from a cold start (collider/mpr.py:763-819),
 * 1 safe GJK + EPA by one lane with the LDS polytope slot (collider/gjk.py:161-437), 2 the same on the full-capacity polytope record, 3 / 4 the
 * cooperative query the collision kernel runs (the 4 lanes of a quad on one query) with an LDS slot / on the full-capacity record, 5 / 6 the same
 * with 16 lanes per query.  Synchronous.
 * Lets the closed-form checks of tests/test_gjk_epa.py run against the HIP implementation, not only against the oracle. */
int go2sim_debug_narrowphase(go2sim_t* h, int which, int i_ga, int i_gb, const float* pos_a, const float* quat_a, const float* pos_b,
                             const float* quat_b, float* out8);

/* hipEvent-timed duration (ms) of each kernel class accumulated since the last call with reset=1;
 * out[0..7] = dyn, collide, solve, integrate, env_pre, env_post, misc, total ; counts in cnt[0..7].
 * Used by bench.py for the live roofline measurement. Returns GO2SIM_E_BADARG if timing is disabled. */
int go2sim_enable_timing(go2sim_t* h, int enable);
int go2sim_read_timing(go2sim_t* h, float* ms_out8, int* cnt_out8, int reset);

/* ---- CPU twin (test oracle; oracle/libgo2sim_cpu.so).  Same semantics, host pointers, the `stream`
 * argument is ignored.  NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it. */
#define GO2SIM_CPU_DECL(name) go2sim_cpu_##name

#ifdef __cplusplus
}
#endif
#endif /* GO2SIM_H */
