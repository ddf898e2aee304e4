// go2sim_cpu.cpp -- CPU ORACLE (test infrastructure, NOT part of the product).
//
// Plain C++ restatement of the reference hot path
//   Go2Env.step -> gs.Scene.step -> RigidSolver.substep -> Go2Env observation/reward/reset
// of saifahmadgit/go2-sim2real-locomotion-rl (Genesis v0.4.0 fork), float32, one env at a time,
// same loop and summation order as the reference's serial (`backend == gs.cpu`) branches.
// Every function cites the reference file:line it follows (paths relative to the reference root,
// `R/` = genesis/engine/solvers/rigid/).
//
// PARITY STATUS: "parity unpinned".  The reference cannot be executed in this pipeline (quadrants,
// mujoco, trimesh, numba absent; no network) and ships no golden vectors for this path (SURVEY.md
// section 8c), so this oracle is pinned only by the analytic known-answer tests re-expressed in
// tests/ (free fall, static weight = contact force, unit quaternions, mass-matrix symmetry/energy, ...).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// Exports the go2sim_cpu_* twin of include/go2sim.h (host pointers everywhere).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/go2sim.h"
#include "../include/go2sim_detmath.h"
#include "gjk_epa_cpu.h"

namespace {

typedef float real;
constexpr int NL = GO2SIM_NL, ND = GO2SIM_ND, NQ = GO2SIM_NQ, NG = GO2SIM_NG, NJ = GO2SIM_NJ;
constexpr int NPAIR = GO2SIM_NPAIR_MAX, MAXC = GO2SIM_MAX_CONTACTS, MAXB = GO2SIM_MAX_BROAD, MAXR = GO2SIM_MAX_ROWS;
constexpr int JOINT_FIXED = 0, JOINT_REVOLUTE = 1, JOINT_FREE = 4;
constexpr int GEOM_SPHERE = 1, GEOM_CYLINDER = 3, GEOM_BOX = 5, GEOM_TERRAIN = 7;
constexpr int CTRL_FORCE = 0, CTRL_VELOCITY = 1, CTRL_POSITION = 2;

// ---------------------------------------------------------------------------------------------
// vector math with the evaluation order of the quadrants/Taichi vector ops used by the reference
// ---------------------------------------------------------------------------------------------
struct V3 { real x, y, z; };
struct Q4 { real w, x, y, z; };
struct M3 { real m[3][3]; };

inline V3 v3(real x, real y, real z) { V3 r = {x, y, z}; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, real s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(real s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, real s) { return v3(a.x / s, a.y / s, a.z / s); }
inline real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline real norm_sqr(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline real norm(V3 a) { return dm_sqrt(norm_sqr(a)); }
inline V3 normalized(V3 a) { real inv = 1.0f / norm(a); return inv * a; }  // taichi Vector.normalized
inline real vget(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline void vset(V3& a, int i, real v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
inline V3 vmin(V3 a, V3 b) { return v3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }
inline real clampf(real x, real lo, real hi) { return std::min(hi, std::max(lo, x)); }
inline bool isnanf_(real x) { return x != x; }

inline Q4 q4(real w, real x, real y, real z) { Q4 r = {w, x, y, z}; return r; }
inline real norm_sqr(Q4 q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Q4 operator*(Q4 q, real s) { return q4(q.w * s, q.x * s, q.y * s, q.z * s); }
inline Q4 qident() { return q4(1.0f, 0.0f, 0.0f, 0.0f); }
inline Q4 inv_quat(Q4 q) { return q4(q.w, -q.x, -q.y, -q.z); }  // geom.py:218

inline V3 mul(const M3& A, V3 v) {
  return v3(A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z);
}
inline M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
inline M3 transpose(const M3& A) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i];
  return C;
}
inline M3 operator+(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][j] + B.m[i][j];
  return C;
}
inline V3 mcol(const M3& A, int j) { return v3(A.m[0][j], A.m[1][j], A.m[2][j]); }

// geom.py:236-242
inline Q4 quat_mul(Q4 u, Q4 v) {
  real w = u.w * v.w - u.x * v.x - u.y * v.y - u.z * v.z;
  real x = u.w * v.x + u.x * v.w + u.y * v.z - u.z * v.y;
  real y = u.w * v.y - u.x * v.z + u.y * v.w + u.z * v.x;
  real z = u.w * v.z + u.x * v.y - u.y * v.x + u.z * v.w;
  return q4(w, x, y, z);
}
// geom.py:245-252  (quat_mul(u, v) normalised)
inline Q4 transform_quat_by_quat(Q4 v, Q4 u) {
  Q4 q = quat_mul(u, v);
  real inv = 1.0f / dm_sqrt(norm_sqr(q));
  return q4(inv * q.w, inv * q.x, inv * q.y, inv * q.z);
}
// geom.py:255-270
inline V3 transform_by_quat(V3 v, Q4 q) {
  real q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_wx = q.x * q.w;
  real q_yy = q.y * q.y, q_yz = q.y * q.z, q_wy = q.y * q.w;
  real q_zz = q.z * q.z, q_wz = q.z * q.w;
  real q_ww = q.w * q.w;
  V3 r = v3(v.x * (q_xx + q_ww - q_yy - q_zz) + v.y * (2.0f * q_xy - 2.0f * q_wz) + v.z * (2.0f * q_xz + 2.0f * q_wy),
            v.x * (2.0f * q_wz + 2.0f * q_xy) + v.y * (q_ww - q_xx + q_yy - q_zz) + v.z * (-2.0f * q_wx + 2.0f * q_yz),
            v.x * (-2.0f * q_wy + 2.0f * q_xz) + v.y * (2.0f * q_wx + 2.0f * q_yz) + v.z * (q_ww - q_xx - q_yy + q_zz));
  return r / (q_ww + q_xx + q_yy + q_zz);
}
inline V3 inv_transform_by_quat(V3 v, Q4 q) { return transform_by_quat(v, inv_quat(q)); }       // geom.py:273
inline V3 transform_by_trans_quat(V3 p, V3 t, Q4 q) { return transform_by_quat(p, q) + t; }      // geom.py:288
// geom.py:313-317
inline void transform_pos_quat_by_trans_quat(V3 pos, Q4 quat, V3 t_trans, Q4 t_quat, V3& opos, Q4& oquat) {
  opos = t_trans + transform_by_quat(pos, t_quat);
  oquat = transform_quat_by_quat(quat, t_quat);
}
// geom.py:136-161
inline M3 quat_to_R(Q4 q, real eps) {
  M3 R = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  real d = norm_sqr(q);
  if (d > eps) {
    real s = 2.0f / d;
    real xs = q.x * s, ys = q.y * s, zs = q.z * s;
    real wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    real xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    real yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    R.m[0][0] = 1.0f - (yy + zz); R.m[0][1] = xy - wz; R.m[0][2] = xz + wy;
    R.m[1][0] = xy + wz; R.m[1][1] = 1.0f - (xx + zz); R.m[1][2] = yz - wx;
    R.m[2][0] = xz - wy; R.m[2][1] = yz + wx; R.m[2][2] = 1.0f - (xx + yy);
  }
  return R;
}
// geom.py:110-133
inline Q4 rotvec_to_quat(V3 rv, real eps) {
  Q4 q = q4(0, 0, 0, 0);
  real thetasq = norm_sqr(rv);
  if (thetasq > eps * eps) {
    real theta = dm_sqrt(thetasq);
    real theta_half = 0.5f * theta;
    real s, c;
    dm_sincos(theta_half, &s, &c);
    q.w = c;
    V3 xyz = (s / theta) * rv;
    q.x = xyz.x; q.y = xyz.y; q.z = xyz.z;
    real k = 0.5f * (3.0f - norm_sqr(q));
    q = q * k;
  } else {
    q.w = 1.0f;
  }
  return q;
}
// geom.py:320-336
inline void transform_inertia_by_trans_quat(const M3& I, real mass, V3 t, Q4 quat, real eps, M3& oI, V3& opos) {
  real xx = t.x * t.x, xy = t.x * t.y, xz = t.x * t.z, yy = t.y * t.y, yz = t.y * t.z, zz = t.z * t.z;
  M3 hhT = {{{yy + zz, -xy, -xz}, {-xy, xx + zz, -yz}, {-xz, -yz, xx + yy}}};
  M3 R = quat_to_R(quat, eps);
  M3 RI = mul(mul(R, I), transpose(R));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) oI.m[i][j] = RI.m[i][j] + hhT.m[i][j] * mass;
  opos = t * mass;
}
// geom.py:365-383
inline void inertial_mul(V3 pos, const M3& I, real mass, V3 vel, V3 ang, V3& oang, V3& ovel) {
  oang = mul(I, ang) + cross(pos, vel);
  ovel = mass * vel - cross(pos, ang);
}
inline void motion_cross_force(V3 m_ang, V3 m_vel, V3 f_ang, V3 f_vel, V3& oang, V3& ovel) {
  ovel = cross(m_ang, f_vel);
  oang = cross(m_ang, f_ang) + cross(m_vel, f_vel);
}
inline void motion_cross_motion(V3 s_ang, V3 s_vel, V3 m_ang, V3 m_vel, V3& oang, V3& ovel) {
  ovel = cross(s_ang, m_vel) + cross(s_vel, m_ang);
  oang = cross(s_ang, m_ang);
}
// geom.py:386-401
inline void orthogonals(V3 a, V3& b, V3& c) {
  if (dm_abs(a.y) < 0.5f) {
    b = v3(-a.x * a.y, 1.0f - a.y * a.y, -a.z * a.y);
  } else {
    b = v3(-a.x * a.z, -a.y * a.z, 1.0f - a.z * a.z);
  }
  b = normalized(b);
  c = cross(a, b);
}
// geom.py:404-422
inline void imp_aref(const real* p, real neg_penetration, real vel, real pos, real& imp, real& aref) {
  real timeconst = p[0], dampratio = p[1], dmin = p[2], dmax = p[3], width = p[4], mid = p[5], power = p[6];
  real imp_x = dm_abs(neg_penetration) / width;
  real imp_a = (1.0f / dm_pow(mid, power - 1.0f)) * dm_pow(imp_x, power);
  real imp_b = 1.0f - (1.0f / dm_pow(1.0f - mid, power - 1.0f)) * dm_pow(1.0f - imp_x, power);
  real imp_y = (imp_x < mid) ? imp_a : imp_b;
  imp = dmin + imp_y * (dmax - dmin);
  imp = clampf(imp, dmin, dmax);
  imp = (imp_x > 1.0f) ? dmax : imp;
  real b = 2.0f / (dmax * timeconst);
  real k = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  aref = -b * vel - k * imp * pos;
}

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
struct Link {
  int parent, root, entity, is_fixed, joint_start, joint_end, dof_start, dof_end, q_start, q_end, n_dofs, geom_start, geom_end;
  V3 pos; Q4 quat; V3 inertial_pos; Q4 inertial_quat; M3 inertial_i; real mass; real invweight[2];
};
struct Joint { int type, link, q_start, dof_start, dof_end; V3 pos; real sol_params[7]; };
struct Dof {
  V3 motion_ang, motion_vel; real limit[2], invweight, armature, damping, stiffness, frictionloss, kp, kv, force_range[2];
};
struct Geom {
  int type, link, is_convex; V3 pos; Q4 quat; real data[7], friction, sol_params[7]; V3 center; V3 aabb[8]; real rim[32][2];
};
struct Entity { int link_start, link_end, dof_start, dof_end, geom_start, geom_end; };
struct Model {
  int n_links, n_joints, n_dofs, n_qs, n_geoms, n_entities, n_pairs, max_collision_pairs, max_contact_pairs, max_broad_pairs,
      n_contacts_per_pair, iterations, ls_iterations, ccd_iterations, support_res;
  real substep_dt; V3 gravity; real eps, tolerance, ls_tolerance, meaninertia, mc_perturbation, mc_tolerance, mpr_to_gjk_ratio,
      ccd_eps, ccd_tolerance;
  Link links[NL]; Joint joints[NJ]; Dof dofs[ND]; Geom geoms[NG]; Entity entities[2];
  real qpos0[NQ]; real mass_parent_mask[ND][ND]; int pair_idx[NG][NG]; int theta_to_ring[180];
  int arrow_mode;   // derived (dm_arrow_mode): numbering of the four leg chains, 0 = no arrow form
  // heightfield terrain replacing the ground slab (go2sim_cpu_set_terrain; collider.py:374-394)
  int terrain_enabled, terrain_rows, terrain_cols; real terrain_hs; real terrain_xyz_maxmin[6]; std::vector<real> terrain_hf;
};

bool parse_model(const void* blob, size_t nbytes, Model& m) {
  m.terrain_enabled = 0; m.terrain_rows = m.terrain_cols = 0; m.terrain_hs = 0.0f;
  if (nbytes < 128) return false;
  const int32_t* H = (const int32_t*)blob;
  if (H[0] != GO2SIM_MODEL_MAGIC || H[1] != GO2SIM_MODEL_VERSION) return false;
  m.n_links = H[2]; m.n_joints = H[3]; m.n_dofs = H[4]; m.n_qs = H[5]; m.n_geoms = H[6]; m.n_entities = H[7];
  m.n_pairs = H[8]; m.max_collision_pairs = H[9]; m.max_contact_pairs = H[10]; m.max_broad_pairs = H[11];
  m.n_contacts_per_pair = H[12]; m.iterations = H[13]; m.ls_iterations = H[14]; m.ccd_iterations = H[15]; m.support_res = H[16];
  int nf = H[18], ni = H[19];
  if (m.n_links != NL || m.n_joints != NJ || m.n_dofs != ND || m.n_qs != NQ || m.n_geoms != NG || m.n_entities != 2) return false;
  if (m.n_pairs > NPAIR || m.max_contact_pairs > MAXC || m.max_broad_pairs > MAXB || m.support_res != 180 || H[17] != 32) return false;
  if (nbytes < 128 + (size_t)4 * (nf + ni)) return false;
  const float* F = (const float*)((const char*)blob + 128);
  const int32_t* I = (const int32_t*)(F + nf);
  const float* f = F;
  m.substep_dt = f[0]; m.gravity = v3(f[1], f[2], f[3]); m.eps = f[4]; m.tolerance = f[5]; m.ls_tolerance = f[6];
  m.meaninertia = f[7]; m.mc_perturbation = f[8]; m.mc_tolerance = f[9]; m.mpr_to_gjk_ratio = f[10]; m.ccd_eps = f[11];
  m.ccd_tolerance = f[12];
  f += 16;
  for (int i = 0; i < NL; ++i, f += 26) {
    Link& l = m.links[i];
    l.pos = v3(f[0], f[1], f[2]); l.quat = q4(f[3], f[4], f[5], f[6]); l.inertial_pos = v3(f[7], f[8], f[9]);
    l.inertial_quat = q4(f[10], f[11], f[12], f[13]);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) l.inertial_i.m[a][b] = f[14 + 3 * a + b];
    l.mass = f[23]; l.invweight[0] = f[24]; l.invweight[1] = f[25];
  }
  for (int i = 0; i < NJ; ++i, f += 10) {
    m.joints[i].pos = v3(f[0], f[1], f[2]);
    for (int a = 0; a < 7; ++a) m.joints[i].sol_params[a] = f[3 + a];
  }
  for (int i = 0; i < ND; ++i, f += 17) {
    Dof& d = m.dofs[i];
    d.motion_ang = v3(f[0], f[1], f[2]); d.motion_vel = v3(f[3], f[4], f[5]); d.limit[0] = f[6]; d.limit[1] = f[7];
    d.invweight = f[8]; d.armature = f[9]; d.damping = f[10]; d.stiffness = f[11]; d.frictionloss = f[12];
    d.kp = f[13]; d.kv = f[14]; d.force_range[0] = f[15]; d.force_range[1] = f[16];
  }
  for (int i = 0; i < NQ; ++i) m.qpos0[i] = f[i];
  f += NQ;
  for (int i = 0; i < NG; ++i, f += 113) {
    Geom& g = m.geoms[i];
    g.pos = v3(f[0], f[1], f[2]); g.quat = q4(f[3], f[4], f[5], f[6]);
    for (int a = 0; a < 7; ++a) g.data[a] = f[7 + a];
    g.friction = f[14];
    for (int a = 0; a < 7; ++a) g.sol_params[a] = f[15 + a];
    g.center = v3(f[22], f[23], f[24]);
    for (int a = 0; a < 8; ++a) g.aabb[a] = v3(f[25 + 3 * a], f[26 + 3 * a], f[27 + 3 * a]);
    for (int a = 0; a < 32; ++a) { g.rim[a][0] = f[49 + 2 * a]; g.rim[a][1] = f[50 + 2 * a]; }
  }
  for (int i = 0; i < ND; ++i) for (int j = 0; j < ND; ++j) m.mass_parent_mask[i][j] = f[i * ND + j];
  m.arrow_mode = getenv("GO2SIM_NO_ARROW") ? 0 : dm_arrow_mode(&m.mass_parent_mask[0][0], ND);   // (GO2SIM_NO_ARROW=1: diagnostic switch, keeps the row-form factorisation under test)
  f += ND * ND;
  if (f - F != nf) return false;
  const int32_t* p = I;
  for (int i = 0; i < NL; ++i, p += 13) {
    Link& l = m.links[i];
    l.parent = p[0]; l.root = p[1]; l.entity = p[2]; l.is_fixed = p[3]; l.joint_start = p[4]; l.joint_end = p[5];
    l.dof_start = p[6]; l.dof_end = p[7]; l.q_start = p[8]; l.q_end = p[9]; l.n_dofs = p[10]; l.geom_start = p[11]; l.geom_end = p[12];
  }
  for (int i = 0; i < NJ; ++i, p += 5) {
    Joint& j = m.joints[i];
    j.type = p[0]; j.link = p[1]; j.q_start = p[2]; j.dof_start = p[3]; j.dof_end = p[4];
  }
  for (int i = 0; i < NG; ++i, p += 3) { m.geoms[i].type = p[0]; m.geoms[i].link = p[1]; m.geoms[i].is_convex = p[2]; }
  for (int i = 0; i < 2; ++i, p += 6) {
    Entity& e = m.entities[i];
    e.link_start = p[0]; e.link_end = p[1]; e.dof_start = p[2]; e.dof_end = p[3]; e.geom_start = p[4]; e.geom_end = p[5];
  }
  for (int i = 0; i < NG; ++i) for (int j = 0; j < NG; ++j) m.pair_idx[i][j] = p[i * NG + j];
  p += NG * NG;
  for (int i = 0; i < 180; ++i) m.theta_to_ring[i] = p[i];
  p += 180;
  return (p - I) == ni;
}

// ---------------------------------------------------------------------------------------------
// per-env state (AoS; the product uses SoA [feature][n_envs])
// ---------------------------------------------------------------------------------------------
struct Contact {
  int geom_a, geom_b, link_a, link_b; V3 pos, normal, force; real penetration, friction, sol_params[7];
};
// FAST ORDER, arrow form of a Cholesky factor (arrow_factor below): per leg its dofs, the reciprocal pivots, l10 l20 l21 and W (6 x 3); the base factor with
// reciprocal pivots on its diagonal
struct ArrowFactor { int p[4][3]; real i[4][3], l[4][3], w[4][6][3], b[6][6]; };
struct Env {
  // persistent rigid state
  real qpos[NQ], vel[ND], acc[ND], qacc_ws[ND];
  int is_warmstart, err, first_time;
  real ctrl_force[ND], ctrl_pos[ND], ctrl_vel[ND]; int ctrl_mode[ND];
  V3 ext_ang[NL], ext_vel[NL];
  real mass_shift[NL]; V3 com_shift[NL]; real friction_ratio[NG]; real geom_friction[NG];
  real sort_value[2 * NG]; int sort_ig[2 * NG]; int sort_ismax[2 * NG]; int active_buf[NG];
  V3 normal_cache[NPAIR];
  // kinematics
  V3 l_pos[NL]; Q4 l_quat[NL]; V3 i_pos_bw[NL], i_pos[NL]; Q4 i_quat[NL]; V3 root_com_bw[NL], root_com[NL]; real mass_sum[NL];
  M3 cinr_inertial[NL]; V3 cinr_pos[NL]; real cinr_mass[NL];
  V3 xanchor[NJ], xaxis[NJ];
  real dof_pos[ND];
  V3 cdof_ang[ND], cdof_vel[ND], cdofd_ang[ND], cdofd_vel[ND];
  V3 cd_vel[NL], cd_ang[NL];
  V3 g_pos[NG]; Q4 g_quat[NG]; V3 aabb_min[NG], aabb_max[NG];
  // dynamics
  M3 crb_inertial[NL]; V3 crb_pos[NL]; real crb_mass[NL];
  V3 f_ang[ND], f_vel[ND];
  real mass_mat[ND][ND], mass_L[ND][ND], mass_Dinv[ND];
  real qf_applied[ND], qf_passive[ND], qf_bias[ND], force[ND], qf_smooth[ND], acc_smooth[ND], qf_constraint[ND];
  V3 cdd_vel[NL], cdd_ang[NL], cfrc_vel[NL], cfrc_ang[NL];
  // collision
  int n_broad; int broad[MAXB][2];
  int n_contacts; Contact contacts[MAXC];
  V3 mpr_v[4], mpr_v1[4], mpr_v2[4];
  V3 prism[6]; real xyz_max_min[6];   // collider_state.prism / xyz_max_min of the terrain narrow phase
  int gjk_fallback_count;  // number of pairs that switched from MPR to the safe GJK + EPA
  // constraints
  int n_con;
  real jac[MAXR][ND], diag[MAXR], aref[MAXR], efc_D[MAXR], Jaref[MAXR], jv[MAXR], efc_force[MAXR];
  int active[MAXR], prev_active[MAXR];
  real qacc[ND], Ma[ND], grad[ND], Mgrad[ND], search[ND], mv[ND], qfrc_constraint[ND], nt_vec[ND];
  real H[ND][ND];
  real Hunf[ND][ND];   // FAST ORDER: the unfactored Hessian of the running Newton solve (lower triangle), updated by the rows that flip
  // FAST ORDER, arrow form of a factor (arrow_factor)
  bool arrow; ArrowFactor af;      // ... of the Newton Hessian
  bool mass_arrow; ArrowFactor maf;   // ... of the mass matrix (factor_mass / solve_mass)
  real cost, prev_cost, gauss, quad_gauss[3], gtol; int ls_it, ls_result, improved, solver_iters;
  V3 contact_force[NL];
  real vel_next[ND], qpos_next[NQ];
};

// ---------------------------------------------------------------------------------------------
// kinematics  (R/abd/forward_kinematics.py)
// ---------------------------------------------------------------------------------------------
// func_forward_kinematics_entity, forward_kinematics.py:463-618
void forward_kinematics_entity(const Model& m, Env& e, int i_e) {
  const Entity& en = m.entities[i_e];
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
    const Link& L = m.links[i_l];
    V3 pos = L.pos; Q4 quat = L.quat;
    if (L.parent != -1) {
      pos = e.l_pos[L.parent] + transform_by_quat(L.pos, e.l_quat[L.parent]);
      quat = transform_quat_by_quat(L.quat, e.l_quat[L.parent]);
    }
    for (int i_j = L.joint_start; i_j < L.joint_end; ++i_j) {
      const Joint& J = m.joints[i_j];
      int q_start = J.q_start, dof_start = J.dof_start;
      if (J.type == JOINT_FREE) {
        e.xanchor[i_j] = v3(e.qpos[q_start], e.qpos[q_start + 1], e.qpos[q_start + 2]);
        e.xaxis[i_j] = v3(0, 0, 1);
        V3 pos_ = v3(e.qpos[q_start], e.qpos[q_start + 1], e.qpos[q_start + 2]);
        Q4 quat_ = q4(e.qpos[q_start + 3], e.qpos[q_start + 4], e.qpos[q_start + 5], e.qpos[q_start + 6]);
        real n = dm_sqrt(norm_sqr(quat_));
        quat_ = q4(quat_.w / n, quat_.x / n, quat_.y / n, quat_.z / n);
        pos = pos_; quat = quat_;
        // dofs_state.pos of the free joint: linear part = position; the Euler-angle part
        // (forward_kinematics.py:571-574) feeds only POSITION control of the base / set_dofs_position
        // round trips, neither used on this path; it is evaluated lazily by the getter.
        e.dof_pos[dof_start + 0] = pos.x; e.dof_pos[dof_start + 1] = pos.y; e.dof_pos[dof_start + 2] = pos.z;
      } else if (J.type == JOINT_REVOLUTE) {
        V3 axis = m.dofs[dof_start].motion_ang;
        e.xanchor[i_j] = transform_by_quat(J.pos, quat) + pos;
        e.xaxis[i_j] = transform_by_quat(axis, quat);
        e.dof_pos[dof_start] = e.qpos[q_start] - m.qpos0[q_start];
        Q4 qloc = rotvec_to_quat(axis * e.dof_pos[dof_start], m.eps);
        quat = transform_quat_by_quat(qloc, quat);
        pos = e.xanchor[i_j] - transform_by_quat(J.pos, quat);
      }
    }
    if (!(L.parent == -1 && L.is_fixed)) { e.l_pos[i_l] = pos; e.l_quat[i_l] = quat; }
  }
}

// func_COM_links_entity, forward_kinematics.py:224-459
void com_links_entity(const Model& m, Env& e, int i_e) {
  const Entity& en = m.entities[i_e];
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) { e.root_com_bw[i_l] = v3(0, 0, 0); e.mass_sum[i_l] = 0.0f; }
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
    const Link& L = m.links[i_l];
    real mass = L.mass + e.mass_shift[i_l];
    transform_pos_quat_by_trans_quat(L.inertial_pos + e.com_shift[i_l], L.inertial_quat, e.l_pos[i_l], e.l_quat[i_l], e.i_pos_bw[i_l],
                                     e.i_quat[i_l]);
    int i_r = L.root;
    e.mass_sum[i_r] = e.mass_sum[i_r] + mass;
    e.root_com_bw[i_r] = e.root_com_bw[i_r] + mass * e.i_pos_bw[i_l];
  }
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l)
    if (m.links[i_l].root == i_l) e.root_com[i_l] = e.root_com_bw[i_l] / e.mass_sum[i_l];
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) e.root_com[i_l] = e.root_com[m.links[i_l].root];
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
    const Link& L = m.links[i_l];
    e.i_pos[i_l] = e.i_pos_bw[i_l] - e.root_com[i_l];
    real i_mass = L.mass + e.mass_shift[i_l];
    transform_inertia_by_trans_quat(L.inertial_i, i_mass, e.i_pos[i_l], e.i_quat[i_l], m.eps, e.cinr_inertial[i_l], e.cinr_pos[i_l]);
    e.cinr_mass[i_l] = i_mass;
  }
  // (j_pos / j_quat, forward_kinematics.py:332-396, are not consumed on this path)
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
    const Link& L = m.links[i_l];
    if (L.n_dofs == 0) continue;
    for (int i_j = L.joint_start; i_j < L.joint_end; ++i_j) {
      const Joint& J = m.joints[i_j];
      V3 offset_pos = e.root_com[i_l] - e.xanchor[i_j];
      int ds = J.dof_start;
      if (J.type == JOINT_REVOLUTE) {
        e.cdof_ang[ds] = e.xaxis[i_j];
        e.cdof_vel[ds] = cross(e.xaxis[i_j], offset_pos);
      } else if (J.type == JOINT_FREE) {
        for (int i = 0; i < 3; ++i) {
          e.cdof_ang[i + ds] = v3(0, 0, 0);
          e.cdof_vel[i + ds] = v3(0, 0, 0);
          vset(e.cdof_vel[i + ds], i, 1.0f);
        }
        M3 xmat_T = transpose(quat_to_R(e.l_quat[i_l], m.eps));
        for (int i = 0; i < 3; ++i) {
          V3 row = v3(xmat_T.m[i][0], xmat_T.m[i][1], xmat_T.m[i][2]);
          e.cdof_ang[i + ds + 3] = row;
          e.cdof_vel[i + ds + 3] = cross(row, offset_pos);
        }
      }
      // cdofvel_* (forward_kinematics.py:447-459) is not consumed on this path
    }
  }
}

// func_update_geoms_entity, forward_kinematics.py:709-744
void update_geoms_entity(const Model& m, Env& e, int i_e, bool force_update_fixed) {
  const Entity& en = m.entities[i_e];
  for (int i_g = en.geom_start; i_g < en.geom_end; ++i_g) {
    const Geom& G = m.geoms[i_g];
    bool is_fixed = m.links[G.link].is_fixed;
    if (force_update_fixed || !is_fixed)
      transform_pos_quat_by_trans_quat(G.pos, G.quat, e.l_pos[G.link], e.l_quat[G.link], e.g_pos[i_g], e.g_quat[i_g]);
  }
}

// func_forward_velocity_entity, forward_kinematics.py:871-994
void forward_velocity_entity(const Model& m, Env& e, int i_e) {
  const Entity& en = m.entities[i_e];
  for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
    const Link& L = m.links[i_l];
    V3 cvel_vel = v3(0, 0, 0), cvel_ang = v3(0, 0, 0);
    if (L.parent != -1) { cvel_vel = e.cd_vel[L.parent]; cvel_ang = e.cd_ang[L.parent]; }
    for (int i_j = L.joint_start; i_j < L.joint_end; ++i_j) {
      const Joint& J = m.joints[i_j];
      int ds = J.dof_start;
      if (J.type == JOINT_FREE) {
        for (int i = 0; i < 3; ++i) {
          cvel_vel = cvel_vel + e.cdof_vel[ds + i] * e.vel[ds + i];
          cvel_ang = cvel_ang + e.cdof_ang[ds + i] * e.vel[ds + i];
        }
        for (int i = 0; i < 3; ++i) {
          e.cdofd_ang[ds + i] = v3(0, 0, 0); e.cdofd_vel[ds + i] = v3(0, 0, 0);
          motion_cross_motion(cvel_ang, cvel_vel, e.cdof_ang[ds + i + 3], e.cdof_vel[ds + i + 3], e.cdofd_ang[ds + i + 3],
                              e.cdofd_vel[ds + i + 3]);
        }
        for (int i = 0; i < 3; ++i) {
          cvel_vel = cvel_vel + e.cdof_vel[ds + i + 3] * e.vel[ds + i + 3];
          cvel_ang = cvel_ang + e.cdof_ang[ds + i + 3] * e.vel[ds + i + 3];
        }
      } else {
        for (int i_d = ds; i_d < J.dof_end; ++i_d)
          motion_cross_motion(cvel_ang, cvel_vel, e.cdof_ang[i_d], e.cdof_vel[i_d], e.cdofd_ang[i_d], e.cdofd_vel[i_d]);
        for (int i_d = ds; i_d < J.dof_end; ++i_d) {
          cvel_vel = cvel_vel + e.cdof_vel[i_d] * e.vel[i_d];
          cvel_ang = cvel_ang + e.cdof_ang[i_d] * e.vel[i_d];
        }
      }
    }
    e.cd_vel[i_l] = cvel_vel; e.cd_ang[i_l] = cvel_ang;
  }
}

// func_update_cartesian_space + func_forward_velocity (forward_kinematics.py:1494-1566, 1048-1088)
void update_cartesian_space(const Model& m, Env& e, bool force_update_fixed) {
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    forward_kinematics_entity(m, e, i_e);
    com_links_entity(m, e, i_e);
    update_geoms_entity(m, e, i_e, force_update_fixed);
  }
}
void forward_velocity(const Model& m, Env& e) {
  for (int i_e = 0; i_e < m.n_entities; ++i_e) forward_velocity_entity(m, e, i_e);
}

// ---------------------------------------------------------------------------------------------
// forward dynamics  (R/abd/forward_dynamics.py)
// ---------------------------------------------------------------------------------------------
// func_compute_mass_matrix, forward_dynamics.py:291-541 (implicit_damping = approximate_implicitfast)
void compute_mass_matrix(const Model& m, Env& e, bool implicit_damping) {
  for (int i_l = 0; i_l < NL; ++i_l) {
    e.crb_inertial[i_l] = e.cinr_inertial[i_l]; e.crb_pos[i_l] = e.cinr_pos[i_l]; e.crb_mass[i_l] = e.cinr_mass[i_l];
  }
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    int n = en.link_end - en.link_start;
    for (int i = 0; i < n; ++i) {
      int i_l = en.link_end - 1 - i, i_p = m.links[i_l].parent;
      if (i_p != -1) {
        e.crb_inertial[i_p] = e.crb_inertial[i_p] + e.crb_inertial[i_l];
        e.crb_mass[i_p] = e.crb_mass[i_p] + e.crb_mass[i_l];
        e.crb_pos[i_p] = e.crb_pos[i_p] + e.crb_pos[i_l];
      }
    }
  }
  for (int i_l = 0; i_l < NL; ++i_l)
    for (int i_d = m.links[i_l].dof_start; i_d < m.links[i_l].dof_end; ++i_d)
      inertial_mul(e.crb_pos[i_l], e.crb_inertial[i_l], e.crb_mass[i_l], e.cdof_vel[i_d], e.cdof_ang[i_d], e.f_ang[i_d], e.f_vel[i_d]);
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    for (int i_d = en.dof_start; i_d < en.dof_end; ++i_d)
      for (int j_d = en.dof_start; j_d < en.dof_end; ++j_d)
        e.mass_mat[i_d][j_d] = (dot(e.f_ang[i_d], e.cdof_ang[j_d]) + dot(e.f_vel[i_d], e.cdof_vel[j_d])) * m.mass_parent_mask[i_d][j_d];
    for (int i_d = en.dof_start; i_d < en.dof_end; ++i_d)
      for (int j_d = i_d + 1; j_d < en.dof_end; ++j_d) e.mass_mat[i_d][j_d] = e.mass_mat[j_d][i_d];
  }
  for (int i_d = 0; i_d < ND; ++i_d) e.mass_mat[i_d][i_d] = e.mass_mat[i_d][i_d] + m.dofs[i_d].armature;
  if (implicit_damping) {
    for (int i_d = 0; i_d < ND; ++i_d) {
      e.mass_mat[i_d][i_d] = e.mass_mat[i_d][i_d] + m.dofs[i_d].damping * m.substep_dt;
      if (e.ctrl_mode[i_d] == CTRL_POSITION || e.ctrl_mode[i_d] == CTRL_VELOCITY)
        e.mass_mat[i_d][i_d] = e.mass_mat[i_d][i_d] + m.dofs[i_d].kv * m.substep_dt;
    }
  }
}

#ifdef GO2SIM_FAST_ORDER
// arrow_factor / arrow_solve of csrc/go2sim.hip: a symmetric positive definite 18 x 18 matrix (lower triangle in A) whose leg blocks (dofs dm_arrow_dof(mode, l, 0..2))
// couple only to the six base dofs: the legs are eliminated first -- four 3 x 3 factorisations, W_l = C_l^T L_l^-T, the 6 x 6 Schur complement of the base with
// the legs added as (l0 + l2) + (l1 + l3) -- with reciprocal pivots sqrt(e) * (1 / e) and fused multiply-adds
void arrow_factor(int mode, real eps, const real (*A)[ND], ArrowFactor& f) {
  real sc[4][6][6];
  for (int l = 0; l < 4; ++l) {
    const int p0 = dm_arrow_dof(mode, l, 0), p1 = dm_arrow_dof(mode, l, 1), p2 = dm_arrow_dof(mode, l, 2);
    const real a00 = A[p0][p0], a10 = A[p1][p0], a11 = A[p1][p1], a20 = A[p2][p0], a21 = A[p2][p1], a22 = A[p2][p2];
    const real e0 = std::max(a00, eps), i0 = dm_sqrt(e0) * (1.0f / e0);
    const real l10 = a10 * i0, l20 = a20 * i0;
    const real e1 = std::max(std::fma(-l10, l10, a11), eps), i1 = dm_sqrt(e1) * (1.0f / e1);
    const real l21 = std::fma(-l20, l10, a21) * i1;
    const real e2 = std::max(std::fma(-l21, l21, std::fma(-l20, l20, a22)), eps), i2 = dm_sqrt(e2) * (1.0f / e2);
    f.p[l][0] = p0; f.p[l][1] = p1; f.p[l][2] = p2;
    f.i[l][0] = i0; f.i[l][1] = i1; f.i[l][2] = i2; f.l[l][0] = l10; f.l[l][1] = l20; f.l[l][2] = l21;
    for (int b = 0; b < 6; ++b) {
      const real c0 = A[p0][b], c1 = A[p1][b], c2 = A[p2][b];
      const real w0 = c0 * i0, w1 = std::fma(-w0, l10, c1) * i1, w2 = std::fma(-w1, l21, std::fma(-w0, l20, c2)) * i2;
      f.w[l][b][0] = w0; f.w[l][b][1] = w1; f.w[l][b][2] = w2;
    }
    for (int u = 0; u < 6; ++u)
      for (int j = 0; j < 6; ++j) sc[l][u][j] = std::fma(f.w[l][u][2], f.w[l][j][2], std::fma(f.w[l][u][1], f.w[l][j][1], f.w[l][u][0] * f.w[l][j][0]));
  }
  real a[6][6];
  for (int u = 0; u < 6; ++u)
    for (int j = 0; j <= u; ++j) a[u][j] = A[u][j] - ((sc[0][u][j] + sc[2][u][j]) + (sc[1][u][j] + sc[3][u][j]));
  for (int k = 0; k < 6; ++k) {
    const real ee = std::max(a[k][k], eps), ik = dm_sqrt(ee) * (1.0f / ee);
    a[k][k] = ik;
    for (int j = k + 1; j < 6; ++j) a[j][k] = a[j][k] * ik;
    for (int j = k + 1; j < 6; ++j)
      for (int i = k + 1; i <= j; ++i) a[j][i] = std::fma(-a[j][k], a[i][k], a[j][i]);
  }
  for (int k = 0; k < 6; ++k) for (int j = 0; j <= k; ++j) f.b[k][j] = a[k][j];
}
void arrow_solve(const ArrowFactor& f, const real* g, real* out) {
  real y[4][3], z[4][6], x[6];
  for (int l = 0; l < 4; ++l) {
    const int p0 = f.p[l][0], p1 = f.p[l][1], p2 = f.p[l][2];
    const real* iv = f.i[l]; const real* lv = f.l[l];
    y[l][0] = g[p0] * iv[0];
    y[l][1] = std::fma(-lv[0], y[l][0], g[p1]) * iv[1];
    y[l][2] = std::fma(-lv[2], y[l][1], std::fma(-lv[1], y[l][0], g[p2])) * iv[2];
    for (int b = 0; b < 6; ++b) z[l][b] = std::fma(f.w[l][b][2], y[l][2], std::fma(f.w[l][b][1], y[l][1], f.w[l][b][0] * y[l][0]));
  }
  for (int b = 0; b < 6; ++b) x[b] = g[b] - ((z[0][b] + z[2][b]) + (z[1][b] + z[3][b]));
  for (int k = 0; k < 6; ++k) {
    real acc = x[k];
    for (int j = 0; j < k; ++j) acc = std::fma(-f.b[k][j], x[j], acc);
    x[k] = acc * f.b[k][k];
  }
  for (int k = 5; k >= 0; --k) {
    real acc = x[k];
    for (int j = 5; j > k; --j) acc = std::fma(-f.b[j][k], x[j], acc);
    x[k] = acc * f.b[k][k];
  }
  real xl[4][3];
  for (int l = 0; l < 4; ++l) {
    const real* iv = f.i[l]; const real* lv = f.l[l];
    real v0 = y[l][0], v1 = y[l][1], v2 = y[l][2];
    for (int b = 0; b < 6; ++b) { v0 = std::fma(-f.w[l][b][0], x[b], v0); v1 = std::fma(-f.w[l][b][1], x[b], v1); v2 = std::fma(-f.w[l][b][2], x[b], v2); }
    xl[l][2] = v2 * iv[2]; xl[l][1] = std::fma(-lv[2], xl[l][2], v1) * iv[1]; xl[l][0] = std::fma(-lv[0], xl[l][1], std::fma(-lv[1], xl[l][2], v0)) * iv[0];
  }
  for (int b = 0; b < 6; ++b) out[b] = x[b];                       // (g and out may be the same array)
  for (int l = 0; l < 4; ++l) for (int t = 0; t < 3; ++t) out[f.p[l][t]] = xl[l][t];
}
#endif
// func_factor_mass, serial branch, forward_dynamics.py:560-604 (implicit_damping=False)
void factor_mass(const Model& m, Env& e) {
#ifdef GO2SIM_FAST_ORDER
  // tk_dynamics of csrc/go2sim.hip: the mass matrix of a floating base with four legs always has the arrow shape; acc_smooth comes from its arrow-form Cholesky factor
  e.mass_arrow = ND == 18 && m.arrow_mode != 0;
  if (e.mass_arrow) { arrow_factor(m.arrow_mode, m.eps, e.mass_mat, e.maf); return; }
#endif
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    int ds = en.dof_start, de = en.dof_end, n = de - ds;
    for (int i_d = ds; i_d < de; ++i_d)
      for (int j_d = ds; j_d < i_d + 1; ++j_d) e.mass_L[i_d][j_d] = e.mass_mat[i_d][j_d];
    for (int i_d_ = 0; i_d_ < n; ++i_d_) {
      int i_d = de - i_d_ - 1;
      real D_inv = 1.0f / e.mass_L[i_d][i_d];
      e.mass_Dinv[i_d] = D_inv;
      for (int j_d_ = 0; j_d_ < i_d - ds; ++j_d_) {
        int j_d = i_d - j_d_ - 1;
        real a = e.mass_L[i_d][j_d] * D_inv;
        for (int k_d = ds; k_d < j_d + 1; ++k_d) e.mass_L[j_d][k_d] -= a * e.mass_L[i_d][k_d];
        e.mass_L[i_d][j_d] = a;
      }
      e.mass_L[i_d][i_d] = 1.0f;
    }
  }
}

// func_solve_mass_entity, forward_dynamics.py:818-900
void solve_mass(const Model& m, const Env& e, const real* vec, real* out) {
#ifdef GO2SIM_FAST_ORDER
  if (e.mass_arrow) { arrow_solve(e.maf, vec, out); return; }
#endif
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    int ds = en.dof_start, de = en.dof_end, n = de - ds;
    for (int i_d_ = 0; i_d_ < n; ++i_d_) {
      int i_d = de - i_d_ - 1;
      real cur = vec[i_d];
      for (int j_d = i_d + 1; j_d < de; ++j_d) cur = cur - e.mass_L[j_d][i_d] * out[j_d];
      out[i_d] = cur;
    }
    for (int i_d = ds; i_d < de; ++i_d) out[i_d] = out[i_d] * e.mass_Dinv[i_d];
    for (int i_d = ds; i_d < de; ++i_d) {
      real cur = out[i_d];
      for (int j_d = ds; j_d < i_d; ++j_d) cur = cur - e.mass_L[i_d][j_d] * out[j_d];
      out[i_d] = cur;
    }
  }
}

// func_torque_and_passive_force, forward_dynamics.py:961-1174
void torque_and_passive_force(const Model& m, Env& e) {
  for (int i_l = 0; i_l < NL; ++i_l) {
    const Link& L = m.links[i_l];
    if (L.n_dofs == 0) continue;
    int joint_type = m.joints[L.joint_start].type;
    for (int i_d = L.dof_start; i_d < L.dof_end; ++i_d) {
      const Dof& D = m.dofs[i_d];
      real force = 0.0f;
      if (e.ctrl_mode[i_d] == CTRL_FORCE) force = e.ctrl_force[i_d];
      else if (e.ctrl_mode[i_d] == CTRL_VELOCITY) force = D.kv * (e.ctrl_vel[i_d] - e.vel[i_d]);
      else if (e.ctrl_mode[i_d] == CTRL_POSITION && !(joint_type == JOINT_FREE && i_d >= L.dof_start + 3))
        force = D.kp * (e.ctrl_pos[i_d] - e.dof_pos[i_d]) + D.kv * (e.ctrl_vel[i_d] - e.vel[i_d]);
      e.qf_applied[i_d] = clampf(force, D.force_range[0], D.force_range[1]);
    }
    // (POSITION control of the free joint's angular dofs, :1028-1068, is never enabled on this path)
  }
  for (int i_d = 0; i_d < ND; ++i_d) e.qf_passive[i_d] = -m.dofs[i_d].damping * e.vel[i_d];
  for (int i_l = 0; i_l < NL; ++i_l) {
    const Link& L = m.links[i_l];
    if (L.n_dofs == 0) continue;
    int joint_type = m.joints[L.joint_start].type;
    if (joint_type != JOINT_FREE && joint_type != JOINT_FIXED)
      for (int j_d = L.dof_start; j_d < L.dof_end; ++j_d) e.qf_passive[j_d] = e.qf_passive[j_d] + (-e.dof_pos[j_d] * m.dofs[j_d].stiffness);
  }
}

// func_update_acc(update_cacc=False), forward_dynamics.py:1177-1277
void update_acc(const Model& m, Env& e) {
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
      const Link& L = m.links[i_l];
      if (L.parent == -1) {
        e.cdd_vel[i_l] = -m.gravity * (1.0f - 0.0f);
        e.cdd_ang[i_l] = v3(0, 0, 0);
      } else {
        e.cdd_vel[i_l] = e.cdd_vel[L.parent]; e.cdd_ang[i_l] = e.cdd_ang[L.parent];
      }
      for (int i_d = L.dof_start; i_d < L.dof_end; ++i_d) {
        V3 local_cdd_vel = e.cdofd_vel[i_d] * e.vel[i_d];
        V3 local_cdd_ang = e.cdofd_ang[i_d] * e.vel[i_d];
        e.cdd_vel[i_l] = e.cdd_vel[i_l] + local_cdd_vel;
        e.cdd_ang[i_l] = e.cdd_ang[i_l] + local_cdd_ang;
      }
    }
  }
}

// func_update_force, forward_dynamics.py:1280-1392
void update_force(const Model& m, Env& e) {
  for (int i_l = 0; i_l < NL; ++i_l) {
    V3 f1_ang, f1_vel, f2_ang, f2_vel, f3_ang, f3_vel;
    inertial_mul(e.cinr_pos[i_l], e.cinr_inertial[i_l], e.cinr_mass[i_l], e.cdd_vel[i_l], e.cdd_ang[i_l], f1_ang, f1_vel);
    inertial_mul(e.cinr_pos[i_l], e.cinr_inertial[i_l], e.cinr_mass[i_l], e.cd_vel[i_l], e.cd_ang[i_l], f2_ang, f2_vel);
    motion_cross_force(e.cd_ang[i_l], e.cd_vel[i_l], f2_ang, f2_vel, f3_ang, f3_vel);
    // cfrc_coupling_* is identically zero on this path (no coupler)
    e.cfrc_vel[i_l] = f1_vel + f3_vel + e.ext_vel[i_l] + v3(0, 0, 0);
    e.cfrc_ang[i_l] = f1_ang + f3_ang + e.ext_ang[i_l] + v3(0, 0, 0);
  }
  for (int i_e = 0; i_e < m.n_entities; ++i_e) {
    const Entity& en = m.entities[i_e];
    int n = en.link_end - en.link_start;
    for (int i = 0; i < n; ++i) {
      int i_l = en.link_end - 1 - i, i_p = m.links[i_l].parent;
      if (i_p != -1) {
        e.cfrc_vel[i_p] = e.cfrc_vel[i_p] + e.cfrc_vel[i_l];
        e.cfrc_ang[i_p] = e.cfrc_ang[i_p] + e.cfrc_ang[i_l];
      }
    }
  }
}

// func_bias_force, forward_dynamics.py:1419-1478
void bias_force(const Model& m, Env& e) {
  for (int i_l = 0; i_l < NL; ++i_l)
    for (int i_d = m.links[i_l].dof_start; i_d < m.links[i_l].dof_end; ++i_d) {
      e.qf_bias[i_d] = dot(e.cdof_ang[i_d], e.cfrc_ang[i_l]) + dot(e.cdof_vel[i_d], e.cfrc_vel[i_l]);
      e.force[i_d] = e.qf_passive[i_d] - e.qf_bias[i_d] + e.qf_applied[i_d];
      e.qf_smooth[i_d] = e.force[i_d];
    }
}

// func_forward_dynamics, forward_dynamics.py:146-... (kernel_step_1 without the FK refresh, rigid_solver.py:3008-3069)
void forward_dynamics(const Model& m, Env& e) {
  compute_mass_matrix(m, e, true);
  factor_mass(m, e);
  torque_and_passive_force(m, e);
  update_acc(m, e);
  update_force(m, e);
  bias_force(m, e);
  solve_mass(m, e, e.force, e.acc_smooth);                       // func_compute_qacc, :1498-1555
  for (int i_d = 0; i_d < ND; ++i_d) e.acc[i_d] = e.acc_smooth[i_d];
}

// ---------------------------------------------------------------------------------------------
// collision detection  (R/collider/*.py)
// ---------------------------------------------------------------------------------------------
// kernel_update_geom_aabbs, forward_kinematics.py:1171-1193
void update_geom_aabbs(const Model& m, Env& e) {
  const real inf = dm_bits2f(0x7f800000u);
  for (int i_g = 0; i_g < NG; ++i_g) {
    V3 lower = v3(inf, inf, inf), upper = v3(-inf, -inf, -inf);
    for (int c = 0; c < 8; ++c) {
      V3 corner = transform_by_trans_quat(m.geoms[i_g].aabb[c], e.g_pos[i_g], e.g_quat[i_g]);
      lower = vmin(lower, corner); upper = vmax(upper, corner);
    }
    e.aabb_min[i_g] = lower; e.aabb_max[i_g] = upper;
  }
}

// func_is_geom_aabbs_overlap, collider/utils.py:102-107
inline bool aabbs_overlap(const Env& e, int a, int b) {
  bool any1 = (e.aabb_max[a].x <= e.aabb_min[b].x) || (e.aabb_max[a].y <= e.aabb_min[b].y) || (e.aabb_max[a].z <= e.aabb_min[b].z);
  bool any2 = (e.aabb_min[a].x >= e.aabb_max[b].x) || (e.aabb_min[a].y >= e.aabb_max[b].y) || (e.aabb_min[a].z >= e.aabb_max[b].z);
  return !(any1 || any2);
}

// func_collision_clear + func_broad_phase, collider/broadphase.py:73-138,141-396 (no hibernation)
void broad_phase(const Model& m, Env& e) {
  for (int i_c = 0; i_c < e.n_contacts; ++i_c) {
    Contact& c = e.contacts[i_c];
    c.link_a = c.link_b = c.geom_a = c.geom_b = -1;
    c.penetration = 0.0f; c.pos = v3(0, 0, 0); c.normal = v3(0, 0, 0); c.force = v3(0, 0, 0);
  }
  e.n_contacts = 0;

  const int axis = 0;
  int env_n_geoms = NG;
  if (e.first_time) {
    int i_buffer = 0;
    for (int i_l = 0; i_l < NL; ++i_l)
      for (int i_g = m.links[i_l].geom_start; i_g < m.links[i_l].geom_end; ++i_g) {
        e.sort_value[2 * i_buffer] = vget(e.aabb_min[i_g], axis); e.sort_ig[2 * i_buffer] = i_g; e.sort_ismax[2 * i_buffer] = 0;
        e.sort_value[2 * i_buffer + 1] = vget(e.aabb_max[i_g], axis); e.sort_ig[2 * i_buffer + 1] = i_g; e.sort_ismax[2 * i_buffer + 1] = 1;
        i_buffer++;
      }
    e.first_time = 0;
  } else {
    for (int i = 0; i < env_n_geoms * 2; ++i)
      e.sort_value[i] = e.sort_ismax[i] ? vget(e.aabb_max[e.sort_ig[i]], axis) : vget(e.aabb_min[e.sort_ig[i]], axis);
  }
  for (int i = 1; i < 2 * env_n_geoms; ++i) {
    real key_value = e.sort_value[i]; int key_is_max = e.sort_ismax[i], key_i_g = e.sort_ig[i];
    int j = i - 1;
    while (j >= 0 && key_value < e.sort_value[j]) {
      e.sort_value[j + 1] = e.sort_value[j]; e.sort_ismax[j + 1] = e.sort_ismax[j]; e.sort_ig[j + 1] = e.sort_ig[j];
      j -= 1;
    }
    e.sort_value[j + 1] = key_value; e.sort_ismax[j + 1] = key_is_max; e.sort_ig[j + 1] = key_i_g;
  }
  int n_broad = 0, n_active = 0;
  for (int i = 0; i < 2 * env_n_geoms; ++i) {
    if (!e.sort_ismax[i]) {
      for (int j = 0; j < n_active; ++j) {
        int i_ga = e.active_buf[j], i_gb = e.sort_ig[i];
        if (i_ga > i_gb) std::swap(i_ga, i_gb);
        if (m.pair_idx[i_ga][i_gb] == -1) continue;  // func_check_collision_valid (no dynamic welds)
        if (!aabbs_overlap(e, i_ga, i_gb)) {
          e.normal_cache[m.pair_idx[i_ga][i_gb]] = v3(0, 0, 0);
          continue;
        }
        if (n_broad == m.max_broad_pairs) { e.err |= GO2SIM_ERR_OVERFLOW_CANDIDATE_CONTACTS; break; }
        e.broad[n_broad][0] = i_ga; e.broad[n_broad][1] = i_gb;
        n_broad++;
      }
      e.active_buf[n_active] = e.sort_ig[i];
      n_active++;
    } else {
      int rm = e.sort_ig[i];
      for (int j = 0; j < n_active; ++j)
        if (e.active_buf[j] == rm) {
          if (j < n_active - 1)
            for (int k = j; k < n_active - 1; ++k) e.active_buf[k] = e.active_buf[k + 1];
          n_active--;
          break;
        }
    }
  }
  e.n_broad = n_broad;
}

// ---- support functions, collider/support_field.py --------------------------------------------
// _func_support_mesh for a cylinder: the 180x180 direction-grid table of support_field.py:37-89 is
// reproduced analytically (vertex set = 32-gon ring x {+h/2,-h/2}; see tools/compile_go2_model.py).
// `int(x % support_res)` of support_field.py:157-159 for x in [0,180]; NaN-safe (never indexes out of range)
inline int wrap180(real x) {
  if (!(x >= 0.0f)) return 0;
  int i = (x >= 180.0f) ? (int)(x - 180.0f) : (int)x;
  return (i > 179) ? 179 : i;
}
inline int clampidx(real x) {  // int(clamp(x, 0, 179)), NaN-safe
  if (!(x >= 0.0f)) return 0;
  return (x >= 179.0f) ? 179 : (int)x;
}
V3 support_cylinder_local(const Model& m, const Geom& G, V3 d_mesh, int* vid_out = nullptr) {
  const real PI = 3.14159265358979323846f;
  real theta = dm_atan2(d_mesh.y, d_mesh.x);
  real phi = dm_acos(d_mesh.z);
  const real support_res = 180.0f;
  real ii = (theta + PI) / PI / 2.0f * support_res;
  real jj = phi / PI * support_res;
  real dot_max = -1e20f;
  V3 v = v3(0, 0, 0);
  int vid = 0;
  real half = 0.5f * G.data[1];
  for (int i4 = 0; i4 < 4; ++i4) {
    int i, j;
    if (i4 % 2) i = wrap180(dm_ceil(ii)); else i = wrap180(dm_floor(ii));
    if (i4 / 2 > 0) {
      j = clampidx(dm_ceil(jj));
      if (j == 179) j = 178;
    } else {
      j = clampidx(dm_floor(jj));
      if (j == 0) j = 1;
    }
    int k = m.theta_to_ring[i];
    V3 pos = v3(G.rim[k][0], G.rim[k][1], (j <= 90) ? half : -half);
    real d = dot(pos, d_mesh);
    if (d > dot_max) { v = pos; dot_max = d; vid = k + ((j <= 90) ? 0 : 32); }
  }
  if (vid_out) *vid_out = vid;
  return v;
}
// _func_support_prism, support_field.py:262-280: the terrain geom is represented by the current 6-vertex prism
static thread_local const V3* g_prism = nullptr;
inline V3 support_prism(const V3* prism, V3 d) {
  int istart = 3;
  if (d.z < 0) istart = 0;
  int ibest = istart;
  real best = dot(prism[istart], d);
  for (int i = istart + 1; i < istart + 3; ++i) { real dt_ = dot(prism[i], d); if (dt_ > best) { ibest = i; best = dt_; } }
  return prism[ibest];
}
// support_driver, collider/mpr.py:146-176 (sphere / box / table-driven mesh / terrain prism)
V3 support_driver(const Model& m, V3 direction, int i_g, V3 pos, Q4 quat) {
  const Geom& G = m.geoms[i_g];
  if (G.type == GEOM_TERRAIN) return support_prism(g_prism, direction);
  if (G.type == GEOM_SPHERE) {                                   // support_field.py:183-206
    return pos + direction * G.data[0];
  } else if (G.type == GEOM_BOX) {                               // support_field.py:285-306
    V3 d_box = inv_transform_by_quat(direction, quat);
    V3 v_ = v3((d_box.x < 0.0f ? -1.0f : 1.0f) * G.data[0] * 0.5f, (d_box.y < 0.0f ? -1.0f : 1.0f) * G.data[1] * 0.5f,
               (d_box.z < 0.0f ? -1.0f : 1.0f) * G.data[2] * 0.5f);
    return transform_by_trans_quat(v_, pos, quat);
  } else {                                                       // support_field.py:120-135
    V3 d_mesh = transform_by_quat(direction, inv_quat(quat));
    V3 v_ = support_cylinder_local(m, G, d_mesh);
    return transform_by_trans_quat(v_, pos, quat);
  }
}
// ---- safe GJK + EPA fallback (oracle/gjk_epa_cpu.h): geometric queries of collider/gjk_support.py:62-186, support_field.py:183-306,
//      gjk.py:1652-1700,1854-1907 ----
// vertex ids only need to be unique per (geom, vertex): 64 ids are reserved per geom
inline V3 gjk_support_driver(const Model& m, V3 direction, int i_g, V3 pos, Q4 quat, int& vid) {
  const Geom& G = m.geoms[i_g];
  if (G.type == GEOM_SPHERE) {                                   // _func_support_sphere (shrink = False): vid = -1
    vid = -1;
    return pos + direction * G.data[0];
  } else if (G.type == GEOM_BOX) {                               // _func_support_box
    V3 d_box = inv_transform_by_quat(direction, quat);
    V3 v_ = v3((d_box.x < 0.0f ? -1.0f : 1.0f) * G.data[0] * 0.5f, (d_box.y < 0.0f ? -1.0f : 1.0f) * G.data[1] * 0.5f,
               (d_box.z < 0.0f ? -1.0f : 1.0f) * G.data[2] * 0.5f);
    vid = (v_.x > 0.0f) * 1 + (v_.y > 0.0f) * 2 + (v_.z > 0.0f) * 4 + 64 * i_g;
    return transform_by_trans_quat(v_, pos, quat);
  } else {                                                       // _func_support_world
    V3 d_mesh = transform_by_quat(direction, inv_quat(quat));
    int k = 0;
    V3 v_ = support_cylinder_local(m, G, d_mesh, &k);
    vid = k + 64 * i_g;
    return transform_by_trans_quat(v_, pos, quat);
  }
}
struct GjkSup {
  const Model& m; int i_ga, i_gb; V3 pos_a; Q4 quat_a; V3 pos_b; Q4 quat_b; float eps; bool discrete; int nverts_a, nverts_b;
  static G3 to_g(V3 v) { return g3(v.x, v.y, v.z); }
  void support(G3 d, G3& o1, G3& o2, int& id1, int& id2) const {   // func_support, gjk_support.py:113-186
    V3 dv = v3(d.x, d.y, d.z);
    o1 = to_g(gjk_support_driver(m, dv, i_ga, pos_a, quat_a, id1));
    o2 = to_g(gjk_support_driver(m, -dv, i_gb, pos_b, quat_b, id2));
  }
  int count_one(V3 d, int i_g, Q4 quat) const {                   // count_support_driver, gjk.py:1854-1877
    if (m.geoms[i_g].type == GEOM_BOX) {                          // _func_count_supports_box
      V3 d_box = inv_transform_by_quat(d, quat);
      int zeros = (d_box.x == 0.0f) + (d_box.y == 0.0f) + (d_box.z == 0.0f);
      return 1 << zeros;
    }
    return 1;   // spheres; the tessellated cylinders are GEOM_TYPE.CYLINDER, not MESH
  }
  int count(G3 d) const { V3 dv = v3(d.x, d.y, d.z); return count_one(dv, i_ga, quat_a) * count_one(-dv, i_gb, quat_b); }
  void discrete_vertex(int which, int i_v, G3& obj, int& id) const {   // func_get_discrete_geom_vertex (BOX), gjk.py:1666-1700
    int i_g = which == 0 ? i_ga : i_gb;
    const Geom& G = m.geoms[i_g];
    V3 v_ = v3(((i_v & 1) ? 1.0f : -1.0f) * G.data[0] * 0.5f, ((i_v & 2) ? 1.0f : -1.0f) * G.data[1] * 0.5f, ((i_v & 4) ? 1.0f : -1.0f) * G.data[2] * 0.5f);
    obj = to_g(transform_by_trans_quat(v_, which == 0 ? pos_a : pos_b, which == 0 ? quat_a : quat_b));
    id = 64 * i_g + i_v;
  }
};
inline GjkResult gjk_contact_pair(const Model& m, GjkScratch& scratch, int i_ga, int i_gb, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b) {
  bool disc = m.geoms[i_ga].type == GEOM_BOX && m.geoms[i_gb].type == GEOM_BOX;   // func_is_discrete_geoms, collider/utils.py:105-126
  GjkSup sup{m, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, m.eps, disc, 8, 8};
  return gjk_contact(sup, scratch);
}

// compute_support, collider/mpr.py:179-202
inline void compute_support(const Model& m, V3 direction, int i_ga, int i_gb, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, V3& v, V3& v1, V3& v2) {
  v1 = support_driver(m, direction, i_ga, pos_a, quat_a);
  v2 = support_driver(m, -direction, i_gb, pos_b, quat_b);
  v = v1 - v2;
}

// ---- MPR, collider/mpr.py ------------------------------------------------------------------------
inline V3 mpr_portal_dir(const Env& e) {                         // mpr.py:112-117
  V3 v2v1 = e.mpr_v[2] - e.mpr_v[1], v3v1 = e.mpr_v[3] - e.mpr_v[1];
  return normalized(cross(v2v1, v3v1));
}
inline bool mpr_portal_reach_tolerance(const Model& m, const Env& e, V3 v, V3 direction) {  // mpr.py:134-143
  real dv1 = dot(e.mpr_v[1], direction), dv2 = dot(e.mpr_v[2], direction), dv3 = dot(e.mpr_v[3], direction), dv4 = dot(v, direction);
  real dot1 = std::min(std::min(dv4 - dv1, dv4 - dv2), dv4 - dv3);
  return dot1 < m.ccd_tolerance + m.ccd_eps * std::max(1.0f, dot1);
}
inline void mpr_expand_portal(Env& e, V3 v, V3 v1, V3 v2) {      // mpr.py:426-442
  V3 v4v0 = cross(v, e.mpr_v[0]);
  real d = dot(e.mpr_v[1], v4v0);
  int i_s;
  if (d > 0) { d = dot(e.mpr_v[2], v4v0); i_s = (d > 0) ? 1 : 3; }
  else { d = dot(e.mpr_v[3], v4v0); i_s = (d > 0) ? 2 : 1; }
  e.mpr_v1[i_s] = v1; e.mpr_v2[i_s] = v2; e.mpr_v[i_s] = v;
}
inline void mpr_swap(Env& e, int i, int j) {
  std::swap(e.mpr_v1[i], e.mpr_v1[j]); std::swap(e.mpr_v2[i], e.mpr_v2[j]); std::swap(e.mpr_v[i], e.mpr_v[j]);
}
// mpr_discover_portal, mpr.py:445-598
int mpr_discover_portal(const Model& m, Env& e, int i_ga, int i_gb, V3 center_a, V3 center_b, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b) {
  const real EPSC = m.ccd_eps;
  e.mpr_v1[0] = center_a; e.mpr_v2[0] = center_b; e.mpr_v[0] = center_a - center_b;
  int simplex_size = 1;
  if (dm_abs(e.mpr_v[0].x) < EPSC && dm_abs(e.mpr_v[0].y) < EPSC && dm_abs(e.mpr_v[0].z) < EPSC) e.mpr_v[0].x += 10.0f * EPSC;
  V3 direction = -normalized(e.mpr_v[0]);
  V3 v, v1, v2;
  compute_support(m, direction, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, v, v1, v2);
  e.mpr_v1[1] = v1; e.mpr_v2[1] = v2; e.mpr_v[1] = v;
  simplex_size = 2;
  real d = dot(v, direction);
  int ret = 0;
  if (d < EPSC) {
    ret = -1;
  } else {
    direction = cross(e.mpr_v[0], e.mpr_v[1]);
    if (dot(direction, direction) < EPSC) {
      if (dm_abs(e.mpr_v[1].x) < EPSC && dm_abs(e.mpr_v[1].y) < EPSC && dm_abs(e.mpr_v[1].z) < EPSC) ret = 1; else ret = 2;
    } else {
      direction = normalized(direction);
      compute_support(m, direction, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, v, v1, v2);
      d = dot(v, direction);
      if (d < EPSC) {
        ret = -1;
      } else {
        e.mpr_v1[2] = v1; e.mpr_v2[2] = v2; e.mpr_v[2] = v;
        simplex_size = 3;
        V3 va = e.mpr_v[1] - e.mpr_v[0], vb = e.mpr_v[2] - e.mpr_v[0];
        direction = normalized(cross(va, vb));
        d = dot(direction, e.mpr_v[0]);
        if (d > 0) { mpr_swap(e, 1, 2); direction = -direction; }
        int num_trials = 0;
        while (simplex_size < 4) {
          compute_support(m, direction, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, v, v1, v2);
          d = dot(v, direction);
          if (d < EPSC) { ret = -1; break; }
          bool cont = false;
          va = cross(e.mpr_v[1], v);
          d = dot(va, e.mpr_v[0]);
          if (d < -EPSC) { e.mpr_v1[2] = v1; e.mpr_v2[2] = v2; e.mpr_v[2] = v; cont = true; }
          if (!cont) {
            va = cross(v, e.mpr_v[2]);
            d = dot(va, e.mpr_v[0]);
            if (d < -EPSC) { e.mpr_v1[1] = v1; e.mpr_v2[1] = v2; e.mpr_v[1] = v; cont = true; }
          }
          if (cont) {
            va = e.mpr_v[1] - e.mpr_v[0]; vb = e.mpr_v[2] - e.mpr_v[0];
            direction = normalized(cross(va, vb));
            num_trials++;
            if (num_trials == 15) { ret = -1; break; }
          } else {
            e.mpr_v1[3] = v1; e.mpr_v2[3] = v2; e.mpr_v[3] = v;
            simplex_size = 4;
          }
        }
      }
    }
  }
  return ret;
}
// mpr_refine_portal, mpr.py:232-278
int mpr_refine_portal(const Model& m, Env& e, int i_ga, int i_gb, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b) {
  int ret = 1;
  while (true) {
    V3 direction = mpr_portal_dir(e);
    if (dot(e.mpr_v[1], direction) > -m.ccd_eps) { ret = 0; break; }
    V3 v, v1, v2;
    compute_support(m, direction, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, v, v1, v2);
    if (!(dot(v, direction) > -m.ccd_eps) || mpr_portal_reach_tolerance(m, e, v, direction)) { ret = -1; break; }
    mpr_expand_portal(e, v, v1, v2);
  }
  return ret;
}
// mpr_find_pos (non-mujoco branch), mpr.py:281-316
V3 mpr_find_pos(const Model& m, const Env& e) {
  real b[4] = {0, 0, 0, 0};
  real sum_ = ((b[0] + b[1]) + b[2]) + b[3];
  if (sum_ < m.ccd_eps) {
    V3 direction = mpr_portal_dir(e);
    b[0] = 0.0f;
    for (int i = 1; i < 4; ++i) {
      int i1 = i % 3 + 1, i2 = (i + 1) % 3 + 1;
      b[i] = dot(cross(e.mpr_v[i1], e.mpr_v[i2]), direction);
    }
    sum_ = ((b[0] + b[1]) + b[2]) + b[3];
  }
  V3 p1 = v3(0, 0, 0), p2 = v3(0, 0, 0);
  for (int i = 0; i < 4; ++i) { p1 = p1 + b[i] * e.mpr_v1[i]; p2 = p2 + b[i] * e.mpr_v2[i]; }
  return (0.5f / sum_) * (p1 + p2);
}
// mpr_find_penetration, mpr.py:338-423
void mpr_find_penetration(const Model& m, Env& e, int i_ga, int i_gb, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, bool& is_col, V3& normal,
                          real& penetration, V3& pos) {
  int iterations = 0;
  while (true) {
    V3 direction = mpr_portal_dir(e);
    V3 v, v1, v2;
    compute_support(m, direction, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, v, v1, v2);
    if (mpr_portal_reach_tolerance(m, e, v, direction) || iterations > m.ccd_iterations) {
      penetration = dot(direction, e.mpr_v[1]);
      normal = -direction;
      is_col = true;
      pos = mpr_find_pos(m, e);
      break;
    }
    mpr_expand_portal(e, v, v1, v2);
    iterations++;
  }
}
// guess_geoms_center, mpr.py:601-683
void guess_geoms_center(const Model& m, int i_ga, int i_gb, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, V3 normal_ws, V3& center_a, V3& center_b) {
  const Geom& A = m.geoms[i_ga]; const Geom& B = m.geoms[i_gb];
  center_a = transform_by_trans_quat(A.center, pos_a, quat_a);
  center_b = transform_by_trans_quat(B.center, pos_b, quat_b);
  if (dm_abs(normal_ws.x) > m.ccd_eps || dm_abs(normal_ws.y) > m.ccd_eps || dm_abs(normal_ws.z) > m.ccd_eps) {
    V3 center_a_local = 0.5f * (A.aabb[7] + A.aabb[0]);
    center_a = transform_by_trans_quat(center_a_local, pos_a, quat_a);
    V3 center_b_local = 0.5f * (B.aabb[7] + B.aabb[0]);
    center_b = transform_by_trans_quat(center_b_local, pos_b, quat_b);
    V3 delta = center_a - center_b;
    V3 normal = normalized(delta);
    if (norm(cross(normal_ws, normal)) > 0.01f) {
      V3 offset = dot(delta, normal_ws) * normal_ws - delta;
      real offset_norm = norm(offset);
      if (offset_norm > m.eps) {
        V3 dir_offset = offset / offset_norm;
        V3 dla = inv_transform_by_quat(dir_offset, quat_a), dlb = inv_transform_by_quat(dir_offset, quat_b);
        V3 box_size_a = A.aabb[7] - A.aabb[0], box_size_b = B.aabb[7] - B.aabb[0];
        real length_a = dot(box_size_a, v3(dm_abs(dla.x), dm_abs(dla.y), dm_abs(dla.z)));
        real length_b = dot(box_size_b, v3(dm_abs(dlb.x), dm_abs(dlb.y), dm_abs(dlb.z)));
        real offset_ratio = std::min(offset_norm / (length_a + length_b), 0.5f);
        center_a = center_a + dir_offset * length_a * offset_ratio;
        center_b = center_b - dir_offset * length_b * offset_ratio;
      }
    }
  }
}
// func_mpr_contact -> func_mpr_contact_from_centers, mpr.py:686-819
// func_mpr_contact_from_centers, mpr.py:686-760
void mpr_contact_from_centers(const Model& m, Env& e, int i_ga, int i_gb, V3 center_a, V3 center_b, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, bool& is_col,
                              V3& normal, real& penetration, V3& pos);
void mpr_contact(const Model& m, Env& e, int i_ga, int i_gb, V3 normal_ws, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, bool& is_col, V3& normal,
                 real& penetration, V3& pos) {
  V3 center_a, center_b;
  guess_geoms_center(m, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, normal_ws, center_a, center_b);
  mpr_contact_from_centers(m, e, i_ga, i_gb, center_a, center_b, pos_a, quat_a, pos_b, quat_b, is_col, normal, penetration, pos);
}
void mpr_contact_from_centers(const Model& m, Env& e, int i_ga, int i_gb, V3 center_a, V3 center_b, V3 pos_a, Q4 quat_a, V3 pos_b, Q4 quat_b, bool& is_col,
                              V3& normal, real& penetration, V3& pos) {
  int res = mpr_discover_portal(m, e, i_ga, i_gb, center_a, center_b, pos_a, quat_a, pos_b, quat_b);
  is_col = false; pos = v3(0, 0, 0); normal = v3(0, 0, 0); penetration = 0.0f;
  if (res == 1) {                                                // mpr_find_penetr_touch
    is_col = true; penetration = 0.0f; normal = -normalized(e.mpr_v[0]); pos = (e.mpr_v1[1] + e.mpr_v2[1]) * 0.5f;
  } else if (res == 2) {                                         // mpr_find_penetr_segment
    is_col = true; penetration = norm(e.mpr_v[1]); normal = -normalized(e.mpr_v[1]); pos = (e.mpr_v1[1] + e.mpr_v2[1]) * 0.5f;
  } else if (res == 0) {
    res = mpr_refine_portal(m, e, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b);
    if (res >= 0) mpr_find_penetration(m, e, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, is_col, normal, penetration, pos);
  }
}

// ---- contact bookkeeping, collider/contact.py ---------------------------------------------------
// func_add_contact, contact.py:165-199
void add_contact(const Model& m, Env& e, int i_ga, int i_gb, V3 normal, V3 contact_pos, real penetration) {
  int i_c = e.n_contacts;
  if (i_c < m.max_contact_pairs) {
    real friction_a = e.geom_friction[i_ga] * e.friction_ratio[i_ga];
    real friction_b = e.geom_friction[i_gb] * e.friction_ratio[i_gb];
    Contact& c = e.contacts[i_c];
    c.geom_a = i_ga; c.geom_b = i_gb; c.normal = normal; c.pos = contact_pos; c.penetration = penetration;
    c.friction = std::max(std::max(friction_a, friction_b), 1e-2f);
    for (int k = 0; k < 7; ++k) c.sol_params[k] = 0.5f * (m.geoms[i_ga].sol_params[k] + m.geoms[i_gb].sol_params[k]);
    c.link_a = m.geoms[i_ga].link; c.link_b = m.geoms[i_gb].link;
    e.n_contacts = i_c + 1;
  } else {
    e.err |= GO2SIM_ERR_OVERFLOW_COLLISION_PAIRS;
  }
}
// func_compute_tolerance, contact.py:264-283
inline real compute_tolerance(const Model& m, int i_ga, int i_gb, real tolerance) {
  real size_b = norm(m.geoms[i_gb].aabb[7] - m.geoms[i_gb].aabb[0]);
  real size_a = norm(m.geoms[i_ga].aabb[7] - m.geoms[i_ga].aabb[0]);
  return 0.5f * tolerance * std::min(size_a, size_b);
}
// func_contact_orthogonals (non-mujoco branch), contact.py:286-345
void contact_orthogonals(const Model& m, const Env& e, int i_ga, int i_gb, V3 normal, V3& axis_0, V3& axis_1) {
  V3 size_ga = m.geoms[i_ga].aabb[7], size_gb = m.geoms[i_gb].aabb[7];
  real volume_ga = size_ga.x * size_ga.y * size_ga.z, volume_gb = size_gb.x * size_gb.y * size_gb.z;
  int i_g = (volume_ga < volume_gb) ? i_ga : i_gb;
  int i_l = m.geoms[i_g].link;
  M3 rot = quat_to_R(e.i_quat[i_l], m.eps);
  int axis_idx = 0; real axis_angle_max = 0.0f;
  for (int i = 0; i < 3; ++i) {
    real axis_angle = dm_abs(dot(mcol(rot, i), normal));
    if (axis_angle > axis_angle_max) { axis_angle_max = axis_angle; axis_idx = i; }
  }
  axis_idx = (axis_idx + 1) % 3;
  axis_0 = mcol(rot, axis_idx);
  axis_0 = normalized(axis_0 - dot(normal, axis_0) * normal);
  axis_1 = cross(normal, axis_0);
}
// func_rotate_frame, contact.py:348-369
inline void rotate_frame(V3 pos, Q4 quat, V3 contact_pos, Q4 qrot, V3& new_pos, Q4& new_quat) {
  new_quat = transform_quat_by_quat(quat, qrot);
  V3 rel = contact_pos - pos;
  V3 vec = transform_by_quat(rel, qrot);
  vec = vec - rel;
  new_pos = pos - vec;
}

// func_convex_convex_contact (CCD_ALGORITHM_CODE.MPR, non-plane, non-capsule branch), narrowphase.py:514-961
void convex_convex_contact(const Model& m, Env& e, int i_ga, int i_gb) {
  const real EPS = m.eps;
  int type_a = m.geoms[i_ga].type, type_b = m.geoms[i_gb].type;
  bool multi_contact = (type_a != GEOM_SPHERE) && (type_b != GEOM_SPHERE);
  real tolerance = compute_tolerance(m, i_ga, i_gb, m.mc_tolerance);
  V3 ga_pos_o = e.g_pos[i_ga], gb_pos_o = e.g_pos[i_gb]; Q4 ga_quat_o = e.g_quat[i_ga], gb_quat_o = e.g_quat[i_gb];
  V3 ga_pos = ga_pos_o, gb_pos = gb_pos_o; Q4 ga_quat = ga_quat_o, gb_quat = gb_quat_o;
  bool is_col_0 = false; real penetration_0 = 0.0f; V3 normal_0 = v3(0, 0, 0), contact_pos_0 = v3(0, 0, 0);
  bool is_col = false; real penetration = 0.0f; V3 normal = v3(0, 0, 0), contact_pos = v3(0, 0, 0);
  int n_con = 0;
  V3 axis_0 = v3(0, 0, 0), axis_1 = v3(0, 0, 0); Q4 qrot = q4(0, 0, 0, 0);
  int i_pair = (i_ga > i_gb) ? m.pair_idx[i_gb][i_ga] : m.pair_idx[i_ga][i_gb];
  for (int i_detection = 0; i_detection < 5; ++i_detection) {
    bool prefer_gjk = false;
    if (multi_contact && is_col_0) {
      V3 axis = (real)(2 * (i_detection % 2) - 1) * axis_0 + (real)(1 - 2 * ((i_detection / 2) % 2)) * axis_1;
      qrot = rotvec_to_quat(m.mc_perturbation * axis, EPS);
      rotate_frame(ga_pos_o, ga_quat_o, contact_pos_0, qrot, ga_pos, ga_quat);
      rotate_frame(gb_pos_o, gb_quat_o, contact_pos_0, inv_quat(qrot), gb_pos, gb_quat);
    }
    if ((multi_contact && is_col_0) || (i_detection == 0)) {
      bool is_mpr_updated = false;
      V3 normal_ws = e.normal_cache[i_pair];
      bool guess_available = (dm_abs(normal_ws.x) > EPS) || (dm_abs(normal_ws.y) > EPS) || (dm_abs(normal_ws.z) > EPS);
      for (int i_mpr = 0; i_mpr < 2; ++i_mpr) {
        if (i_mpr == 1) {
          if ((i_detection == 0) && !is_col && guess_available) {
            normal_ws = v3(0, 0, 0); guess_available = false; is_mpr_updated = false;
          }
        }
        if (!is_mpr_updated) {
          mpr_contact(m, e, i_ga, i_gb, normal_ws, ga_pos, ga_quat, gb_pos, gb_quat, is_col, normal, penetration, contact_pos);
          is_mpr_updated = true;
        }
      }
      if (penetration > tolerance)
        prefer_gjk = !guess_available || (m.mc_tolerance * penetration >= m.mpr_to_gjk_ratio * tolerance);
      if (prefer_gjk) {                                            // narrowphase.py:734-845: safe GJK + EPA replaces the MPR answer
        e.gjk_fallback_count++;
        static thread_local GjkScratch gjk_scratch;   // working memory of one GJK/EPA query
        GjkResult gr = gjk_contact_pair(m, gjk_scratch, i_ga, i_gb, ga_pos, ga_quat, gb_pos, gb_quat);
        is_col = gr.is_col != 0;
        penetration = gr.penetration;
        if (is_col) { contact_pos = v3(gr.pos.x, gr.pos.y, gr.pos.z); normal = v3(gr.normal.x, gr.normal.y, gr.normal.z); }
      }
    }
    if (i_detection == 0) {
      is_col_0 = is_col; normal_0 = normal; penetration_0 = penetration; contact_pos_0 = contact_pos;
      if (is_col_0) {
        add_contact(m, e, i_ga, i_gb, normal_0, contact_pos_0, penetration_0);
        if (multi_contact) { contact_orthogonals(m, e, i_ga, i_gb, normal, axis_0, axis_1); n_con = 1; }
        e.normal_cache[i_pair] = normal;
      } else {
        e.normal_cache[i_pair] = v3(0, 0, 0);
      }
    } else if (multi_contact && is_col) {
      V3 contact_point_a = transform_by_quat((contact_pos - 0.5f * penetration * normal) - contact_pos_0, inv_quat(qrot)) + contact_pos_0;
      V3 contact_point_b = transform_by_quat((contact_pos + 0.5f * penetration * normal) - contact_pos_0, qrot) + contact_pos_0;
      contact_pos = 0.5f * (contact_point_a + contact_point_b);
      V3 tw = cross(normal, normal_0);
      V3 twist_rotvec = v3(clampf(tw.x, -m.mc_perturbation, m.mc_perturbation), clampf(tw.y, -m.mc_perturbation, m.mc_perturbation),
                           clampf(tw.z, -m.mc_perturbation, m.mc_perturbation));
      normal = normal + cross(twist_rotvec, normal);
      penetration = dot(normal, contact_point_b - contact_point_a);
      bool repeated = false;
      for (int i_c = 0; i_c < n_con; ++i_c)
        if (!repeated) {
          int idx_prev = e.n_contacts - 1 - i_c;
          if (norm(contact_pos - e.contacts[idx_prev].pos) < tolerance) repeated = true;
        }
      if (!repeated && penetration > -tolerance) {
        penetration = std::max(penetration, 0.0f);
        add_contact(m, e, i_ga, i_gb, normal, contact_pos, penetration);
        n_con++;
      }
    }
  }
}

// func_add_prism_vert, narrowphase.py:493-512
inline void add_prism_vert(Env& e, real x, real y, real z) {
  e.prism[0] = e.prism[1]; e.prism[1] = e.prism[2]; e.prism[3] = e.prism[4]; e.prism[4] = e.prism[5];
  e.prism[2].x = x; e.prism[5].x = x; e.prism[2].y = y; e.prism[5].y = y; e.prism[5].z = z;
}
// func_contact_mpr_terrain, narrowphase.py:345-490: geom i_ga against the heightfield cells under its bounding box
void contact_mpr_terrain(const Model& m, Env& e, int i_ga, int i_gb) {
  V3 ga_pos = e.g_pos[i_ga], gb_pos = e.g_pos[i_gb]; Q4 ga_quat = e.g_quat[i_ga], gb_quat = e.g_quat[i_gb];
  const real margin = 0.0f;
  bool is_return = false;
  real tolerance = compute_tolerance(m, i_ga, i_gb, m.mc_tolerance);
  V3 ga_pos_t; Q4 ga_quat_t;
  transform_pos_quat_by_trans_quat(ga_pos - gb_pos, ga_quat, v3(0, 0, 0), inv_quat(gb_quat), ga_pos_t, ga_quat_t);
  V3 gb_pos_t = v3(0, 0, 0); Q4 gb_quat_t = q4(1, 0, 0, 0);
  V3 center_a = transform_by_trans_quat(m.geoms[i_ga].center, ga_pos_t, ga_quat_t);
  for (int i_axis = 0; i_axis < 3; ++i_axis)
    for (int i_m = 0; i_m < 2; ++i_m) {
      V3 direction = v3(0, 0, 0);
      vset(direction, i_axis, (i_m == 0) ? 1.0f : -1.0f);
      V3 v1 = support_driver(m, direction, i_ga, ga_pos_t, ga_quat_t);
      e.xyz_max_min[3 * i_m + i_axis] = vget(v1, i_axis);
    }
  const real* tmm = m.terrain_xyz_maxmin;
  for (int i = 0; i < 3; ++i) {
    e.prism[i].z = tmm[5];
    if (tmm[i] < e.xyz_max_min[i + 3] - margin || tmm[i + 3] > e.xyz_max_min[i] + margin) is_return = true;
  }
  if (is_return) return;
  const real sh = m.terrain_hs;
  int r_min = (int)dm_floor((e.xyz_max_min[3] - tmm[3]) / sh);
  int r_max = (int)dm_ceil((e.xyz_max_min[0] - tmm[3]) / sh);
  int c_min = (int)dm_floor((e.xyz_max_min[4] - tmm[4]) / sh);
  int c_max = (int)dm_ceil((e.xyz_max_min[1] - tmm[4]) / sh);
  r_min = std::max(0, r_min); c_min = std::max(0, c_min);
  r_max = std::min(m.terrain_rows - 1, r_max); c_max = std::min(m.terrain_cols - 1, c_max);
  int n_con = 0;
  g_prism = e.prism;
  for (int r = r_min; r < r_max; ++r) {
    int nvert = 0;
    for (int c = c_min; c < c_max + 1; ++c)
      for (int i = 0; i < 2; ++i)
        if (n_con < m.n_contacts_per_pair) {
          nvert = nvert + 1;
          add_prism_vert(e, sh * (real)(r + i) + tmm[3], sh * (real)c + tmm[4], m.terrain_hf[(size_t)(r + i) * m.terrain_cols + c] + margin);
          if (nvert > 2 && (e.prism[3].z >= e.xyz_max_min[5] || e.prism[4].z >= e.xyz_max_min[5] || e.prism[5].z >= e.xyz_max_min[5])) {
            V3 center_b = v3(0, 0, 0);
            for (int i_p = 0; i_p < 6; ++i_p) center_b = center_b + e.prism[i_p];
            center_b = center_b / 6.0f;
            bool is_col; V3 normal, contact_pos; real penetration;
            mpr_contact_from_centers(m, e, i_ga, i_gb, center_a, center_b, ga_pos_t, ga_quat_t, gb_pos_t, gb_quat_t, is_col, normal, penetration, contact_pos);
            if (is_col) {
              normal = transform_by_quat(normal, gb_quat);
              contact_pos = transform_by_quat(contact_pos, gb_quat);
              contact_pos = contact_pos + gb_pos;
              bool valid = true;
              int i_c = e.n_contacts;
              for (int j = 0; j < n_con; ++j)
                if (norm(contact_pos - e.contacts[i_c - j - 1].pos) < tolerance) { valid = false; break; }
              if (valid) { add_contact(m, e, i_ga, i_gb, normal, contact_pos, penetration); n_con = n_con + 1; }
            }
          }
        }
  }
  g_prism = nullptr;
}

// func_narrow_phase_convex_vs_convex (narrowphase.py:964-1068) then func_narrow_phase_any_vs_terrain (:1197-1244), collider.py:436-528
void narrow_phase(const Model& m, Env& e) {
  for (int i_pair = 0; i_pair < e.n_broad; ++i_pair) {
    int i_ga = e.broad[i_pair][0], i_gb = e.broad[i_pair][1];
    if (m.geoms[i_ga].type > m.geoms[i_gb].type) std::swap(i_ga, i_gb);
    if (m.geoms[i_gb].type == GEOM_TERRAIN) continue;
    convex_convex_contact(m, e, i_ga, i_gb);
  }
  if (!m.terrain_enabled) return;
  for (int i_pair = 0; i_pair < e.n_broad; ++i_pair) {
    int i_ga = e.broad[i_pair][0], i_gb = e.broad[i_pair][1];
    if (m.geoms[i_ga].type == GEOM_TERRAIN) std::swap(i_ga, i_gb);
    if (m.geoms[i_gb].type == GEOM_TERRAIN) contact_mpr_terrain(m, e, i_ga, i_gb);
  }
}

// ---------------------------------------------------------------------------------------------
// constraints + Newton solver  (R/constraint/solver.py)
// ---------------------------------------------------------------------------------------------
// add_collision_constraints, solver.py:498-595
void add_collision_constraints(const Model& m, Env& e) {
  for (int i_col = 0; i_col < e.n_contacts; ++i_col) {
    const Contact& c = e.contacts[i_col];
    V3 d1, d2;
    orthogonals(c.normal, d1, d2);
    real invweight = m.links[c.link_a].invweight[0];
    if (c.link_b > -1) invweight = invweight + m.links[c.link_b].invweight[0];
    for (int i = 0; i < 4; ++i) {
      V3 d = (real)(2 * (i % 2) - 1) * ((i < 2) ? d1 : d2);
      V3 n = d * c.friction - c.normal;
      int n_con = e.n_con++;
      for (int i_d = 0; i_d < ND; ++i_d) e.jac[n_con][i_d] = 0.0f;
      real jac_qvel = 0.0f;
      for (int i_ab = 0; i_ab < 2; ++i_ab) {
        real sign = -1.0f; int link = c.link_a;
        if (i_ab == 1) { sign = 1.0f; link = c.link_b; }
        while (link > -1) {
          const Link& L = m.links[link];
          for (int i_d_ = 0; i_d_ < L.n_dofs; ++i_d_) {
            int i_d = L.dof_end - 1 - i_d_;
            V3 t_pos = c.pos - e.root_com[link];
            // qd_transform_motion_by_trans_quat with the identity quaternion (geom.py:298-303): the rotation is
            // the exact identity (up to the sign of zero), so only the translation part is evaluated.
            V3 vel = e.cdof_vel[i_d] - cross(t_pos, e.cdof_ang[i_d]);
            V3 diff = sign * vel;
            real jac = dot(diff, n);
            jac_qvel = jac_qvel + jac * e.vel[i_d];
            e.jac[n_con][i_d] = e.jac[n_con][i_d] + jac;
          }
          link = L.parent;
        }
      }
      real imp, aref;
      imp_aref(c.sol_params, -c.penetration, jac_qvel, -c.penetration, imp, aref);
      real diag = invweight + c.friction * c.friction * invweight;
      diag *= 2.0f * c.friction * c.friction * (1.0f - imp) / imp;
      diag = std::max(diag, m.eps);
      e.diag[n_con] = diag; e.aref[n_con] = aref; e.efc_D[n_con] = 1.0f / diag;
    }
  }
}
// add_joint_limit_constraints, solver.py:1088-1143
void add_joint_limit_constraints(const Model& m, Env& e) {
  for (int i_l = 0; i_l < NL; ++i_l)
    for (int i_j = m.links[i_l].joint_start; i_j < m.links[i_l].joint_end; ++i_j) {
      const Joint& J = m.joints[i_j];
      if (J.type != JOINT_REVOLUTE) continue;
      int i_q = J.q_start, i_d = J.dof_start;
      real pos_delta_min = e.qpos[i_q] - m.dofs[i_d].limit[0];
      real pos_delta_max = m.dofs[i_d].limit[1] - e.qpos[i_q];
      real pos_delta = std::min(pos_delta_min, pos_delta_max);
      if (pos_delta < 0) {
        real jac = (real)((pos_delta_min < pos_delta_max) * 2 - 1);
        real jac_qvel = jac * e.vel[i_d];
        real imp, aref;
        imp_aref(J.sol_params, pos_delta, jac_qvel, pos_delta, imp, aref);
        real diag = std::max(m.dofs[i_d].invweight * (1.0f - imp) / imp, m.eps);
        int n_con = e.n_con++;
        e.diag[n_con] = diag; e.aref[n_con] = aref; e.efc_D[n_con] = 1.0f / diag;
        for (int i_d2 = 0; i_d2 < ND; ++i_d2) e.jac[n_con][i_d2] = 0.0f;
        e.jac[n_con][i_d] = jac;
      }
    }
}

// func_hessian_direct_batch, solver.py:1285-1343
// ---------------------------------------------------------------------------------------------
// GO2SIM_FAST_ORDER: the summation order / reciprocal forms of the HIP solver (csrc/go2sim.hip, "FAST ORDER").  The strict build follows the
// reference's CPU (serial) variants; north_star allows a float32 tolerance on floats, and the HIP product uses it for the reductions that sit on
// the critical path of the Newton solve: sums over constraint rows / dofs are lane-parallel butterfly trees instead of first-to-last chains,
// the triangular solves multiply by a stored reciprocal diagonal (as the reference's own LDL^T path does for the mass matrix,
// forward_dynamics.py:545-687) and run column-oriented, and the rank-1 Cholesky rotations use reciprocals that are formed off the critical
// path.  This build mirrors that arithmetic operation for operation, so that HIP == libgo2sim_cpu_fast.so bit for bit;
// tests/test_fast_order.py bounds fast vs strict.
// ---------------------------------------------------------------------------------------------
#ifdef GO2SIM_FAST_ORDER
#ifndef GO2SIM_REBUILD_FLIPS
#define GO2SIM_REBUILD_FLIPS 1
#endif
inline int fast_team(const Model& m) { return m.terrain_enabled ? 64 : 32; }   // lanes per env of k_constraint_solve_team (flat: 32, heightfield: 64)
// sum of `n` terms spread over the W lanes of a team: lane l adds its terms l, l + W, ... first to last, then a butterfly over the lanes
// (xor 1, xor 2, mirror in 8, mirror in 16 inside rows of 16 lanes; the rows are added pairwise)
template <class F>
inline real team_tree_sum(int W, int n, F term) {
  real y[64], z[64];
  for (int l = 0; l < W; ++l) {
    real p = 0.0f; bool first = true;
    for (int c = l; c < n; c += W) { real t = term(c); p = first ? t : p + t; first = false; }
    y[l] = p;
  }
  for (int l = 0; l < W; ++l) z[l] = y[l] + y[l ^ 1];
  for (int l = 0; l < W; ++l) y[l] = z[l] + z[l ^ 2];
  for (int l = 0; l < W; ++l) z[l] = y[l] + y[l ^ 7];
  for (int l = 0; l < W; ++l) y[l] = z[l] + z[l ^ 15];
  return (W == 32) ? (y[0] + y[16]) : ((y[0] + y[16]) + (y[32] + y[48]));
}
#endif

void hessian_direct(const Model& m, Env& e) {
#ifdef GO2SIM_FAST_ORDER
  // ts_hessian_direct of csrc/go2sim.hip: row factor J[c][i] * (D[c] * active[c]) (0 where the reference skips the row), fused multiply-add chain over
  // the rows; teams of 64 lanes (heightfield) sum the rows c = g mod 3 separately and add the partial sums as (p0 + p1) + p2
  const int G = fast_team(m) == 64 ? 3 : 1;
  for (int i = 0; i < ND; ++i)
    for (int j = 0; j < i + 1; ++j) {
      real part[3] = {0.0f, 0.0f, 0.0f};
      for (int g = 0; g < G; ++g) {
        real h = 0.0f;
        for (int c = g; c < e.n_con; c += G) {
          const real j1 = e.jac[c][i];
          const real jd = (dm_abs(j1) > m.eps) ? j1 * (e.efc_D[c] * (real)e.active[c]) : 0.0f;
          h = std::fma(e.jac[c][j], jd, h);
        }
        part[g] = h;
      }
      e.H[i][j] = ((G == 1) ? part[0] : (part[0] + part[1]) + part[2]) + e.mass_mat[i][j];
      e.Hunf[i][j] = e.H[i][j];
    }
  return;
#endif
  for (int i = 0; i < ND; ++i) for (int j = 0; j < i + 1; ++j) e.H[i][j] = 0.0f;
  for (int i_d1 = 0; i_d1 < ND; ++i_d1)
    for (int i_c = 0; i_c < e.n_con; ++i_c)
      if (dm_abs(e.jac[i_c][i_d1]) > m.eps)
        for (int i_d2 = 0; i_d2 < i_d1 + 1; ++i_d2)
          e.H[i_d1][i_d2] = e.H[i_d1][i_d2] + e.jac[i_c][i_d2] * e.jac[i_c][i_d1] * e.efc_D[i_c] * (real)e.active[i_c];
  for (int i_e = 0; i_e < m.n_entities; ++i_e)
    for (int i_d1 = m.entities[i_e].dof_start; i_d1 < m.entities[i_e].dof_end; ++i_d1)
      for (int i_d2 = m.entities[i_e].dof_start; i_d2 < i_d1 + 1; ++i_d2) e.H[i_d1][i_d2] = e.H[i_d1][i_d2] + e.mass_mat[i_d1][i_d2];
}
#ifdef GO2SIM_FAST_ORDER
// ts_hessian_update of csrc/go2sim.hip: after a change of the active set the rows that flipped are added to / subtracted from the stored Hessian
// (first to last, fused multiply-adds) instead of summing all rows again
void hessian_update(const Model& m, Env& e) {
  for (int c = 0; c < e.n_con; ++c) {
    if ((e.active[c] != 0) == (e.prev_active[c] != 0)) continue;
    const real sg = e.active[c] ? 1.0f : -1.0f;
    for (int i = 0; i < ND; ++i) {
      const real j1 = e.jac[c][i];
      const real jd = (dm_abs(j1) > m.eps) ? sg * (j1 * e.efc_D[c]) : 0.0f;
      for (int j = 0; j < i + 1; ++j) e.Hunf[i][j] = std::fma(e.jac[c][j], jd, e.Hunf[i][j]);
    }
  }
  for (int i = 0; i < ND; ++i) for (int j = 0; j < i + 1; ++j) e.H[i][j] = e.Hunf[i][j];
}
#endif
#ifdef GO2SIM_FAST_ORDER
// ts_cholesky_factor_arrow of csrc/go2sim.hip: when no contact joins links of two different legs (dm_arrow_leg), the legs are eliminated
// first -- four 3 x 3 factorisations, W_l = C_l^T L_l^-T, the 6 x 6 Schur complement of the base with the legs added as (l0 + l2) + (l1 + l3) -- with
// reciprocal pivots sqrt(e) * (1 / e) and fused multiply-adds
bool rows_uncoupled(const Model& m, const Env& e) {                 // no contact joins links of two different legs (the link chains a contact row walks up, ts_solve)
  if (ND != 18 || m.arrow_mode == 0) return false;
#if GO2SIM_REBUILD_FLIPS > 1
  return false;
#endif
  for (int i_c = 0; i_c < e.n_contacts; ++i_c) {
    unsigned legs = 0u;
    for (int i_ab = 0; i_ab < 2; ++i_ab)
      for (int link = i_ab ? e.contacts[i_c].link_b : e.contacts[i_c].link_a; link > -1; link = m.links[link].parent)
        if (m.links[link].n_dofs > 0 && m.links[link].dof_end > 6) legs |= 1u << dm_arrow_leg(m.arrow_mode, m.links[link].dof_end - 1);
    if (legs & (legs - 1u)) return false;
  }
  return true;
}
void cholesky_factor_arrow(const Model& m, Env& e) { arrow_factor(m.arrow_mode, m.eps, e.H, e.af); }
void cholesky_solve_arrow(Env& e) { arrow_solve(e.af, e.grad, e.Mgrad); }
#endif
// func_cholesky_factor_direct_batch, solver.py:1467-1494
void cholesky_factor_direct(const Model& m, Env& e) {
#ifdef GO2SIM_FAST_ORDER
  e.arrow = rows_uncoupled(m, e);
  if (e.arrow) { cholesky_factor_arrow(m, e); return; }
  // ts_cholesky_factor of csrc/go2sim.hip: right-looking, the column scaled by the reciprocal diagonal, fused multiply-add updates of the rows below
  for (int k = 0; k < ND; ++k) {
    const real d = dm_sqrt(std::max(e.H[k][k], m.eps));
    const real inv = 1.0f / d;
    e.H[k][k] = d;
    for (int j = k + 1; j < ND; ++j) e.H[j][k] = e.H[j][k] * inv;
    for (int j = k + 1; j < ND; ++j)
      for (int i = k + 1; i <= j; ++i) e.H[j][i] = std::fma(-e.H[j][k], e.H[i][k], e.H[j][i]);
  }
  return;
#endif
  for (int i_d = 0; i_d < ND; ++i_d) {
    real tmp = e.H[i_d][i_d];
    for (int j_d = 0; j_d < i_d; ++j_d) tmp = tmp - e.H[i_d][j_d] * e.H[i_d][j_d];
    e.H[i_d][i_d] = dm_sqrt(std::max(tmp, m.eps));
    tmp = 1.0f / e.H[i_d][i_d];
    for (int j_d = i_d + 1; j_d < ND; ++j_d) {
      real dotv = 0.0f;
      for (int k_d = 0; k_d < i_d; ++k_d) dotv = dotv + e.H[j_d][k_d] * e.H[i_d][k_d];
      e.H[j_d][i_d] = (e.H[j_d][i_d] - dotv) * tmp;
    }
  }
}
// func_hessian_and_cholesky_factor_incremental_dense_batch, solver.py:1632-1675
bool cholesky_incremental(const Model& m, Env& e) {
  bool is_degenerated = false;
#ifdef GO2SIM_FAST_ORDER
  {                                                                // ts_cholesky_incremental: REBUILD_FLIPS or more flipped rows -> the caller's rebuild path
    int n_flip = 0;
    for (int i_c = 0; i_c < e.n_con; ++i_c) n_flip += ((e.active[i_c] != 0) != (e.prev_active[i_c] != 0)) ? 1 : 0;
    if (n_flip >= GO2SIM_REBUILD_FLIPS) return true;
  }
  real invd[ND];                                                   // reciprocal diagonal of the factor, carried through the updates
  for (int k = 0; k < ND; ++k) invd[k] = 1.0f / e.H[k][k];
#endif
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    bool is_active = e.active[i_c] != 0, is_active_prev = e.prev_active[i_c] != 0;
    if (is_active ^ is_active_prev) {
      real sign = is_active ? 1.0f : -1.0f;
      real efc_D_sqrt = dm_sqrt(e.efc_D[i_c]);
      for (int i_d = 0; i_d < ND; ++i_d) e.nt_vec[i_d] = e.jac[i_c][i_d] * efc_D_sqrt;
      for (int k = 0; k < ND; ++k) {
        if (dm_abs(e.nt_vec[k]) > m.eps) {
          real Lkk = e.H[k][k];
          real tmp = Lkk * Lkk + sign * (e.nt_vec[k] * e.nt_vec[k]);
          if (tmp < m.eps) { is_degenerated = true; break; }
          real r = dm_sqrt(tmp);
#ifdef GO2SIM_FAST_ORDER
          real rinv = r * (1.0f / tmp);                              // 1 / r without a division after the square root
          real c = r * invd[k];
          real cinv = Lkk * rinv;
          real s = e.nt_vec[k] * invd[k];
          invd[k] = rinv;
#else
          real c = r / Lkk;
          real cinv = 1.0f / c;
          real s = e.nt_vec[k] / Lkk;
#endif
          e.H[k][k] = r;
          for (int i = k + 1; i < ND; ++i) e.H[i][k] = (e.H[i][k] + s * e.nt_vec[i] * sign) * cinv;
          for (int i = k + 1; i < ND; ++i) e.nt_vec[i] = e.nt_vec[i] * c - s * e.H[i][k];
        }
      }
    }
  }
  return is_degenerated;
}
// func_cholesky_solve_batch, solver.py:1747-1765
void cholesky_solve(Env& e) {
#ifdef GO2SIM_FAST_ORDER
  if (e.arrow) { cholesky_solve_arrow(e); return; }
  real linv[ND], cur[ND];                                          // column-oriented substitutions with the reciprocal diagonal
  for (int i = 0; i < ND; ++i) { linv[i] = 1.0f / e.H[i][i]; cur[i] = e.grad[i]; }
  for (int j = 0; j < ND; ++j) {
    real yj = cur[j] * linv[j];
    cur[j] = yj;
    for (int i = j + 1; i < ND; ++i) cur[i] = cur[i] - e.H[i][j] * yj;
  }
  for (int j = ND - 1; j >= 0; --j) {
    real xj = cur[j] * linv[j];
    cur[j] = xj;
    for (int i = 0; i < j; ++i) cur[i] = cur[i] - e.H[j][i] * xj;
  }
  for (int i = 0; i < ND; ++i) e.Mgrad[i] = cur[i];
  return;
#endif
  for (int i_d = 0; i_d < ND; ++i_d) {
    real cur = e.grad[i_d];
    for (int j_d = 0; j_d < i_d; ++j_d) cur = cur - e.H[i_d][j_d] * e.Mgrad[j_d];
    e.Mgrad[i_d] = cur / e.H[i_d][i_d];
  }
  for (int i_d_ = 0; i_d_ < ND; ++i_d_) {
    int i_d = ND - 1 - i_d_;
    real cur = e.Mgrad[i_d];
    for (int j_d = i_d + 1; j_d < ND; ++j_d) cur = cur - e.H[j_d][i_d] * e.Mgrad[j_d];
    e.Mgrad[i_d] = cur / e.H[i_d][i_d];
  }
}

// func_update_constraint_batch, solver.py:2428-2503 (ne = nef = 0: no equality / frictionloss rows for Go2)
void update_constraint(const Model& m, Env& e) {
  e.prev_cost = e.cost;
  real cost_i = 0.0f, gauss_i = 0.0f;
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    e.prev_active[i_c] = e.active[i_c];
    e.active[i_c] = 1;
    real floss_force = 0.0f;
    e.active[i_c] = e.Jaref[i_c] < 0.0f;
    e.efc_force[i_c] = floss_force + (-e.Jaref[i_c] * e.efc_D[i_c] * (real)e.active[i_c]);
  }
  for (int i_d = 0; i_d < ND; ++i_d) {
    real q = 0.0f;
    for (int i_c = 0; i_c < e.n_con; ++i_c) q = q + e.jac[i_c][i_d] * e.efc_force[i_c];
    e.qfrc_constraint[i_d] = q;
  }
#ifdef GO2SIM_FAST_ORDER
  const int W = fast_team(m);
  gauss_i = team_tree_sum(W, ND, [&](int i_d) { return 0.5f * (e.Ma[i_d] - e.force[i_d]) * (e.qacc[i_d] - e.acc_smooth[i_d]); });
  cost_i = team_tree_sum(W, e.n_con, [&](int i_c) { return 0.5f * (e.Jaref[i_c] * e.Jaref[i_c] * (e.efc_D[i_c] * (real)e.active[i_c])); }) + gauss_i;
#else
  for (int i_d = 0; i_d < ND; ++i_d) {
    real v = 0.5f * (e.Ma[i_d] - e.force[i_d]) * (e.qacc[i_d] - e.acc_smooth[i_d]);
    gauss_i = gauss_i + v;
    cost_i = cost_i + v;
  }
  for (int i_c = 0; i_c < e.n_con; ++i_c) cost_i = cost_i + 0.5f * (e.Jaref[i_c] * e.Jaref[i_c] * e.efc_D[i_c] * (real)e.active[i_c]);
#endif
  e.gauss = gauss_i;
  e.cost = cost_i;
}
// func_update_gradient_batch (Newton), solver.py:2530-2559
void update_gradient(Env& e) {
  for (int i_d = 0; i_d < ND; ++i_d) e.grad[i_d] = e.Ma[i_d] - e.force[i_d] - e.qfrc_constraint[i_d];
  cholesky_solve(e);
}

// ---- exact line search, solver.py:1888-2417 ------------------------------------------------------
struct LsPoint { real alpha, cost, grad, hess; };
// func_ls_init_and_eval_p0_opt, solver.py:1888-2006
LsPoint ls_init_and_eval_p0(const Model& m, Env& e) {
  for (int i_e = 0; i_e < m.n_entities; ++i_e)
    for (int i_d1 = m.entities[i_e].dof_start; i_d1 < m.entities[i_e].dof_end; ++i_d1) {
      real mv = 0.0f;
      for (int i_d2 = m.entities[i_e].dof_start; i_d2 < m.entities[i_e].dof_end; ++i_d2) mv = mv + e.mass_mat[i_d1][i_d2] * e.search[i_d2];
      e.mv[i_d1] = mv;
    }
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    real jv = 0.0f;
    for (int i_d = 0; i_d < ND; ++i_d) jv = jv + e.jac[i_c][i_d] * e.search[i_d];
    e.jv[i_c] = jv;
  }
  real qg1 = 0.0f, qg2 = 0.0f;
#ifdef GO2SIM_FAST_ORDER
  const int W = fast_team(m);
  qg1 = team_tree_sum(W, ND, [&](int i_d) { return e.search[i_d] * e.Ma[i_d] - e.search[i_d] * e.force[i_d]; });
  qg2 = team_tree_sum(W, ND, [&](int i_d) { return 0.5f * e.search[i_d] * e.mv[i_d]; });
  e.quad_gauss[0] = e.gauss; e.quad_gauss[1] = qg1; e.quad_gauss[2] = qg2;
  real t0, t1, t2;
  {
    auto qf = [&](int i_c, int k) {
      real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
      real q = (k == 0) ? D * (0.5f * Ja * Ja) : ((k == 1) ? D * (jv * Ja) : D * (0.5f * jv * jv));
      return q * (real)(Ja < 0.0f);
    };
    t0 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 0); }) + e.gauss;
    t1 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 1); }) + qg1;
    t2 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 2); }) + qg2;
  }
#else
  for (int i_d = 0; i_d < ND; ++i_d) {
    qg1 = qg1 + (e.search[i_d] * e.Ma[i_d] - e.search[i_d] * e.force[i_d]);
    qg2 = qg2 + 0.5f * e.search[i_d] * e.mv[i_d];
  }
  e.quad_gauss[0] = e.gauss; e.quad_gauss[1] = qg1; e.quad_gauss[2] = qg2;
  real t0 = e.gauss, t1 = qg1, t2 = qg2;
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
    real qf_0 = D * (0.5f * Ja * Ja), qf_1 = D * (jv * Ja), qf_2 = D * (0.5f * jv * jv);
    real active = (real)(Ja < 0.0f);
    t0 = t0 + qf_0 * active; t1 = t1 + qf_1 * active; t2 = t2 + qf_2 * active;
  }
#endif
  LsPoint p; p.alpha = 0.0f; p.cost = t0; p.grad = t1; p.hess = 2.0f * t2;
  if (p.hess <= 0.0f) p.hess = m.eps;
  e.ls_it = 1;
  return p;
}
// func_ls_point_fn_opt, solver.py:2009-2077
LsPoint ls_point_fn(const Model& m, Env& e, real alpha) {
  real t0 = e.quad_gauss[0] + 0.0f, t1 = e.quad_gauss[1] + 0.0f, t2 = e.quad_gauss[2] + 0.0f;  // + eq_sum (zero)
#ifdef GO2SIM_FAST_ORDER
  {
    const int W = fast_team(m);
    auto qf = [&](int i_c, int k) {
      real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
      real q = (k == 0) ? D * (0.5f * Ja * Ja) : ((k == 1) ? D * (jv * Ja) : D * (0.5f * jv * jv));
      return q * (real)((Ja + alpha * jv) < 0.0f);
    };
    t0 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 0); }) + t0;
    t1 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 1); }) + t1;
    t2 = team_tree_sum(W, e.n_con, [&](int c) { return qf(c, 2); }) + t2;
  }
#else
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
    real x = Ja + alpha * jv;
    real active = (real)(x < 0.0f);
    real qf_0 = D * (0.5f * Ja * Ja), qf_1 = D * (jv * Ja), qf_2 = D * (0.5f * jv * jv);
    t0 = t0 + qf_0 * active; t1 = t1 + qf_1 * active; t2 = t2 + qf_2 * active;
  }
#endif
  LsPoint p; p.alpha = alpha;
  p.cost = alpha * alpha * t2 + alpha * t1 + t0;
  p.grad = 2.0f * alpha * t2 + t1;
  p.hess = 2.0f * t2;
  if (p.hess <= 0.0f) p.hess = m.eps;
  e.ls_it = e.ls_it + 1;
  return p;
}
// func_ls_point_fn_3alphas_opt, solver.py:2080-2209
void ls_point_fn_3(const Model& m, Env& e, const real a[3], real costs[3], real grads[3], real hess[3]) {
  real b0 = e.quad_gauss[0] + 0.0f, b1 = e.quad_gauss[1] + 0.0f, b2 = e.quad_gauss[2] + 0.0f;
  real t[3][3] = {{b0, b1, b2}, {b0, b1, b2}, {b0, b1, b2}};
#ifdef GO2SIM_FAST_ORDER
  {
    const int W = fast_team(m);
    for (int k = 0; k < 3; ++k)
      for (int q = 0; q < 3; ++q)
        t[k][q] = team_tree_sum(W, e.n_con, [&](int i_c) {
          real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
          real qf = (q == 0) ? D * (0.5f * Ja * Ja) : ((q == 1) ? D * (jv * Ja) : D * (0.5f * jv * jv));
          return qf * (real)((Ja + a[k] * jv) < 0.0f);
        }) + t[k][q];
  }
#else
  for (int i_c = 0; i_c < e.n_con; ++i_c) {
    real Ja = e.Jaref[i_c], jv = e.jv[i_c], D = e.efc_D[i_c];
    real qf_0 = D * (0.5f * Ja * Ja), qf_1 = D * (jv * Ja), qf_2 = D * (0.5f * jv * jv);
    for (int k = 0; k < 3; ++k) {
      real x = Ja + a[k] * jv;
      real act = (real)(x < 0.0f);
      t[k][0] = t[k][0] + qf_0 * act; t[k][1] = t[k][1] + qf_1 * act; t[k][2] = t[k][2] + qf_2 * act;
    }
  }
#endif
  for (int k = 0; k < 3; ++k) {
    costs[k] = a[k] * a[k] * t[k][2] + a[k] * t[k][1] + t[k][0];
    grads[k] = 2.0f * a[k] * t[k][2] + t[k][1];
    hess[k] = 2.0f * t[k][2];
    if (hess[k] <= 0.0f) hess[k] = m.eps;
  }
  e.ls_it = e.ls_it + 3;
}
// update_bracket_no_eval_local, solver.py:2212-2243
int update_bracket(LsPoint& p, const real alphas[3], const real costs[3], const real grads[3], const real hess[3], real& p_next_alpha) {
  int flag = 0;
  for (int i = 0; i < 3; ++i) {
    if (p.grad < 0 && grads[i] < 0 && p.grad < grads[i]) {
      p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 1;
    } else if (p.grad > 0 && grads[i] > 0 && p.grad > grads[i]) {
      p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 2;
    }
  }
  p_next_alpha = p.alpha;
  if (flag > 0) p_next_alpha = p.alpha - p.grad / p.hess;
  return flag;
}
// func_linesearch_batch, solver.py:2246-2417
real linesearch(const Model& m, Env& e) {
  real snorm = 0.0f;
#ifdef GO2SIM_FAST_ORDER
  snorm = team_tree_sum(fast_team(m), ND, [&](int jd) { return e.search[jd] * e.search[jd]; });
#else
  for (int jd = 0; jd < ND; ++jd) snorm = snorm + e.search[jd] * e.search[jd];
#endif
  snorm = dm_sqrt(snorm);
  real scale = m.meaninertia * (real)std::max(1, ND);
  real gtol = m.tolerance * m.ls_tolerance * snorm * scale;
  e.gtol = gtol;
  e.ls_it = 0; e.ls_result = 0;
  real res_alpha = 0.0f;
  bool done = false;
  if (snorm < m.eps) {
    e.ls_result = 1; res_alpha = 0.0f;
  } else {
    LsPoint p0 = ls_init_and_eval_p0(m, e);
    LsPoint p1 = ls_point_fn(m, e, p0.alpha - p0.grad / p0.hess);
    if (p0.cost < p1.cost) p1 = p0;
    if (dm_abs(p1.grad) < gtol) {
      e.ls_result = (dm_abs(p1.alpha) < m.eps) ? 2 : 0;
      res_alpha = p1.alpha;
    } else {
      int direction = (p1.grad < 0) * 2 - 1;
      int p2update = 0;
      LsPoint p2 = p1;
      while (p1.grad * (real)direction <= -gtol && e.ls_it < m.ls_iterations) {
        p2 = p1; p2update = 1;
        p1 = ls_point_fn(m, e, p1.alpha - p1.grad / p1.hess);
        if (dm_abs(p1.grad) < gtol) { res_alpha = p1.alpha; done = true; break; }
      }
      if (!done) {
        if (e.ls_it >= m.ls_iterations) { e.ls_result = 3; res_alpha = p1.alpha; done = true; }
        if (!p2update && !done) { e.ls_result = 6; res_alpha = p1.alpha; done = true; }
        if (!done) {
          real al[3];
          al[0] = p1.alpha - p1.grad / p1.hess; al[1] = p1.alpha; al[2] = (p1.alpha + p2.alpha) * 0.5f;
          while (e.ls_it < m.ls_iterations) {
            real costs[3], grads[3], hess[3];
            ls_point_fn_3(m, e, al, costs, grads, hess);
            real p1_next_alpha = al[0], p2_next_alpha = al[1];
            real best_alpha = 0.0f, best_cost = 0.0f; bool best_found = false;
            for (int i = 0; i < 3; ++i)
              if (dm_abs(grads[i]) < gtol && (!best_found || costs[i] < best_cost)) { best_alpha = al[i]; best_cost = costs[i]; best_found = true; }
            if (best_found) {
              res_alpha = best_alpha; done = true;
            } else {
              int b1 = update_bracket(p1, al, costs, grads, hess, p1_next_alpha);
              int b2 = update_bracket(p2, al, costs, grads, hess, p2_next_alpha);
              if (b1 == 0 && b2 == 0) {
                e.ls_result = (costs[2] < p0.cost) ? 0 : 7;
                res_alpha = al[2]; done = true;
              }
            }
            if (done) break;
            al[0] = p1_next_alpha; al[1] = p2_next_alpha; al[2] = (p1.alpha + p2.alpha) * 0.5f;
          }
          if (!done) {
            if (p1.cost <= p2.cost && p1.cost < p0.cost) { e.ls_result = 4; res_alpha = p1.alpha; }
            else if (p2.cost <= p1.cost && p2.cost < p0.cost) { e.ls_result = 4; res_alpha = p2.alpha; }
            else { e.ls_result = 5; res_alpha = 0.0f; }
          }
        }
      }
    }
  }
  return res_alpha;
}

// func_solve_init (non-mujoco branch), solver.py:2739-2859
void solve_init(const Model& m, Env& e) {
  for (int i_d = 0; i_d < ND; ++i_d) e.qacc[i_d] = (e.n_con > 0 && e.is_warmstart) ? e.qacc_ws[i_d] : e.acc_smooth[i_d];
  for (int i_d1 = 0; i_d1 < ND; ++i_d1) {                        // initialize_Ma, :2714-2733
    real Ma_ = 0.0f;
    for (int i_d2 = 0; i_d2 < ND; ++i_d2) Ma_ = Ma_ + e.mass_mat[i_d1][i_d2] * e.qacc[i_d2];
    e.Ma[i_d1] = Ma_;
  }
  for (int i_c = 0; i_c < e.n_con; ++i_c) {                      // initialize_Jaref, :2691-2711
    real Jaref = -e.aref[i_c];
    for (int i_d = 0; i_d < ND; ++i_d) Jaref = Jaref + e.jac[i_c][i_d] * e.qacc[i_d];
    e.Jaref[i_c] = Jaref;
  }
  update_constraint(m, e);
  hessian_direct(m, e);
  cholesky_factor_direct(m, e);
  update_gradient(e);
  for (int i_d = 0; i_d < ND; ++i_d) e.search[i_d] = -e.Mgrad[i_d];
}
// func_solve_iter, solver.py:2862-2938
void solve_iter(const Model& m, Env& e) {
  real alpha = linesearch(m, e);
  if (dm_abs(alpha) < m.eps) {
    e.improved = 0;
  } else {
    for (int i_d = 0; i_d < ND; ++i_d) {
      e.qacc[i_d] = e.qacc[i_d] + e.search[i_d] * alpha;
      e.Ma[i_d] = e.Ma[i_d] + e.mv[i_d] * alpha;
    }
    for (int i_c = 0; i_c < e.n_con; ++i_c) e.Jaref[i_c] = e.Jaref[i_c] + e.jv[i_c] * alpha;
    update_constraint(m, e);
#ifdef GO2SIM_FAST_ORDER
    if (cholesky_incremental(m, e)) { hessian_update(m, e); cholesky_factor_direct(m, e); }
#else
    if (cholesky_incremental(m, e)) { hessian_direct(m, e); cholesky_factor_direct(m, e); }
#endif
    update_gradient(e);
    // func_terminate_or_update_descent_batch, :2645-2688
    real tol_scaled = (m.meaninertia * (real)std::max(1, ND)) * m.tolerance;
    real improvement = e.prev_cost - e.cost;
    real grad_norm = 0.0f;
#ifdef GO2SIM_FAST_ORDER
    grad_norm = team_tree_sum(fast_team(m), ND, [&](int i_d) { return e.grad[i_d] * e.grad[i_d]; });
#else
    for (int i_d = 0; i_d < ND; ++i_d) grad_norm = grad_norm + e.grad[i_d] * e.grad[i_d];
#endif
    grad_norm = dm_sqrt(grad_norm);
    e.improved = (grad_norm > tol_scaled) && (improvement > tol_scaled);
    if (e.improved)
      for (int i_d = 0; i_d < ND; ++i_d) e.search[i_d] = -e.Mgrad[i_d];
  }
}
// ConstraintSolver.resolve, solver.py:177-209
void resolve(const Model& m, Env& e) {
  solve_init(m, e);
  e.solver_iters = 0;
  if (e.n_con > 0) {                                             // func_solve_body, :2941-2966
    for (int it = 0; it < m.iterations; ++it) {
      solve_iter(m, e);
      e.solver_iters++;
      if (!e.improved) break;
    }
  } else {
    e.improved = 0;
  }
  for (int i_d = 0; i_d < ND; ++i_d) {                           // func_update_qacc, :3016-3037
    e.acc[i_d] = e.qacc[i_d];
    e.qf_constraint[i_d] = e.qfrc_constraint[i_d];
    e.force[i_d] = e.qf_smooth[i_d] + e.qfrc_constraint[i_d];
    e.qacc_ws[i_d] = e.qacc[i_d];
    if (isnanf_(e.qacc[i_d])) e.err |= GO2SIM_ERR_INVALID_FORCE_NAN;
  }
  e.is_warmstart = 1;
  for (int i_l = 0; i_l < NL; ++i_l) e.contact_force[i_l] = v3(0, 0, 0);   // func_update_contact_force, :2974-3013
  for (int i_c = 0; i_c < e.n_contacts; ++i_c) {
    Contact& c = e.contacts[i_c];
    V3 force = v3(0, 0, 0), d1, d2;
    orthogonals(c.normal, d1, d2);
    for (int i_dir = 0; i_dir < 4; ++i_dir) {
      V3 d = (real)(2 * (i_dir % 2) - 1) * ((i_dir < 2) ? d1 : d2);
      V3 n = d * c.friction - c.normal;
      force = force + n * e.efc_force[i_c * 4 + i_dir];
    }
    c.force = force;
    e.contact_force[c.link_a] = e.contact_force[c.link_a] - force;
    e.contact_force[c.link_b] = e.contact_force[c.link_b] + force;
  }
}

// ---------------------------------------------------------------------------------------------
// integration  (forward_dynamics.py:1558-1699, abd/diff.py:25-54)
// ---------------------------------------------------------------------------------------------
void integrate(const Model& m, Env& e) {
  for (int i_d = 0; i_d < ND; ++i_d) e.vel_next[i_d] = e.vel[i_d] + e.acc[i_d] * m.substep_dt;
  for (int i_l = 0; i_l < NL; ++i_l) {
    const Link& L = m.links[i_l];
    if (L.n_dofs == 0) continue;
    int ds = L.dof_start, qs = L.q_start;
    int joint_type = m.joints[L.joint_start].type;
    if (joint_type == JOINT_FREE) {
      V3 pos = v3(e.qpos[qs], e.qpos[qs + 1], e.qpos[qs + 2]);
      V3 vel = v3(e.vel_next[ds], e.vel_next[ds + 1], e.vel_next[ds + 2]);
      pos = pos + vel * m.substep_dt;
      e.qpos_next[qs] = pos.x; e.qpos_next[qs + 1] = pos.y; e.qpos_next[qs + 2] = pos.z;
      Q4 rot0 = q4(e.qpos[qs + 3], e.qpos[qs + 4], e.qpos[qs + 5], e.qpos[qs + 6]);
      V3 ang = v3(e.vel_next[ds + 3], e.vel_next[ds + 4], e.vel_next[ds + 5]) * m.substep_dt;
      Q4 qrot = rotvec_to_quat(ang, m.eps);
      Q4 rot = transform_quat_by_quat(qrot, rot0);
      e.qpos_next[qs + 3] = rot.w; e.qpos_next[qs + 4] = rot.x; e.qpos_next[qs + 5] = rot.y; e.qpos_next[qs + 6] = rot.z;
    } else {
      for (int j_ = 0; j_ < L.q_end - qs; ++j_) e.qpos_next[qs + j_] = e.qpos[qs + j_] + e.vel_next[ds + j_] * m.substep_dt;
    }
  }
  bool is_valid = true;                                          // func_copy_next_to_curr
  for (int i_d = 0; i_d < ND; ++i_d) is_valid &= !isnanf_(e.vel_next[i_d]);
  for (int i_q = 0; i_q < NQ; ++i_q) is_valid &= !isnanf_(e.qpos_next[i_q]);
  if (is_valid) {
    for (int i_d = 0; i_d < ND; ++i_d) e.vel[i_d] = e.vel_next[i_d];
    for (int i_q = 0; i_q < NQ; ++i_q) e.qpos[i_q] = e.qpos_next[i_q];
  } else {
    e.err |= GO2SIM_ERR_INVALID_ACC_NAN;
  }
}

// RigidSolver.substep, rigid_solver.py:1116-1184 (kernel_step_1 -> _func_constraint_force -> kernel_step_2)
void substep(const Model& m, Env& e) {
  forward_dynamics(m, e);                                        // kernel_step_1 (FK is already fresh)
  e.n_con = 0;                                                   // add_equality_constraints, solver.py:791-809
  update_geom_aabbs(m, e);                                       // Collider.detection, collider.py:436-528
  broad_phase(m, e);
  narrow_phase(m, e);
  add_collision_constraints(m, e);                               // add_inequality_constraints, solver.py:852-892
  add_joint_limit_constraints(m, e);
  resolve(m, e);
  integrate(m, e);                                               // kernel_step_2, rigid_solver.py:3072-3180
  update_cartesian_space(m, e, false);
  forward_velocity(m, e);
}

// ---------------------------------------------------------------------------------------------
// Go2Env (walk)  -- examples/locomotion/final/go2_env_walk.py
// ---------------------------------------------------------------------------------------------
constexpr int NA = 16, NM = 12, NOBS_MAX = 64, NPRIV_MAX = 192, NREW = 32;

struct EnvBuf {
  real actions[NA], last_actions[NA], applied_actions[NA], action_history[GO2SIM_ACTION_RING_MAX][NA]; int delay_steps;
  real target_dof_pos[NM], dof_pos[NM], dof_vel[NM], last_dof_vel[NM], torque[NM];
  real base_pos[3], base_quat[4], base_lin_vel[3], base_ang_vel[3], projected_gravity[3], base_euler[3];
  real commands[3]; int episode_length, reset_buf; real time_out;
  real base_vel_world[3];   // robot.get_vel() (world frame), used by the base-env rewards
  int terrain_row; real last_base_pos_x; unsigned terrain_key;   // go2_env_stair.py: _env_terrain_row, _last_base_pos_x; key of the row shuffle
  real kp_factors[NM], kd_factors[NM], motor_strength[NM], gravity_offset[3], current_push_force[3];
  real push_stored_force[3]; int push_remaining;
  int foot_contact[4], last_foot_contact[4]; real feet_air_time[4];
  real episode_sums[NREW], rew_terms[NREW], rew;
  real obs[NOBS_MAX], priv[NPRIV_MAX];
};

// torch-side quaternion helpers of genesis/utils/geom.py used by Go2Env (evaluation order of the torch code)
inline Q4 tc_quat_mul(Q4 u, Q4 v) {                              // geom.py:989-1007
  real w1 = u.w, x1 = u.x, y1 = u.y, z1 = u.z, w2 = v.w, x2 = v.x, y2 = v.y, z2 = v.z;
  real ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
  real xx = ww + yy + zz;
  real qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  Q4 o = q4(qq - ww + (z1 - y1) * (y2 - z2), qq - xx + (x1 + w1) * (x2 + w2), qq - yy + (w1 - x1) * (y2 + z2), qq - zz + (z1 + y1) * (w2 - x2));
  real n = dm_sqrt(norm_sqr(o));
  return q4(o.w / n, o.x / n, o.y / n, o.z / n);
}
inline V3 tc_transform_by_quat(V3 v, Q4 q) {                     // geom.py:1052-1070
  real q_ww = q.w * q.w, q_wx = q.w * q.x, q_wy = q.w * q.y, q_wz = q.w * q.z;
  real q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_yy = q.y * q.y, q_yz = q.y * q.z, q_zz = q.z * q.z;
  real den = q_ww + q_xx + q_yy + q_zz;
  real vx = v.x / den, vy = v.y / den, vz = v.z / den;
  return v3(vx * (q_xx + q_ww - q_yy - q_zz) + vy * (2.0f * q_xy - 2.0f * q_wz) + vz * (2.0f * q_xz + 2.0f * q_wy),
            vx * (2.0f * q_wz + 2.0f * q_xy) + vy * (q_ww - q_xx + q_yy - q_zz) + vz * (2.0f * q_yz - 2.0f * q_wx),
            vx * (2.0f * q_xz - 2.0f * q_wy) + vy * (2.0f * q_wx + 2.0f * q_yz) + vz * (q_ww - q_xx - q_yy + q_zz));
}
inline V3 tc_quat_to_xyz_rpy_deg(Q4 q, real eps) {               // geom.py:717-762 with rpy=True, then rad2deg
  real q_ww = q.w * q.w, q_wx = q.w * q.x, q_wy = q.w * q.y, q_wz = q.w * q.z;
  real q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_yy = q.y * q.y, q_yz = q.y * q.z, q_zz = q.z * q.z;
  real sinp = q_wy - q_xz, sinrcosp = q_wx + q_yz, sinycosp = q_wz + q_xy;
  real cosrcosp = (q_ww - q_xx - q_yy + q_zz) / 2.0f, cosycosp = (q_ww + q_xx - q_yy - q_zz) / 2.0f;
  real cosp = dm_sqrt(cosycosp * cosycosp + sinycosp * sinycosp);
  real x = dm_atan2(sinrcosp, cosrcosp), y = dm_atan2(sinp, cosp), z = dm_atan2(sinycosp, cosycosp);
  if (cosp < eps) {
    x = 0.0f;
    z = dm_atan2(q_wz - q_xy, (q_ww - q_xx + q_yy - q_zz) / 2.0f);
  }
  const real R2D = 57.29577951308232f;
  return v3(x * R2D, y * R2D, z * R2D);
}

// d[]: the host scalars in double (include/go2sim.h enum go2sim_fcfg, entries below GO2SIM_FC_N_HOST); f[]: every entry rounded to float32
struct Cfg { double d[GO2SIM_FC_N_HOST]; float f[GO2SIM_FC_COUNT]; int i[GO2SIM_IC_COUNT]; bool set; };

inline double clamp01d(double x) { return std::max(0.0, std::min(1.0, x)); }
inline double lerpd(double a, double b, double t) { t = clamp01d(t); return a + (b - a) * t; }

}  // namespace

struct go2sim {
  Model m;
  int B;
  uint64_t seed;
  std::vector<Env> envs;
  std::vector<EnvBuf> eb;
  Cfg cfg;
  go2sim_env_globals_t g;
  double acc_timeouts, acc_tracking, acc_ep[NREW];  // deterministic (env-order) accumulators of one reset call
};

namespace {

inline dm_u4 rng4(const go2sim* h, uint32_t purpose, uint32_t env, uint32_t step, uint32_t idx) {
#ifdef GO2SIM_RNG_CONST   // diagnostic build (include/go2sim_detmath.h): every word of a draw is its stream key
  dm_u4 o; o.v[0] = o.v[1] = o.v[2] = o.v[3] = step; (void)purpose; (void)env; (void)idx; (void)h; return o;
#else
  return dm_philox(env, step, purpose, idx, (uint32_t)h->seed, (uint32_t)(h->seed >> 32));
#endif
}
enum { RNG_ACTION_NOISE = 1, RNG_PUSH = 2, RNG_CMD = 3, RNG_OBS_NOISE = 4, RNG_RESET_DR = 5, RNG_GLOBAL_DR = 6, RNG_RESET_CMD = 7, RNG_RESET_POSE = 8, RNG_TERRAIN_ROW = 9, RNG_TERRAIN_PERM = 10 };
// gs_rand_float, go2_env_walk.py:7-8: `(upper - lower) * torch.rand(...) + lower` with python-float bounds: the difference is formed in float64 and
// both scalars are rounded to float32 where they meet the float32 tensor
inline real rand_float(double lower, double upper, uint32_t r) { return (real)(upper - lower) * dm_u01(r) + (real)lower; }
#ifdef GO2SIM_RNG_CONST
inline int rand_int(int lower, int upper, uint32_t r) { return lower + (int)(dm_rng_const_u(r) * (float)(upper - lower + 1)); }
#else
inline int rand_int(int lower, int upper, uint32_t r) { return lower + (int)(r % (uint32_t)(upper - lower + 1)); }   // gs_rand_int, :11-13
#endif

// Go2Env._apply_curriculum_level, go2_env_walk.py:628-686 (python float64 arithmetic)
// _get_dr_level, go2_env_stair.py:972-988 (two-phase DR schedule coupled to the terrain level)
inline double dr_level(const Cfg& c, double terrain_level) {
  if (!c.i[GO2SIM_IC_DR_SCHEDULE]) return terrain_level;
  double gate = c.d[GO2SIM_FC_DR_TERRAIN_GATE], p1 = c.d[GO2SIM_FC_DR_PHASE1_LEVEL];
  if (terrain_level < gate) return p1;
  double progress = clamp01d((terrain_level - gate) / std::max(1e-6, 1.0 - gate));
  return lerpd(p1, 1.0, progress);
}
// heightfield lookup of the env code (_get_terrain_height, go2_env_stair.py:758-770): truncation toward zero, then clamping
inline real terrain_height(const go2sim* h, real x, real y) {
  const Model& m = h->m; const Cfg& c = h->cfg;
  if (!c.i[GO2SIM_IC_USE_TERRAIN] || !m.terrain_enabled) return 0.0f;
  long col = (long)((x - c.f[GO2SIM_FC_TERRAIN_ORIGIN_X]) / c.f[GO2SIM_FC_TERRAIN_H_SCALE]);
  long row = (long)((y - c.f[GO2SIM_FC_TERRAIN_ORIGIN_Y]) / c.f[GO2SIM_FC_TERRAIN_H_SCALE]);
  col = std::min(std::max(col, 0L), (long)m.terrain_rows - 1); row = std::min(std::max(row, 0L), (long)m.terrain_cols - 1);
  return m.terrain_hf[(size_t)col * m.terrain_cols + row];
}
void apply_curriculum_level(go2sim* h) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  double lvl_terrain = c.i[GO2SIM_IC_CURR_ENABLED] ? g.level : 1.0;
  double lvl = dr_level(c, lvl_terrain);   // noise / pushes / delay follow the DR level; the command ranges follow the curriculum level
  g.obs_noise_level_cur = lerpd(0.0, c.i[GO2SIM_IC_HAS_OBS_NOISE] ? c.d[GO2SIM_FC_OBS_NOISE_LEVEL_MAX] : 0.0, lvl);
  g.action_noise_std_cur = lerpd(0.0, c.d[GO2SIM_FC_ACTION_NOISE_STD_MAX], lvl);
  double dt = c.d[GO2SIM_FC_DT];
  if (!c.i[GO2SIM_IC_HAS_PUSH]) {
    g.push_enable = 0; g.push_force_lo = g.push_force_hi = 0.0; g.push_interval = 1000000000;
  } else {
    double push_start = c.d[GO2SIM_FC_PUSH_START];
    if (lvl < push_start) {
      g.push_enable = 0; g.push_force_lo = g.push_force_hi = 0.0;
      g.push_interval = (int)(c.d[GO2SIM_FC_PUSH_INTERVAL_S_EASY] / dt);
    } else {
      double s = clamp01d((lvl - push_start) / std::max(1e-6, 1.0 - push_start));
      g.push_force_lo = c.d[GO2SIM_FC_PUSH_FORCE_LO] * s;
      g.push_force_hi = c.d[GO2SIM_FC_PUSH_FORCE_HI] * s;
      double interval_s = lerpd(c.d[GO2SIM_FC_PUSH_INTERVAL_S_EASY], c.d[GO2SIM_FC_PUSH_INTERVAL_S_HARD], s);
      g.push_interval = std::max(1, (int)(interval_s / dt));
      g.push_enable = 1;
    }
  }
  g.delay_max_cur = (int)nearbyint(lerpd((double)c.i[GO2SIM_IC_DELAY_EASY_MAX], (double)c.i[GO2SIM_IC_MAX_DELAY], lvl));
  double frac = c.i[GO2SIM_IC_CMD_CURRICULUM] ? lerpd(c.d[GO2SIM_FC_CMD_START_FRAC], 1.0, lvl_terrain) : 1.0;
  const int lo_idx[3] = {GO2SIM_FC_CMD_X_LO, GO2SIM_FC_CMD_Y_LO, GO2SIM_FC_CMD_YAW_LO};
  double* out[3][2] = {{&g.cmd_x_lo, &g.cmd_x_hi}, {&g.cmd_y_lo, &g.cmd_y_hi}, {&g.cmd_yaw_lo, &g.cmd_yaw_hi}};
  for (int k = 0; k < 3; ++k) {
    double lo = c.d[lo_idx[k]], hi = c.d[lo_idx[k] + 1];
    double center = (lo + hi) / 2.0, half = (hi - lo) / 2.0;
    *out[k][0] = center - half * frac; *out[k][1] = center + half * frac;
  }
}

// CurriculumManager.update, go2_env_walk.py:101-142
bool curriculum_update(go2sim* h, double timeout_rate, double tracking_per_sec, double fall_rate) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  double a = c.d[GO2SIM_FC_CURR_EMA_ALPHA];
  if (!g.ema_valid) { g.timeout_rate_ema = timeout_rate; g.tracking_ema = tracking_per_sec; g.fall_rate_ema = fall_rate; g.ema_valid = 1; }
  else {
    g.timeout_rate_ema = (1.0 - a) * g.timeout_rate_ema + a * timeout_rate;
    g.tracking_ema = (1.0 - a) * g.tracking_ema + a * tracking_per_sec;
    g.fall_rate_ema = (1.0 - a) * g.fall_rate_ema + a * fall_rate;
  }
  if (g.cooldown > 0) g.cooldown -= 1;
  bool ready = g.timeout_rate_ema >= c.d[GO2SIM_FC_CURR_READY_TIMEOUT_RATE] && g.tracking_ema >= c.d[GO2SIM_FC_CURR_READY_TRACKING] &&
               g.fall_rate_ema <= c.d[GO2SIM_FC_CURR_READY_FALL_RATE];
  bool hard = g.fall_rate_ema >= c.d[GO2SIM_FC_CURR_HARD_FALL_RATE];
  g.ready_streak = ready ? g.ready_streak + 1 : 0;
  g.hard_streak = hard ? g.hard_streak + 1 : 0;
  double old_level = g.level;
  if (g.hard_streak >= c.i[GO2SIM_IC_CURR_HARD_STREAK]) {
    g.level = std::max(c.d[GO2SIM_FC_CURR_LEVEL_MIN], g.level - c.d[GO2SIM_FC_CURR_STEP_DOWN]);
    g.hard_streak = 0; g.ready_streak = 0; g.cooldown = c.i[GO2SIM_IC_CURR_COOLDOWN];
  } else if (g.ready_streak >= c.i[GO2SIM_IC_CURR_READY_STREAK] && g.cooldown == 0) {
    g.level = std::min(c.d[GO2SIM_FC_CURR_LEVEL_MAX], g.level + c.d[GO2SIM_FC_CURR_STEP_UP]);
    g.ready_streak = 0; g.cooldown = c.i[GO2SIM_IC_CURR_COOLDOWN];
  }
  g.level = clamp01d(g.level);
  return g.level != old_level;
}

// _lerp_range(easy, hard, t_sample), go2_env_walk.py:37-39: python floats
inline double lerp_lo(const Cfg& c, int easy_lo, double t) { return lerpd(c.d[easy_lo], c.d[easy_lo + 2], t); }
inline double lerp_hi(const Cfg& c, int easy_lo, double t) { return lerpd(c.d[easy_lo + 1], c.d[easy_lo + 3], t); }
// Go2Env._resample_commands, go2_env_walk.py:927-963: all three components (compound commands), or ONE component chosen by `randint(0, 3)` with the
// other two left at zero (:948-958); the first `rel_standing_envs * num_envs` envs always stand (:367-368, :960-963)
inline void sample_commands(const Cfg& c, const go2sim_env_globals_t& g, const dm_u4& r, int b, real* cmd) {
  cmd[0] = rand_float(g.cmd_x_lo, g.cmd_x_hi, r.v[0]); cmd[1] = rand_float(g.cmd_y_lo, g.cmd_y_hi, r.v[1]); cmd[2] = rand_float(g.cmd_yaw_lo, g.cmd_yaw_hi, r.v[2]);
  if (!c.i[GO2SIM_IC_COMPOUND_COMMANDS]) {
    const int choice = rand_int(0, 2, r.v[3]);
    for (int k = 0; k < 3; ++k) if (k != choice) cmd[k] = 0.0f;
  }
  if (b < c.i[GO2SIM_IC_N_STANDING]) cmd[0] = cmd[1] = cmd[2] = 0.0f;
}

// Go2Env.step, pre-physics part: go2_env_walk.py:985-1023 (+ _apply_push :872-906)
void env_pre(go2sim* h, int b, const real* actions) {
  const Model& m = h->m; const Cfg& c = h->cfg; const go2sim_env_globals_t& g = h->g;
  Env& e = h->envs[b]; EnvBuf& x = h->eb[b];
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS];
  real clip = c.f[GO2SIM_FC_CLIP_ACTIONS];
  for (int i = 0; i < na; ++i) x.actions[i] = std::min(std::max(actions[i], -clip), clip);
  real delayed[NA];
  if (c.i[GO2SIM_IC_ENV_KIND] == 1) {
    // go2_env_base.py:124-125: exec_actions = last_actions (one step of latency) or the fresh actions.  last_actions is zeroed by reset_idx (:225)
    // and then overwritten with the step's actions (:187), so the step after a reset executes the action of the reset step -- there is no ring.
    for (int i = 0; i < na; ++i) { delayed[i] = c.i[GO2SIM_IC_MAX_DELAY] > 0 ? x.last_actions[i] : x.actions[i]; x.applied_actions[i] = delayed[i]; }
  } else {
    const int depth = c.i[GO2SIM_IC_MAX_DELAY] + 1;                                       // _delay_buf_size = max_delay_steps + 1, :375
    int w = g.action_write_idx;
    for (int i = 0; i < na; ++i) x.action_history[w][i] = x.actions[i];                 // _store_action :916-918
    int w_after = (w + 1) % depth;
    int read_idx = (((w_after - 1 - x.delay_steps) % depth) + depth) % depth;            // _get_delayed_action :920-923
    for (int i = 0; i < na; ++i) { delayed[i] = x.action_history[read_idx][i]; x.applied_actions[i] = delayed[i]; }
  }
  real target[NM];
  for (int i = 0; i < NM; ++i) target[i] = delayed[i] * c.f[GO2SIM_FC_ACTION_SCALE] + c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i];
  if (g.action_noise_std_cur > 0.0) {                                                  // randn_like(target) * python float: the scalar meets the tensor as float32
    for (int blk = 0; blk < 3; ++blk) {
      dm_u4 r = rng4(h, RNG_ACTION_NOISE, b, g.step_count, blk);
      real n0, n1, n2, n3;
      dm_normal2(r.v[0], r.v[1], &n0, &n1); dm_normal2(r.v[2], r.v[3], &n2, &n3);
      real nn[4] = {n0, n1, n2, n3};
      for (int k = 0; k < 4; ++k) target[4 * blk + k] = target[4 * blk + k] + nn[k] * (real)g.action_noise_std_cur;
    }
  }
  for (int i = 0; i < NM; ++i) x.target_dof_pos[i] = target[i];
  real eff_kp[NM], eff_kd[NM];
  if (c.i[GO2SIM_IC_PLS_ENABLE]) {                                                    // _compute_pls_kp_kd :969-979
    for (int leg = 0; leg < 4; ++leg) {
      real kp_leg = c.f[GO2SIM_FC_PLS_KP_DEFAULT] + delayed[NM + leg] * c.f[GO2SIM_FC_PLS_KP_ACTION_SCALE];
      kp_leg = std::min(std::max(kp_leg, c.f[GO2SIM_FC_PLS_KP_MIN]), c.f[GO2SIM_FC_PLS_KP_MAX]);
      real kd_j = 0.2f * dm_sqrt(kp_leg);
      for (int k = 0; k < 3; ++k) {
        int i = 3 * leg + k;
        eff_kp[i] = kp_leg * x.kp_factors[i] * x.motor_strength[i];
        eff_kd[i] = kd_j * x.kd_factors[i];
      }
    }
  } else {
    for (int i = 0; i < NM; ++i) { eff_kp[i] = c.f[GO2SIM_FC_KP] * x.kp_factors[i]; eff_kd[i] = c.f[GO2SIM_FC_KD] * x.kd_factors[i]; }
  }
  if (!c.i[GO2SIM_IC_MANUAL_PD]) {                                                      // go2_env_base.py:127: control_dofs_position (engine PD)
    for (int i = 0; i < NM; ++i) {
      int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
      e.ctrl_mode[d] = CTRL_POSITION; e.ctrl_pos[d] = target[i]; e.ctrl_vel[d] = 0.0f; e.ctrl_force[d] = 0.0f;
      x.torque[i] = 0.0f;
    }
  } else
  for (int i = 0; i < NM; ++i) {                                                       // :1012-1019
    real pos_error = target[i] - x.dof_pos[i];
    real torque = eff_kp[i] * pos_error - eff_kd[i] * x.dof_vel[i];
    real lim = c.f[GO2SIM_FC_TORQUE_LIMIT0 + i];
    torque = std::min(std::max(torque, -lim), lim);
    x.torque[i] = torque;
    int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
    e.ctrl_mode[d] = CTRL_FORCE; e.ctrl_force[d] = torque;
  }
  // _apply_push :872-906
  if (!c.i[GO2SIM_IC_HAS_PUSH] || !g.push_enable) {
    x.current_push_force[0] = x.current_push_force[1] = x.current_push_force[2] = 0.0f;
  } else {
    if (g.push_counter % g.push_interval == 0) {
      dm_u4 r = rng4(h, RNG_PUSH, b, g.step_count, 0);
      x.push_stored_force[0] = rand_float(g.push_force_lo, g.push_force_hi, r.v[0]);
      x.push_stored_force[1] = rand_float(g.push_force_lo, g.push_force_hi, r.v[1]);
      x.push_stored_force[2] = 0.0f;
      x.push_remaining = rand_int(c.i[GO2SIM_IC_PUSH_DUR_LO], c.i[GO2SIM_IC_PUSH_DUR_HI], r.v[2]);
    }
    real active = (x.push_remaining > 0) ? 1.0f : 0.0f;
    V3 force = v3(x.push_stored_force[0] * active, x.push_stored_force[1] * active, x.push_stored_force[2] * active);
    x.current_push_force[0] = force.x; x.current_push_force[1] = force.y; x.current_push_force[2] = force.z;
    x.push_remaining = std::max(x.push_remaining - 1, 0);
    int l = c.i[GO2SIM_IC_PUSH_LINK];                                                   // func_apply_link_external_force ref=link_origin, misc.py:695-715
    V3 torque = cross(e.l_pos[l] - e.root_com[l], force);
    e.ext_vel[l] = e.ext_vel[l] - force;
    e.ext_ang[l] = e.ext_ang[l] - torque;
  }
}

inline real fmaxr(real a, real b) { return (a < b) ? b : a; }   // torch.clamp(min=) on non-NaN input
inline real fminr(real a, real b) { return (b < a) ? b : a; }
inline real reward_term(const go2sim* h, int b, int id, const real* link_vel_xy /*[4][2]*/, const real* foot_z, const real* foot_xy = nullptr /*[4][2]*/) {
  const Cfg& c = h->cfg; const Env& e = h->envs[b]; EnvBuf& x = const_cast<EnvBuf&>(h->eb[b]);
  const real dt = c.f[GO2SIM_FC_DT];
  real cmd_norm = dm_sqrt(x.commands[0] * x.commands[0] + x.commands[1] * x.commands[1] + x.commands[2] * x.commands[2]);
  real still = (cmd_norm < 0.1f) ? 1.0f : 0.0f;
  real moving = (dm_sqrt(x.commands[0] * x.commands[0] + x.commands[1] * x.commands[1]) > 0.1f) ? 1.0f : 0.0f;
  switch (id) {
    case GO2SIM_R_TRACKING_LIN_VEL: {                                                   // :1251-1253
      real d0 = x.commands[0] - x.base_lin_vel[0], d1 = x.commands[1] - x.base_lin_vel[1];
      return dm_exp(-(d0 * d0 + d1 * d1) / c.f[GO2SIM_FC_TRACKING_SIGMA]);
    }
    case GO2SIM_R_TRACKING_ANG_VEL: { real d = x.commands[2] - x.base_ang_vel[2]; return dm_exp(-(d * d) / c.f[GO2SIM_FC_TRACKING_SIGMA]); }
    case GO2SIM_R_LIN_VEL_Z: {                                                           // go2_env_stair.py:1615-1626 (deadzone 0 = walk env)
      real dz = c.f[GO2SIM_FC_LIN_VEL_Z_DEADZONE];
      if (dz > 0.0f) { real ex = fmaxr(dm_abs(x.base_lin_vel[2]) - dz, 0.0f); return ex * ex; }
      return x.base_lin_vel[2] * x.base_lin_vel[2];
    }
    case GO2SIM_R_ACTION_RATE: { real s = 0.0f; for (int i = 0; i < c.i[GO2SIM_IC_NUM_ACTIONS]; ++i) { real d = x.last_actions[i] - x.actions[i]; s = s + d * d; } return s; }
    case GO2SIM_R_SIMILAR_TO_DEFAULT: { real s = 0.0f; for (int i = 0; i < NM; ++i) s = s + dm_abs(x.dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]); return s; }
    case GO2SIM_R_BASE_HEIGHT: {                                                         // go2_env_stair.py:1634-1648: height above the local terrain
      real hgt = x.base_pos[2];
      if (c.i[GO2SIM_IC_USE_TERRAIN]) hgt = x.base_pos[2] - terrain_height(h, x.base_pos[0], x.base_pos[1]);
      real d = hgt - c.f[GO2SIM_FC_BASE_HEIGHT_TARGET]; return d * d;
    }
    case GO2SIM_R_DOF_ACC: { real s = 0.0f; for (int i = 0; i < NM; ++i) { real a = (x.dof_vel[i] - x.last_dof_vel[i]) / dt; s = s + a * a; } return s; }
    case GO2SIM_R_DOF_VEL: { real s = 0.0f; for (int i = 0; i < NM; ++i) s = s + x.dof_vel[i] * x.dof_vel[i]; return s; }
    case GO2SIM_R_ORIENTATION_PENALTY: return x.projected_gravity[0] * x.projected_gravity[0] + x.projected_gravity[1] * x.projected_gravity[1];
    case GO2SIM_R_ANG_VEL_XY: return x.base_ang_vel[0] * x.base_ang_vel[0] + x.base_ang_vel[1] * x.base_ang_vel[1];
    case GO2SIM_R_STAND_STILL: { real s = 0.0f; for (int i = 0; i < NM; ++i) s = s + dm_abs(x.dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]); return s * still; }
    case GO2SIM_R_STAND_STILL_VEL: {
      real lin = x.base_lin_vel[0] * x.base_lin_vel[0] + x.base_lin_vel[1] * x.base_lin_vel[1];
      real ang = x.base_ang_vel[2] * x.base_ang_vel[2];
      return (lin + 0.5f * ang) * still;
    }
    case GO2SIM_R_FEET_STANCE: {                                                         // :1296-1301
      real sa = 0.0f, sn = 0.0f;
      for (int i = 0; i < 4; ++i) { sa = sa + x.feet_air_time[i]; sn = sn + (x.foot_contact[i] ? 0.0f : 1.0f); }
      return (sa + sn) * still;
    }
    case GO2SIM_R_FEET_AIR_TIME: {                                                       // :1303-1314 (mutates _feet_air_time)
      real first[4];
      for (int i = 0; i < 4; ++i) first[i] = (x.feet_air_time[i] > 0.0f && x.foot_contact[i]) ? 1.0f : 0.0f;
      for (int i = 0; i < 4; ++i) { x.feet_air_time[i] = x.feet_air_time[i] + dt; x.feet_air_time[i] = x.feet_air_time[i] * (x.foot_contact[i] ? 0.0f : 1.0f); }
      real s = 0.0f;
      for (int i = 0; i < 4; ++i) s = s + (x.feet_air_time[i] - c.f[GO2SIM_FC_FEET_AIR_TIME_TARGET]) * first[i];
      return s * moving;
    }
    case GO2SIM_R_FOOT_SLIP: {                                                           // :1316-1325
      real slip = 0.0f;
      for (int i = 0; i < 4; ++i) { real vx = link_vel_xy[2 * i], vy = link_vel_xy[2 * i + 1]; slip = slip + (x.foot_contact[i] ? 1.0f : 0.0f) * (vx * vx + vy * vy); }
      return slip;
    }
    case GO2SIM_R_FOOT_CLEARANCE: {                                                      // :1331-1355
      real pen = 0.0f;
      for (int i = 0; i < 4; ++i) {
        real vx = link_vel_xy[2 * i], vy = link_vel_xy[2 * i + 1];
        real vn = dm_sqrt(vx * vx + vy * vy);
        real fz = foot_z[i];
        if (c.i[GO2SIM_IC_USE_TERRAIN]) fz = foot_z[i] - terrain_height(h, foot_xy[2 * i], foot_xy[2 * i + 1]);   // go2_env_stair.py:1742-1747
        real he = c.f[GO2SIM_FC_FEET_HEIGHT_TARGET] - fz; he = he * he;
        pen = pen + (x.foot_contact[i] ? 0.0f : 1.0f) * he * vn;
      }
      return pen * moving;
    }
    case GO2SIM_R_JOINT_TRACKING: { real s = 0.0f; for (int i = 0; i < NM; ++i) { real d = x.target_dof_pos[i] - x.dof_pos[i]; s = s + d * d; } return s; }
    case GO2SIM_R_ENERGY: case GO2SIM_R_TORQUE_LOAD: {                                   // :1360-1366, get_dofs_control_force (accessor.py:848-875)
      real s = 0.0f;
      for (int i = 0; i < NM; ++i) {
        int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
        real tau = clampf(e.ctrl_force[d], h->m.dofs[d].force_range[0], h->m.dofs[d].force_range[1]);
        s = s + ((id == GO2SIM_R_ENERGY) ? dm_abs(tau * x.dof_vel[i]) : dm_abs(tau));
      }
      return s;
    }
    // ---- go2_env_base.py:246-390 (crouch / jump) ----
    case GO2SIM_R_JUMP_IMPULSE: { real gate = (x.base_pos[2] < 0.50f) ? 1.0f : 0.0f; return gate * fmaxr(x.base_lin_vel[2], 0.0f); }
    case GO2SIM_R_JUMP_APEX: { real q = (x.base_pos[2] - c.f[GO2SIM_FC_JUMP_APEX_HEIGHT]) / c.f[GO2SIM_FC_JUMP_APEX_SIGMA]; return dm_exp(-(q * q)); }
    case GO2SIM_R_XY_STABILITY: return -(x.base_vel_world[0] * x.base_vel_world[0] + x.base_vel_world[1] * x.base_vel_world[1]);
    case GO2SIM_R_ORIENTATION: return -x.projected_gravity[2];
    case GO2SIM_R_NO_SHAKE: return -((x.base_ang_vel[0] * x.base_ang_vel[0] + x.base_ang_vel[1] * x.base_ang_vel[1]) + x.base_ang_vel[2] * x.base_ang_vel[2]) / 1.0f;
    case GO2SIM_R_CROUCH: return (x.base_pos[2] < 0.25f) ? 1.0f : 0.0f;
    case GO2SIM_R_CROUCH_2: return (x.base_pos[2] <= 0.30f && x.base_pos[2] >= 0.20f) ? 1.0f : 0.0f;
    case GO2SIM_R_GROUND_PENALTY: { real v = (0.15f - x.base_pos[2]) / 0.1f; v = fminr(fmaxr(v, 0.0f), 1.0f); return -(v * v); }
    case GO2SIM_R_CROUCH_TARGET: { real q = (x.base_pos[2] - 0.15f) / 0.03f; return dm_exp(-(q * q)); }
    case GO2SIM_R_NO_FALL: { real dn = fmaxr(-x.base_lin_vel[2] - 0.5f, 0.0f); return -(dn * dn); }
    case GO2SIM_R_Y_STABILITY: return -(x.base_vel_world[1] * x.base_vel_world[1]);
    case GO2SIM_R_TORQUE_LOAD_BASE: {                                                    // get_dofs_control_force of the current state, accessor.py:848-875
      real s = 0.0f;
      for (int i = 0; i < NM; ++i) {
        int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
        const Dof& D = h->m.dofs[d];
        real force = 0.0f;
        if (e.ctrl_mode[d] == CTRL_FORCE) force = e.ctrl_force[d];
        else if (e.ctrl_mode[d] == CTRL_VELOCITY) force = D.kv * (e.ctrl_vel[d] - e.vel[d]);
        else if (e.ctrl_mode[d] == CTRL_POSITION) force = D.kp * (e.ctrl_pos[d] - e.dof_pos[d]) + D.kv * (e.ctrl_vel[d] - e.vel[d]);
        s = s + dm_abs(clampf(force, D.force_range[0], D.force_range[1]));
      }
      return -0.001f * s;
    }
    case GO2SIM_R_CROUCH_PROGRESS: return fmaxr(0.35f - x.base_pos[2], 0.0f);
    case GO2SIM_R_CROUCH_SPEED: return -(x.base_lin_vel[2] * x.base_lin_vel[2]);
    // ---- go2_env_stair.py:1659-1771 ----
    case GO2SIM_R_ORIENTATION_ROLL_ONLY: return x.projected_gravity[1] * x.projected_gravity[1];
    case GO2SIM_R_FORWARD_PROGRESS: { real dx = x.base_pos[0] - x.last_base_pos_x; x.last_base_pos_x = x.base_pos[0]; return dx; }   // mutates _last_base_pos_x
  }
  return 0.0f;
}

// Go2Env.step post-physics, part A: state read-back, commands, termination, rewards  (go2_env_walk.py:1026-1077)
void env_post_a(go2sim* h, int b) {
  const Model& m = h->m; const Cfg& c = h->cfg; const go2sim_env_globals_t& g = h->g;
  Env& e = h->envs[b]; EnvBuf& x = h->eb[b];
  x.episode_length += 1;
  int bl = c.i[GO2SIM_IC_BASE_LINK];
  x.base_pos[0] = e.l_pos[bl].x; x.base_pos[1] = e.l_pos[bl].y; x.base_pos[2] = e.l_pos[bl].z;
  Q4 bq = e.l_quat[bl];
  x.base_quat[0] = bq.w; x.base_quat[1] = bq.x; x.base_quat[2] = bq.y; x.base_quat[3] = bq.z;
  Q4 inv_init = inv_quat(q4(c.f[GO2SIM_FC_BASE_INIT_QUAT0], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 1], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 2], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 3]));
  V3 eul = tc_quat_to_xyz_rpy_deg(tc_quat_mul(bq, inv_init), m.eps);
  x.base_euler[0] = eul.x; x.base_euler[1] = eul.y; x.base_euler[2] = eul.z;
  Q4 inv_bq = inv_quat(bq);
  V3 vel = e.cd_vel[bl] + cross(e.cd_ang[bl], e.l_pos[bl] - e.root_com[bl]);             // get_vel: kernel_get_links_vel ref=link_origin
  x.base_vel_world[0] = vel.x; x.base_vel_world[1] = vel.y; x.base_vel_world[2] = vel.z;
  V3 blv = tc_transform_by_quat(vel, inv_bq), bav = tc_transform_by_quat(e.cd_ang[bl], inv_bq);
  V3 pg = tc_transform_by_quat(v3(0.0f, 0.0f, -1.0f), inv_bq);
  x.base_lin_vel[0] = blv.x; x.base_lin_vel[1] = blv.y; x.base_lin_vel[2] = blv.z;
  x.base_ang_vel[0] = bav.x; x.base_ang_vel[1] = bav.y; x.base_ang_vel[2] = bav.z;
  x.projected_gravity[0] = pg.x; x.projected_gravity[1] = pg.y; x.projected_gravity[2] = pg.z;
  for (int i = 0; i < NM; ++i) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i]; x.dof_pos[i] = e.dof_pos[d]; x.dof_vel[i] = e.vel[d]; }
  real link_vel_xy[8], foot_z[4], foot_xy[8];
  for (int i = 0; i < 4; ++i) {                                                         // _update_foot_contacts :599-605
    int l = c.i[GO2SIM_IC_FOOT_LINK0 + i];
    x.last_foot_contact[i] = x.foot_contact[i];
    x.foot_contact[i] = dm_abs(e.contact_force[l].z) > c.f[GO2SIM_FC_FOOT_CONTACT_THRESHOLD];
    V3 lv = e.cd_vel[l] + cross(e.cd_ang[l], e.l_pos[l] - e.root_com[l]);
    link_vel_xy[2 * i] = lv.x; link_vel_xy[2 * i + 1] = lv.y; foot_z[i] = e.l_pos[l].z; foot_xy[2 * i] = e.l_pos[l].x; foot_xy[2 * i + 1] = e.l_pos[l].y;
  }
  if (x.episode_length % c.i[GO2SIM_IC_RESAMPLE_STEPS] == 0) {                          // _resample_commands :927-963
    dm_u4 r = rng4(h, RNG_CMD, b, g.step_count, 0);
    sample_commands(c, g, r, b, x.commands);
  }
  int maxlen = c.i[GO2SIM_IC_MAX_EPISODE_LENGTH];
  int rst = x.episode_length > maxlen;                                                   // :1062-1070
  rst |= dm_abs(x.base_euler[1]) > c.f[GO2SIM_FC_TERM_PITCH_DEG];
  rst |= dm_abs(x.base_euler[0]) > c.f[GO2SIM_FC_TERM_ROLL_DEG];
  rst |= dm_abs(x.base_lin_vel[2]) > c.f[GO2SIM_FC_TERM_ZVEL];
  rst |= dm_abs(x.base_lin_vel[1]) > c.f[GO2SIM_FC_TERM_YVEL];
  x.reset_buf = rst;
  x.time_out = (x.episode_length > maxlen) ? 1.0f : 0.0f;
  if (c.i[GO2SIM_IC_ENV_KIND] == 1) return;                                              // base env: rewards follow the reset (go2_env_base.py:165-172)
  x.rew = 0.0f;                                                                          // :1072-1077
  for (int k = 0; k < c.i[GO2SIM_IC_N_REWARDS]; ++k) {
    real r = reward_term(h, b, c.i[GO2SIM_IC_REWARD_ID0 + k], link_vel_xy, foot_z, foot_xy) * c.f[GO2SIM_FC_REWARD_SCALE0 + k];
    x.rew_terms[k] = r;
    x.rew = x.rew + r;
    x.episode_sums[k] = x.episode_sums[k] + r;
  }
}

// reset-call statistics of one env (go2_env_walk.py:688-715,1228-1235), accumulated in env order
void env_reset_stats(go2sim* h, int b) {
  const Cfg& c = h->cfg; EnvBuf& x = h->eb[b];
  real ep_steps = std::max((real)x.episode_length, 1.0f);
  real ep_seconds = ep_steps * c.f[GO2SIM_FC_DT];
  real tracking_int = 0.0f;
  for (int k = 0; k < c.i[GO2SIM_IC_N_REWARDS]; ++k) {
    int id = c.i[GO2SIM_IC_REWARD_ID0 + k];
    if (id == GO2SIM_R_TRACKING_LIN_VEL || id == GO2SIM_R_TRACKING_ANG_VEL) tracking_int = tracking_int + x.episode_sums[k];
    if (c.i[GO2SIM_IC_ENV_KIND] == 1) h->acc_ep[k] += (double)x.episode_sums[k];          // go2_env_base.py:232-236: mean(sum) / episode_length_s
    else h->acc_ep[k] += (double)(x.episode_sums[k] / ep_seconds);
  }
  h->acc_tracking += (double)(tracking_int / ep_seconds);
  h->acc_timeouts += (double)x.time_out;
  h->g.n_reset_now += 1;
}

// the update step of _maybe_update_curriculum_on_reset (go2_env_walk.py:717-729) on the accumulated counters
void globals_curriculum_check(go2sim* h) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  if (g.curr_ep_total < c.i[GO2SIM_IC_CURR_UPDATE_EVERY]) return;
  double timeout_rate = g.curr_timeout_total / std::max(1, g.curr_ep_total);
  double fall_rate = 1.0 - timeout_rate;
  double tracking_avg = g.curr_tracking_sum / std::max(1, g.curr_tracking_n);
  if (curriculum_update(h, timeout_rate, tracking_avg, fall_rate)) apply_curriculum_level(h);
  g.curr_ep_total = 0; g.curr_timeout_total = 0.0; g.curr_tracking_sum = 0.0; g.curr_tracking_n = 0;
}
// t_sample (CurriculumManager.sample_level :85-93) and the "global" DR draws (:737-756, 803-848) of one reset call; `n_throttle` resets are counted
// for the friction throttle, `key` numbers the call in the Philox stream
void globals_draws(go2sim* h, int n_throttle, uint32_t key) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  dm_u4 r0 = rng4(h, RNG_GLOBAL_DR, 0xffffffffu, key, 0);
  dm_u4 r1 = rng4(h, RNG_GLOBAL_DR, 0xffffffffu, key, 1);
  dm_u4 r2 = rng4(h, RNG_GLOBAL_DR, 0xffffffffu, key, 2);
  double t;                                                                             // CurriculumManager.sample_level :85-93
  if (c.i[GO2SIM_IC_DR_SCHEDULE]) t = dr_level(c, c.i[GO2SIM_IC_CURR_ENABLED] ? g.level : 1.0);   // go2_env_stair.py:1506-1507
  else if (!c.i[GO2SIM_IC_CURR_ENABLED]) t = 1.0;
  else if (dm_u01(r0.v[0]) < c.f[GO2SIM_FC_CURR_MIX_PROB_CURRENT]) t = clamp01d(g.level);
  else {
    double hi = std::min(g.level, c.d[GO2SIM_FC_CURR_MIX_LEVEL_HIGH]);
    double lo = std::min(c.d[GO2SIM_FC_CURR_MIX_LEVEL_LOW], hi);
    t = clamp01d(lo + (hi - lo) * (double)dm_u01(r0.v[1]));
  }
  g.t_sample = t;
  double ts = g.t_sample;
  if (c.i[GO2SIM_IC_HAS_FRICTION_DR]) {                                                  // _randomize_friction :737-756
    g.global_dr_reset_counter += n_throttle;
    if (g.global_dr_reset_counter >= c.i[GO2SIM_IC_GLOBAL_DR_INTERVAL]) {
      g.global_dr_reset_counter = 0;
      g.friction = rand_float(lerp_lo(c, GO2SIM_FC_FRICTION_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_FRICTION_EASY_LO, ts), r0.v[2]);
    }
  }
  if (c.i[GO2SIM_IC_HAS_MASS_DR])                                                        // _randomize_mass :803-822
    g.mass_shift = rand_float(lerp_lo(c, GO2SIM_FC_MASS_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MASS_EASY_LO, ts), r0.v[3]);
  if (c.i[GO2SIM_IC_HAS_COM_DR])
    for (int k = 0; k < 3; ++k) g.com_shift[k] = rand_float(lerp_lo(c, GO2SIM_FC_COM_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_COM_EASY_LO, ts), r1.v[k]);
  if (c.i[GO2SIM_IC_HAS_LEGM_DR])                                                        // _randomize_leg_mass :834-848
    for (int k = 0; k < 4; ++k) g.leg_mass_shift[k] = rand_float(lerp_lo(c, GO2SIM_FC_LEGM_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_LEGM_EASY_LO, ts), r2.v[k]);
}

// single-instance part of reset_idx: curriculum, t_sample, "global" DR  (go2_env_walk.py:688-756,803-848,1160-1171)
void env_globals_update(go2sim* h, bool count_push) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  if (count_push && c.i[GO2SIM_IC_HAS_PUSH] && g.push_enable) g.push_counter += 1;
  int n = g.n_reset_now;
  if (n > 0) {
    // `float(tensor.sum().item())` of float32 tensors (:698, :710) added to python floats
    const double timeouts = (double)(float)h->acc_timeouts, tracking = (double)(float)h->acc_tracking;
    if (c.i[GO2SIM_IC_SHARED_GLOBALS]) {              // one shard of a larger batch: the increments are combined by the host (go2sim_env_sync_*)
      g.shard_counters[0] += n; g.shard_counters[1] += timeouts; g.shard_counters[2] += tracking; g.shard_counters[3] += n;
      if (!(g.sync_calls > 0 && g.reset_calls == 0)) g.shard_counters[4] += n;   // (the constructor's reset after an initial sync: that sync already counted these envs for the friction throttle)
    } else {
      if (c.i[GO2SIM_IC_CURR_ENABLED] && !c.i[GO2SIM_IC_FREEZE_CURRICULUM]) {                // _maybe_update_curriculum_on_reset
        g.curr_ep_total += n; g.curr_timeout_total += timeouts; g.curr_tracking_sum += tracking; g.curr_tracking_n += n;
        globals_curriculum_check(h);
      }
      globals_draws(h, n, g.reset_calls);
    }
    g.last_reset_count = n;
    for (int k = 0; k < NREW; ++k)
      g.last_episode_rew[k] = (c.i[GO2SIM_IC_ENV_KIND] == 1) ? (float)((double)(float)(h->acc_ep[k] / (double)n) / c.d[GO2SIM_FC_EPISODE_LENGTH_S]) : (float)(h->acc_ep[k] / (double)n);
    g.reset_calls += 1;
  }
}

// per-env part of reset_idx  (go2_env_walk.py:1156-1240)
void env_reset_one(go2sim* h, int b) {
  const Model& m = h->m; const Cfg& c = h->cfg; const go2sim_env_globals_t& g = h->g;
  Env& e = h->envs[b]; EnvBuf& x = h->eb[b];
  uint32_t rc = g.reset_calls - 1;  // id of the current reset call (already advanced by env_globals_update)
  double ts = g.t_sample;
  if (c.i[GO2SIM_IC_HAS_KPF_DR])                                                          // _randomize_kp_kd (PLS branch) :763-773
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(h, RNG_RESET_DR, b, rc, blk); for (int k = 0; k < 4; ++k) x.kp_factors[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_KPF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_KPF_EASY_LO, ts), r.v[k]); }
  if (c.i[GO2SIM_IC_HAS_KDF_DR])
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(h, RNG_RESET_DR, b, rc, 3 + blk); for (int k = 0; k < 4; ++k) x.kd_factors[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_KDF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_KDF_EASY_LO, ts), r.v[k]); }
  if (c.i[GO2SIM_IC_HAS_GOFF_DR]) {                                                        // _randomize_gravity_offset :824-832
    dm_u4 r = rng4(h, RNG_RESET_DR, b, rc, 6);
    for (int k = 0; k < 3; ++k) x.gravity_offset[k] = rand_float(lerp_lo(c, GO2SIM_FC_GOFF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_GOFF_EASY_LO, ts), r.v[k]);
  }
  if (c.i[GO2SIM_IC_HAS_MSTR_DR])                                                          // _randomize_motor_strength :850-858
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(h, RNG_RESET_DR, b, rc, 7 + blk); for (int k = 0; k < 4; ++k) x.motor_strength[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_MSTR_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MSTR_EASY_LO, ts), r.v[k]); }
  if (c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR]) {   // extension (BASELINE configs[4], not in the reference): the friction / base-mass scalars are drawn per env
    dm_u4 r = rng4(h, RNG_RESET_DR, b, rc, 10);
    if (c.i[GO2SIM_IC_HAS_FRICTION_DR]) {
      real mu = rand_float(lerp_lo(c, GO2SIM_FC_FRICTION_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_FRICTION_EASY_LO, ts), r.v[0]);
      for (int i = 0; i < NG; ++i) e.geom_friction[i] = mu;
    }
    if (c.i[GO2SIM_IC_HAS_MASS_DR])
      e.mass_shift[c.i[GO2SIM_IC_BASE_LINK]] = rand_float(lerp_lo(c, GO2SIM_FC_MASS_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MASS_EASY_LO, ts), r.v[1]);
  }
  dm_u4 rp = rng4(h, RNG_RESET_POSE, b, rc, 0);
  {                                                                                        // _randomize_delay :860-866
    int max_d = std::max(c.i[GO2SIM_IC_MIN_DELAY], std::min(g.delay_max_cur, c.i[GO2SIM_IC_MAX_DELAY]));
    x.delay_steps = rand_int(c.i[GO2SIM_IC_MIN_DELAY], max_d, rp.v[3]);
  }
  // reset dofs / base (set_dofs_position, set_pos, set_quat, zero_all_dofs_velocity :1173-1205)
  for (int i = 0; i < NM; ++i) {
    x.dof_pos[i] = c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]; x.dof_vel[i] = 0.0f;
    int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
    int q = d + 1;                                                                         // revolute dof d <-> qpos index d+1 (free joint has 7 q / 6 dofs)
    e.qpos[q] = m.qpos0[q] + x.dof_pos[i];
  }
  for (int d = 0; d < ND; ++d) e.vel[d] = 0.0f;
  e.err = 0; e.is_warmstart = 0;                                                           // rigid_solver.py:2403-2410
  for (int d = 0; d < ND; ++d) e.qacc_ws[d] = 0.0f;
  for (int p = 0; p < NPAIR; ++p) e.normal_cache[p] = v3(0, 0, 0);
  x.base_pos[0] = c.f[GO2SIM_FC_BASE_INIT_POS0]; x.base_pos[1] = c.f[GO2SIM_FC_BASE_INIT_POS0 + 1]; x.base_pos[2] = c.f[GO2SIM_FC_BASE_INIT_POS0 + 2];
  for (int k = 0; k < 4; ++k) x.base_quat[k] = c.f[GO2SIM_FC_BASE_INIT_QUAT0 + k];
  if (c.i[GO2SIM_IC_USE_TERRAIN]) {                                                         // _get_terrain_spawn_pos, go2_env_stair.py:856-871, :1531-1540
    const float* rcn = &c.f[GO2SIM_FC_ROW_CENTER0 + 3 * x.terrain_row];
    real init_z = c.f[GO2SIM_FC_BASE_INIT_POS0 + 2];
    real spawn_z = rcn[2] + init_z;
    x.base_pos[0] = rcn[0]; x.base_pos[1] = rcn[1]; x.base_pos[2] = spawn_z;
    if (c.i[GO2SIM_IC_HAS_INIT_Z]) x.base_pos[2] = ((spawn_z + rand_float(c.d[GO2SIM_FC_INIT_Z_LO], c.d[GO2SIM_FC_INIT_Z_HI], rp.v[0])) - init_z) + init_z;
  } else
  if (c.i[GO2SIM_IC_HAS_INIT_Z]) x.base_pos[2] = rand_float(c.d[GO2SIM_FC_INIT_Z_LO], c.d[GO2SIM_FC_INIT_Z_HI], rp.v[0]);
  if (c.i[GO2SIM_IC_HAS_INIT_EULER]) {                                                     // :1191-1199, euler_to_quat_wxyz :16-25
    const double D2R = 3.141592653589793 / 180.0;                                          // math.radians: x * (pi / 180) in float64
    double lo = c.d[GO2SIM_FC_INIT_EULER_LO_DEG] * D2R, hi = c.d[GO2SIM_FC_INIT_EULER_HI_DEG] * D2R;
    real roll = rand_float(lo, hi, rp.v[1]), pitch = rand_float(lo, hi, rp.v[2]), yaw = 0.0f;
    real sr, cr, sp, cp, sy, cy;
    dm_sincos(roll / 2.0f, &sr, &cr); dm_sincos(pitch / 2.0f, &sp, &cp); dm_sincos(yaw / 2.0f, &sy, &cy);
    x.base_quat[0] = cr * cp * cy + sr * sp * sy; x.base_quat[1] = sr * cp * cy - cr * sp * sy;
    x.base_quat[2] = cr * sp * cy + sr * cp * sy; x.base_quat[3] = cr * cp * sy - sr * sp * cy;
  }
  for (int k = 0; k < 3; ++k) e.qpos[k] = x.base_pos[k];
  for (int k = 0; k < 4; ++k) e.qpos[3 + k] = x.base_quat[k];
  for (int k = 0; k < 3; ++k) { x.base_lin_vel[k] = 0.0f; x.base_ang_vel[k] = 0.0f; }
  for (int i = 0; i < NA; ++i) { x.last_actions[i] = 0.0f; x.applied_actions[i] = 0.0f; for (int k = 0; k < GO2SIM_ACTION_RING_MAX; ++k) x.action_history[k][i] = 0.0f; }
  for (int i = 0; i < NM; ++i) x.last_dof_vel[i] = 0.0f;
  x.last_base_pos_x = x.base_pos[0];                                                       // go2_env_stair.py:1557
  for (int k = 0; k < 3; ++k) x.push_stored_force[k] = 0.0f;
  x.push_remaining = 0;
  for (int i = 0; i < 4; ++i) { x.feet_air_time[i] = 0.0f; x.foot_contact[i] = 0; x.last_foot_contact[i] = 0; }
  for (int k = 0; k < NREW; ++k) x.episode_sums[k] = 0.0f;
  x.episode_length = 0; x.reset_buf = 1;
  dm_u4 r = rng4(h, RNG_RESET_CMD, b, rc, 0);                                              // _resample_commands(envs_idx) :1240
  sample_commands(c, g, r, b, x.commands);
}

// broadcast of the "global" DR scalars to one env + full-batch FK refresh done by the reference's
// set_dofs_position / set_pos / set_quat / zero_all_dofs_velocity calls (rigid_solver.py:1928-1943,2412-2427)
void env_apply_globals_and_fk(go2sim* h, int b) {
  const Model& m = h->m; const Cfg& c = h->cfg; const go2sim_env_globals_t& g = h->g;
  Env& e = h->envs[b];
  const bool per_env = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] != 0;   // per-env draws were applied by env_reset_one
  if (c.i[GO2SIM_IC_HAS_FRICTION_DR] && !per_env) for (int i = 0; i < NG; ++i) e.geom_friction[i] = g.friction;
  int bl = c.i[GO2SIM_IC_BASE_LINK];
  if (c.i[GO2SIM_IC_HAS_MASS_DR] && !per_env) e.mass_shift[bl] = g.mass_shift;
  if (c.i[GO2SIM_IC_HAS_COM_DR]) e.com_shift[bl] = v3(g.com_shift[0], g.com_shift[1], g.com_shift[2]);
  if (c.i[GO2SIM_IC_HAS_LEGM_DR]) for (int k = 0; k < 4; ++k) e.mass_shift[c.i[GO2SIM_IC_HIP_LINK0 + k]] = g.leg_mass_shift[k];
  update_cartesian_space(m, e, true);
  forward_velocity(m, e);
}

// Go2Env.step post-physics, part B: observations  (go2_env_walk.py:1082-1141)
void env_post_b(go2sim* h, int b, real* obs, real* priv) {
  const Cfg& c = h->cfg; const go2sim_env_globals_t& g = h->g;
  EnvBuf& x = h->eb[b];
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS], nobs = c.i[GO2SIM_IC_NUM_OBS], npriv = c.i[GO2SIM_IC_NUM_PRIV_OBS];
  real* o = x.obs;
  const real cs[3] = {c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL], c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL], c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL]};
  if (c.i[GO2SIM_IC_ENV_KIND] == 1) {                                                       // go2_env_base.py:165-196
    if (x.reset_buf) { x.base_vel_world[0] = x.base_vel_world[1] = x.base_vel_world[2] = 0.0f; }   // get_vel() after zero_all_dofs_velocity
    x.rew = 0.0f;
    for (int k = 0; k < c.i[GO2SIM_IC_N_REWARDS]; ++k) {
      real r = reward_term(h, b, c.i[GO2SIM_IC_REWARD_ID0 + k], nullptr, nullptr) * c.f[GO2SIM_FC_REWARD_SCALE0 + k];
      x.rew_terms[k] = r;
      x.rew = x.rew + r;
      x.episode_sums[k] = x.episode_sums[k] + r;
    }
    for (int k = 0; k < 3; ++k) o[k] = x.base_ang_vel[k] * c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL];
    for (int k = 0; k < 3; ++k) o[3 + k] = x.projected_gravity[k];
    for (int k = 0; k < 3; ++k) o[6 + k] = x.commands[k] * cs[k];
    for (int i = 0; i < NM; ++i) o[9 + i] = (x.dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]) * c.f[GO2SIM_FC_OBS_SCALE_DOF_POS];
    for (int i = 0; i < NM; ++i) o[21 + i] = x.dof_vel[i] * c.f[GO2SIM_FC_OBS_SCALE_DOF_VEL];
    for (int i = 0; i < na; ++i) o[33 + i] = x.actions[i];
    for (int i = 0; i < nobs; ++i) x.priv[i] = o[i];
    for (int i = 0; i < na; ++i) x.last_actions[i] = x.actions[i];
    for (int i = 0; i < NM; ++i) x.last_dof_vel[i] = x.dof_vel[i];
    if (obs) for (int i = 0; i < nobs; ++i) obs[i] = o[i];
    if (priv) for (int i = 0; i < npriv; ++i) priv[i] = x.priv[i];
    return;
  }
  for (int k = 0; k < 3; ++k) o[k] = x.base_ang_vel[k] * c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL];
  for (int k = 0; k < 3; ++k) o[3 + k] = x.projected_gravity[k] + x.gravity_offset[k];
  for (int k = 0; k < 3; ++k) o[6 + k] = x.commands[k] * cs[k];
  for (int i = 0; i < NM; ++i) o[9 + i] = (x.dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]) * c.f[GO2SIM_FC_OBS_SCALE_DOF_POS];
  for (int i = 0; i < NM; ++i) o[21 + i] = x.dof_vel[i] * c.f[GO2SIM_FC_OBS_SCALE_DOF_VEL];
  for (int i = 0; i < na; ++i) o[33 + i] = x.applied_actions[i];
  if (c.i[GO2SIM_IC_HAS_OBS_NOISE] && c.d[GO2SIM_FC_OBS_NOISE_LEVEL_MAX] > 0.0) {          // _add_obs_noise :908-910, _rebuild_obs_noise_vec :611-626
    double lvl = g.obs_noise_level_cur;                                                     // python-float products, rounded on assignment into the float32 vector
    for (int blk = 0; blk * 4 < nobs; ++blk) {
      dm_u4 r = rng4(h, RNG_OBS_NOISE, b, g.step_count, blk);
      real n[4];
      dm_normal2(r.v[0], r.v[1], &n[0], &n[1]); dm_normal2(r.v[2], r.v[3], &n[2], &n[3]);
      for (int k = 0; k < 4; ++k) {
        int i = 4 * blk + k;
        if (i >= nobs) break;
        real nv = 0.0f;
        if (i < 3) nv = (real)(c.d[GO2SIM_FC_OBS_NOISE_ANG_VEL] * c.d[GO2SIM_FC_OBS_SCALE_ANG_VEL] * lvl);
        else if (i < 6) nv = (real)(c.d[GO2SIM_FC_OBS_NOISE_GRAVITY] * lvl);
        else if (i < 9) nv = 0.0f;
        else if (i < 21) nv = (real)(c.d[GO2SIM_FC_OBS_NOISE_DOF_POS] * c.d[GO2SIM_FC_OBS_SCALE_DOF_POS] * lvl);
        else if (i < 33) nv = (real)(c.d[GO2SIM_FC_OBS_NOISE_DOF_VEL] * c.d[GO2SIM_FC_OBS_SCALE_DOF_VEL] * lvl);
        o[i] = o[i] + n[k] * nv;
      }
    }
  }
  real* p = x.priv;                                                                         // _build_privileged_obs :1115-1141
  for (int i = 0; i < nobs; ++i) p[i] = o[i];
  int idx = nobs;
  for (int k = 0; k < 3; ++k) p[idx + k] = x.base_lin_vel[k] * c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL];
  idx += 3;
  p[idx] = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] ? h->envs[b].geom_friction[NG - 1] : g.friction; idx += 1;
  for (int i = 0; i < NM; ++i) p[idx + i] = x.kp_factors[i];
  idx += 12;
  for (int i = 0; i < NM; ++i) p[idx + i] = x.kd_factors[i];
  idx += 12;
  for (int i = 0; i < NM; ++i) p[idx + i] = x.motor_strength[i];
  idx += 12;
  p[idx] = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] ? h->envs[b].mass_shift[c.i[GO2SIM_IC_BASE_LINK]] : g.mass_shift; idx += 1;
  for (int k = 0; k < 3; ++k) p[idx + k] = g.com_shift[k];
  idx += 3;
  for (int k = 0; k < 4; ++k) p[idx + k] = g.leg_mass_shift[k];
  idx += 4;
  for (int k = 0; k < 3; ++k) p[idx + k] = x.gravity_offset[k];
  idx += 3;
  for (int k = 0; k < 3; ++k) p[idx + k] = x.current_push_force[k];
  idx += 3;
  if (c.i[GO2SIM_IC_MAX_DELAY] > 0) p[idx] = (real)x.delay_steps / (real)c.i[GO2SIM_IC_MAX_DELAY];
  idx += 1;
  if (c.i[GO2SIM_IC_USE_TERRAIN]) {                                                         // go2_env_stair.py:1466-1480
    if (idx < npriv) { p[idx] = (real)x.terrain_row / (real)std::max(1, c.i[GO2SIM_IC_N_TERRAIN_ROWS] - 1); idx += 1; }
    const int scan_n = c.i[GO2SIM_IC_SCAN_N];
    if (scan_n > 0 && idx + scan_n <= npriv) {                                                // _compute_height_scan :772-803
      real qw = x.base_quat[0], qx = x.base_quat[1], qy = x.base_quat[2], qz = x.base_quat[3];
      real yaw = dm_atan2(2.0f * (qw * qz + qx * qy), 1.0f - 2.0f * (qy * qy + qz * qz));
      real sy, cy;
      dm_sincos(yaw, &sy, &cy);
      for (int k = 0; k < scan_n; ++k) {
        real lx = c.f[GO2SIM_FC_SCAN_X0 + k], ly = c.f[GO2SIM_FC_SCAN_Y0 + k];
        real wx = x.base_pos[0] + cy * lx - sy * ly;
        real wy = x.base_pos[1] + sy * lx + cy * ly;
        p[idx + k] = terrain_height(h, wx, wy) - x.base_pos[2];
      }
      idx += scan_n;
    }
  }
  for (int i = idx; i < npriv; ++i) p[i] = 0.0f;
  for (int i = 0; i < na; ++i) x.last_actions[i] = x.actions[i];                            // :1103-1104
  for (int i = 0; i < NM; ++i) x.last_dof_vel[i] = x.dof_vel[i];
  if (obs) for (int i = 0; i < nobs; ++i) obs[i] = o[i];
  if (priv) for (int i = 0; i < npriv; ++i) priv[i] = p[i];
}

void init_env_state(go2sim* h, int b) {
  const Model& m = h->m; Env& e = h->envs[b];
  memset(&e, 0, sizeof(Env));
  for (int i = 0; i < NQ; ++i) e.qpos[i] = m.qpos0[i];
  for (int i = 0; i < NG; ++i) { e.friction_ratio[i] = 1.0f; e.geom_friction[i] = m.geoms[i].friction; }
  for (int i = 0; i < NL; ++i) { e.l_pos[i] = m.links[i].pos; e.l_quat[i] = m.links[i].quat; }
  e.first_time = 1;
  update_cartesian_space(m, e, true);
  forward_velocity(m, e);
}

}  // namespace

// =============================================================================================
// C API (go2sim_cpu_* twin of include/go2sim.h)
// =============================================================================================
extern "C" {

int go2sim_cpu_create(const void* blob, size_t nbytes, int n_envs, int device, uint64_t seed, go2sim** out) {
  (void)device;
  if (!blob || !out || n_envs <= 0) return GO2SIM_E_BADARG;
  go2sim* h = new go2sim();
  if (!parse_model(blob, nbytes, h->m)) { delete h; return GO2SIM_E_BADMODEL; }
  h->B = n_envs; h->seed = seed;
  h->envs.resize(n_envs); h->eb.resize(n_envs);
  memset(&h->g, 0, sizeof(h->g)); memset(&h->cfg, 0, sizeof(h->cfg));
  h->g.friction = 1.0f;
  for (int b = 0; b < n_envs; ++b) { init_env_state(h, b); memset(&h->eb[b], 0, sizeof(EnvBuf)); }
  *out = h;
  return GO2SIM_E_OK;
}
int go2sim_cpu_destroy(go2sim* h) { delete h; return GO2SIM_E_OK; }
int go2sim_cpu_n_envs(const go2sim* h) { return h ? h->B : GO2SIM_E_BADARG; }

int go2sim_cpu_scene_reset(go2sim* h, void*) {
  if (!h) return GO2SIM_E_BADARG;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) {
    real mass_shift[NL]; V3 com_shift[NL]; real fr[NG], gf[NG];
    Env& e = h->envs[b];
    memcpy(mass_shift, e.mass_shift, sizeof(mass_shift)); memcpy(com_shift, e.com_shift, sizeof(com_shift));
    memcpy(fr, e.friction_ratio, sizeof(fr)); memcpy(gf, e.geom_friction, sizeof(gf));
    init_env_state(h, b);
    memcpy(e.mass_shift, mass_shift, sizeof(mass_shift)); memcpy(e.com_shift, com_shift, sizeof(com_shift));
    memcpy(e.friction_ratio, fr, sizeof(fr)); memcpy(e.geom_friction, gf, sizeof(gf));
    update_cartesian_space(h->m, e, true); forward_velocity(h->m, e);
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_substep(go2sim* h, void*) {
  if (!h) return GO2SIM_E_BADARG;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) substep(h->m, h->envs[b]);
  return GO2SIM_E_OK;
}
int go2sim_cpu_scene_step(go2sim* h, int substeps, void*) {
  if (!h) return GO2SIM_E_BADARG;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) {
    Env& e = h->envs[b];
    for (int s = 0; s < substeps; ++s) substep(h->m, e);
    for (int l = 0; l < NL; ++l) { e.ext_ang[l] = v3(0, 0, 0); e.ext_vel[l] = v3(0, 0, 0); }   // kernel_clear_external_force
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_forward_kinematics(go2sim* h, void*) {
  if (!h) return GO2SIM_E_BADARG;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) { update_cartesian_space(h->m, h->envs[b], true); forward_velocity(h->m, h->envs[b]); }
  return GO2SIM_E_OK;
}

int go2sim_cpu_field_size(int field, int* k, int* is_int) {
  int kk = -1, ii = 0;
  switch (field) {
    case GO2SIM_F_QPOS: kk = NQ; break;
    case GO2SIM_F_VEL: case GO2SIM_F_ACC: case GO2SIM_F_QACC_WS: case GO2SIM_F_CTRL_FORCE: case GO2SIM_F_FORCE: case GO2SIM_F_ACC_SMOOTH:
    case GO2SIM_F_QFRC_CONSTRAINT: case GO2SIM_F_CTRL_POS: case GO2SIM_F_CTRL_VEL: case GO2SIM_F_DOF_POS: kk = ND; break;
    case GO2SIM_F_EXT_FORCE: kk = NL * 6; break;
    case GO2SIM_F_MASS_SHIFT: kk = NL; break;
    case GO2SIM_F_COM_SHIFT: case GO2SIM_F_LINK_POS: case GO2SIM_F_LINK_CDVEL: case GO2SIM_F_LINK_CDANG: case GO2SIM_F_CONTACT_FORCE: kk = NL * 3; break;
    case GO2SIM_F_FRICTION_RATIO: case GO2SIM_F_GEOM_FRICTION: kk = NG; break;
    case GO2SIM_F_LINK_QUAT: kk = NL * 4; break;
    case GO2SIM_F_ROOT_COM: kk = 3; break;
    case GO2SIM_F_MASS_MAT: kk = ND * ND; break;
    case GO2SIM_F_CONTACT_POS: case GO2SIM_F_CONTACT_NORMAL: kk = MAXC * 3; break;
    case GO2SIM_F_CONTACT_PEN: kk = MAXC; break;
    case GO2SIM_F_NORMAL_CACHE: kk = NPAIR * 3; break;
    case GO2SIM_F_SORT_VALUE: kk = 2 * NG; break;
    case GO2SIM_F_EFC_FORCE: kk = MAXR; break;
    case GO2SIM_I_N_CONTACTS: case GO2SIM_I_N_CONSTRAINTS: case GO2SIM_I_ERRNO: case GO2SIM_I_IS_WARMSTART: case GO2SIM_I_FIRST_TIME:
    case GO2SIM_I_N_BROAD: case GO2SIM_I_SOLVER_ITERS: kk = 1; ii = 1; break;
    case GO2SIM_I_CONTACT_GEOMS: kk = 2 * MAXC; ii = 1; break;
    case GO2SIM_I_SORT_IG: kk = 2 * NG; ii = 1; break;
    case GO2SIM_I_CTRL_MODE: kk = ND; ii = 1; break;
    default: return GO2SIM_E_BADARG;
  }
  if (k) *k = kk;
  if (is_int) *is_int = ii;
  return GO2SIM_E_OK;
}

}  // extern "C"

namespace {
// element accessor used by get/set_field: pointer to the j-th scalar of `field` in env e (float or int)
void* field_elem(Env& e, int field, int j) {
  switch (field) {
    case GO2SIM_F_QPOS: return &e.qpos[j];
    case GO2SIM_F_VEL: return &e.vel[j];
    case GO2SIM_F_ACC: return &e.acc[j];
    case GO2SIM_F_QACC_WS: return &e.qacc_ws[j];
    case GO2SIM_F_CTRL_FORCE: return &e.ctrl_force[j];
    case GO2SIM_F_EXT_FORCE: { int l = j / 6, k = j % 6; return (k < 3) ? ((real*)&e.ext_ang[l]) + k : ((real*)&e.ext_vel[l]) + (k - 3); }
    case GO2SIM_F_MASS_SHIFT: return &e.mass_shift[j];
    case GO2SIM_F_COM_SHIFT: return ((real*)&e.com_shift[j / 3]) + j % 3;
    case GO2SIM_F_FRICTION_RATIO: return &e.friction_ratio[j];
    case GO2SIM_F_GEOM_FRICTION: return &e.geom_friction[j];
    case GO2SIM_F_LINK_POS: return ((real*)&e.l_pos[j / 3]) + j % 3;
    case GO2SIM_F_LINK_QUAT: return ((real*)&e.l_quat[j / 4]) + j % 4;
    case GO2SIM_F_LINK_CDVEL: return ((real*)&e.cd_vel[j / 3]) + j % 3;
    case GO2SIM_F_LINK_CDANG: return ((real*)&e.cd_ang[j / 3]) + j % 3;
    case GO2SIM_F_ROOT_COM: return ((real*)&e.root_com[1]) + j;
    case GO2SIM_F_CONTACT_FORCE: return ((real*)&e.contact_force[j / 3]) + j % 3;
    case GO2SIM_F_MASS_MAT: return &e.mass_mat[j / ND][j % ND];
    case GO2SIM_F_FORCE: return &e.qf_smooth[j];
    case GO2SIM_F_ACC_SMOOTH: return &e.acc_smooth[j];
    case GO2SIM_F_QFRC_CONSTRAINT: return &e.qfrc_constraint[j];
    case GO2SIM_F_CTRL_POS: return &e.ctrl_pos[j];
    case GO2SIM_F_CTRL_VEL: return &e.ctrl_vel[j];
    case GO2SIM_F_DOF_POS: return &e.dof_pos[j];
    case GO2SIM_F_CONTACT_POS: return ((real*)&e.contacts[j / 3].pos) + j % 3;
    case GO2SIM_F_CONTACT_NORMAL: return ((real*)&e.contacts[j / 3].normal) + j % 3;
    case GO2SIM_F_CONTACT_PEN: return &e.contacts[j].penetration;
    case GO2SIM_F_NORMAL_CACHE: return ((real*)&e.normal_cache[j / 3]) + j % 3;
    case GO2SIM_F_SORT_VALUE: return &e.sort_value[j];
    case GO2SIM_F_EFC_FORCE: return &e.efc_force[j];
    case GO2SIM_I_N_CONTACTS: return &e.n_contacts;
    case GO2SIM_I_N_CONSTRAINTS: return &e.n_con;
    case GO2SIM_I_ERRNO: return &e.err;
    case GO2SIM_I_IS_WARMSTART: return &e.is_warmstart;
    case GO2SIM_I_FIRST_TIME: return &e.first_time;
    case GO2SIM_I_N_BROAD: return &e.n_broad;
    case GO2SIM_I_SOLVER_ITERS: return &e.solver_iters;
    case GO2SIM_I_CONTACT_GEOMS: return (j < MAXC) ? &e.contacts[j].geom_a : &e.contacts[j - MAXC].geom_b;
    case GO2SIM_I_CTRL_MODE: return &e.ctrl_mode[j];
  }
  return nullptr;
}
}  // namespace

extern "C" {

int go2sim_cpu_get_field(go2sim* h, int field, void* dst, void*) {
  int k, is_int;
  if (!h || !dst || go2sim_cpu_field_size(field, &k, &is_int)) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b)
    for (int j = 0; j < k; ++j) {
      if (field == GO2SIM_I_SORT_IG) { ((int*)dst)[(size_t)j * h->B + b] = h->envs[b].sort_ig[j] | (h->envs[b].sort_ismax[j] << 8); continue; }
      void* p = field_elem(h->envs[b], field, j);
      if (is_int) ((int*)dst)[(size_t)j * h->B + b] = *(int*)p; else ((float*)dst)[(size_t)j * h->B + b] = *(float*)p;
    }
  return GO2SIM_E_OK;
}
int go2sim_cpu_set_field(go2sim* h, int field, const void* src, void*) {
  int k, is_int;
  if (!h || !src || go2sim_cpu_field_size(field, &k, &is_int)) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b)
    for (int j = 0; j < k; ++j) {
      if (field == GO2SIM_I_SORT_IG) { int v = ((const int*)src)[(size_t)j * h->B + b]; h->envs[b].sort_ig[j] = v & 0xff; h->envs[b].sort_ismax[j] = (v >> 8) & 1; continue; }
      void* p = field_elem(h->envs[b], field, j);
      if (is_int) *(int*)p = ((const int*)src)[(size_t)j * h->B + b]; else *(float*)p = ((const float*)src)[(size_t)j * h->B + b];
    }
  return GO2SIM_E_OK;
}
int go2sim_cpu_field_ptr(go2sim*, int, void** ptr) { if (ptr) *ptr = nullptr; return GO2SIM_E_BADARG; }  // AoS oracle: no SoA view

int go2sim_cpu_reset_caches(go2sim* h, const int* envs_idx, int n_sel, void*) {
  if (!h) return GO2SIM_E_BADARG;
  int n = envs_idx ? n_sel : h->B;
  for (int i = 0; i < n; ++i) {
    int b = envs_idx ? envs_idx[i] : i;
    if (b < 0 || b >= h->B) return GO2SIM_E_BADARG;
    Env& e = h->envs[b];
    e.err = 0; e.is_warmstart = 0;
    for (int d = 0; d < ND; ++d) e.qacc_ws[d] = 0.0f;
    for (int p = 0; p < NPAIR; ++p) e.normal_cache[p] = v3(0, 0, 0);
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_set_friction(go2sim* h, float mu, void*) {
  if (!h) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b) for (int i = 0; i < NG; ++i) h->envs[b].geom_friction[i] = mu;
  h->g.friction = mu;
  return GO2SIM_E_OK;
}
int go2sim_cpu_set_dof_gains(go2sim* h, int d, float kp, float kv, float flo, float fhi) {
  if (!h || d < 0 || d >= ND) return GO2SIM_E_BADARG;
  h->m.dofs[d].kp = kp; h->m.dofs[d].kv = kv; h->m.dofs[d].force_range[0] = flo; h->m.dofs[d].force_range[1] = fhi;
  return GO2SIM_E_OK;
}
int go2sim_cpu_check_errno(go2sim* h, int* out, void*) {
  if (!h || !out) return GO2SIM_E_BADARG;
  int v = 0;
  for (int b = 0; b < h->B; ++b) v |= h->envs[b].err;
  *out = v;
  return GO2SIM_E_OK;
}

int go2sim_cpu_env_configure(go2sim* h, const double* f, int nf, const int* i, int ni) {
  if (!h || !f || !i || nf != GO2SIM_FC_COUNT || ni != GO2SIM_IC_COUNT) return GO2SIM_E_BADARG;
  for (int k = 0; k < GO2SIM_FC_N_HOST; ++k) h->cfg.d[k] = f[k];
  for (int k = 0; k < nf; ++k) h->cfg.f[k] = (float)f[k];
  memcpy(h->cfg.i, i, sizeof(int) * ni); h->cfg.set = true;
  const Cfg& c = h->cfg;
  if (c.i[GO2SIM_IC_NUM_ACTIONS] > NA || c.i[GO2SIM_IC_NUM_OBS] > NOBS_MAX || c.i[GO2SIM_IC_NUM_PRIV_OBS] > NPRIV_MAX || c.i[GO2SIM_IC_N_REWARDS] > NREW ||
      c.i[GO2SIM_IC_MAX_DELAY] >= GO2SIM_ACTION_RING_MAX || c.i[GO2SIM_IC_MAX_DELAY] < 0)
    return GO2SIM_E_BADARG;
  memset(&h->g, 0, sizeof(h->g));
  h->g.level = c.d[GO2SIM_FC_CURR_LEVEL_INIT];
  h->g.friction = 1.0f;
  apply_curriculum_level(h);
  for (int b = 0; b < h->B; ++b) {
    EnvBuf& x = h->eb[b];
    memset(&x, 0, sizeof(EnvBuf));
    for (int k = 0; k < NM; ++k) { x.kp_factors[k] = 1.0f; x.kd_factors[k] = 1.0f; x.motor_strength[k] = 1.0f; }
    x.delay_steps = 1; x.reset_buf = 1;
    if (c.i[GO2SIM_IC_MANUAL_PD]) for (int k = 0; k < NM; ++k) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + k]; h->envs[b].ctrl_mode[d] = CTRL_FORCE; }
  }
  if (c.i[GO2SIM_IC_MANUAL_PD]) for (int k = 0; k < NM; ++k) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + k]; h->m.dofs[d].kp = 0.0f; h->m.dofs[d].kv = 0.0f; }
  else for (int k = 0; k < NM; ++k) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + k]; h->m.dofs[d].kp = c.f[GO2SIM_FC_KP]; h->m.dofs[d].kv = c.f[GO2SIM_FC_KD]; }
  return GO2SIM_E_OK;
}

// _assign_terrain_rows, go2_env_stair.py:809-854: 40 % of the reset envs on the frontier row, 30 % just below it, 30 % on easy rows,
// shuffled.  The shuffle is the rank of a per-env Philox key; the j-th reset env (env order) receives rows[perm[j]].
static void assign_terrain_rows(go2sim* h) {
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  const int n_rows = c.i[GO2SIM_IC_N_TERRAIN_ROWS];
  uint32_t rc = g.reset_calls - 1;
  std::vector<int> idx;
  for (int b = 0; b < h->B; ++b) if (h->eb[b].reset_buf) idx.push_back(b);
  const int n = (int)idx.size();
  double mean_row = 0.0;
  if (n_rows > 1 && !g.lock_terrain_rows) {                        // `if not self._lock_terrain_rows`, go2_env_stair.py:1513
    double level = c.i[GO2SIM_IC_CURR_ENABLED] ? g.level : 1.0;
    int max_row = (int)(level * (double)(n_rows - 1));
    max_row = std::max(0, std::min(max_row, n_rows - 1));
    int n_frontier = (int)((double)n * 0.40), n_near = (int)((double)n * 0.30);
    for (int j = 0; j < n; ++j) h->eb[idx[j]].terrain_key = rng4(h, RNG_TERRAIN_PERM, idx[j], rc, 0).v[0];
    for (int j = 0; j < n; ++j) {
      unsigned kj = h->eb[idx[j]].terrain_key;
      int p = 0;
      for (int j2 = 0; j2 < n; ++j2) { unsigned k2 = h->eb[idx[j2]].terrain_key; p += (k2 < kj) || (k2 == kj && j2 < j); }
      dm_u4 r = rng4(h, RNG_TERRAIN_ROW, (uint32_t)p, rc, 0);
      int row;
      if (p < n_frontier) row = max_row;
      else if (p < n_frontier + n_near) row = (max_row >= 2) ? rand_int(std::max(0, max_row - 2), std::max(0, max_row - 1), r.v[0]) : max_row;
      else row = rand_int(0, (max_row >= 3) ? max_row - 3 : 0, r.v[1]);
      h->eb[idx[j]].terrain_row = row;
    }
  }
  int row_sum = 0;
  for (int j = 0; j < n; ++j) row_sum += h->eb[idx[j]].terrain_row;
  (void)mean_row;
  g.terrain_row_sum = row_sum;
  g.terrain_mean_row = n > 0 ? (float)((double)row_sum / (double)n) : 0.0f;
}

static void reset_call(go2sim* h, bool count_push) {
  h->acc_timeouts = 0.0; h->acc_tracking = 0.0;
  for (int k = 0; k < NREW; ++k) h->acc_ep[k] = 0.0;
  h->g.n_reset_now = 0;
  for (int b = 0; b < h->B; ++b) if (h->eb[b].reset_buf) env_reset_stats(h, b);
  bool any = h->g.n_reset_now > 0;
  env_globals_update(h, count_push);
  if (any && h->cfg.i[GO2SIM_IC_USE_TERRAIN]) assign_terrain_rows(h);
  if (any) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < h->B; ++b) {
      if (h->eb[b].reset_buf) env_reset_one(h, b);
      env_apply_globals_and_fk(h, b);
    }
  }
}

int go2sim_cpu_env_step(go2sim* h, const float* actions, float* obs, float* priv, float* rew, uint8_t* reset, float* timeout, void*) {
  if (!h || !h->cfg.set || !actions) return GO2SIM_E_BADARG;
  const Cfg& c = h->cfg;
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS], nobs = c.i[GO2SIM_IC_NUM_OBS], npriv = c.i[GO2SIM_IC_NUM_PRIV_OBS], substeps = c.i[GO2SIM_IC_SUBSTEPS];
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) {
    Env& e = h->envs[b];
    env_pre(h, b, actions + (size_t)b * na);
    for (int s = 0; s < substeps; ++s) substep(h->m, e);
    for (int l = 0; l < NL; ++l) { e.ext_ang[l] = v3(0, 0, 0); e.ext_vel[l] = v3(0, 0, 0); }
    env_post_a(h, b);
  }
  h->g.action_write_idx = (h->g.action_write_idx + 1) % (c.i[GO2SIM_IC_MAX_DELAY] + 1);
  reset_call(h, true);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) {
    env_post_b(h, b, obs ? obs + (size_t)b * nobs : nullptr, priv ? priv + (size_t)b * npriv : nullptr);
    if (rew) rew[b] = h->eb[b].rew;
    if (reset) reset[b] = (uint8_t)h->eb[b].reset_buf;
    if (timeout) timeout[b] = h->eb[b].time_out;
  }
  h->g.step_count += 1;
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_reset(go2sim* h, void*) {
  if (!h || !h->cfg.set) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b) h->eb[b].reset_buf = 1;
  reset_call(h, false);
  return GO2SIM_E_OK;
}
// Go2Env.reset_idx(envs_idx), go2_env_walk.py:1156-1240
int go2sim_cpu_env_reset_idx(go2sim* h, const int* envs_idx, int n_sel, void*) {
  if (!h || !h->cfg.set || n_sel < 0 || (n_sel > 0 && !envs_idx)) return GO2SIM_E_BADARG;
  if (n_sel == 0) return GO2SIM_E_OK;
  for (int b = 0; b < h->B; ++b) h->eb[b].reset_buf = 0;
  for (int t = 0; t < n_sel; ++t) { int b = envs_idx[t]; if (b >= 0 && b < h->B) h->eb[b].reset_buf = 1; }
  reset_call(h, false);
  return GO2SIM_E_OK;
}
// respawn_at_start (go2_eval_stairs.py:314-361) / respawn_on_tile (go2_eval_walk.py:399-480)
int go2sim_cpu_env_respawn(go2sim* h, const int* envs_idx, int n_sel, const float* pos, const float* quat, int clear_buffers, void*) {
  if (!h || !h->cfg.set || n_sel < 0 || (n_sel > 0 && (!envs_idx || !pos))) return GO2SIM_E_BADARG;
  if (n_sel == 0) return GO2SIM_E_OK;
  const Model& m = h->m; const Cfg& c = h->cfg;
  for (int t = 0; t < n_sel; ++t) {
    int b = envs_idx[t];
    if (b < 0 || b >= h->B) continue;
    Env& e = h->envs[b]; EnvBuf& x = h->eb[b];
    for (int i = 0; i < NM; ++i) {                                 // robot.set_dofs_position(default, zero_velocity=True)
      real dp = c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i];
      x.dof_pos[i] = dp; x.dof_vel[i] = 0.0f;
      int q = c.i[GO2SIM_IC_MOTOR_DOF0 + i] + 1;
      e.qpos[q] = m.qpos0[q] + dp;
    }
    for (int d = 0; d < ND; ++d) e.vel[d] = 0.0f;
    e.err = 0; e.is_warmstart = 0;                                 // rigid_solver.py:2403-2410
    for (int d = 0; d < ND; ++d) e.qacc_ws[d] = 0.0f;
    for (int p = 0; p < NPAIR; ++p) e.normal_cache[p] = v3(0, 0, 0);
    for (int k = 0; k < 3; ++k) { x.base_pos[k] = pos[3 * t + k]; e.qpos[k] = pos[3 * t + k]; }
    for (int k = 0; k < 4; ++k) { real q = quat ? quat[4 * t + k] : c.f[GO2SIM_FC_BASE_INIT_QUAT0 + k]; x.base_quat[k] = q; e.qpos[3 + k] = q; }
    if (clear_buffers) {
      for (int k = 0; k < 3; ++k) { x.base_lin_vel[k] = 0.0f; x.base_ang_vel[k] = 0.0f; }
      for (int i = 0; i < NA; ++i) { x.last_actions[i] = 0.0f; x.applied_actions[i] = 0.0f; for (int k = 0; k < GO2SIM_ACTION_RING_MAX; ++k) x.action_history[k][i] = 0.0f; }
      for (int i = 0; i < NM; ++i) x.last_dof_vel[i] = 0.0f;
      x.last_base_pos_x = pos[3 * t];
    }
  }
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) { update_cartesian_space(m, h->envs[b], true); forward_velocity(m, h->envs[b]); }   // full-batch FK, rigid_solver.py:1928-1943
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_lock_terrain_rows(go2sim* h, int lock, void*) {
  if (!h || !h->cfg.set) return GO2SIM_E_BADARG;
  h->g.lock_terrain_rows = lock != 0;
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_set_terrain_rows(go2sim* h, const int* rows, void*) {
  if (!h || !h->cfg.set || !rows) return GO2SIM_E_BADARG;
  int n_rows = std::max(1, h->cfg.i[GO2SIM_IC_N_TERRAIN_ROWS]);
  for (int b = 0; b < h->B; ++b) h->eb[b].terrain_row = std::max(0, std::min(rows[b], n_rows - 1));
  return GO2SIM_E_OK;
}
// the host-side twin has nothing asynchronous: the poll is ready at once
int go2sim_cpu_errno_poll_begin(go2sim* h, void*) { return h ? GO2SIM_E_OK : GO2SIM_E_BADARG; }
int go2sim_cpu_errno_poll_result(go2sim* h, int* errno_host, int* ready) {
  if (!h || !errno_host || !ready) return GO2SIM_E_BADARG;
  *ready = 1;
  return go2sim_cpu_check_errno(h, errno_host, nullptr);
}
int go2sim_cpu_errno_poll_wait(go2sim* h, int* errno_host) { return (h && errno_host) ? go2sim_cpu_check_errno(h, errno_host, nullptr) : GO2SIM_E_BADARG; }
int go2sim_cpu_graph_status(go2sim* h, int* using_graph, int* n_fallbacks) {
  if (!h) return GO2SIM_E_BADARG;
  if (using_graph) *using_graph = 0;
  if (n_fallbacks) *n_fallbacks = 0;
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_get(go2sim* h, int buf, void* dst, void*) {
  if (!h || !dst) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b) {
    const EnvBuf& x = h->eb[b];
    float* f = (float*)dst; int* ip = (int*)dst;
    switch (buf) {
      case GO2SIM_EB_COMMANDS: memcpy(f + 3 * b, x.commands, 12); break;
      case GO2SIM_EB_EPISODE_LENGTH: ip[b] = x.episode_length; break;
      case GO2SIM_EB_BASE_LIN_VEL: memcpy(f + 3 * b, x.base_lin_vel, 12); break;
      case GO2SIM_EB_BASE_ANG_VEL: memcpy(f + 3 * b, x.base_ang_vel, 12); break;
      case GO2SIM_EB_PROJECTED_GRAVITY: memcpy(f + 3 * b, x.projected_gravity, 12); break;
      case GO2SIM_EB_DOF_POS: memcpy(f + 12 * b, x.dof_pos, 48); break;
      case GO2SIM_EB_DOF_VEL: memcpy(f + 12 * b, x.dof_vel, 48); break;
      case GO2SIM_EB_BASE_POS: memcpy(f + 3 * b, x.base_pos, 12); break;
      case GO2SIM_EB_BASE_QUAT: memcpy(f + 4 * b, x.base_quat, 16); break;
      case GO2SIM_EB_BASE_EULER: memcpy(f + 3 * b, x.base_euler, 12); break;
      case GO2SIM_EB_EPISODE_SUMS: memcpy(f + NREW * b, x.episode_sums, 4 * NREW); break;
      case GO2SIM_EB_FOOT_CONTACT: memcpy(ip + 4 * b, x.foot_contact, 16); break;
      case GO2SIM_EB_FEET_AIR_TIME: memcpy(f + 4 * b, x.feet_air_time, 16); break;
      case GO2SIM_EB_REW_TERMS: memcpy(f + NREW * b, x.rew_terms, 4 * NREW); break;
      case GO2SIM_EB_TORQUE: memcpy(f + 12 * b, x.torque, 48); break;
      case GO2SIM_EB_TERRAIN_ROW: ip[b] = x.terrain_row; break;
      default: return GO2SIM_E_BADARG;
    }
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_set_episode_length(go2sim* h, const int* ep, void*) {
  if (!h || !ep) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b) h->eb[b].episode_length = ep[b];
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_set_commands(go2sim* h, const float* cmd, void*) {
  if (!h || !cmd) return GO2SIM_E_BADARG;
  for (int b = 0; b < h->B; ++b) memcpy(h->eb[b].commands, cmd + 3 * b, 12);
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_globals(go2sim* h, go2sim_env_globals_t* out, void*) {
  if (!h || !out) return GO2SIM_E_BADARG;
  *out = h->g;
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_globals_ptr(go2sim* h, void** ptr_out) {
  if (!h || !ptr_out) return GO2SIM_E_BADARG;
  *ptr_out = &h->g;
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_set_level(go2sim* h, double level, void*) {
  if (!h || !h->cfg.set) return GO2SIM_E_BADARG;
  h->g.level = level;
  apply_curriculum_level(h);
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_sync_counters(go2sim* h, double* out5, void*) {
  if (!h || !h->cfg.set || !out5) return GO2SIM_E_BADARG;
  for (int k = 0; k < 5; ++k) { out5[k] = h->g.shard_counters[k]; h->g.shard_counters[k] = 0.0; }
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_sync_apply(go2sim* h, const double* s5, double* dr_out10, void*) {
  if (!h || !h->cfg.set || !s5) return GO2SIM_E_BADARG;
  const Cfg& c = h->cfg; go2sim_env_globals_t& g = h->g;
  const int n = (int)s5[0];
  const bool first = g.sync_calls == 0;            // the first apply draws even without a counted reset (include/go2sim.h)
  if (n > 0 || first) {
    if (n > 0 && c.i[GO2SIM_IC_CURR_ENABLED] && !c.i[GO2SIM_IC_FREEZE_CURRICULUM]) {
      g.curr_ep_total += n; g.curr_timeout_total += s5[1]; g.curr_tracking_sum += s5[2]; g.curr_tracking_n += (int)s5[3];
      globals_curriculum_check(h);
    }
    globals_draws(h, (int)s5[4], (uint32_t)g.sync_calls);
    g.sync_calls += 1;
  }
  if (dr_out10) {
    dr_out10[0] = g.friction; dr_out10[1] = g.mass_shift;
    for (int k = 0; k < 3; ++k) dr_out10[2 + k] = g.com_shift[k];
    for (int k = 0; k < 4; ++k) dr_out10[5 + k] = g.leg_mass_shift[k];
    dr_out10[9] = g.t_sample;
  }
  return GO2SIM_E_OK;
}
int go2sim_cpu_env_set_global_dr(go2sim* h, const double* dr10, void*) {
  if (!h || !h->cfg.set || !dr10) return GO2SIM_E_BADARG;
  go2sim_env_globals_t& g = h->g;
  g.friction = (float)dr10[0]; g.mass_shift = (float)dr10[1];
  for (int k = 0; k < 3; ++k) g.com_shift[k] = (float)dr10[2 + k];
  for (int k = 0; k < 4; ++k) g.leg_mass_shift[k] = (float)dr10[5 + k];
  g.t_sample = dr10[9];
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) env_apply_globals_and_fk(h, b);
  return GO2SIM_E_OK;
}
// the "device array" forms of the HIP library: host arrays here
int go2sim_cpu_env_sync_counters_dev(go2sim* h, double* out5, void* s) { return go2sim_cpu_env_sync_counters(h, out5, s); }
int go2sim_cpu_env_sync_apply_dev(go2sim* h, const double* s5, double* dr_out10, void* s) { return (dr_out10 == nullptr) ? GO2SIM_E_BADARG : go2sim_cpu_env_sync_apply(h, s5, dr_out10, s); }
int go2sim_cpu_env_set_global_dr_dev(go2sim* h, const double* dr10, void* s) { return go2sim_cpu_env_set_global_dr(h, dr10, s); }
int go2sim_cpu_enable_timing(go2sim*, int) { return GO2SIM_E_BADARG; }
int go2sim_cpu_read_timing(go2sim*, float*, int*, int) { return GO2SIM_E_BADARG; }

// Terrain morph (rigid_entity.py:505-552, utils/terrain.py:228-330, collider.py:374-394): the ground slab becomes a heightfield geom
int go2sim_cpu_set_terrain(go2sim* h, const int16_t* hf, int rows, int cols, float horizontal_scale, float vertical_scale, const float* origin, void*) {
  if (!h || !hf || rows < 2 || cols < 2 || !origin || !(horizontal_scale > 0.0f)) return GO2SIM_E_BADARG;
  Model& m = h->m;
  m.terrain_enabled = 1; m.terrain_rows = rows; m.terrain_cols = cols; m.terrain_hs = horizontal_scale;
  m.terrain_hf.resize((size_t)rows * cols);
  real hmax = -1e30f, hmin = 1e30f;
  for (size_t k = 0; k < (size_t)rows * cols; ++k) { real v = (real)hf[k] * vertical_scale; m.terrain_hf[k] = v; hmax = std::max(hmax, v); hmin = std::min(hmin, v); }
  m.terrain_xyz_maxmin[0] = (real)rows * horizontal_scale; m.terrain_xyz_maxmin[1] = (real)cols * horizontal_scale; m.terrain_xyz_maxmin[2] = hmax;
  m.terrain_xyz_maxmin[3] = 0.0f; m.terrain_xyz_maxmin[4] = 0.0f; m.terrain_xyz_maxmin[5] = hmin - 1.0f;
  Geom& G = m.geoms[0];
  G.type = GEOM_TERRAIN; G.pos = v3(0, 0, 0); G.quat = q4(1, 0, 0, 0); G.center = v3(0, 0, 0);
  real x1 = (real)(rows - 1) * horizontal_scale, y1 = (real)(cols - 1) * horizontal_scale, z0 = hmin - 1.0f, z1 = hmax;
  for (int c = 0; c < 8; ++c) G.aabb[c] = v3((c & 4) ? x1 : 0.0f, (c & 2) ? y1 : 0.0f, (c & 1) ? z1 : z0);
  m.links[0].pos = v3(origin[0], origin[1], origin[2]); m.links[0].quat = q4(1, 0, 0, 0);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < h->B; ++b) {
    Env& e = h->envs[b];
    e.l_pos[0] = m.links[0].pos; e.l_quat[0] = m.links[0].quat;
    e.first_time = 1; e.is_warmstart = 0;
    for (int p = 0; p < NPAIR; ++p) e.normal_cache[p] = v3(0, 0, 0);
    update_cartesian_space(m, e, true);
    forward_velocity(m, e);
  }
  return GO2SIM_E_OK;
}

// extra oracle-only diagnostics
// one narrow-phase query on explicit poses: which = 0 -> MPR (cold start), 1 / 2 -> safe GJK + EPA (the device library distinguishes its two polytope stores).  out = {is_col, penetration, normal[3], pos[3]}
int go2sim_cpu_debug_narrowphase(go2sim* h, int which, int i_ga, int i_gb, const float* pa, const float* qa, const float* pb, const float* qb, float* out8) {
  if (!h || !out8 || i_ga < 0 || i_gb < 0 || i_ga >= NG || i_gb >= NG) return GO2SIM_E_BADARG;
  V3 pos_a = v3(pa[0], pa[1], pa[2]), pos_b = v3(pb[0], pb[1], pb[2]);
  Q4 quat_a = q4(qa[0], qa[1], qa[2], qa[3]), quat_b = q4(qb[0], qb[1], qb[2], qb[3]);
  bool is_col = false; V3 normal = v3(0, 0, 0), pos = v3(0, 0, 0); real pen = 0.0f;
  if (which == 0) {
    mpr_contact(h->m, h->envs[0], i_ga, i_gb, v3(0, 0, 0), pos_a, quat_a, pos_b, quat_b, is_col, normal, pen, pos);
  } else {
    static thread_local GjkScratch scratch;
    GjkResult r = gjk_contact_pair(h->m, scratch, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b);
    is_col = r.is_col != 0; pen = r.penetration; normal = v3(r.normal.x, r.normal.y, r.normal.z); pos = v3(r.pos.x, r.pos.y, r.pos.z);
  }
  out8[0] = is_col ? 1.0f : 0.0f; out8[1] = pen; out8[2] = normal.x; out8[3] = normal.y; out8[4] = normal.z; out8[5] = pos.x; out8[6] = pos.y; out8[7] = pos.z;
  return GO2SIM_E_OK;
}
int go2sim_cpu_gjk_fallback_count(go2sim* h, long long* out) {
  if (!h || !out) return GO2SIM_E_BADARG;
  long long s = 0;
  for (int b = 0; b < h->B; ++b) s += h->envs[b].gjk_fallback_count;
  *out = s;
  return GO2SIM_E_OK;
}

// development/test aid: copy a solver-internal array of every env into dst laid out [k][n_envs]
int go2sim_cpu_debug_get(go2sim* h, const char* name, float* dst, int* k_out) {
  if (!h || !name) return GO2SIM_E_BADARG;
  int k = 0;
  for (int b = 0; b < h->B; ++b) {
    Env& e = h->envs[b];
    const float* src = nullptr;
    if (!strcmp(name, "jac")) { src = &e.jac[0][0]; k = MAXR * ND; }
    else if (!strcmp(name, "diag")) { src = e.diag; k = MAXR; }
    else if (!strcmp(name, "aref")) { src = e.aref; k = MAXR; }
    else if (!strcmp(name, "efc_D")) { src = e.efc_D; k = MAXR; }
    else if (!strcmp(name, "Jaref")) { src = e.Jaref; k = MAXR; }
    else if (!strcmp(name, "jv")) { src = e.jv; k = MAXR; }
    else if (!strcmp(name, "H")) { src = &e.H[0][0]; k = ND * ND; }
    else if (!strcmp(name, "grad")) { src = e.grad; k = ND; }
    else if (!strcmp(name, "Mgrad")) { src = e.Mgrad; k = ND; }
    else if (!strcmp(name, "search")) { src = e.search; k = ND; }
    else if (!strcmp(name, "qacc")) { src = e.qacc; k = ND; }
    else if (!strcmp(name, "Ma")) { src = e.Ma; k = ND; }
    else if (!strcmp(name, "mv")) { src = e.mv; k = ND; }
    else if (!strcmp(name, "mass_L")) { src = &e.mass_L[0][0]; k = ND * ND; }
    else if (!strcmp(name, "cdof_ang")) { src = (const float*)e.cdof_ang; k = ND * 3; }
    else if (!strcmp(name, "cdof_vel")) { src = (const float*)e.cdof_vel; k = ND * 3; }
    else return GO2SIM_E_BADARG;
    if (dst) for (int j = 0; j < k; ++j) dst[(size_t)j * h->B + b] = src[j];
  }
  if (k_out) *k_out = k;
  return GO2SIM_E_OK;
}

// test aids for the arrow form (tests/test_arrow_form.py): the numbering rule on a mask, and factor + solve on a matrix handed in (FAST ORDER build, ND = 18)
int go2sim_cpu_debug_arrow_mode(const float* mask, int nd) { return mask ? dm_arrow_mode(mask, nd) : -1; }
int go2sim_cpu_debug_arrow_solve(int mode, float eps, const float* A, const float* g, float* x) {
#ifdef GO2SIM_FAST_ORDER
  if (ND != 18 || (mode != 1 && mode != 2) || !A || !g || !x) return GO2SIM_E_BADARG;
  static real Am[ND][ND];
  ArrowFactor f;
  real gg[ND], xx[ND];
  for (int i = 0; i < ND; ++i) { gg[i] = g[i]; for (int j = 0; j < ND; ++j) Am[i][j] = A[i * ND + j]; }
  arrow_factor(mode, eps, Am, f);
  arrow_solve(f, gg, xx);
  for (int i = 0; i < ND; ++i) x[i] = xx[i];
  return GO2SIM_E_OK;
#else
  (void)mode; (void)eps; (void)A; (void)g; (void)x;
  return GO2SIM_E_BADARG;
#endif
}

}  // extern "C"
