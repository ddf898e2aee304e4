// go2sim.hip -- MI355X (gfx950) implementation of the go2sim C ABI (include/go2sim.h).
//
// The vectorised Go2 locomotion environment of saifahmadgit/go2-sim2real-locomotion-rl (a Genesis
// v0.4.0 fork) as hand-written HIP: batched rigid-body substep (articulated forward dynamics,
// sweep-and-prune + MPR contact generation, Newton constraint solve with exact line search,
// semi-implicit integration) and the Go2Env walk step (action latency, PLS PD torques, pushes,
// observations, 19 reward terms, command resampling, termination, reset with domain randomisation and
// metric-gated curriculum) with NO host synchronisation inside a step.
//
// Data layout: two pools.  SoA rows [feature][n_envs] hold what the lane-per-env Go2Env kernels touch; AoS records [n_envs][record]
// hold the physics-internal arrays of the team kernels (see GO2SIM_FLOAT_FIELDS / GO2SIM_AOS_FLOAT_FIELDS below and DESIGN.md).
// Execution model: the physics kernels (k_*_team) give T lanes to each environment (64/T environments per single-wavefront
// workgroup) and keep an environment's working set in LDS; the Go2Env bookkeeping kernels run one environment per lane.
//
// Reference citations use `R/` = genesis/engine/solvers/rigid/ and `E/` = examples/locomotion/final/ of the reference tree.
// Arithmetic follows the reference's serial (`backend == gs.cpu`) evaluation order -- independent outputs are spread over lanes, chained
// sums stay first-to-last -- so that results are bit-identical to the CPU oracle (oracle/go2sim_cpu.cpp) when both are built with
// -ffp-contract=off.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/go2sim.h"
#include "../../include/go2sim_detmath.h"

#define DEV __device__ __forceinline__
#define DEVN __device__ __noinline__

namespace {

constexpr int NL = GO2SIM_NL, ND = GO2SIM_ND, NQ = GO2SIM_NQ, NG = GO2SIM_NG, NJ = GO2SIM_NJ;
constexpr int NPAIR = GO2SIM_NPAIR_MAX, MAXC = GO2SIM_MAX_CONTACTS, MAXB = GO2SIM_MAX_BROAD, MAXR = GO2SIM_MAX_ROWS;
constexpr int JOINT_FIXED = 0, JOINT_REVOLUTE = 1, JOINT_FREE = 4;
constexpr int GEOM_SPHERE = 1, GEOM_CYLINDER = 3, GEOM_BOX = 5, GEOM_TERRAIN = 7;
constexpr int TERRAIN_CB = 4;   // vertices per side of a block of the coarse maximum map of the heightfield
constexpr int CTRL_FORCE = 0, CTRL_VELOCITY = 1, CTRL_POSITION = 2;
constexpr int NA = 16, NM = 12, NOBS_MAX = 64, NPRIV_MAX = 192, NREW = 32;
constexpr int WG = 64;  // one wavefront per workgroup

// ---------------------------------------------------------------------------------------------
// small vector types (registers)
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };
struct M3 { float m[3][3]; };

constexpr int GJK_SLOTS_MAX = 4;   // LDS polytope slots per environment (one per foot: the landing case asks for four queries at once)
DEV float fmn(float a, float b) { return (b < a) ? b : a; }   // std::min semantics
DEV float fmx(float a, float b) { return (a < b) ? b : a; }   // std::max semantics
DEV int imn(int a, int b) { return (b < a) ? b : a; }
DEV int imx(int a, int b) { return (a < b) ? b : a; }
DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
static inline V3 v3h(float x, float y, float z) { V3 r = {x, y, z}; return r; }   // host-side constructor
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float norm_sqr(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV float norm(V3 a) { return dm_sqrt(norm_sqr(a)); }
DEV V3 normalized(V3 a) { float inv = 1.0f / norm(a); return inv * a; }
DEV float vget(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
DEV void vset(V3& a, int i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
DEV V3 vmin(V3 a, V3 b) { return v3(fmn(a.x, b.x), fmn(a.y, b.y), fmn(a.z, b.z)); }
DEV V3 vmax(V3 a, V3 b) { return v3(fmx(a.x, b.x), fmx(a.y, b.y), fmx(a.z, b.z)); }
DEV float clampf(float x, float lo, float hi) { return fmn(hi, fmx(lo, x)); }
DEV bool isnan_(float x) { return x != x; }
DEV Q4 q4(float w, float x, float y, float z) { Q4 r = {w, x, y, z}; return r; }
DEV float norm_sqr(Q4 q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
DEV Q4 operator*(Q4 q, float s) { return q4(q.w * s, q.x * s, q.y * s, q.z * s); }
DEV Q4 qident() { return q4(1.0f, 0.0f, 0.0f, 0.0f); }
DEV Q4 inv_quat(Q4 q) { return q4(q.w, -q.x, -q.y, -q.z); }
DEV V3 mul(const M3& A, V3 v) {
  return v3(A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z);
}
DEV M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
DEV M3 transpose(const M3& A) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i];
  return C;
}
DEV M3 operator+(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][j] + B.m[i][j];
  return C;
}
DEV V3 mcol(const M3& A, int j) { return v3(j == 0 ? A.m[0][0] : (j == 1 ? A.m[0][1] : A.m[0][2]), j == 0 ? A.m[1][0] : (j == 1 ? A.m[1][1] : A.m[1][2]),
                                          j == 0 ? A.m[2][0] : (j == 1 ? A.m[2][1] : A.m[2][2])); }

// genesis/utils/geom.py:236-242
DEV Q4 quat_mul(Q4 u, Q4 v) {
  float w = u.w * v.w - u.x * v.x - u.y * v.y - u.z * v.z;
  float x = u.w * v.x + u.x * v.w + u.y * v.z - u.z * v.y;
  float y = u.w * v.y - u.x * v.z + u.y * v.w + u.z * v.x;
  float z = u.w * v.z + u.x * v.y - u.y * v.x + u.z * v.w;
  return q4(w, x, y, z);
}
// geom.py:245-252
DEV Q4 transform_quat_by_quat(Q4 v, Q4 u) {
  Q4 q = quat_mul(u, v);
  float inv = 1.0f / dm_sqrt(norm_sqr(q));
  return q4(inv * q.w, inv * q.x, inv * q.y, inv * q.z);
}
// geom.py:255-270
DEV V3 transform_by_quat(V3 v, Q4 q) {
  float q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_wx = q.x * q.w;
  float q_yy = q.y * q.y, q_yz = q.y * q.z, q_wy = q.y * q.w;
  float q_zz = q.z * q.z, q_wz = q.z * q.w;
  float q_ww = q.w * q.w;
  V3 r = v3(v.x * (q_xx + q_ww - q_yy - q_zz) + v.y * (2.0f * q_xy - 2.0f * q_wz) + v.z * (2.0f * q_xz + 2.0f * q_wy),
            v.x * (2.0f * q_wz + 2.0f * q_xy) + v.y * (q_ww - q_xx + q_yy - q_zz) + v.z * (-2.0f * q_wx + 2.0f * q_yz),
            v.x * (-2.0f * q_wy + 2.0f * q_xz) + v.y * (2.0f * q_wx + 2.0f * q_yz) + v.z * (q_ww - q_xx - q_yy + q_zz));
  return r / (q_ww + q_xx + q_yy + q_zz);
}
DEV V3 inv_transform_by_quat(V3 v, Q4 q) { return transform_by_quat(v, inv_quat(q)); }
// transform_by_quat with the quaternion-only part factored out: the nine coefficients and the denominator of geom.py:255-270 evaluated once per
// pose, in the same expression order, so rot_apply(make_rot(q), v) == transform_by_quat(v, q) bit for bit.  The inverse rotation uses the
// transposed coefficients: for q' = (w, -x, -y, -z) the products x'y', x'z', y'z' and the squares are unchanged and w x', w y', w z' only change
// sign (exactly), which turns every coefficient of q' into the transposed coefficient of q (sums commute).  A support query of the narrow phase
// rotates a direction into the geom frame and a vertex back some twenty times per pair with the same two poses.  When the denominator
// |q|^2 is exactly 1 (the ground slab's identity pose) the division is the identity and is skipped.
struct Rot { float c00, c01, c02, c10, c11, c12, c20, c21, c22, den; };
DEV Rot make_rot(Q4 q) {
  float q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_wx = q.x * q.w;
  float q_yy = q.y * q.y, q_yz = q.y * q.z, q_wy = q.y * q.w;
  float q_zz = q.z * q.z, q_wz = q.z * q.w;
  float q_ww = q.w * q.w;
  Rot R;
  R.c00 = q_xx + q_ww - q_yy - q_zz; R.c01 = 2.0f * q_xy - 2.0f * q_wz; R.c02 = 2.0f * q_xz + 2.0f * q_wy;
  R.c10 = 2.0f * q_wz + 2.0f * q_xy; R.c11 = q_ww - q_xx + q_yy - q_zz; R.c12 = -2.0f * q_wx + 2.0f * q_yz;
  R.c20 = -2.0f * q_wy + 2.0f * q_xz; R.c21 = 2.0f * q_wx + 2.0f * q_yz; R.c22 = q_ww - q_xx - q_yy + q_zz;
  R.den = q_ww + q_xx + q_yy + q_zz;
  return R;
}
DEV V3 rot_finish(V3 r, float den) { return (den == 1.0f) ? r : r / den; }
DEV V3 rot_apply(const Rot& R, V3 v) {        // == transform_by_quat(v, q)
  return rot_finish(v3(v.x * R.c00 + v.y * R.c01 + v.z * R.c02, v.x * R.c10 + v.y * R.c11 + v.z * R.c12, v.x * R.c20 + v.y * R.c21 + v.z * R.c22), R.den);
}
DEV V3 rot_apply_inv(const Rot& R, V3 v) {    // == transform_by_quat(v, inv_quat(q))
  return rot_finish(v3(v.x * R.c00 + v.y * R.c10 + v.z * R.c20, v.x * R.c01 + v.y * R.c11 + v.z * R.c21, v.x * R.c02 + v.y * R.c12 + v.z * R.c22), R.den);
}
DEV V3 transform_by_trans_quat(V3 p, V3 t, Q4 q) { return transform_by_quat(p, q) + t; }
DEV void transform_pos_quat_by_trans_quat(V3 pos, Q4 quat, V3 t_trans, Q4 t_quat, V3& opos, Q4& oquat) {
  opos = t_trans + transform_by_quat(pos, t_quat);
  oquat = transform_quat_by_quat(quat, t_quat);
}
// geom.py:136-161
DEV M3 quat_to_R(Q4 q, float eps) {
  M3 R = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  float d = norm_sqr(q);
  if (d > eps) {
    float s = 2.0f / d;
    float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    R.m[0][0] = 1.0f - (yy + zz); R.m[0][1] = xy - wz; R.m[0][2] = xz + wy;
    R.m[1][0] = xy + wz; R.m[1][1] = 1.0f - (xx + zz); R.m[1][2] = yz - wx;
    R.m[2][0] = xz - wy; R.m[2][1] = yz + wx; R.m[2][2] = 1.0f - (xx + yy);
  }
  return R;
}
// geom.py:110-133
DEV Q4 rotvec_to_quat(V3 rv, float eps) {
  Q4 q = q4(0, 0, 0, 0);
  float thetasq = norm_sqr(rv);
  if (thetasq > eps * eps) {
    float theta = dm_sqrt(thetasq);
    float theta_half = 0.5f * theta;
    float s, c;
    dm_sincos(theta_half, &s, &c);
    q.w = c;
    V3 xyz = (s / theta) * rv;
    q.x = xyz.x; q.y = xyz.y; q.z = xyz.z;
    float k = 0.5f * (3.0f - norm_sqr(q));
    q = q * k;
  } else {
    q.w = 1.0f;
  }
  return q;
}
// geom.py:320-336
DEV void transform_inertia_by_trans_quat(const M3& I, float mass, V3 t, Q4 quat, float eps, M3& oI, V3& opos) {
  float xx = t.x * t.x, xy = t.x * t.y, xz = t.x * t.z, yy = t.y * t.y, yz = t.y * t.z, zz = t.z * t.z;
  M3 hhT = {{{yy + zz, -xy, -xz}, {-xy, xx + zz, -yz}, {-xz, -yz, xx + yy}}};
  M3 R = quat_to_R(quat, eps);
  M3 RI = mul(mul(R, I), transpose(R));
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) oI.m[i][j] = RI.m[i][j] + hhT.m[i][j] * mass;
  opos = t * mass;
}
// geom.py:365-383
DEV void inertial_mul(V3 pos, const M3& I, float mass, V3 vel, V3 ang, V3& oang, V3& ovel) {
  oang = mul(I, ang) + cross(pos, vel);
  ovel = mass * vel - cross(pos, ang);
}
DEV void motion_cross_force(V3 m_ang, V3 m_vel, V3 f_ang, V3 f_vel, V3& oang, V3& ovel) {
  ovel = cross(m_ang, f_vel);
  oang = cross(m_ang, f_ang) + cross(m_vel, f_vel);
}
DEV void motion_cross_motion(V3 s_ang, V3 s_vel, V3 m_ang, V3 m_vel, V3& oang, V3& ovel) {
  ovel = cross(s_ang, m_vel) + cross(s_vel, m_ang);
  oang = cross(s_ang, m_ang);
}
// geom.py:386-401
DEV void orthogonals(V3 a, V3& b, V3& c) {
  if (dm_abs(a.y) < 0.5f) b = v3(-a.x * a.y, 1.0f - a.y * a.y, -a.z * a.y);
  else b = v3(-a.x * a.z, -a.y * a.z, 1.0f - a.z * a.z);
  b = normalized(b);
  c = cross(a, b);
}
// geom.py:404-422
DEV void imp_aref(const float* p, float neg_penetration, float vel, float pos, float& imp, float& aref) {
  float timeconst = p[0], dampratio = p[1], dmin = p[2], dmax = p[3], width = p[4], mid = p[5], power = p[6];
  float imp_x = dm_abs(neg_penetration) / width;
  float imp_a = (1.0f / dm_pow(mid, power - 1.0f)) * dm_pow(imp_x, power);
  float imp_b = 1.0f - (1.0f / dm_pow(1.0f - mid, power - 1.0f)) * dm_pow(1.0f - imp_x, power);
  float imp_y = (imp_x < mid) ? imp_a : imp_b;
  imp = dmin + imp_y * (dmax - dmin);
  imp = clampf(imp, dmin, dmax);
  imp = (imp_x > 1.0f) ? dmax : imp;
  float b = 2.0f / (dmax * timeconst);
  float k = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  aref = -b * vel - k * imp * pos;
}

// ---------------------------------------------------------------------------------------------
// model tables (device global memory, read-mostly: L2 / Infinity-Cache resident)
// ---------------------------------------------------------------------------------------------
struct Link {
  int parent, root, entity, is_fixed, joint_start, joint_end, dof_start, dof_end, q_start, q_end, n_dofs, geom_start, geom_end;
  V3 pos; Q4 quat; V3 inertial_pos; Q4 inertial_quat; M3 inertial_i; float mass; float invweight[2];
};
struct Joint { int type, link, q_start, dof_start, dof_end; V3 pos; float sol_params[7]; };
struct Dof { V3 motion_ang, motion_vel; float limit[2], invweight, armature, damping, stiffness, frictionloss, kp, kv, force_range[2]; };
struct Geom { int type, link, is_convex; V3 pos; Q4 quat; float data[7], friction, sol_params[7]; V3 center; V3 aabb[8]; float rim[32][2]; };
struct Entity { int link_start, link_end, dof_start, dof_end, geom_start, geom_end; };
struct Model {
  int n_links, n_joints, n_dofs, n_qs, n_geoms, n_entities, n_pairs, max_collision_pairs, max_contact_pairs, max_broad_pairs,
      n_contacts_per_pair, iterations, ls_iterations, ccd_iterations, support_res;
  float substep_dt; V3 gravity; float eps, tolerance, ls_tolerance, meaninertia, mc_perturbation, mc_tolerance, mpr_to_gjk_ratio, ccd_eps,
      ccd_tolerance;
  Link links[NL]; Joint joints[NJ]; Dof dofs[ND]; Geom geoms[NG]; Entity entities[2];
  float qpos0[NQ]; float mass_parent_mask[ND][ND]; int pair_idx[NG][NG]; int theta_to_ring[180];
  int pair_list[NPAIR];   // derived: valid pair p -> geom_a | geom_b << 8 (a < b)
  // heightfield terrain replacing the ground slab (go2sim_set_terrain; collider.py:374-394); hf is a device pointer
  int terrain_enabled, terrain_rows, terrain_cols; float terrain_hs; float terrain_xyz_maxmin[6]; const float* terrain_hf;
  // derived: maximum height over blocks of TERRAIN_CB x TERRAIN_CB vertices (vertex (r, c) lies in block (r / CB, c / CB)); device pointer
  const float* terrain_cmax; int terrain_crows, terrain_ccols;
  // derived tree tables: links grouped by depth, children of every link in DESCENDING index order (the order in which the
  // reference's leaf->root loops add them to the parent), link of every dof
  int n_levels, level_start[NL + 1], level_links[NL], child_start[NL + 1], child_list[NL], dof_link[ND];
  int arrow_mode;   // derived: numbering of the four leg chains (dm_arrow_mode; 0 = the Hessian has no arrow form)
};

bool parse_model(const void* blob, size_t nbytes, Model& m) {
  m.terrain_enabled = 0; m.terrain_rows = m.terrain_cols = 0; m.terrain_hs = 0.0f; m.terrain_hf = nullptr;
  m.terrain_cmax = nullptr; m.terrain_crows = m.terrain_ccols = 0;
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
  m.substep_dt = f[0]; m.gravity = {f[1], f[2], f[3]}; m.eps = f[4]; m.tolerance = f[5]; m.ls_tolerance = f[6]; m.meaninertia = f[7];
  m.mc_perturbation = f[8]; m.mc_tolerance = f[9]; m.mpr_to_gjk_ratio = f[10]; m.ccd_eps = f[11]; m.ccd_tolerance = f[12];
  f += 16;
  for (int i = 0; i < NL; ++i, f += 26) {
    Link& l = m.links[i];
    l.pos = {f[0], f[1], f[2]}; l.quat = {f[3], f[4], f[5], f[6]}; l.inertial_pos = {f[7], f[8], f[9]}; l.inertial_quat = {f[10], f[11], f[12], f[13]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) l.inertial_i.m[a][b] = f[14 + 3 * a + b];
    l.mass = f[23]; l.invweight[0] = f[24]; l.invweight[1] = f[25];
  }
  for (int i = 0; i < NJ; ++i, f += 10) { m.joints[i].pos = {f[0], f[1], f[2]}; for (int a = 0; a < 7; ++a) m.joints[i].sol_params[a] = f[3 + a]; }
  for (int i = 0; i < ND; ++i, f += 17) {
    Dof& d = m.dofs[i];
    d.motion_ang = {f[0], f[1], f[2]}; d.motion_vel = {f[3], f[4], f[5]}; d.limit[0] = f[6]; d.limit[1] = f[7]; d.invweight = f[8];
    d.armature = f[9]; d.damping = f[10]; d.stiffness = f[11]; d.frictionloss = f[12]; d.kp = f[13]; d.kv = f[14]; d.force_range[0] = f[15];
    d.force_range[1] = f[16];
  }
  for (int i = 0; i < NQ; ++i) m.qpos0[i] = f[i];
  f += NQ;
  for (int i = 0; i < NG; ++i, f += 113) {
    Geom& g = m.geoms[i];
    g.pos = {f[0], f[1], f[2]}; g.quat = {f[3], f[4], f[5], f[6]};
    for (int a = 0; a < 7; ++a) g.data[a] = f[7 + a];
    g.friction = f[14];
    for (int a = 0; a < 7; ++a) g.sol_params[a] = f[15 + a];
    g.center = {f[22], f[23], f[24]};
    for (int a = 0; a < 8; ++a) g.aabb[a] = {f[25 + 3 * a], f[26 + 3 * a], f[27 + 3 * a]};
    for (int a = 0; a < 32; ++a) { g.rim[a][0] = f[49 + 2 * a]; g.rim[a][1] = f[50 + 2 * a]; }
  }
  for (int i = 0; i < ND; ++i) for (int j = 0; j < ND; ++j) m.mass_parent_mask[i][j] = f[i * ND + j];
  m.arrow_mode = getenv("GO2SIM_NO_ARROW") ? 0 : dm_arrow_mode(&m.mass_parent_mask[0][0], ND);   // (GO2SIM_NO_ARROW=1: diagnostic switch, keeps the row-form factorisation under test)
  f += ND * ND;
  if (f - F != nf) return false;
  const int32_t* p = I;
  for (int i = 0; i < NL; ++i, p += 13) {
    Link& l = m.links[i];
    l.parent = p[0]; l.root = p[1]; l.entity = p[2]; l.is_fixed = p[3]; l.joint_start = p[4]; l.joint_end = p[5]; l.dof_start = p[6];
    l.dof_end = p[7]; l.q_start = p[8]; l.q_end = p[9]; l.n_dofs = p[10]; l.geom_start = p[11]; l.geom_end = p[12];
  }
  for (int i = 0; i < NJ; ++i, p += 5) { Joint& j = m.joints[i]; j.type = p[0]; j.link = p[1]; j.q_start = p[2]; j.dof_start = p[3]; j.dof_end = p[4]; }
  for (int i = 0; i < NG; ++i, p += 3) { m.geoms[i].type = p[0]; m.geoms[i].link = p[1]; m.geoms[i].is_convex = p[2]; }
  for (int i = 0; i < 2; ++i, p += 6) {
    Entity& e = m.entities[i];
    e.link_start = p[0]; e.link_end = p[1]; e.dof_start = p[2]; e.dof_end = p[3]; e.geom_start = p[4]; e.geom_end = p[5];
  }
  for (int i = 0; i < NG; ++i) for (int j = 0; j < NG; ++j) m.pair_idx[i][j] = p[i * NG + j];
  p += NG * NG;
  for (int i = 0; i < 180; ++i) m.theta_to_ring[i] = p[i];
  p += 180;
  if ((p - I) != ni) return false;
  for (int i = 0; i < NPAIR; ++i) m.pair_list[i] = 0;
  int n_found = 0;
  for (int a = 0; a < NG; ++a) for (int b = a + 1; b < NG; ++b) {
    int pi = m.pair_idx[a][b];
    if (pi < 0) continue;
    if (pi >= m.n_pairs) return false;
    m.pair_list[pi] = a | (b << 8); n_found++;
  }
  if (n_found != m.n_pairs) return false;
  {
    int depth[NL];
    int max_depth = 0;
    for (int i = 0; i < NL; ++i) {
      int par = m.links[i].parent;
      if (par >= i) return false;   // parents precede children
      depth[i] = (par < 0) ? 0 : depth[par] + 1;
      if (depth[i] > max_depth) max_depth = depth[i];
    }
    m.n_levels = max_depth + 1;
    int k = 0;
    for (int lev = 0; lev <= max_depth; ++lev) {
      m.level_start[lev] = k;
      for (int i = 0; i < NL; ++i) if (depth[i] == lev) m.level_links[k++] = i;
    }
    for (int lev = max_depth + 1; lev <= NL; ++lev) m.level_start[lev] = k;
    k = 0;
    for (int par = 0; par < NL; ++par) {
      m.child_start[par] = k;
      for (int i = NL - 1; i > par; --i) if (m.links[i].parent == par) m.child_list[k++] = i;
    }
    m.child_start[NL] = k;
    for (int d = 0; d < ND; ++d) m.dof_link[d] = -1;
    for (int i = 0; i < NL; ++i) for (int d = m.links[i].dof_start; d < m.links[i].dof_end; ++d) m.dof_link[d] = i;
    for (int d = 0; d < ND; ++d) if (m.dof_link[d] < 0) return false;
  }
  int g_next = 0;   // geoms must be stored link-major (the SAP buffer is initialised in (link, geom) order)
  for (int i = 0; i < NL; ++i) { if (m.links[i].geom_start != g_next) return false; g_next = m.links[i].geom_end; }
  return g_next == NG;
}


// ---------------------------------------------------------------------------------------------
// ModelS: the small tables of the model (tree topology, link / joint / dof constants, geom frames, derived level and
// child lists, lower-triangle index LUT).  Team kernels copy it to LDS once per workgroup so that the lane-varying table
// walks (parent chains, per-level link lists) are LDS reads instead of dependent global loads.  Member names match Model.
// ---------------------------------------------------------------------------------------------
struct LinkS {
  int parent, entity, is_fixed, joint_start, joint_end, dof_start, dof_end, q_start, q_end, n_dofs;
  V3 pos; Q4 quat; V3 inertial_pos; Q4 inertial_quat; M3 inertial_i; float mass; float invweight[2];
};
struct GeomS { int type, link; V3 pos; Q4 quat; };
struct JLim { int q_start, dof_start; float lo, hi; };   // revolute joints; q_start = -1 for the others (add_joint_limit_constraints, solver.py:1088-1143)
constexpr int NTRI = ND * (ND + 1) / 2;
struct alignas(16) ModelS {
  int n_levels, iterations, ls_iterations, arrow_mode;
  float substep_dt; V3 gravity; float eps, tolerance, ls_tolerance, meaninertia;
  // ---- block staged by the constraint solver (contiguous: links, triangle LUT, joint-limit table) ----
  LinkS links[NL];
  unsigned char tri_i[NTRI + 1], tri_j[NTRI + 1];
  JLim jlim[NJ];
  // ---- the rest is only needed by the kinematics / dynamics kernels ----
  Joint joints[NJ]; Dof dofs[ND]; GeomS geoms[NG]; Entity entities[2];
  float qpos0[NQ]; unsigned mass_mask_bits[ND];
  int level_start[NL + 1], level_links[NL], child_start[NL + 1], child_list[NL], dof_link[ND];
};
constexpr int SOLVER_BLOCK_BYTES = (int)(sizeof(LinkS) * NL + 2 * (NTRI + 1) + sizeof(JLim) * NJ);
static_assert(sizeof(ModelS) % 16 == 0 && (2 * (NTRI + 1)) % 4 == 0 && offsetof(ModelS, links) % 16 == 0 && offsetof(ModelS, jlim) == offsetof(ModelS, links) + sizeof(LinkS) * NL + 2 * (NTRI + 1), "ModelS is copied in 16-byte granules");

bool build_model_s(const Model& m, ModelS& o) {
  memset(&o, 0, sizeof(o));
  o.n_levels = m.n_levels; o.iterations = m.iterations; o.ls_iterations = m.ls_iterations; o.arrow_mode = m.arrow_mode;
  o.substep_dt = m.substep_dt; o.gravity = m.gravity; o.eps = m.eps; o.tolerance = m.tolerance; o.ls_tolerance = m.ls_tolerance; o.meaninertia = m.meaninertia;
  for (int i = 0; i < NL; ++i) {
    const Link& a = m.links[i]; LinkS& b = o.links[i];
    b.parent = a.parent; b.entity = a.entity; b.is_fixed = a.is_fixed; b.joint_start = a.joint_start; b.joint_end = a.joint_end; b.dof_start = a.dof_start;
    b.dof_end = a.dof_end; b.q_start = a.q_start; b.q_end = a.q_end; b.n_dofs = a.n_dofs; b.pos = a.pos; b.quat = a.quat; b.inertial_pos = a.inertial_pos;
    b.inertial_quat = a.inertial_quat; b.inertial_i = a.inertial_i; b.mass = a.mass; b.invweight[0] = a.invweight[0]; b.invweight[1] = a.invweight[1];
  }
  for (int i = 0; i < NJ; ++i) o.joints[i] = m.joints[i];
  for (int i = 0; i < ND; ++i) o.dofs[i] = m.dofs[i];
  for (int i = 0; i < NG; ++i) { o.geoms[i].type = m.geoms[i].type; o.geoms[i].link = m.geoms[i].link; o.geoms[i].pos = m.geoms[i].pos; o.geoms[i].quat = m.geoms[i].quat; }
  for (int i = 0; i < 2; ++i) o.entities[i] = m.entities[i];
  for (int i = 0; i < NQ; ++i) o.qpos0[i] = m.qpos0[i];
  for (int i = 0; i < ND; ++i) {
    unsigned bits = 0;
    for (int j = 0; j < ND; ++j) {
      float v = m.mass_parent_mask[i][j];
      if (v == 1.0f) bits |= 1u << j; else if (v != 0.0f) return false;   // the mask must be a 0/1 matrix
    }
    o.mass_mask_bits[i] = bits;
  }
  for (int i = 0; i <= NL; ++i) { o.level_start[i] = m.level_start[i]; o.child_start[i] = m.child_start[i]; }
  for (int i = 0; i < NL; ++i) { o.level_links[i] = m.level_links[i]; o.child_list[i] = m.child_list[i]; }
  for (int i = 0; i < ND; ++i) o.dof_link[i] = m.dof_link[i];
  int k = 0;
  for (int i = 0; i < ND; ++i) for (int j = 0; j <= i; ++j) { o.tri_i[k] = (unsigned char)i; o.tri_j[k] = (unsigned char)j; k++; }
  for (int i = 0; i < NJ; ++i) {
    const Joint& J = m.joints[i];
    bool rev = J.type == JOINT_REVOLUTE;
    o.jlim[i].q_start = rev ? J.q_start : -1; o.jlim[i].dof_start = J.dof_start;
    o.jlim[i].lo = rev ? m.dofs[J.dof_start].limit[0] : 0.0f; o.jlim[i].hi = rev ? m.dofs[J.dof_start].limit[1] : 0.0f;
  }
  return true;
}
DEV float mass_mask(const ModelS& m, int i, int j) { return (float)((m.mass_mask_bits[i] >> j) & 1u); }
DEV void tri_index(const ModelS& m, int idx, int& i, int& j) { i = m.tri_i[idx]; j = m.tri_j[idx]; }

// View used by the team kernels: scalars come from the global copy through uniform (scalar) loads and live in SGPRs, tables point into LDS
struct ModelView {
  int n_levels, iterations, ls_iterations, arrow_mode; float substep_dt; V3 gravity; float eps, tolerance, ls_tolerance, meaninertia;
  const LinkS* links; const Joint* joints; const Dof* dofs; const GeomS* geoms; const Entity* entities; const float* qpos0;
  const unsigned* mass_mask_bits; const int *level_start, *level_links, *child_start, *child_list, *dof_link; const unsigned char *tri_i, *tri_j;
  DEV ModelView(const ModelS* t, const ModelS* __restrict__ g)
      : n_levels(g->n_levels), iterations(g->iterations), ls_iterations(g->ls_iterations), arrow_mode(g->arrow_mode), substep_dt(g->substep_dt), gravity(g->gravity), eps(g->eps),
        tolerance(g->tolerance), ls_tolerance(g->ls_tolerance), meaninertia(g->meaninertia), links(t->links), joints(t->joints), dofs(t->dofs),
        geoms(t->geoms), entities(t->entities), qpos0(t->qpos0), mass_mask_bits(t->mass_mask_bits), level_start(t->level_start),
        level_links(t->level_links), child_start(t->child_start), child_list(t->child_list), dof_link(t->dof_link), tri_i(t->tri_i), tri_j(t->tri_j) {}
  // solver flavour: only the link table and the triangle LUT are staged in LDS; joint / dof constants are read with uniform indices and stay
  // behind scalar loads of the global model
  DEV ModelView(const LinkS* lds_links, const unsigned char* lds_tri_i, const unsigned char* lds_tri_j, const Model* __restrict__ g)
      : n_levels(g->n_levels), iterations(g->iterations), ls_iterations(g->ls_iterations), arrow_mode(g->arrow_mode), substep_dt(g->substep_dt), gravity(g->gravity), eps(g->eps),
        tolerance(g->tolerance), ls_tolerance(g->ls_tolerance), meaninertia(g->meaninertia), links(lds_links), joints(g->joints), dofs(g->dofs),
        geoms(nullptr), entities(g->entities), qpos0(g->qpos0), mass_mask_bits(nullptr), level_start(nullptr), level_links(nullptr), child_start(nullptr),
        child_list(nullptr), dof_link(nullptr), tri_i(lds_tri_i), tri_j(lds_tri_j) {}
};
DEV float mass_mask(const ModelView& m, int i, int j) { return (float)((m.mass_mask_bits[i] >> j) & 1u); }
DEV void tri_index(const ModelView& m, int idx, int& i, int& j) { i = m.tri_i[idx]; j = m.tri_j[idx]; }
// workgroup-cooperative copy of the compact model into LDS (64 threads)
// Global -> LDS copy of a read-only table by the memory pipeline itself (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction,
// no staging registers).  Asynchronous: it is issued first, the kernel's other staging loads follow, and the barrier that ends the staging
// (team_sync = s_waitcnt vmcnt(0) + s_barrier) makes the table visible.  Source and destination are padded to whole KiB.
constexpr int lds_dma_bytes(int nbytes) { return (nbytes + 1023) / 1024 * 1024; }
template <int NBYTES>
DEV void wg_dma_to_lds(void* lds_dst, const void* __restrict__ src, int lane = (int)threadIdx.x) {   // one wavefront; `lane` = its lane index (workgroups of several wavefronts pass threadIdx.x % 64)
#pragma unroll
  for (int k = 0; k < lds_dma_bytes(NBYTES) / 1024; ++k)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)src + (k * 64 + lane) * 16),
                                     (__attribute__((address_space(3))) void*)((char*)lds_dst + k * 1024), 16, 0, 0);
}
constexpr int MODELS_LDS_BYTES = lds_dma_bytes((int)sizeof(ModelS));

// ---------------------------------------------------------------------------------------------
// SoA state pool.  X(name, floats_per_env).  Order of the first group matches enum go2sim_field so
// that go2sim_get_field / go2sim_set_field are plain device copies.
// ---------------------------------------------------------------------------------------------
// Two pools.  SoA rows [feature][n_envs] hold what the lane-per-env Go2Env kernels touch (coalesced across envs); the AoS records
// [n_envs][record] hold the physics-internal arrays that only the team kernels (and the field API) touch: the lanes of a team then
// read / write consecutive words of one record (coalesced across the team) instead of one 64-byte sector per word.
#define GO2SIM_FLOAT_FIELDS(X)                                                                                       \
  X(qpos, NQ) X(vel, ND) X(ctrl_force, ND) X(ext, NL * 6) X(mass_shift, NL) X(com_shift, NL * 3)                        \
  X(friction_ratio, NG) X(l_pos, NL * 3) X(l_quat, NL * 4) X(cd_vel, NL * 3) X(cd_ang, NL * 3) X(root_com, NL * 3)     \
  X(contact_force, NL * 3) X(geom_friction, NG) X(ctrl_pos, ND) X(ctrl_vel, ND) X(dof_pos, ND) \
  /* ---- Go2Env buffers ---- */                                                                                     \
  X(actions, NA) X(last_actions, NA) X(applied_actions, NA) X(action_history, GO2SIM_ACTION_RING_MAX * NA) X(target_dof_pos, NM)            \
  X(e_dof_pos, NM) X(e_dof_vel, NM) X(last_dof_vel, NM) X(torque, NM) X(base_pos, 3) X(base_quat, 4) X(base_lin_vel, 3) \
  X(base_ang_vel, 3) X(projected_gravity, 3) X(base_euler, 3) X(commands, 3) X(time_out, 1) X(kp_factors, NM)          \
  X(kd_factors, NM) X(motor_strength, NM) X(gravity_offset, 3) X(current_push_force, 3) X(push_stored_force, 3)       \
  X(feet_air_time, 4) X(base_vel_world, 3) X(last_base_pos_x, 1) X(episode_sums, NREW) X(rew_terms, NREW) X(rew, 1) X(obs, NOBS_MAX) X(priv, NPRIV_MAX)

#define GO2SIM_INT_FIELDS(X)                                                                                         \
  X(n_contacts, 1) X(n_con, 1) X(err, 1) X(is_warmstart, 1) X(first_time, 1) X(n_broad, 1) X(solver_iters, 1) X(ctrl_mode, ND) \
  X(gjk_fallback, 1) X(delay_steps, 1) X(episode_length, 1) X(reset_buf, 1) X(push_remaining, 1) X(foot_contact, 4)    \
  X(last_foot_contact, 4) X(terrain_row, 1) X(terrain_key, 1)

#define GO2SIM_AOS_FLOAT_FIELDS(X)                                                                                   \
  X(acc, ND) X(qacc_ws, ND) X(force, ND) X(qf_smooth, ND) X(acc_smooth, ND) X(qfrc_constraint, ND) X(mass_mat, NTRI)  \
  X(cdof_ang, ND * 3) X(cdof_vel, ND * 3) X(cdofd_ang, ND * 3) X(cdofd_vel, ND * 3) X(cinr_inertial, NL * 9)            \
  X(cinr_pos, NL * 3) X(cinr_mass, NL) X(i_pos, NL * 3) X(i_quat, NL * 4) X(g_pos, NG * 3) X(g_quat, NG * 4)              \
  X(sort_value, 2 * NG) X(c_pos, MAXC * 3) X(c_normal, MAXC * 3) X(c_pen, MAXC) X(c_friction, MAXC) X(c_sol, MAXC * 7)    \
  X(c_force, MAXC * 3) X(efc_force, MAXR) X(normal_cache, NPAIR * 3)

// ncache_valid: one bit per geom pair, set = normal_cache[pair] holds the contact normal of the last narrow phase; clear = the cache entry reads as the
// zero vector (func_broad_phase / func_convex_convex_contact write zeros there: here they clear a bit of a mask the kernel keeps in LDS)
constexpr int NCV = (NPAIR + 31) / 32;
#define GO2SIM_AOS_INT_FIELDS(X) X(sort_ig, 2 * NG) X(broad, MAXB * 2) X(c_geom, 2 * MAXC) X(c_link, 2 * MAXC) X(ncache_valid, NCV)

enum FOff : int {
#define X(n, c) FO_##n##_, FO_##n##_end = FO_##n##_ + (c) - 1,
  GO2SIM_FLOAT_FIELDS(X)
#undef X
  FO_TOTAL
};
enum IOff : int {
#define X(n, c) IO_##n##_, IO_##n##_end = IO_##n##_ + (c) - 1,
  GO2SIM_INT_FIELDS(X)
#undef X
  IO_TOTAL
};
enum AOff : int {
#define X(n, c) AO_##n##_, AO_##n##_end = AO_##n##_ + (c) - 1,
  GO2SIM_AOS_FLOAT_FIELDS(X)
#undef X
  AO_TOTAL
};
enum AIOff : int {
#define X(n, c) AIO_##n##_, AIO_##n##_end = AIO_##n##_ + (c) - 1,
  GO2SIM_AOS_INT_FIELDS(X)
#undef X
  AIO_TOTAL
};
#define FO(n) FO_##n##_
#define IO(n) IO_##n##_
#define AO(n) AO_##n##_
#define AIO(n) AIO_##n##_
constexpr int ASTRIDE = (AO_TOTAL + 15) / 16 * 16;     // floats per AoS record (64-byte aligned records)
constexpr int AISTRIDE = (AIO_TOTAL + 15) / 16 * 16;   // ints per AoS record

// per-lane view of the pool
template <typename T>
struct Arr {
  T* p; int B;
  DEV T& operator[](int i) const { return p[(size_t)i * B]; }
};
struct V3Ref {
  float* p; int B;
  DEV operator V3() const { return v3(p[0], p[B], p[2 * (size_t)B]); }
  DEV const V3Ref& operator=(V3 v) const { p[0] = v.x; p[B] = v.y; p[2 * (size_t)B] = v.z; return *this; }
  DEV const V3Ref& operator=(const V3Ref& o) const { V3 v = o; return *this = v; }
};
struct Q4Ref {
  float* p; int B;
  DEV operator Q4() const { return q4(p[0], p[B], p[2 * (size_t)B], p[3 * (size_t)B]); }
  DEV const Q4Ref& operator=(Q4 q) const { p[0] = q.w; p[B] = q.x; p[2 * (size_t)B] = q.y; p[3 * (size_t)B] = q.z; return *this; }
  DEV const Q4Ref& operator=(const Q4Ref& o) const { Q4 v = o; return *this = v; }
};
struct M3Ref {
  float* p; int B;
  DEV operator M3() const { M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i / 3][i % 3] = p[(size_t)i * B];
    return r; }
  DEV const M3Ref& operator=(const M3& r) const {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[(size_t)i * B] = r.m[i / 3][i % 3];
    return *this; }
  DEV const M3Ref& operator=(const M3Ref& o) const { M3 v = o; return *this = v; }
};
struct Arr3 { float* p; int B; DEV V3Ref operator[](int i) const { return V3Ref{p + (size_t)(3 * i) * B, B}; } };
struct Arr4 { float* p; int B; DEV Q4Ref operator[](int i) const { return Q4Ref{p + (size_t)(4 * i) * B, B}; } };
struct Arr9 { float* p; int B; DEV M3Ref operator[](int i) const { return M3Ref{p + (size_t)(9 * i) * B, B}; } };
template <int W>
struct Arr2 {
  float* p; int B;
  DEV Arr<float> operator[](int i) const { return Arr<float>{p + (size_t)(W * i) * B, B}; }
};

struct Pool { float* f; int* i; int B; float* fa; int* ia; };

// Env view: field accessors of lane b
struct E {
  float* f; int* i; int B; int b; float* fa; int* ia;
  DEV E(const Pool& P, int b_) : f(P.f + b_), i(P.i + b_), B(P.B), b(b_), fa(P.fa + (size_t)b_ * ASTRIDE), ia(P.ia + (size_t)b_ * AISTRIDE) {}
#define FA(name) DEV Arr<float> name() const { return Arr<float>{f + (size_t)FO(name) * B, B}; }
#define FA3(name) DEV Arr3 name() const { return Arr3{f + (size_t)FO(name) * B, B}; }
#define FA4(name) DEV Arr4 name() const { return Arr4{f + (size_t)FO(name) * B, B}; }
#define FA2(name, W) DEV Arr2<W> name() const { return Arr2<W>{f + (size_t)FO(name) * B, B}; }
#define IA(name) DEV Arr<int> name() const { return Arr<int>{i + (size_t)IO(name) * B, B}; }
#define AA(name) DEV Arr<float> name() const { return Arr<float>{fa + AO(name), 1}; }
#define AA3(name) DEV Arr3 name() const { return Arr3{fa + AO(name), 1}; }
#define AA4(name) DEV Arr4 name() const { return Arr4{fa + AO(name), 1}; }
#define AA9(name) DEV Arr9 name() const { return Arr9{fa + AO(name), 1}; }
#define AA2(name, W) DEV Arr2<W> name() const { return Arr2<W>{fa + AO(name), 1}; }
#define AIA(name) DEV Arr<int> name() const { return Arr<int>{ia + AIO(name), 1}; }
  FA(qpos) FA(vel) FA(ctrl_force) FA(ctrl_pos) FA(ctrl_vel) FA(ext) FA(mass_shift) FA3(com_shift) FA(friction_ratio)
  FA(geom_friction) FA3(l_pos) FA4(l_quat) FA3(root_com) FA(dof_pos) FA3(cd_vel) FA3(cd_ang) FA3(contact_force)
  FA(actions) FA(last_actions) FA(applied_actions) FA2(action_history, NA) FA(target_dof_pos) FA(e_dof_pos) FA(e_dof_vel) FA(last_dof_vel)
  FA(torque) FA(base_pos) FA(base_quat) FA(base_lin_vel) FA(base_ang_vel) FA(projected_gravity) FA(base_euler) FA(commands) FA(time_out)
  FA(kp_factors) FA(kd_factors) FA(motor_strength) FA(gravity_offset) FA(current_push_force) FA(push_stored_force) FA(feet_air_time) FA(base_vel_world) FA(last_base_pos_x)
  FA(episode_sums) FA(rew_terms) FA(rew) FA(obs) FA(priv)
  IA(n_contacts) IA(n_con) IA(err) IA(is_warmstart) IA(first_time) IA(n_broad) IA(solver_iters) IA(ctrl_mode)
  IA(gjk_fallback) IA(delay_steps) IA(episode_length) IA(reset_buf) IA(push_remaining) IA(foot_contact) IA(last_foot_contact) IA(terrain_row) IA(terrain_key)
  AA(acc) AA(qacc_ws) AA(force) AA(qf_smooth) AA(acc_smooth) AA(qfrc_constraint) AA3(cdof_ang) AA3(cdof_vel) AA3(cdofd_ang)
  AA3(cdofd_vel) AA9(cinr_inertial) AA3(cinr_pos) AA(cinr_mass) AA3(i_pos) AA4(i_quat) AA3(g_pos) AA4(g_quat) AA(sort_value) AA3(c_pos) AA3(c_normal)
  AA(c_pen) AA(c_friction) AA2(c_sol, 7) AA3(c_force) AA(efc_force) AA3(normal_cache)
  AIA(sort_ig) AIA(broad) AIA(c_geom) AIA(c_link) AIA(ncache_valid)
#undef FA
#undef FA3
#undef FA4
#undef FA2
#undef IA
#undef AA
#undef AA3
#undef AA4
#undef AA9
#undef AA2
#undef AIA
};
// Ordering point between the phases of a team.  Workgroups are single wavefronts and a wavefront executes its LDS / memory instructions in
// order, so what is needed is (a) that the compiler does not move accesses across the point and (b) that outstanding memory operations have
// completed (s_waitcnt) -- a workgroup-scope release / acquire fence pair around a wave barrier.  No s_barrier: teams of one wavefront reach
// these points under team-divergent control flow (different Newton / line-search trip counts, early exits), where a hardware barrier would be
// undefined behaviour.
#ifdef GO2SIM_TEAM_SYNC_SBARRIER
DEV void team_sync() { __syncthreads(); }
#else
DEV void team_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif
DEV float gload(const E& e, int off, int k) { return e.f[(size_t)(off + k) * e.B]; }
DEV void gstore(const E& e, int off, int k, float v) { e.f[(size_t)(off + k) * e.B] = v; }
DEV float aload(const E& e, int off, int k) { return e.fa[off + k]; }              // AoS record word
DEV void astore(const E& e, int off, int k, float v) { e.fa[off + k] = v; }


// Workgroup-cooperative staging between the SoA pool and per-env LDS blocks: adjacent lanes address adjacent environments of the same
// pool row, so one wave instruction touches 64/EPW rows with EPW contiguous floats each instead of 64 separate sectors.
// statically unrolled team loop: k = tl, tl + T, ... < N.  With compile-time bounds the loads of all rounds are issued before the first
// wait; a dynamic `for (k = tl; k < N; k += T)` loop pays one memory round trip per round instead.
template <int N, int T, class F>
DEV void team_for(int tl, F f) {
#pragma unroll
  for (int k0 = 0; k0 < N; k0 += T) { int k = k0 + tl; if (k < N) f(k); }
}
// branch-free staging: out-of-range lanes redo the last element (same value to the same address), so neither the loads nor the stores
// sit behind a branch and the loads of consecutive team_stage calls are all in flight together.
// (written as a recursion: round r loads, the deeper rounds run, then round r stores -- no private array for the compiler to spill)
template <int N, int STRIDE, int R, class FL, class FS>
DEV void team_stage_r(int k0, FL& ld, FS& st) {
  if constexpr (R * STRIDE < N) {
    int k = R * STRIDE + k0;
    k = k < N ? k : N - 1;
    const float v = ld(k);
    team_stage_r<N, STRIDE, R + 1>(k0, ld, st);
    st(k, v);
  }
}
template <int N, int T, class FL, class FS>
DEV void team_stage(int tl, FL ld, FS st) { team_stage_r<N, T, 0>(tl, ld, st); }
template <int EPW, int COUNT, class F>
DEV void wg_load(const Pool& P, int b0, int off, F put) {
  const int ev = threadIdx.x % EPW, k0 = threadIdx.x / EPW;
  const int b = (b0 + ev < P.B) ? b0 + ev : P.B - 1;
  auto ld = [&](int k) { return P.f[(size_t)(off + k) * P.B + b]; };
  auto st = [&](int k, float v) { put(ev, k, v); };
  team_stage_r<COUNT, 64 / EPW, 0>(k0, ld, st);
}

// ---- optional per-phase cycle accounting (build with -DGO2SIM_PHASE_PROFILE; development aid, see tools/phase_profile.py) ----
#ifdef GO2SIM_PHASE_PROFILE
// one row of 64 counters per workgroup (no atomics, no sharing: the former single row of atomically updated counters cost more than the
// phases it measured); phase ids are disjoint between kernels, so the kernels of a step share the rows by blockIdx
constexpr int PH_MAX_WG = 8192;
__device__ unsigned long long g_phase_cycles[PH_MAX_WG * 64];
#define PH_BEGIN unsigned long long ph_t = __builtin_readcyclecounter();
#define PH(i) { __builtin_amdgcn_s_waitcnt(0); unsigned long long ph_n = __builtin_readcyclecounter(); if (threadIdx.x == 0 && blockIdx.x < PH_MAX_WG) g_phase_cycles[blockIdx.x * 64 + (i)] += ph_n - ph_t; ph_t = __builtin_readcyclecounter(); }
// the same inside lane-divergent code: the first active lane of the wavefront accounts the section (all active lanes run it together)
#define PHD_BEGIN unsigned long long phd_t = __builtin_readcyclecounter();
#define PHD(i) { unsigned long long phd_n = __builtin_readcyclecounter(); const unsigned long long phd_a = __ballot(1); if ((int)threadIdx.x == __ffsll((long long)phd_a) - 1 && blockIdx.x < PH_MAX_WG) { atomicAdd(&g_phase_cycles[blockIdx.x * 64 + (i)], phd_n - phd_t); atomicAdd(&g_phase_cycles[blockIdx.x * 64 + (i) + 1], 1ull); } phd_t = __builtin_readcyclecounter(); }
#define PHC(i, n) { if (threadIdx.x == 0 && blockIdx.x < PH_MAX_WG) g_phase_cycles[blockIdx.x * 64 + (i)] += (unsigned long long)(n); }   // event counter
#else
#define PH_BEGIN
#define PH(i)
#define PHD_BEGIN
#define PHD(i)
#define PHC(i, n)
#endif

// ---- optional wall-clock stamps per workgroup (build with -DGO2SIM_STAMP; tools/launch_overhead.py): every workgroup of the step kernels records
// s_memrealtime (the 100 MHz constant clock shared by all CUs) at entry and after its last memory operation has retired.  From one step's stamps the
// host gets, per launch, the span between the earliest workgroup start and the latest workgroup end, and between launches the time in which NO
// workgroup runs (end-of-kernel write-back + dispatch of the next launch): the fixed cost of a launch that rocprof's kernel durations hide. ----
#ifdef GO2SIM_STAMP
constexpr int ST_KINDS = 8, ST_MAX_WG = 4096, ST_DEPTH = 4;
__device__ unsigned long long g_stamp[ST_KINDS * ST_MAX_WG * ST_DEPTH * 2];
__device__ unsigned g_stamp_cnt[ST_KINDS * ST_MAX_WG];
struct StampScope {
  int kind; unsigned long long t0;
  DEV StampScope(int k) : kind(k) { t0 = __builtin_amdgcn_s_memrealtime(); }
  DEV ~StampScope() {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < ST_MAX_WG) {
      const int w = kind * ST_MAX_WG + (int)blockIdx.x;
      const unsigned c = g_stamp_cnt[w];                               // only this workgroup id of this kind touches the slot: no atomics
      g_stamp[(w * ST_DEPTH + (int)(c % ST_DEPTH)) * 2] = t0; g_stamp[(w * ST_DEPTH + (int)(c % ST_DEPTH)) * 2 + 1] = t1;
      g_stamp_cnt[w] = c + 1;
    }
  }
};
#define STAMP(kind) StampScope stamp_scope_(kind);
#else
#define STAMP(kind)
#endif
enum { STK_PRE_DYN = 0, STK_COLLIDE, STK_SOLVE, STK_INT_FK_DYN, STK_INT_FK, STK_POST_A, STK_POST_B, STK_OTHER };

// -DGO2SIM_REPEAT_PHASE=k (profiling builds, tools/repeat_probe.py): phase k runs twice; every such phase is idempotent, so the results
// are unchanged and the time difference prices the phase.  Solver: 0 stage, 1 rows (13 contact rows, 14 joint-limit rows), 3 Hessian +
// factorisation, 4 Hessian, 5 gradient, 6 line search, 11 commit.  Collision: 30 AABBs, 31 endpoint sort, 32 candidate pairs, 34 GJK / EPA
// query, 35 MPR query.
// scalar slots
enum { SV_COST = 0, SV_PREV_COST, SV_GAUSS, SV_QG0, SV_QG1, SV_QG2, SV_GTOL };
enum { SI_LS_IT = 0, SI_LS_RESULT, SI_IMPROVED };

// ---------------------------------------------------------------------------------------------
// Team kinematics / dynamics: T lanes per environment, link tree processed level by level
// (Go2: base | 4 hips | 4 thighs | 4 calves), independent links / dofs / matrix entries spread over the lanes, every
// chained sum evaluated in the serial order of the reference.
// ---------------------------------------------------------------------------------------------
DEV V3 ld3(const float* p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
DEV void st3(float* p, int i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
DEV Q4 ld4(const float* p, int i) { return q4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]); }
DEV void st4(float* p, int i, Q4 q) { p[4 * i] = q.w; p[4 * i + 1] = q.x; p[4 * i + 2] = q.y; p[4 * i + 3] = q.z; }
DEV M3 ld9(const float* p, int i) { M3 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.m[k / 3][k % 3] = p[9 * i + k];
  return r; }
DEV void st9(float* p, int i, const M3& r) {
#pragma unroll
  for (int k = 0; k < 9; ++k) p[9 * i + k] = r.m[k / 3][k % 3]; }

// Workgroups are dealt to the 8 XCDs round-robin (workgroups i and i + 8 share an XCD) and every XCD has its own L2.  With the plain blockIdx -> env
// map the 8 (T = 16) or 16 (T = 32) workgroups whose envs share one 128-byte line of a [feature][env] row sit on 8 different XCDs: every L2 fetches
// the whole line for 4 or 8 of its bytes and writes it back piecemeal.  Logical block ids are handed out so that the workgroups of one XCD cover one
// contiguous range of envs (speed only: any placement gives the same results).
DEV int xcd_block() {
#ifdef GO2SIM_NO_XCD_REMAP
  return (int)blockIdx.x;
#endif
  const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- heaviest-first dispatch of the constraint solver ------------------------------------------------------------------------------------
// A solver launch lasts as long as its last wavefront.  On the heightfield (one env per wavefront, two residency rounds) an expensive env that is
// dispatched late finishes long after everything else.  k_collide_team therefore files every env under (XCD of its solver workgroup, contact count)
// with one atomic per env, and solver workgroup w takes the w-th env of its XCD in descending contact count instead of env w: the expensive solves
// start first (and envs of similar cost share a wavefront at T = 32).  Which workgroup solves an env does not enter its result.
// Record: cnt[8][LPT_CLS] followed by env[8][LPT_CLS][cap].  Two records alternate between consecutive collide / solve pairs; the solver clears the
// record of the next pair.  A record whose per-XCD total is not what the solver grid expects (stale, filled twice) is ignored: identity mapping.
constexpr int LPT_CLS = 32;
DEV int lpt_solver_xcd(int b, int B, int epw_s) {                          // XCD of the solver workgroup that holds env b under the identity mapping (inverse of xcd_block)
  const int n = (B + epw_s - 1) / epw_s, q = n >> 3, r = n & 7, L = b / epw_s;
  if (L < r * (q + 1)) return L / (q + 1);
  return r + (L - r * (q + 1)) / (q > 0 ? q : 1);
}
// (the contact count alone is the better key: weighting it with the Newton iterations of the previous solve measured 2-3 % slower)
// files solver block L (= its epw_s consecutive envs) under the largest contact count of its envs
DEV void lpt_file(int* __restrict__ rec, int cap, int L, int B, int epw_s, int nc) {
  const int x = lpt_solver_xcd(L * epw_s, B, epw_s), cls = LPT_CLS - 1 - imn(LPT_CLS - 1, nc);
  const int pos = atomicAdd(&rec[x * LPT_CLS + cls], 1);
  if (pos < cap) rec[8 * LPT_CLS + (x * LPT_CLS + cls) * cap + pos] = L;
}
// env of (workgroup, slot) in the solver grid; B (= no env) past the end of the XCD's list
DEV int lpt_take(const int* __restrict__ rec, int cap, int B, int epw, int slot) {
  const int ident = xcd_block() * epw + slot;
#ifdef GO2SIM_NO_XCD_REMAP
  return ident;
#endif
  if (rec == nullptr) return ident;
  const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = (int)blockIdx.x & 7;
  const int start = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q), nblk = q + (x < r ? 1 : 0);
  (void)start;
  const int expected = nblk;                                             // blocks of this XCD
  const int local = (int)blockIdx.x >> 3;
  int total = 0, cls = -1, off = 0;
#pragma unroll
  for (int c = 0; c < LPT_CLS; ++c) {
    const int k = rec[x * LPT_CLS + c];
    if (cls < 0 && local < total + k) { cls = c; off = local - total; }
    total += k;
  }
  if (total != expected) return ident;
  if (cls < 0 || off >= cap) return B;
  return rec[8 * LPT_CLS + (x * LPT_CLS + cls) * cap + off] * epw + slot;
}

struct KinData {
  float qpos[NQ], vel[ND], qpos_next[NQ], vel_next[ND];
  float l_pos[NL * 3], l_quat[NL * 4], i_pos[NL * 3], i_quat[NL * 4];
  float xanchor[NJ * 3], xaxis[NJ * 3];
  float cdof_ang[ND * 3], cdof_vel[ND * 3];
  float cd_vel[NL * 3], cd_ang[NL * 3];
  float mass[NL];
  int valid;
};

#ifndef GO2SIM_FAST_ORDER
#define GO2SIM_FAST_ORDER 1
#endif
#ifndef REBUILD_FLIPS
#define REBUILD_FLIPS 1
#endif
template <int CTRL>
DEV float dpp_perm(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true)); }
#if GO2SIM_FAST_ORDER
// ---- FAST ORDER, arrow form ------------------------------------------------------------------------------------------------------------------
// A floating base with four legs gives the mass matrix M -- and the Newton Hessian H = M + J^T D J as long as no contact joins links of two legs -- an
// arrow shape: the 3 x 3 leg blocks A_l (dofs dm_arrow_dof(mode, l, 0..2); Model::arrow_mode from the mass-matrix mask) couple only to the 6 base dofs
// (C_l, 3 x 6), never to each other.  Eliminating the legs FIRST keeps that shape (no fill-in), so the factorisation of the permuted matrix [legs..., base] is
//     A_l = L_l L_l^T  (four 3 x 3 factorisations side by side),   W_l = C_l^T L_l^-T  (6 x 3),   B' = B - sum_l W_l W_l^T,   B' = L_b L_b^T  (6 x 6)
// -- 3 + 6 dependent pivots instead of 18, and small enough to run without any exchange between the lanes but two LDS round trips:
// lane (l, u) = (tl / 8, tl % 8) of a team factorises the block of leg l (redundantly with the 7 other lanes of the leg), forms row u of W_l and of
// W_l W_l^T, the four legs are added across the lanes ((l0 + l2) + (l1 + l3): v_permlane16_swap, row_ror:8), and every lane factorises the 6 x 6
// Schur complement in registers.  Only reciprocal pivots are kept (1 / sqrt(e) = sqrt(e) * (1 / e): the division is issued beside the square root).
// The triangular solves (arrow_solve) walk the same structure: 3 + 6 + 6 + 3 dependent steps instead of 36, in registers.
// Layout of the factor F (176 floats, 16-byte aligned; may lie over the matrix it is computed from, which is consumed first):
//   leg l at 32 l:  [0..2] reciprocal pivots, [4..6] l10 l20 l21, [8 + 4 b .. +2] row b of W_l (b = 0..5);   base at 128 + 8 k: row k of L_b, the
//   reciprocal pivot in place of the diagonal element.
// Users: the constraint solver (ts_solve: a solve none of whose contacts joins links of two different legs -- found while the rows are built from the link
// chains a row walks up; otherwise ts_cholesky_factor_rows / the row-form solves) and the forward dynamics (tk_dynamics: acc_smooth = M^-1 force instead of
// the reverse LDL^T of the strict build).  The FAST ORDER oracle mirrors the arithmetic operation for operation (arrow_factor / arrow_solve in
// oracle/go2sim_cpu.cpp).
constexpr bool ARROW_SHAPE = (ND == 18);
constexpr bool ARROW_SOLVER = ARROW_SHAPE && (REBUILD_FLIPS <= 1);    // (rank-1 updates of the Newton factor exist for the row form only)
DEV float leg_sum4(float x) {                                          // (x_l + x_(l^2)) + (x_(l^1) + x_(l^3)) over the four 8-lane groups of 32 lanes
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float p = __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lane ^ 16
  return p + dpp_perm<0x128>(p);                                       // row_ror:8 = lane ^ 8 inside a row of 16
}
DEV float4 lds4(const float* p) { return *(const float4*)p; }
// A: lower triangle of the matrix, row stride STRIDE floats
template <int T, int STRIDE>
DEV void arrow_factor(const int mode, const float eps, const float* A, float* F, int tl) {
  static_assert(T == 32 || T == 64, "four groups of eight lanes");
  const int leg = (tl >> 3) & 3, u = tl & 7, ub = u < 6 ? u : 5;
  const int p0 = dm_arrow_dof(mode, leg, 0), p1 = dm_arrow_dof(mode, leg, 1), p2 = dm_arrow_dof(mode, leg, 2);
  const bool w_lane = tl < 32 && u < 6;
  // every input first (one round trip): the lane's leg block, its element of the three coupling rows, its row of the base block
  const float a00 = A[p0 * STRIDE + p0], a10 = A[p1 * STRIDE + p0], a11 = A[p1 * STRIDE + p1];
  const float a20 = A[p2 * STRIDE + p0], a21 = A[p2 * STRIDE + p1], a22 = A[p2 * STRIDE + p2];
  const float c0 = A[p0 * STRIDE + ub], c1 = A[p1 * STRIDE + ub], c2 = A[p2 * STRIDE + ub];
  float br[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) br[j] = A[ub * STRIDE + j];              // (entries right of the diagonal are don't-care values)
  team_sync();                                                         // the factor may be written over the matrix
  const float e0 = fmx(a00, eps), i0 = dm_sqrt(e0) * (1.0f / e0);
  const float l10 = a10 * i0, l20 = a20 * i0;
  const float e1 = fmx(__builtin_fmaf(-l10, l10, a11), eps), i1 = dm_sqrt(e1) * (1.0f / e1);
  const float l21 = __builtin_fmaf(-l20, l10, a21) * i1;
  const float e2 = fmx(__builtin_fmaf(-l21, l21, __builtin_fmaf(-l20, l20, a22)), eps), i2 = dm_sqrt(e2) * (1.0f / e2);
  const float w0 = c0 * i0, w1 = __builtin_fmaf(-w0, l10, c1) * i1, w2 = __builtin_fmaf(-w1, l21, __builtin_fmaf(-w0, l20, c2)) * i2;
  if (w_lane) *(float4*)&F[32 * leg + 8 + 4 * u] = make_float4(w0, w1, w2, 0.0f);
  if (tl < 32 && u == 0) { *(float4*)&F[32 * leg] = make_float4(i0, i1, i2, 0.0f); *(float4*)&F[32 * leg + 4] = make_float4(l10, l20, l21, 0.0f); }
  team_sync();
  float sc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float4 wj = lds4(&F[32 * leg + 8 + 4 * j]);
    sc[j] = __builtin_fmaf(w2, wj.z, __builtin_fmaf(w1, wj.y, w0 * wj.x));
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) sc[j] = leg_sum4(sc[j]);
  const float bp[6] = {br[0] - sc[0], br[1] - sc[1], br[2] - sc[2], br[3] - sc[3], br[4] - sc[4], br[5] - sc[5]};
  if (tl < 6) { *(float4*)&F[128 + 8 * tl] = make_float4(bp[0], bp[1], bp[2], bp[3]); *(float4*)&F[128 + 8 * tl + 4] = make_float4(bp[4], bp[5], 0.0f, 0.0f); }
  team_sync();
  float a[6][6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float4 lo = lds4(&F[128 + 8 * k]); const float2 hi = *(const float2*)&F[128 + 8 * k + 4];
    a[k][0] = lo.x; a[k][1] = lo.y; a[k][2] = lo.z; a[k][3] = lo.w; a[k][4] = hi.x; a[k][5] = hi.y;
  }
  team_sync();                                                         // (rows are rewritten below)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float e = fmx(a[k][k], eps), ik = dm_sqrt(e) * (1.0f / e);
    a[k][k] = ik;
#pragma unroll
    for (int j = k + 1; j < 6; ++j) a[j][k] = a[j][k] * ik;
#pragma unroll
    for (int j = k + 1; j < 6; ++j)
#pragma unroll
      for (int i = k + 1; i <= j; ++i) a[j][i] = __builtin_fmaf(-a[j][k], a[i][k], a[j][i]);
  }
  if (tl == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      *(float4*)&F[128 + 8 * k] = make_float4(a[k][0], k >= 1 ? a[k][1] : 0.0f, k >= 2 ? a[k][2] : 0.0f, k >= 3 ? a[k][3] : 0.0f);
      *(float2*)&F[128 + 8 * k + 4] = make_float2(k >= 4 ? a[k][4] : 0.0f, k >= 5 ? a[k][5] : 0.0f);
    }
  }
  team_sync();
}
// x = (L L^T)^-1 g on the arrow factor F.  The caller hands in the lane's right-hand-side elements (its leg's three dofs, base dof min(tl % 8, 5)); `x6` is a
// 6-float exchange buffer (8-byte aligned); out: the base part in xb[0..5] of every lane, the lane's leg part in xl0..2
template <int T>
DEV void arrow_solve(const float* F, float gl0, float gl1, float gl2, float gb, float* x6, int tl, float (&xb)[6], float& xl0, float& xl1, float& xl2) {
  static_assert(T == 32 || T == 64, "four groups of eight lanes");
  const int leg = (tl >> 3) & 3, u = tl & 7, ub = u < 6 ? u : 5;
  const float4 iv = lds4(&F[32 * leg]), lv = lds4(&F[32 * leg + 4]), wo = lds4(&F[32 * leg + 8 + 4 * ub]);
  float4 W[6];
#pragma unroll
  for (int b = 0; b < 6; ++b) W[b] = lds4(&F[32 * leg + 8 + 4 * b]);
  float L[6][6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float4 lo = lds4(&F[128 + 8 * k]); const float2 hi = *(const float2*)&F[128 + 8 * k + 4];
    L[k][0] = lo.x; L[k][1] = lo.y; L[k][2] = lo.z; L[k][3] = lo.w; L[k][4] = hi.x; L[k][5] = hi.y;
  }
  // forward: the legs, then the base with the legs' part taken out of its right-hand side
  const float y0 = gl0 * iv.x, y1 = __builtin_fmaf(-lv.x, y0, gl1) * iv.y, y2 = __builtin_fmaf(-lv.z, y1, __builtin_fmaf(-lv.y, y0, gl2)) * iv.z;
  const float z = leg_sum4(__builtin_fmaf(wo.z, y2, __builtin_fmaf(wo.y, y1, wo.x * y0)));
  if (tl < 6) x6[tl] = gb - z;
  team_sync();
  const float2 r0 = *(const float2*)&x6[0], r1 = *(const float2*)&x6[2], r2 = *(const float2*)&x6[4];
  float x[6] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
  team_sync();                                                         // (the caller may write its result over the exchange buffer)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float acc = x[k];
#pragma unroll
    for (int j = 0; j < k; ++j) acc = __builtin_fmaf(-L[k][j], x[j], acc);
    x[k] = acc * L[k][k];
  }
#pragma unroll
  for (int k_ = 0; k_ < 6; ++k_) {
    const int k = 5 - k_;
    float acc = x[k];
#pragma unroll
    for (int j_ = 0; j_ < 5 - k; ++j_) { const int j = 5 - j_; acc = __builtin_fmaf(-L[j][k], x[j], acc); }
    x[k] = acc * L[k][k];
  }
  // backward through the legs
  float v0 = y0, v1 = y1, v2 = y2;
#pragma unroll
  for (int b = 0; b < 6; ++b) { v0 = __builtin_fmaf(-W[b].x, x[b], v0); v1 = __builtin_fmaf(-W[b].y, x[b], v1); v2 = __builtin_fmaf(-W[b].z, x[b], v2); }
  xl2 = v2 * iv.z; xl1 = __builtin_fmaf(-lv.z, xl2, v1) * iv.y; xl0 = __builtin_fmaf(-lv.x, xl1, __builtin_fmaf(-lv.y, xl2, v0)) * iv.x;
#pragma unroll
  for (int k = 0; k < 6; ++k) xb[k] = x[k];
}
#endif
struct DynData {
  float cinr_I[NL * 9], cinr_pos[NL * 3], cinr_mass[NL];
  float crb_I[NL * 9], crb_pos[NL * 3], crb_mass[NL];
  float cdof_ang[ND * 3], cdof_vel[ND * 3], cdofd_ang[ND * 3], cdofd_vel[ND * 3];
  float cd_vel[NL * 3], cd_ang[NL * 3], cdd_vel[NL * 3], cdd_ang[NL * 3], cfrc_vel[NL * 3], cfrc_ang[NL * 3];
  float f_ang[ND * 3], f_vel[ND * 3];
  float vel[ND], qf_applied[ND], qf_passive[ND], force[ND], out[ND], Dinv[ND];
  float M[ND * ND], L[ND * ND];
  int ctrl_mode[ND];
};

template <int T>
DEV void tk_stage_links(const E& e, KinData* s, int tl) {
  team_for<NL, T>(tl, [&](int i_l) { st3(s->l_pos, i_l, e.l_pos()[i_l]); st4(s->l_quat, i_l, e.l_quat()[i_l]); });
}
// update_cartesian_space + forward_velocity of the state held in s->qpos / s->vel
// (func_forward_kinematics_entity :463-618, func_COM_links_entity :224-459, func_update_geoms_entity :709-744,
//  func_forward_velocity_entity :871-994 of forward_kinematics.py)
// `dk` (optional): the working set of the forward dynamics that follow in the same kernel (k_integrate_fk_dynamics_team): everything the dynamics would
// otherwise read back from HBM is also left there
// (with `dk`, i.e. between the substeps of an env step, the outputs that only the forward dynamics reads -- cinr_*, cdofd_*, cd_vel / cd_ang, i_pos --
//  are not written to HBM at all: the dynamics half of the same kernel takes them from LDS, and the FK that closes the step writes them all)
template <int T, class MT>
DEV void tk_kinematics(const MT& m, const E& e, KinData* s, int tl, bool force_update_fixed, DynData* dk = nullptr) {
  // s->l_pos / s->l_quat hold the current link poses (tk_stage_links, issued with the kernel's other staging loads)
  PH_BEGIN                                                             // (profiling builds: 18 = link poses by level, 19 = COM / inertia / cdof / geoms, 55 = velocities by level)
  for (int lev = 0; lev < m.n_levels; ++lev) {
    for (int k = m.level_start[lev] + tl; k < m.level_start[lev + 1]; k += T) {
      int i_l = m.level_links[k];
      const auto& L = m.links[i_l];
      V3 pos = L.pos; Q4 quat = L.quat;
      if (L.parent != -1) {
        Q4 pq = ld4(s->l_quat, L.parent);
        pos = ld3(s->l_pos, L.parent) + transform_by_quat(L.pos, pq);
        quat = transform_quat_by_quat(L.quat, pq);
      }
      for (int i_j = L.joint_start; i_j < L.joint_end; ++i_j) {
        const Joint& J = m.joints[i_j];
        int q_start = J.q_start, dof_start = J.dof_start;
        if (J.type == JOINT_FREE) {
          V3 pos_ = v3(s->qpos[q_start], s->qpos[q_start + 1], s->qpos[q_start + 2]);
          st3(s->xanchor, i_j, pos_);
          st3(s->xaxis, i_j, v3(0, 0, 1));
          Q4 quat_ = q4(s->qpos[q_start + 3], s->qpos[q_start + 4], s->qpos[q_start + 5], s->qpos[q_start + 6]);
          float n = dm_sqrt(norm_sqr(quat_));
          quat_ = q4(quat_.w / n, quat_.x / n, quat_.y / n, quat_.z / n);
          pos = pos_; quat = quat_;
          gstore(e, FO(dof_pos), dof_start + 0, pos.x); gstore(e, FO(dof_pos), dof_start + 1, pos.y); gstore(e, FO(dof_pos), dof_start + 2, pos.z);
          if (dk) { dk->out[dof_start] = pos.x; dk->out[dof_start + 1] = pos.y; dk->out[dof_start + 2] = pos.z; }
        } else if (J.type == JOINT_REVOLUTE) {
          V3 axis = m.dofs[dof_start].motion_ang;
          V3 anchor = transform_by_quat(J.pos, quat) + pos;
          st3(s->xanchor, i_j, anchor);
          st3(s->xaxis, i_j, transform_by_quat(axis, quat));
          float dp = s->qpos[q_start] - m.qpos0[q_start];
          gstore(e, FO(dof_pos), dof_start, dp);
          if (dk) dk->out[dof_start] = dp;
          Q4 qloc = rotvec_to_quat(axis * dp, m.eps);
          quat = transform_quat_by_quat(qloc, quat);
          pos = anchor - transform_by_quat(J.pos, quat);
        }
      }
      if (!(L.parent == -1 && L.is_fixed)) { st3(s->l_pos, i_l, pos); st4(s->l_quat, i_l, quat); e.l_pos()[i_l] = pos; e.l_quat()[i_l] = quat; }
    }
    team_sync();
  }
  PH(18)
  // centre of mass of each kinematic tree
  for (int i_l = tl; i_l < NL; i_l += T) {
    const auto& L = m.links[i_l];
    s->mass[i_l] = L.mass + gload(e, FO(mass_shift), i_l);
    V3 ipbw; Q4 iq;
    transform_pos_quat_by_trans_quat(L.inertial_pos + (V3)e.com_shift()[i_l], L.inertial_quat, ld3(s->l_pos, i_l), ld4(s->l_quat, i_l), ipbw, iq);
    st3(s->i_pos, i_l, ipbw); st4(s->i_quat, i_l, iq);
  }
  team_sync();
  V3 rc[2];
#pragma unroll
  for (int i_e = 0; i_e < 2; ++i_e) {
    const Entity& en = m.entities[i_e];
    V3 root_com_bw = v3(0, 0, 0); float mass_sum = 0.0f;
    for (int i_l = en.link_start; i_l < en.link_end; ++i_l) {
      float mass = s->mass[i_l];
      mass_sum = mass_sum + mass;
      root_com_bw = root_com_bw + mass * ld3(s->i_pos, i_l);
    }
    rc[i_e] = root_com_bw / mass_sum;
  }
  team_sync();
  for (int i_l = tl; i_l < NL; i_l += T) {
    const auto& L = m.links[i_l];
    V3 r = (L.entity == 0) ? rc[0] : rc[1];
    e.root_com()[i_l] = r;
    V3 ip = ld3(s->i_pos, i_l) - r;
    Q4 iq = ld4(s->i_quat, i_l);
    if (!dk) e.i_pos()[i_l] = ip;
    e.i_quat()[i_l] = iq;
    float i_mass = s->mass[i_l];
    M3 oI; V3 op;
    transform_inertia_by_trans_quat(L.inertial_i, i_mass, ip, iq, m.eps, oI, op);
    if (!dk) { e.cinr_inertial()[i_l] = oI; e.cinr_pos()[i_l] = op; e.cinr_mass()[i_l] = i_mass; }
    if (dk) { st9(dk->cinr_I, i_l, oI); st9(dk->crb_I, i_l, oI); st3(dk->cinr_pos, i_l, op); st3(dk->crb_pos, i_l, op); dk->cinr_mass[i_l] = i_mass; dk->crb_mass[i_l] = i_mass; }
  }
  for (int i_j = tl; i_j < NJ; i_j += T) {
    const Joint& J = m.joints[i_j];
    const auto& L = m.links[J.link];
    if (L.n_dofs == 0) continue;
    V3 r = (L.entity == 0) ? rc[0] : rc[1];
    V3 offset_pos = r - ld3(s->xanchor, i_j);
    int ds = J.dof_start;
    if (J.type == JOINT_REVOLUTE) {
      V3 ax = ld3(s->xaxis, i_j);
      V3 cv = cross(ax, offset_pos);
      st3(s->cdof_ang, ds, ax); st3(s->cdof_vel, ds, cv);
      e.cdof_ang()[ds] = ax; e.cdof_vel()[ds] = cv;
      if (dk) { st3(dk->cdof_ang, ds, ax); st3(dk->cdof_vel, ds, cv); }
    } else if (J.type == JOINT_FREE) {
      for (int i = 0; i < 3; ++i) {
        V3 cv = v3(0, 0, 0);
        vset(cv, i, 1.0f);
        st3(s->cdof_ang, i + ds, v3(0, 0, 0)); st3(s->cdof_vel, i + ds, cv);
        e.cdof_ang()[i + ds] = v3(0, 0, 0); e.cdof_vel()[i + ds] = cv;
        if (dk) { st3(dk->cdof_ang, i + ds, v3(0, 0, 0)); st3(dk->cdof_vel, i + ds, cv); }
      }
      M3 xmat_T = transpose(quat_to_R(ld4(s->l_quat, J.link), m.eps));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        V3 row = v3(xmat_T.m[i][0], xmat_T.m[i][1], xmat_T.m[i][2]);
        V3 cv = cross(row, offset_pos);
        st3(s->cdof_ang, i + ds + 3, row); st3(s->cdof_vel, i + ds + 3, cv);
        e.cdof_ang()[i + ds + 3] = row; e.cdof_vel()[i + ds + 3] = cv;
        if (dk) { st3(dk->cdof_ang, i + ds + 3, row); st3(dk->cdof_vel, i + ds + 3, cv); }
      }
    }
  }
  for (int i_g = tl; i_g < NG; i_g += T) {
    const auto& G = m.geoms[i_g];
    bool is_fixed = m.links[G.link].is_fixed;
    if (force_update_fixed || !is_fixed) {
      V3 p; Q4 q;
      transform_pos_quat_by_trans_quat(G.pos, G.quat, ld3(s->l_pos, G.link), ld4(s->l_quat, G.link), p, q);
      e.g_pos()[i_g] = p; e.g_quat()[i_g] = q;
    }
  }
  team_sync();
  PH(19)
  // forward velocity
  for (int lev = 0; lev < m.n_levels; ++lev) {
    for (int k = m.level_start[lev] + tl; k < m.level_start[lev + 1]; k += T) {
      int i_l = m.level_links[k];
      const auto& L = m.links[i_l];
      V3 cvel_vel = v3(0, 0, 0), cvel_ang = v3(0, 0, 0);
      if (L.parent != -1) { cvel_vel = ld3(s->cd_vel, L.parent); cvel_ang = ld3(s->cd_ang, L.parent); }
      for (int i_j = L.joint_start; i_j < L.joint_end; ++i_j) {
        const Joint& J = m.joints[i_j];
        int ds = J.dof_start;
        if (J.type == JOINT_FREE) {
          for (int i = 0; i < 3; ++i) {
            float v = s->vel[ds + i];
            cvel_vel = cvel_vel + ld3(s->cdof_vel, ds + i) * v;
            cvel_ang = cvel_ang + ld3(s->cdof_ang, ds + i) * v;
          }
          for (int i = 0; i < 3; ++i) {
            if (!dk) { e.cdofd_ang()[ds + i] = v3(0, 0, 0); e.cdofd_vel()[ds + i] = v3(0, 0, 0); }
            if (dk) { st3(dk->cdofd_ang, ds + i, v3(0, 0, 0)); st3(dk->cdofd_vel, ds + i, v3(0, 0, 0)); }
            V3 oa, ov;
            motion_cross_motion(cvel_ang, cvel_vel, ld3(s->cdof_ang, ds + i + 3), ld3(s->cdof_vel, ds + i + 3), oa, ov);
            if (!dk) { e.cdofd_ang()[ds + i + 3] = oa; e.cdofd_vel()[ds + i + 3] = ov; }
            if (dk) { st3(dk->cdofd_ang, ds + i + 3, oa); st3(dk->cdofd_vel, ds + i + 3, ov); }
          }
          for (int i = 0; i < 3; ++i) {
            float v = s->vel[ds + i + 3];
            cvel_vel = cvel_vel + ld3(s->cdof_vel, ds + i + 3) * v;
            cvel_ang = cvel_ang + ld3(s->cdof_ang, ds + i + 3) * v;
          }
        } else {
          for (int i_d = ds; i_d < J.dof_end; ++i_d) {
            V3 oa, ov;
            motion_cross_motion(cvel_ang, cvel_vel, ld3(s->cdof_ang, i_d), ld3(s->cdof_vel, i_d), oa, ov);
            if (!dk) { e.cdofd_ang()[i_d] = oa; e.cdofd_vel()[i_d] = ov; }
            if (dk) { st3(dk->cdofd_ang, i_d, oa); st3(dk->cdofd_vel, i_d, ov); }
          }
          for (int i_d = ds; i_d < J.dof_end; ++i_d) {
            float v = s->vel[i_d];
            cvel_vel = cvel_vel + ld3(s->cdof_vel, i_d) * v;
            cvel_ang = cvel_ang + ld3(s->cdof_ang, i_d) * v;
          }
        }
      }
      st3(s->cd_vel, i_l, cvel_vel); st3(s->cd_ang, i_l, cvel_ang);
      if (!dk) { e.cd_vel()[i_l] = cvel_vel; e.cd_ang()[i_l] = cvel_ang; }
      if (dk) { st3(dk->cd_vel, i_l, cvel_vel); st3(dk->cd_ang, i_l, cvel_ang); }
    }
    team_sync();
  }
  PH(55)
}

// func_integrate (forward_dynamics.py:1558-1699) + func_copy_next_to_curr (abd/diff.py:25-54) on the staged state: s->vel_next holds v + a dt
template <int T, class MT>
DEV void tk_integrate(const MT& m, const E& e, KinData* s, int tl) {
  for (int i_l = tl; i_l < NL; i_l += T) {
    const auto& L = m.links[i_l];
    if (L.n_dofs == 0) continue;
    int ds = L.dof_start, qs = L.q_start;
    int joint_type = m.joints[L.joint_start].type;
    if (joint_type == JOINT_FREE) {
      V3 pos = v3(s->qpos[qs], s->qpos[qs + 1], s->qpos[qs + 2]);
      V3 v = v3(s->vel_next[ds], s->vel_next[ds + 1], s->vel_next[ds + 2]);
      pos = pos + v * m.substep_dt;
      s->qpos_next[qs] = pos.x; s->qpos_next[qs + 1] = pos.y; s->qpos_next[qs + 2] = pos.z;
      Q4 rot0 = q4(s->qpos[qs + 3], s->qpos[qs + 4], s->qpos[qs + 5], s->qpos[qs + 6]);
      V3 ang = v3(s->vel_next[ds + 3], s->vel_next[ds + 4], s->vel_next[ds + 5]) * m.substep_dt;
      Q4 qrot = rotvec_to_quat(ang, m.eps);
      Q4 rot = transform_quat_by_quat(qrot, rot0);
      s->qpos_next[qs + 3] = rot.w; s->qpos_next[qs + 4] = rot.x; s->qpos_next[qs + 5] = rot.y; s->qpos_next[qs + 6] = rot.z;
    } else {
      for (int j_ = 0; j_ < L.q_end - qs; ++j_) s->qpos_next[qs + j_] = s->qpos[qs + j_] + s->vel_next[ds + j_] * m.substep_dt;
    }
  }
  team_sync();
  bool bad = false;
  for (int d = tl; d < ND; d += T) bad |= isnan_(s->vel_next[d]);
  for (int q = tl; q < NQ; q += T) bad |= isnan_(s->qpos_next[q]);
  if (bad) s->valid = 0;
  team_sync();
  if (s->valid) {
    for (int d = tl; d < ND; d += T) { float v = s->vel_next[d]; s->vel[d] = v; gstore(e, FO(vel), d, v); }
    for (int q = tl; q < NQ; q += T) { float v = s->qpos_next[q]; s->qpos[q] = v; gstore(e, FO(qpos), q, v); }
  } else if (tl == 0) {
    atomicOr(&e.err()[0], GO2SIM_ERR_INVALID_ACC_NAN);
  }
  team_sync();
}

// kernel_step_2 (rigid_solver.py:3072-3180): func_integrate (forward_dynamics.py:1558-1699) + func_copy_next_to_curr
// (abd/diff.py:25-54) + FK / forward velocity of the new state
// `pre` (k_solve_integrate_team): lane d < ND of the team already holds velocity and acceleration of dof d (taken from the solver's LDS block before the
// overlay), so neither is fetched from HBM again
template <int T>
DEV void integrate_fk_body(const Pool& P, const ModelS* __restrict__ mp, int b, KinData* lds, char* ms_raw, bool pre, float pre_vel, float pre_acc) {
  const ModelS& ms = *(const ModelS*)ms_raw;
  wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, mp);
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const ModelView m(&ms, mp);
  E e(P, b < P.B ? b : P.B - 1);
  KinData* s = &lds[slot];
  PH_BEGIN
  if (pre) { if (tl < ND) { s->vel[tl] = pre_vel; s->vel_next[tl] = pre_acc; } }
  else {
    team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(vel), d); }, [&](int d, float v) { s->vel[d] = v; });
    team_stage<ND, T>(tl, [&](int d) { return aload(e, AO(acc), d); }, [&](int d, float a) { s->vel_next[d] = a; });
  }
  team_stage<NQ, T>(tl, [&](int q) { return gload(e, FO(qpos), q); }, [&](int q, float v) { s->qpos[q] = v; });
  team_for<ND, T>(tl, [&](int d) { s->vel_next[d] = s->vel[d] + s->vel_next[d] * m.substep_dt; });
  if (tl == 0) s->valid = 1;
  tk_stage_links<T>(e, s, tl);
  team_sync();
  if (b >= P.B) return;
  tk_integrate<T>(m, e, s, tl);
  PH(40)
  tk_kinematics<T>(m, e, s, tl, false);
  PH(41)
}
template <int T>
__global__ __launch_bounds__(64) void k_integrate_fk_team(Pool P, const ModelS* __restrict__ mp) {
  STAMP(STK_INT_FK)
  constexpr int EPW = 64 / T;
  __shared__ KinData lds[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  integrate_fk_body<T>(P, mp, xcd_block() * EPW + (int)threadIdx.x / T, lds, ms_raw, false, 0.0f, 0.0f);
}

// FK refresh of the current state; `cond` (device) gates the launch body: the reset path only needs it when an env was reset
template <int T>
__global__ __launch_bounds__(64) void k_fk_team(Pool P, const ModelS* __restrict__ mp, int force_update_fixed, const int* __restrict__ cond) {
  constexpr int EPW = 64 / T;
  __shared__ KinData lds[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  const ModelS& ms = *(const ModelS*)ms_raw;
  if (cond && *cond <= 0) return;
  wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, mp);
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const int b = xcd_block() * EPW + slot;
  const ModelView m(&ms, mp);
  E e(P, b < P.B ? b : P.B - 1);
  KinData* s = &lds[slot];
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(vel), d); }, [&](int d, float v) { s->vel[d] = v; });
  team_stage<NQ, T>(tl, [&](int q) { return gload(e, FO(qpos), q); }, [&](int q, float v) { s->qpos[q] = v; });
  tk_stage_links<T>(e, s, tl);
  team_sync();
  if (b >= P.B) return;
  tk_kinematics<T>(m, e, s, tl, force_update_fixed != 0);
}

// kernel_step_1 without the (already fresh) FK, rigid_solver.py:3008-3069: func_compute_mass_matrix (forward_dynamics.py:291-541),
// func_factor_mass :560-604, func_torque_and_passive_force :961-1174, func_update_acc/force/bias_force :1177-1478,
// func_solve_mass_entity :818-900
// (the computation proper; the inputs are in the LDS working set `s`: staged from HBM by k_dynamics_team, or left there by the kinematics of
//  the same kernel in k_integrate_fk_dynamics_team)
template <int T, class MT>
DEV void tk_dynamics(const MT& m, const E& e, DynData* s, int tl, bool env_valid) {
  PH_BEGIN
  // ---- composite rigid bodies, leaf -> root ----
  for (int lev = m.n_levels - 2; lev >= 0; --lev) {
    int n_par = m.level_start[lev + 1] - m.level_start[lev];
    for (int w = tl; w < n_par * 13; w += T) {
      int i_p = m.level_links[m.level_start[lev] + w / 13], comp = w % 13;
      float* dst = (comp < 9) ? &s->crb_I[9 * i_p + comp] : ((comp < 12) ? &s->crb_pos[3 * i_p + comp - 9] : &s->crb_mass[i_p]);
      float a = *dst;
      for (int c = m.child_start[i_p]; c < m.child_start[i_p + 1]; ++c) {
        int i_l = m.child_list[c];
        float v = (comp < 9) ? s->crb_I[9 * i_l + comp] : ((comp < 12) ? s->crb_pos[3 * i_l + comp - 9] : s->crb_mass[i_l]);
        a = a + v;
      }
      *dst = a;
    }
    team_sync();
  }
  PH(21)
  for (int i_d = tl; i_d < ND; i_d += T) {
    int i_l = m.dof_link[i_d];
    V3 oa, ov;
    inertial_mul(ld3(s->crb_pos, i_l), ld9(s->crb_I, i_l), s->crb_mass[i_l], ld3(s->cdof_vel, i_d), ld3(s->cdof_ang, i_d), oa, ov);
    st3(s->f_ang, i_d, oa); st3(s->f_vel, i_d, ov);
  }
  team_sync();
  for (int idx = tl; idx < ND * (ND + 1) / 2; idx += T) {
    int i_d, j_d;
    tri_index(m, idx, i_d, j_d);
    float v = (dot(ld3(s->f_ang, i_d), ld3(s->cdof_ang, j_d)) + dot(ld3(s->f_vel, i_d), ld3(s->cdof_vel, j_d))) * mass_mask(m, i_d, j_d);
    if (i_d == j_d) {
      v = v + m.dofs[i_d].armature;
      v = v + m.dofs[i_d].damping * m.substep_dt;                      // implicit damping (approximate_implicitfast)
      int cm = s->ctrl_mode[i_d];
      if (cm == CTRL_POSITION || cm == CTRL_VELOCITY) v = v + m.dofs[i_d].kv * m.substep_dt;
    }
    s->M[i_d * ND + j_d] = v; s->M[j_d * ND + i_d] = v;
    s->L[i_d * ND + j_d] = v;
  }
  team_sync();
  PH(22)
  // ---- reverse-order LDL^T (func_factor_mass, forward_dynamics.py) ----
#if GO2SIM_FAST_ORDER
  // FAST ORDER: the mass matrix of a floating base with four legs has the arrow shape; acc_smooth = M^-1 force comes from its arrow-form Cholesky
  // factorisation (3 + 6 dependent pivots in registers instead of 18 published rows; below: 18 dependent substitution steps instead of 36)
  const bool arrow = ARROW_SHAPE && (T == 32 || T == 64) && m.arrow_mode != 0;
  float* const AF = (float*)(((uintptr_t)s->L + 15) & ~(uintptr_t)15);  // 176 floats of the 324 of s->L, 16-byte aligned
#else
  const bool arrow = false;
#endif
  if constexpr (T == 32 || T == 64) {
#if GO2SIM_FAST_ORDER
    if (arrow) arrow_factor<T, ND>(m.arrow_mode, m.eps, s->M, AF, tl);
#endif
  }
  if (arrow) {
  } else if constexpr (T >= ND) {
    // Lane j keeps row j of the factor in registers.  Step i (i = ND-1 .. 0): lane i publishes its (final, unscaled) row in its LDS home, every
    // row j < i reads it and does L[j][k] -= (L[i][j] * D_inv) * L[i][k] for k <= j -- the same operands in the same order as the element loop of
    // the reference, one LDS round trip and one fence per step.  Row i's own scaling by D_inv touches nothing a later step reads, so every
    // lane applies it to its registers at the end.  (Register entries right of the diagonal are don't-care values: they are never published.)
    const int row = tl < ND ? tl : ND - 1;
    float Lr[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) Lr[k] = s->L[row * ND + k];
    float dinv_own = 0.0f;
#pragma unroll
    for (int i_d_ = 0; i_d_ < ND; ++i_d_) {
      const int i_d = ND - i_d_ - 1;
      if (i_d_ > 0) {                                                     // (the last row has not changed: its LDS copy is current)
        if (tl == i_d) {
#pragma unroll
          for (int k = 0; k <= i_d; ++k) s->L[i_d * ND + k] = Lr[k];
        }
        team_sync();
      }
      float ri[ND];
#pragma unroll
      for (int k = 0; k <= i_d; ++k) ri[k] = s->L[i_d * ND + k];
      const float lij = s->L[i_d * ND + (row < i_d ? row : 0)];
      const float D_inv = 1.0f / ri[i_d];
      if (row == i_d) dinv_own = D_inv;
      const bool act = row < i_d;
      const float a = lij * D_inv;
#pragma unroll
      for (int k = 0; k < i_d; ++k) { const float nv = Lr[k] - a * ri[k]; Lr[k] = act ? nv : Lr[k]; }
    }
    team_sync();
    if (tl < ND) {
#pragma unroll
      for (int k = 0; k < ND; ++k) if (k < row) s->L[row * ND + k] = Lr[k] * dinv_own;
      s->L[row * ND + row] = 1.0f;
      s->Dinv[row] = dinv_own;
    }
    team_sync();
  } else {
#pragma unroll
    for (int i_d_ = 0; i_d_ < ND; ++i_d_) {
      const int i_d = ND - i_d_ - 1;
      float D_inv = 1.0f / s->L[i_d * ND + i_d];
      for (int idx = tl; idx < i_d * (i_d + 1) / 2; idx += T) {
        int j_d, k_d;
        tri_index(m, idx, j_d, k_d);
        float a = s->L[i_d * ND + j_d] * D_inv;
        s->L[j_d * ND + k_d] -= a * s->L[i_d * ND + k_d];
      }
      team_sync();
      for (int j_d = tl; j_d < i_d; j_d += T) s->L[i_d * ND + j_d] = s->L[i_d * ND + j_d] * D_inv;
      if (tl == 0) { s->Dinv[i_d] = D_inv; s->L[i_d * ND + i_d] = 1.0f; }
      team_sync();
    }
  }
  PH(23)
  // ---- applied / passive joint forces ----
  for (int i_d = tl; i_d < ND; i_d += T) {
    const Dof& D = m.dofs[i_d];
    const auto& L = m.links[m.dof_link[i_d]];
    int joint_type = m.joints[L.joint_start].type;
    float force = 0.0f;
    int cm = s->ctrl_mode[i_d];
    const float in_ctrl_force = s->qf_applied[i_d], in_ctrl_pos = s->qf_passive[i_d], in_ctrl_vel = s->force[i_d], in_dof_pos = s->out[i_d];   // parked by the staging
    if (cm == CTRL_FORCE) force = in_ctrl_force;
    else if (cm == CTRL_VELOCITY) force = D.kv * (in_ctrl_vel - s->vel[i_d]);
    else if (cm == CTRL_POSITION && !(joint_type == JOINT_FREE && i_d >= L.dof_start + 3))
      force = D.kp * (in_ctrl_pos - in_dof_pos) + D.kv * (in_ctrl_vel - s->vel[i_d]);
    s->qf_applied[i_d] = clampf(force, D.force_range[0], D.force_range[1]);
    float qp = -D.damping * s->vel[i_d];
    if (joint_type != JOINT_FREE && joint_type != JOINT_FIXED) qp = qp + (-in_dof_pos * D.stiffness);
    s->qf_passive[i_d] = qp;
  }
  // ---- bias forces: accelerations root -> leaf ----
  for (int lev = 0; lev < m.n_levels; ++lev) {
    for (int k = m.level_start[lev] + tl; k < m.level_start[lev + 1]; k += T) {
      int i_l = m.level_links[k];
      const auto& L = m.links[i_l];
      V3 cv, ca;
      if (L.parent == -1) { cv = -m.gravity * (1.0f - 0.0f); ca = v3(0, 0, 0); }
      else { cv = ld3(s->cdd_vel, L.parent); ca = ld3(s->cdd_ang, L.parent); }
      for (int i_d = L.dof_start; i_d < L.dof_end; ++i_d) {
        float v = s->vel[i_d];
        V3 local_cdd_vel = ld3(s->cdofd_vel, i_d) * v;
        V3 local_cdd_ang = ld3(s->cdofd_ang, i_d) * v;
        cv = cv + local_cdd_vel;
        ca = ca + local_cdd_ang;
      }
      st3(s->cdd_vel, i_l, cv); st3(s->cdd_ang, i_l, ca);
    }
    team_sync();
  }
  for (int i_l = tl; i_l < NL; i_l += T) {
    V3 f1_ang, f1_vel, f2_ang, f2_vel, f3_ang, f3_vel;
    M3 I = ld9(s->cinr_I, i_l); V3 cp = ld3(s->cinr_pos, i_l); float cm = s->cinr_mass[i_l];
    V3 cdv = ld3(s->cd_vel, i_l), cda = ld3(s->cd_ang, i_l);
    inertial_mul(cp, I, cm, ld3(s->cdd_vel, i_l), ld3(s->cdd_ang, i_l), f1_ang, f1_vel);
    inertial_mul(cp, I, cm, cdv, cda, f2_ang, f2_vel);
    motion_cross_force(cda, cdv, f2_ang, f2_vel, f3_ang, f3_vel);
    V3 ext_ang = ld3(s->cfrc_ang, i_l), ext_vel = ld3(s->cfrc_vel, i_l);   // parked by the staging
    st3(s->cfrc_vel, i_l, f1_vel + f3_vel + ext_vel + v3(0, 0, 0));
    st3(s->cfrc_ang, i_l, f1_ang + f3_ang + ext_ang + v3(0, 0, 0));
  }
  team_sync();
  for (int lev = m.n_levels - 2; lev >= 0; --lev) {
    int n_par = m.level_start[lev + 1] - m.level_start[lev];
    for (int w = tl; w < n_par * 6; w += T) {
      int i_p = m.level_links[m.level_start[lev] + w / 6], comp = w % 6;
      float* dst = (comp < 3) ? &s->cfrc_vel[3 * i_p + comp] : &s->cfrc_ang[3 * i_p + comp - 3];
      float a = *dst;
      for (int c = m.child_start[i_p]; c < m.child_start[i_p + 1]; ++c) {
        int i_l = m.child_list[c];
        a = a + ((comp < 3) ? s->cfrc_vel[3 * i_l + comp] : s->cfrc_ang[3 * i_l + comp - 3]);
      }
      *dst = a;
    }
    team_sync();
  }
  for (int i_d = tl; i_d < ND; i_d += T) {
    int i_l = m.dof_link[i_d];
    float qf_bias = dot(ld3(s->cdof_ang, i_d), ld3(s->cfrc_ang, i_l)) + dot(ld3(s->cdof_vel, i_d), ld3(s->cfrc_vel, i_l));
    float f = s->qf_passive[i_d] - qf_bias + s->qf_applied[i_d];
    s->force[i_d] = f;
    if (env_valid) { astore(e, AO(force), i_d, f); astore(e, AO(qf_smooth), i_d, f); }
  }
  team_sync();
  PH(24)
  // ---- acc_smooth = L^-T D^-1 L^-1 force: serial chains, evaluated redundantly by every lane on the LDS copy ----
  bool solved = false;
#if GO2SIM_FAST_ORDER
  if constexpr (T == 32 || T == 64) {
    if (arrow) {
      const int leg = (tl >> 3) & 3, u = tl & 7, ub = u < 6 ? u : 5;
      const int p0 = dm_arrow_dof(m.arrow_mode, leg, 0), p1 = dm_arrow_dof(m.arrow_mode, leg, 1), p2 = dm_arrow_dof(m.arrow_mode, leg, 2);
      float xb[6], x0, x1, x2;
      arrow_solve<T>(AF, s->force[p0], s->force[p1], s->force[p2], s->force[ub], s->Dinv, tl, xb, x0, x1, x2);   // (Dinv is idle in this form: exchange buffer)
      if (tl == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s->out[k] = xb[k];
      }
      if (tl < 32 && u == 0) { s->out[p0] = x0; s->out[p1] = x1; s->out[p2] = x2; }
      solved = true;
    }
  }
#endif
  if (!solved) {
    float y[ND];                                                       // statically unrolled: the running vector stays in registers
#pragma unroll
    for (int i_d_ = 0; i_d_ < ND; ++i_d_) {
      const int i_d = ND - i_d_ - 1;
      float cur = s->force[i_d];
#pragma unroll
      for (int j_d = i_d + 1; j_d < ND; ++j_d) cur = cur - s->L[j_d * ND + i_d] * y[j_d];
      y[i_d] = cur;
    }
#pragma unroll
    for (int i_d = 0; i_d < ND; ++i_d) y[i_d] = y[i_d] * s->Dinv[i_d];
#pragma unroll
    for (int i_d = 0; i_d < ND; ++i_d) {
      float cur = y[i_d];
#pragma unroll
      for (int j_d = 0; j_d < i_d; ++j_d) cur = cur - s->L[i_d * ND + j_d] * y[j_d];
      y[i_d] = cur;
    }
    if (tl == 0) {
#pragma unroll
      for (int i_d = 0; i_d < ND; ++i_d) s->out[i_d] = y[i_d];
    }
  }
  team_sync();
  if (env_valid) for (int i_d = tl; i_d < ND; i_d += T) { float a = s->out[i_d]; astore(e, AO(acc_smooth), i_d, a); astore(e, AO(acc), i_d, a); }
  // the mass matrix is symmetric: its lower triangle goes to the record packed (entry (i, j), j <= i, at i (i + 1) / 2 + j: 684 instead of 1296 bytes per env
  // written here and read by the solver)
  if (env_valid) for (int idx = tl; idx < NTRI; idx += T) { int i, j; tri_index(m, idx, i, j); astore(e, AO(mass_mat), idx, s->M[i * ND + j]); }
  PH(25)
}

template <int T>
__global__ __launch_bounds__(64) void k_dynamics_team(Pool P, const ModelS* __restrict__ mp) {
  constexpr int EPW = 64 / T;
  __shared__ DynData lds[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  const ModelS& ms = *(const ModelS*)ms_raw;
  wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, mp);
  {  // ---- stage the SoA inputs cooperatively (adjacent lanes = adjacent envs), before any lane retires ----
    const int b0 = xcd_block() * EPW;
    wg_load<EPW, NL * 3>(P, b0, FO(cd_vel), [&](int ev, int k, float v) { lds[ev].cd_vel[k] = v; });
    wg_load<EPW, NL * 3>(P, b0, FO(cd_ang), [&](int ev, int k, float v) { lds[ev].cd_ang[k] = v; });
    wg_load<EPW, ND>(P, b0, FO(vel), [&](int ev, int k, float v) { lds[ev].vel[k] = v; });
  }
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const int b = xcd_block() * EPW + slot;
  const bool env_valid = b < P.B;
  const ModelView m(&ms, mp);
  E e(P, env_valid ? b : P.B - 1);
  DynData* s = &lds[slot];
  PH_BEGIN
  // AoS record: the team reads consecutive words
  team_stage<NL * 9, T>(tl, [&](int k) { return aload(e, AO(cinr_inertial), k); }, [&](int k, float v) { s->cinr_I[k] = v; s->crb_I[k] = v; });
  team_stage<NL * 3, T>(tl, [&](int k) { return aload(e, AO(cinr_pos), k); }, [&](int k, float v) { s->cinr_pos[k] = v; s->crb_pos[k] = v; });
  team_stage<NL, T>(tl, [&](int k) { return aload(e, AO(cinr_mass), k); }, [&](int k, float v) { s->cinr_mass[k] = v; s->crb_mass[k] = v; });
  team_stage<ND, T>(tl, [&](int d) { return __int_as_float((int)e.ctrl_mode()[d]); }, [&](int d, float v) { s->ctrl_mode[d] = __float_as_int(v); });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_ang), k); }, [&](int k, float v) { s->cdof_ang[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_vel), k); }, [&](int k, float v) { s->cdof_vel[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdofd_ang), k); }, [&](int k, float v) { s->cdofd_ang[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdofd_vel), k); }, [&](int k, float v) { s->cdofd_vel[k] = v; });
  // inputs of the later phases, parked in LDS slots that are only written afterwards (by the same lane that reads the parked value):
  // control targets -> qf_applied / qf_passive / force / out, external link forces -> cfrc_ang / cfrc_vel
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_force), d); }, [&](int d, float v) { s->qf_applied[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_pos), d); }, [&](int d, float v) { s->qf_passive[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_vel), d); }, [&](int d, float v) { s->force[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(dof_pos), d); }, [&](int d, float v) { s->out[d] = v; });
  team_stage<NL * 6, T>(tl, [&](int k) { return gload(e, FO(ext), k); },
                        [&](int k, float v) { int i_l = k / 6, c = k % 6; if (c < 3) s->cfrc_ang[3 * i_l + c] = v; else s->cfrc_vel[3 * i_l + c - 3] = v; });
  team_sync();
  PH(20)
  tk_dynamics<T>(m, e, s, tl, env_valid);
}


// the kinematics working set of k_integrate_fk_dynamics_team: laid over the M | L words of the dynamics record where it fits (the Go2 shape), else on its own
template <bool OVERLAY, int EPW> struct KinSeparate { KinData k[EPW]; DEV KinData* at(int slot, void*) { return &k[slot]; } };
template <int EPW> struct KinSeparate<true, EPW> { DEV KinData* at(int, void* overlay) { return (KinData*)overlay; } };
// kernel_step_2 of substep i followed by kernel_step_1 of substep i + 1 (rigid_solver.py:3072-3180, 3008-3069) in one launch: integrate, commit, FK,
// COM / cinr / cdof, geoms, forward velocity -- and straight on to the forward dynamics of the new state, whose inputs stay in LDS instead of making
// an HBM round trip between two kernels (the FK outputs are still written out: the collision kernel, the solver and the env kernels read them).
// Between the two halves nothing else runs in a scene step; the control inputs of the dynamics do not depend on the kinematics and are requested at
// the top of the kernel.
constexpr bool KIN_OVERLAY = sizeof(KinData) <= 2 * sizeof(float) * ND * ND;         // (shape variants with few dofs: the kinematics set gets its own LDS block)
template <int T>
DEV void integrate_fk_dynamics_body(const Pool& P, const ModelS* __restrict__ mp, int b, DynData* lds_d, char* ms_raw, KinSeparate<KIN_OVERLAY, 64 / T>& lds_k,
                                    bool pre, float pre_vel, float pre_acc) {
  const ModelS& ms = *(const ModelS*)ms_raw;
  wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, mp);
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const ModelView m(&ms, mp);
  E e(P, b < P.B ? b : P.B - 1);
  DynData* d = &lds_d[slot];
  // the kinematics working set lives in the M | L words of the dynamics record, which the dynamics only start writing (mass matrix, then its factor)
  // after the kinematics are done: same LDS footprint, hence the same 8 workgroups per CU, as k_dynamics_team alone
  static_assert(offsetof(DynData, L) == offsetof(DynData, M) + sizeof(float) * ND * ND, "KinData overlay");
  KinData* s = lds_k.at(slot, d->M);
  PH_BEGIN
  if (pre) { if (tl < ND) { s->vel[tl] = pre_vel; s->vel_next[tl] = pre_acc; } }     // (k_solve_integrate_team: taken from the solver's LDS block)
  else {
    team_stage<ND, T>(tl, [&](int i) { return gload(e, FO(vel), i); }, [&](int i, float v) { s->vel[i] = v; });
    team_stage<ND, T>(tl, [&](int i) { return aload(e, AO(acc), i); }, [&](int i, float a) { s->vel_next[i] = a; });
  }
  team_stage<NQ, T>(tl, [&](int q) { return gload(e, FO(qpos), q); }, [&](int q, float v) { s->qpos[q] = v; });
  // control inputs of the dynamics half (parked exactly as k_dynamics_team parks them)
  team_stage<ND, T>(tl, [&](int i) { return __int_as_float((int)e.ctrl_mode()[i]); }, [&](int i, float v) { d->ctrl_mode[i] = __float_as_int(v); });
  team_stage<ND, T>(tl, [&](int i) { return gload(e, FO(ctrl_force), i); }, [&](int i, float v) { d->qf_applied[i] = v; });
  team_stage<ND, T>(tl, [&](int i) { return gload(e, FO(ctrl_pos), i); }, [&](int i, float v) { d->qf_passive[i] = v; });
  team_stage<ND, T>(tl, [&](int i) { return gload(e, FO(ctrl_vel), i); }, [&](int i, float v) { d->force[i] = v; });
  team_stage<NL * 6, T>(tl, [&](int k) { return gload(e, FO(ext), k); },
                        [&](int k, float v) { int i_l = k / 6, c = k % 6; if (c < 3) d->cfrc_ang[3 * i_l + c] = v; else d->cfrc_vel[3 * i_l + c - 3] = v; });
  team_for<ND, T>(tl, [&](int i) { s->vel_next[i] = s->vel[i] + s->vel_next[i] * m.substep_dt; });
  if (tl == 0) s->valid = 1;
  tk_stage_links<T>(e, s, tl);
  team_sync();
  if (b >= P.B) return;                                               // (team_sync is a fence, not a barrier: a team may leave early)
  tk_integrate<T>(m, e, s, tl);
  PH(40)
  tk_kinematics<T>(m, e, s, tl, false, d);
  team_for<ND, T>(tl, [&](int i) { d->vel[i] = s->vel[i]; });
  team_sync();
  PH(41)
  tk_dynamics<T>(m, e, d, tl, true);
}
template <int T>
__global__ __launch_bounds__(64) void k_integrate_fk_dynamics_team(Pool P, const ModelS* __restrict__ mp) {
  STAMP(STK_INT_FK_DYN)
  constexpr int EPW = 64 / T;
  __shared__ DynData lds_d[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  __shared__ KinSeparate<KIN_OVERLAY, EPW> lds_k;
  integrate_fk_dynamics_body<T>(P, mp, xcd_block() * EPW + (int)threadIdx.x / T, lds_d, ms_raw, lds_k, false, 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// collision detection  (R/collider/*.py)
// ---------------------------------------------------------------------------------------------
// ---- support functions, collider/support_field.py ---------------------------------------------
DEV int wrap180(float x) {
  if (!(x >= 0.0f)) return 0;
  int i = (x >= 180.0f) ? (int)(x - 180.0f) : (int)x;
  return (i > 179) ? 179 : i;
}
DEV int clampidx(float x) {
  if (!(x >= 0.0f)) return 0;
  return (x >= 179.0f) ? 179 : (int)x;
}
// _func_support_mesh for a cylinder (support_field.py:138-180); the 180x180 direction-grid table is
// reproduced analytically: vertex set = 32-gon ring x {+h/2,-h/2} (tools/compile_go2_model.py)
DEV V3 support_cylinder_local(const Model& m, const Geom& G, V3 d_mesh, int* vid_out = nullptr) {
  const float PI = 3.14159265358979323846f;
  float theta = dm_atan2(d_mesh.y, d_mesh.x);
  float phi = dm_acos(d_mesh.z);
  const float support_res = 180.0f;
  float ii = (theta + PI) / PI / 2.0f * support_res;
  float jj = phi / PI * support_res;
  // the four neighbours of the grid cell are (floor | ceil of ii) x (floor | ceil of jj): two azimuth cells -> two ring vertices, whose table
  // entries are fetched side by side (two dependent round trips to the model instead of eight); candidates in the order of the reference's loop
  const int i_lo = wrap180(dm_floor(ii)), i_hi = wrap180(dm_ceil(ii));
  int j_lo = clampidx(dm_floor(jj)); if (j_lo == 0) j_lo = 1;
  int j_hi = clampidx(dm_ceil(jj)); if (j_hi == 179) j_hi = 178;
  const int k_lo = m.theta_to_ring[i_lo], k_hi = m.theta_to_ring[i_hi];
  const float rx_lo = G.rim[k_lo][0], ry_lo = G.rim[k_lo][1], rx_hi = G.rim[k_hi][0], ry_hi = G.rim[k_hi][1];
  const float half = 0.5f * G.data[1];
  float dot_max = -1e20f;
  V3 v = v3(0, 0, 0);
  int vid = 0;
#pragma unroll
  for (int i4 = 0; i4 < 4; ++i4) {
    const bool hi_i = (i4 % 2) != 0, hi_j = (i4 / 2) > 0;
    const int j = hi_j ? j_hi : j_lo, k = hi_i ? k_hi : k_lo;
    V3 pos = v3(hi_i ? rx_hi : rx_lo, hi_i ? ry_hi : ry_lo, (j <= 90) ? half : -half);
    float d = dot(pos, d_mesh);
    if (d > dot_max) { v = pos; dot_max = d; vid = k + ((j <= 90) ? 0 : 32); }
  }
  if (vid_out) *vid_out = vid;
  return v;
}
// support_driver, collider/mpr.py:146-176
// _func_support_prism, support_field.py:262-280: the terrain geom is represented by the current 6-vertex prism
// the lanes of my team that pass p (bit l = lane l of the team)
template <int T> DEV unsigned long long team_ballot(bool p) {
  const unsigned long long b = __ballot(p);
  if constexpr (T == 64) return b;
  else return (b >> (threadIdx.x & (64 - T))) & ((1ull << T) - 1ull);
}
DEV V3 vsel(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
DEV V3 support_prism(const V3* prism, V3 d) {
  // the bottom (0..2) or top (3..5) triangle, then the first vertex with the largest projection; written with value selects and constant
  // indices so that the six vertices stay in registers (an index computed at run time would put the prism in scratch memory)
  const bool bottom = d.z < 0;
  const V3 p0 = vsel(bottom, prism[0], prism[3]), p1 = vsel(bottom, prism[1], prism[4]), p2 = vsel(bottom, prism[2], prism[5]);
  float best = dot(p0, d);
  V3 v = p0;
  const float dt1 = dot(p1, d);
  const bool b1 = dt1 > best;
  v = vsel(b1, p1, v); best = b1 ? dt1 : best;
  const float dt2 = dot(p2, d);
  v = vsel(dt2 > best, p2, v);
  return v;
}
// type and size of a geom, read once per pair so that the MPR iterations do not go back to the model in global memory
struct GeomLite { int type; float d0, d1, d2; };
DEV GeomLite geom_lite(const Model& m, int i_g) { const Geom& G = m.geoms[i_g]; GeomLite r = {G.type, G.data[0], G.data[1], G.data[2]}; return r; }
DEV V3 support_driver(const Model& m, V3 direction, int i_g, const GeomLite& gl, V3 pos, const Rot& rot, const V3* prism = nullptr) {
  if (gl.type == GEOM_TERRAIN) return support_prism(prism, direction);
  if (gl.type == GEOM_SPHERE) {
    return pos + direction * gl.d0;
  } else if (gl.type == GEOM_BOX) {
    V3 d_box = rot_apply_inv(rot, direction);
    V3 v_ = v3((d_box.x < 0.0f ? -1.0f : 1.0f) * gl.d0 * 0.5f, (d_box.y < 0.0f ? -1.0f : 1.0f) * gl.d1 * 0.5f,
               (d_box.z < 0.0f ? -1.0f : 1.0f) * gl.d2 * 0.5f);
    return rot_apply(rot, v_) + pos;
  } else {
    V3 d_mesh = rot_apply_inv(rot, direction);
    V3 v_ = support_cylinder_local(m, m.geoms[i_g], d_mesh);
    return rot_apply(rot, v_) + pos;
  }
}
struct Pair { int i_ga, i_gb; V3 pos_a; Q4 quat_a; V3 pos_b; Q4 quat_b; const V3* prism; GeomLite ga, gb; Rot ra, rb; };   // ra / rb = make_rot(quat_a / quat_b)
DEV void pair_set_rots(Pair& pr) { pr.ra = make_rot(pr.quat_a); pr.rb = make_rot(pr.quat_b); }
// compute_support, collider/mpr.py:179-202
DEV void compute_support(const Model& m, V3 direction, const Pair& pr, V3& v, V3& v1, V3& v2) {
  v1 = support_driver(m, direction, pr.i_ga, pr.ga, pr.pos_a, pr.ra);
  v2 = support_driver(m, -direction, pr.i_gb, pr.gb, pr.pos_b, pr.rb, pr.prism);
  v = v1 - v2;
}


// ---- safe GJK + EPA fallback (csrc/go2sim_gjk_dev.h): geometric queries of collider/gjk_support.py:62-186, support_field.py:183-306,
//      gjk.py:1652-1700,1854-1907.  Vertex ids only need to be unique per (geom, vertex): 64 ids are reserved per geom. ----
// (type and size of the geom come from the per-pair GeomLite record: a GJK / EPA query makes some sixty support calls, and fetching them from
//  the model in global memory every time put a dependent load in front of each one)
DEV V3 gjk_support_driver(const Model& m, V3 direction, int i_g, const GeomLite& gl, V3 pos, const Rot& rot, int& vid) {
  if (gl.type == GEOM_SPHERE) {
    vid = -1;
    return pos + direction * gl.d0;
  } else if (gl.type == GEOM_BOX) {
    V3 d_box = rot_apply_inv(rot, direction);
    V3 v_ = v3((d_box.x < 0.0f ? -1.0f : 1.0f) * gl.d0 * 0.5f, (d_box.y < 0.0f ? -1.0f : 1.0f) * gl.d1 * 0.5f,
               (d_box.z < 0.0f ? -1.0f : 1.0f) * gl.d2 * 0.5f);
    vid = (v_.x > 0.0f) * 1 + (v_.y > 0.0f) * 2 + (v_.z > 0.0f) * 4 + 64 * i_g;
    return rot_apply(rot, v_) + pos;
  } else {
    V3 d_mesh = rot_apply_inv(rot, direction);
    int k = 0;
    V3 v_ = support_cylinder_local(m, m.geoms[i_g], d_mesh, &k);
    vid = k + 64 * i_g;
    return rot_apply(rot, v_) + pos;
  }
}
#include "go2sim_gjk_dev.h"   // device-side safe GJK + EPA (templated on the polytope store: LDS slot or full-capacity global record)

// One GJK / EPA query of the narrow phase.  The lane takes one of its team's LDS polytope slots (bit mask, LDS atomics: no waiting, a lane
// that finds none goes to global memory straight away); a query that outgrows the slot is repeated on the full-capacity record in global
// memory.  Same code, same arithmetic, same answer in all three cases.  The LDS flavour is inlined into the kernel so that the compiler sees
// the address space of the slot (ds_read / ds_write instead of flat accesses); the global flavour is the cold path and stays out of line.
DEVN DgResult gjk_query_global(const DgPair& dp, GjkStoreFull* full, float eps) { return dg_contact(dp, *full, eps); }
DEV DgResult gjk_query(const DgPair& dp, GjkStoreLds* slots, unsigned* slot_mask, GjkStoreFull* full, float eps) {
  int slot = -1;
  if (slots) {
    for (int i = 0; i < GJK_SLOTS_MAX && slot < 0; ++i) {
      const unsigned bit = 1u << i;
      if (!(atomicOr(slot_mask, bit) & bit)) slot = i;
    }
  }
  DgResult r; r.is_col = false; r.overflow = true; r.penetration = 0.0f; r.normal = v3(0, 0, 0); r.pos = v3(0, 0, 0);
  if (slot >= 0) {
    r = dg_contact(dp, slots[slot], eps);
    atomicAnd(slot_mask, ~(1u << slot));
  }
  if (r.overflow) r = gjk_query_global(dp, full, eps);
  return r;
}

// ---- MPR, collider/mpr.py: the 4-vertex portal simplex lives in registers -----------------------
struct Simplex { V3 v[4], v1[4], v2[4]; };
DEV V3 mpr_portal_dir(const Simplex& s) { return normalized(cross(s.v[2] - s.v[1], s.v[3] - s.v[1])); }
DEV bool mpr_portal_reach_tolerance(const Model& m, const Simplex& s, V3 v, V3 direction) {
  float dv1 = dot(s.v[1], direction), dv2 = dot(s.v[2], direction), dv3 = dot(s.v[3], direction), dv4 = dot(v, direction);
  float dot1 = fmn(fmn(dv4 - dv1, dv4 - dv2), dv4 - dv3);
  return dot1 < m.ccd_tolerance + m.ccd_eps * fmx(1.0f, dot1);
}
DEV void simplex_set(Simplex& s, int i, V3 v, V3 v1, V3 v2) {
  // value selects on every slot (not branches around stores): the compiler otherwise turns the three stores into one store through a
  // computed address, which moves the whole simplex to scratch memory and adds two memory round trips to every portal iteration
  const bool c1 = i == 1, c2 = i == 2, c3 = !(c1 || c2);
  s.v[1] = vsel(c1, v, s.v[1]); s.v1[1] = vsel(c1, v1, s.v1[1]); s.v2[1] = vsel(c1, v2, s.v2[1]);
  s.v[2] = vsel(c2, v, s.v[2]); s.v1[2] = vsel(c2, v1, s.v1[2]); s.v2[2] = vsel(c2, v2, s.v2[2]);
  s.v[3] = vsel(c3, v, s.v[3]); s.v1[3] = vsel(c3, v1, s.v1[3]); s.v2[3] = vsel(c3, v2, s.v2[3]);
}
DEV void mpr_expand_portal(Simplex& s, V3 v, V3 v1, V3 v2) {
  V3 v4v0 = cross(v, s.v[0]);
  float d = dot(s.v[1], v4v0);
  int i_s;
  if (d > 0) { d = dot(s.v[2], v4v0); i_s = (d > 0) ? 1 : 3; }
  else { d = dot(s.v[3], v4v0); i_s = (d > 0) ? 2 : 1; }
  simplex_set(s, i_s, v, v1, v2);
}
// The first step of mpr_discover_portal on its own (same expressions): true when the query ends there without a contact (the support point along
// the line of centres does not pass the origin).  The heightfield pass uses it to drop such prisms before the full queries are distributed.
DEV bool mpr_first_support_separates(const Model& m, const Pair& pr, V3 center_a, V3 center_b) {
  const float EPSC = m.ccd_eps;
  V3 v0 = center_a - center_b;
  if (dm_abs(v0.x) < EPSC && dm_abs(v0.y) < EPSC && dm_abs(v0.z) < EPSC) v0.x += 10.0f * EPSC;
  const V3 direction = -normalized(v0);
  V3 v, v1, v2;
  compute_support(m, direction, pr, v, v1, v2);
  return dot(v, direction) < EPSC;
}
// mpr_discover_portal, mpr.py:445-598
DEV int mpr_discover_portal(const Model& m, Simplex& s, const Pair& pr, V3 center_a, V3 center_b) {
  const float EPSC = m.ccd_eps;
  s.v1[0] = center_a; s.v2[0] = center_b; s.v[0] = center_a - center_b;
  int simplex_size = 1;
  if (dm_abs(s.v[0].x) < EPSC && dm_abs(s.v[0].y) < EPSC && dm_abs(s.v[0].z) < EPSC) s.v[0].x += 10.0f * EPSC;
  V3 direction = -normalized(s.v[0]);
  V3 v, v1, v2;
  compute_support(m, direction, pr, v, v1, v2);
  s.v1[1] = v1; s.v2[1] = v2; s.v[1] = v;
  simplex_size = 2;
  float d = dot(v, direction);
  int ret = 0;
  if (d < EPSC) {
    ret = -1;
  } else {
    direction = cross(s.v[0], s.v[1]);
    if (dot(direction, direction) < EPSC) {
      if (dm_abs(s.v[1].x) < EPSC && dm_abs(s.v[1].y) < EPSC && dm_abs(s.v[1].z) < EPSC) ret = 1; else ret = 2;
    } else {
      direction = normalized(direction);
      compute_support(m, direction, pr, v, v1, v2);
      d = dot(v, direction);
      if (d < EPSC) {
        ret = -1;
      } else {
        s.v1[2] = v1; s.v2[2] = v2; s.v[2] = v;
        simplex_size = 3;
        V3 va = s.v[1] - s.v[0], vb = s.v[2] - s.v[0];
        direction = normalized(cross(va, vb));
        d = dot(direction, s.v[0]);
        if (d > 0) {
          V3 t;
          t = s.v[1]; s.v[1] = s.v[2]; s.v[2] = t; t = s.v1[1]; s.v1[1] = s.v1[2]; s.v1[2] = t; t = s.v2[1]; s.v2[1] = s.v2[2]; s.v2[2] = t;
          direction = -direction;
        }
        int num_trials = 0;
        while (simplex_size < 4) {
          compute_support(m, direction, pr, v, v1, v2);
          d = dot(v, direction);
          if (d < EPSC) { ret = -1; break; }
          bool cont = false;
          va = cross(s.v[1], v);
          d = dot(va, s.v[0]);
          if (d < -EPSC) { s.v1[2] = v1; s.v2[2] = v2; s.v[2] = v; cont = true; }
          if (!cont) {
            va = cross(v, s.v[2]);
            d = dot(va, s.v[0]);
            if (d < -EPSC) { s.v1[1] = v1; s.v2[1] = v2; s.v[1] = v; cont = true; }
          }
          if (cont) {
            va = s.v[1] - s.v[0]; vb = s.v[2] - s.v[0];
            direction = normalized(cross(va, vb));
            num_trials++;
            if (num_trials == 15) { ret = -1; break; }
          } else {
            s.v1[3] = v1; s.v2[3] = v2; s.v[3] = v;
            simplex_size = 4;
          }
        }
      }
    }
  }
  return ret;
}
// mpr_refine_portal, mpr.py:232-278
DEV int mpr_refine_portal(const Model& m, Simplex& s, const Pair& pr) {
  int ret = 1;
  while (true) {
    V3 direction = mpr_portal_dir(s);
    if (dot(s.v[1], direction) > -m.ccd_eps) { ret = 0; break; }
    V3 v, v1, v2;
    compute_support(m, direction, pr, v, v1, v2);
    if (!(dot(v, direction) > -m.ccd_eps) || mpr_portal_reach_tolerance(m, s, v, direction)) { ret = -1; break; }
    mpr_expand_portal(s, v, v1, v2);
  }
  return ret;
}
// mpr_find_pos (non-mujoco branch), mpr.py:281-316
DEV V3 mpr_find_pos(const Model& m, const Simplex& s) {
  float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
  float sum_ = ((b0 + b1) + b2) + b3;
  if (sum_ < m.ccd_eps) {
    V3 direction = mpr_portal_dir(s);
    b0 = 0.0f;
    b1 = dot(cross(s.v[2], s.v[3]), direction);   // i=1: i1=2, i2=3
    b2 = dot(cross(s.v[3], s.v[1]), direction);   // i=2: i1=3, i2=1
    b3 = dot(cross(s.v[1], s.v[2]), direction);   // i=3: i1=1, i2=2
    sum_ = ((b0 + b1) + b2) + b3;
  }
  V3 p1 = v3(0, 0, 0), p2 = v3(0, 0, 0);
  p1 = p1 + b0 * s.v1[0]; p2 = p2 + b0 * s.v2[0];
  p1 = p1 + b1 * s.v1[1]; p2 = p2 + b1 * s.v2[1];
  p1 = p1 + b2 * s.v1[2]; p2 = p2 + b2 * s.v2[2];
  p1 = p1 + b3 * s.v1[3]; p2 = p2 + b3 * s.v2[3];
  return (0.5f / sum_) * (p1 + p2);
}
// mpr_find_penetration, mpr.py:338-423
DEV void mpr_find_penetration(const Model& m, Simplex& s, const Pair& pr, bool& is_col, V3& normal, float& penetration, V3& pos) {
  int iterations = 0;
  while (true) {
    V3 direction = mpr_portal_dir(s);
    V3 v, v1, v2;
    compute_support(m, direction, pr, v, v1, v2);
    if (mpr_portal_reach_tolerance(m, s, v, direction) || iterations > m.ccd_iterations) {
      penetration = dot(direction, s.v[1]);
      normal = -direction;
      is_col = true;
      pos = mpr_find_pos(m, s);
      break;
    }
    mpr_expand_portal(s, v, v1, v2);
    iterations++;
  }
}
// guess_geoms_center, mpr.py:601-683
DEV void guess_geoms_center(const Model& m, const Pair& pr, V3 normal_ws, V3& center_a, V3& center_b) {
  const Geom& A = m.geoms[pr.i_ga]; const Geom& Bg = m.geoms[pr.i_gb];
  center_a = transform_by_trans_quat(A.center, pr.pos_a, pr.quat_a);
  center_b = transform_by_trans_quat(Bg.center, pr.pos_b, pr.quat_b);
  if (dm_abs(normal_ws.x) > m.ccd_eps || dm_abs(normal_ws.y) > m.ccd_eps || dm_abs(normal_ws.z) > m.ccd_eps) {
    V3 center_a_local = 0.5f * (A.aabb[7] + A.aabb[0]);
    center_a = transform_by_trans_quat(center_a_local, pr.pos_a, pr.quat_a);
    V3 center_b_local = 0.5f * (Bg.aabb[7] + Bg.aabb[0]);
    center_b = transform_by_trans_quat(center_b_local, pr.pos_b, pr.quat_b);
    V3 delta = center_a - center_b;
    V3 normal = normalized(delta);
    if (norm(cross(normal_ws, normal)) > 0.01f) {
      V3 offset = dot(delta, normal_ws) * normal_ws - delta;
      float offset_norm = norm(offset);
      if (offset_norm > m.eps) {
        V3 dir_offset = offset / offset_norm;
        V3 dla = inv_transform_by_quat(dir_offset, pr.quat_a), dlb = inv_transform_by_quat(dir_offset, pr.quat_b);
        V3 box_size_a = A.aabb[7] - A.aabb[0], box_size_b = Bg.aabb[7] - Bg.aabb[0];
        float length_a = dot(box_size_a, v3(dm_abs(dla.x), dm_abs(dla.y), dm_abs(dla.z)));
        float length_b = dot(box_size_b, v3(dm_abs(dlb.x), dm_abs(dlb.y), dm_abs(dlb.z)));
        float offset_ratio = fmn(offset_norm / (length_a + length_b), 0.5f);
        center_a = center_a + dir_offset * length_a * offset_ratio;
        center_b = center_b - dir_offset * length_b * offset_ratio;
      }
    }
  }
}
// func_mpr_contact -> func_mpr_contact_from_centers, mpr.py:686-819
// func_mpr_contact_from_centers, mpr.py:686-760
DEV void mpr_contact_from_centers(const Model& m, const Pair& pr, V3 center_a, V3 center_b, bool& is_col, V3& normal, float& penetration, V3& pos) {
  Simplex s;
  int res = mpr_discover_portal(m, s, pr, center_a, center_b);
  is_col = false; pos = v3(0, 0, 0); normal = v3(0, 0, 0); penetration = 0.0f;
  if (res == 1) {
    is_col = true; penetration = 0.0f; normal = -normalized(s.v[0]); pos = (s.v1[1] + s.v2[1]) * 0.5f;
  } else if (res == 2) {
    is_col = true; penetration = norm(s.v[1]); normal = -normalized(s.v[1]); pos = (s.v1[1] + s.v2[1]) * 0.5f;
  } else if (res == 0) {
    res = mpr_refine_portal(m, s, pr);
    if (res >= 0) mpr_find_penetration(m, s, pr, is_col, normal, penetration, pos);
  }
}
DEV void mpr_contact(const Model& m, const Pair& pr, V3 normal_ws, bool& is_col, V3& normal, float& penetration, V3& pos) {
  V3 center_a, center_b;
  guess_geoms_center(m, pr, normal_ws, center_a, center_b);
  mpr_contact_from_centers(m, pr, center_a, center_b, is_col, normal, penetration, pos);
}

// func_compute_tolerance, contact.py:264-283
DEV float compute_tolerance(const Model& m, int i_ga, int i_gb, float tolerance) {
  float size_b = norm(m.geoms[i_gb].aabb[7] - m.geoms[i_gb].aabb[0]);
  float size_a = norm(m.geoms[i_ga].aabb[7] - m.geoms[i_ga].aabb[0]);
  return 0.5f * tolerance * fmn(size_a, size_b);
}
// func_contact_orthogonals (non-mujoco branch), contact.py:286-345
DEV void contact_orthogonals(const Model& m, const E& e, int i_ga, int i_gb, V3 normal, V3& axis_0, V3& axis_1) {
  V3 size_ga = m.geoms[i_ga].aabb[7], size_gb = m.geoms[i_gb].aabb[7];
  float volume_ga = size_ga.x * size_ga.y * size_ga.z, volume_gb = size_gb.x * size_gb.y * size_gb.z;
  int i_g = (volume_ga < volume_gb) ? i_ga : i_gb;
  int i_l = m.geoms[i_g].link;
  M3 rot = quat_to_R(e.i_quat()[i_l], m.eps);
  int axis_idx = 0; float axis_angle_max = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float axis_angle = dm_abs(dot(mcol(rot, i), normal));
    if (axis_angle > axis_angle_max) { axis_angle_max = axis_angle; axis_idx = i; }
  }
  axis_idx = (axis_idx + 1) % 3;
  axis_0 = mcol(rot, axis_idx);
  axis_0 = normalized(axis_0 - dot(normal, axis_0) * normal);
  axis_1 = cross(normal, axis_0);
}
// func_rotate_frame, contact.py:348-369
DEV void rotate_frame(V3 pos, Q4 quat, V3 contact_pos, Q4 qrot, V3& new_pos, Q4& new_quat) {
  new_quat = transform_quat_by_quat(quat, qrot);
  V3 rel = contact_pos - pos;
  V3 vec = transform_by_quat(rel, qrot);
  vec = vec - rel;
  new_pos = pos - vec;
}

// ---------------------------------------------------------------------------------------------
// Team collision detection: T lanes per environment.
//   * AABBs: one lane per geom.
//   * Sweep-and-prune: the warm-started insertion sort of the reference is a *stable* sort of the 56 x-endpoints, so its
//     result equals a stable rank sort (rank = #smaller + #equal-before), evaluated one endpoint per lane.  The serial
//     sweep emits the pair (a, s) when min(a) < min(s) < max(a) in sorted order, in the order (position of min(s), position of
//     min(a)); the same list is produced by testing every valid geom pair in parallel and ordering the survivors by that key.
//   * Narrow phase: one lane per broad-phase pair (MPR + multi-contact perturbations are independent per pair); the contacts of a
//     pass of T pairs are compacted in pair order, which is the order in which the serial loop appends them.
// ---------------------------------------------------------------------------------------------
constexpr int GJK_SLOTS = GJK_SLOTS_MAX;
template <int T>
struct CollideData {
  // the three phases of the kernel use disjoint working sets, laid over each other: broad phase -> (barrier) -> convex narrow phase (GJK / EPA
  // polytopes of the lanes that fall back from MPR) -> (barrier) -> terrain pass
  struct Broad {
    alignas(16) float amin[NG * 4], amax[NG * 4];   // xyz + pad: one 128-bit LDS read per corner
    float sval[2 * NG], sval_sorted[2 * NG];
    alignas(16) unsigned long long skey[2 * NG];      // (order-preserving integer image of the endpoint value) << 8 | position before the sort
    int sig[2 * NG], sig_sorted[2 * NG];
    alignas(8) int rank_mm[NG * 2];                  // (rank of the min endpoint, rank of the max endpoint) per geom
    int cand_key[MAXB], cand_pair[MAXB];
  };
  // terrain pass: one slot per (geom, terrain) pair of the broad-phase list
  struct TPair { int i_ga, r_min, r_max, c_min, c_max, n_items, item_off; float zmin, tol; V3 pos_a; Q4 quat_a; V3 center_a; };
  struct Terrain {
    TPair tp[NG];
  };
  union alignas(16) {
    Broad bp;
    GjkStoreLds gjk[GJK_SLOTS];
    Terrain tr;
  };
  unsigned short pair_sorted[MAXB];                   // geom a | geom b << 8
  float stage[T][5][7];
  int cnt[T];
  unsigned gjk_slot_mask;                             // bit i set = gjk[i] is taken
  float gjk_res[T][8];                                // T = 16: answers of the cooperative GJK / EPA queries, indexed by the lane that asked
  unsigned ncv[NCV];                                  // ncache_valid of this env for the duration of the kernel
};
struct ContactStage { float* st; int n; };   // per-lane staging of the (<= 5) contacts of one pair

DEV void stage_contact(ContactStage& cs, V3 normal, V3 pos, float pen) {
  float* p = cs.st + 7 * cs.n;
  p[0] = normal.x; p[1] = normal.y; p[2] = normal.z; p[3] = pos.x; p[4] = pos.y; p[5] = pos.z; p[6] = pen;
  cs.n++;
}

// func_convex_convex_contact (CCD_ALGORITHM_CODE.MPR branch), narrowphase.py:514-961; contacts go to the lane's staging buffer.
// The function is cut in three so that the GJK / EPA fallback of the unperturbed detection -- the one the landing robots need -- can be answered
// by the whole team between the pieces (k_collide_team, T = 16):  cc_detect0 = pair set-up + MPR (+ cold retry) + "prefer GJK" decision;
// [GJK / EPA, by one lane (cc_gjk_lane) or by the team (dgc_contact)];  cc_rest = bookkeeping of detection 0 and the four perturbed detections.
struct CcState {
  Pair pr; V3 ga_pos_o, gb_pos_o; Q4 ga_quat_o, gb_quat_o;
  int i_pair, type_a, type_b; bool multi_contact, want_gjk; float tolerance;
  bool is_col; float penetration; V3 normal, contact_pos;
};
DEV void cc_mpr_with_retry(const Model& m, CcState& c, int i_detection, unsigned* ncv, const Arr3& normal_cache, bool& guess_available) {
  const float EPS = m.eps;
  bool is_mpr_updated = false;
  V3 normal_ws = ((ncv[c.i_pair >> 5] >> (c.i_pair & 31)) & 1u) ? (V3)normal_cache[c.i_pair] : v3(0, 0, 0);
  guess_available = (dm_abs(normal_ws.x) > EPS) || (dm_abs(normal_ws.y) > EPS) || (dm_abs(normal_ws.z) > EPS);
  for (int i_mpr = 0; i_mpr < 2; ++i_mpr) {
    if (i_mpr == 1) {
      if ((i_detection == 0) && !c.is_col && guess_available) { normal_ws = v3(0, 0, 0); guess_available = false; is_mpr_updated = false; }
    }
    if (!is_mpr_updated) {
      PHD_BEGIN
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 35
      mpr_contact(m, c.pr, normal_ws, c.is_col, c.normal, c.penetration, c.contact_pos);
#endif
      mpr_contact(m, c.pr, normal_ws, c.is_col, c.normal, c.penetration, c.contact_pos);
      PHD(36)
      is_mpr_updated = true;
    }
  }
}
DEV bool cc_prefer_gjk(const Model& m, const CcState& c, bool guess_available) {
  if (c.penetration > c.tolerance) return !guess_available || (m.mc_tolerance * c.penetration >= m.mpr_to_gjk_ratio * c.tolerance);
  return false;
}
DEV void cc_detect0(const Model& m, const E& e, int i_ga, int i_gb, unsigned* ncv, CcState& c) {
  c.type_a = m.geoms[i_ga].type; c.type_b = m.geoms[i_gb].type;
  c.multi_contact = (c.type_a != GEOM_SPHERE) && (c.type_b != GEOM_SPHERE);
  c.tolerance = compute_tolerance(m, i_ga, i_gb, m.mc_tolerance);
  c.ga_pos_o = e.g_pos()[i_ga]; c.gb_pos_o = e.g_pos()[i_gb]; c.ga_quat_o = e.g_quat()[i_ga]; c.gb_quat_o = e.g_quat()[i_gb];
  Pair& pr = c.pr;
  pr.i_ga = i_ga; pr.i_gb = i_gb; pr.pos_a = c.ga_pos_o; pr.quat_a = c.ga_quat_o; pr.pos_b = c.gb_pos_o; pr.quat_b = c.gb_quat_o; pr.prism = nullptr; pr.ga = geom_lite(m, i_ga); pr.gb = geom_lite(m, i_gb);
  pair_set_rots(pr);
  c.is_col = false; c.penetration = 0.0f; c.normal = v3(0, 0, 0); c.contact_pos = v3(0, 0, 0);
  c.i_pair = (i_ga > i_gb) ? m.pair_idx[i_gb][i_ga] : m.pair_idx[i_ga][i_gb];
  bool guess_available;
  cc_mpr_with_retry(m, c, 0, ncv, e.normal_cache(), guess_available);
  c.want_gjk = cc_prefer_gjk(m, c, guess_available);
}
// narrowphase.py:734-845: safe GJK + EPA replaces the MPR answer (one lane; LDS polytope slot when `gjk_slots` is given, else the global record)
DEV void cc_gjk_lane(const Model& m, const E& e, CcState& c, GjkStoreLds* gjk_slots, unsigned* gjk_slot_mask, GjkStoreFull* gjk_full) {
  atomicAdd(&e.gjk_fallback()[0], 1);
  const Pair& pr = c.pr;
  DgPair dp;                                                 // the out-of-line callee takes a reference: this record only exists on the cold
  dp.m = &m; dp.i_ga = pr.i_ga; dp.i_gb = pr.i_gb; dp.pos_a = pr.pos_a; dp.quat_a = pr.quat_a; dp.pos_b = pr.pos_b; dp.quat_b = pr.quat_b;   // path, `pr` stays in registers
  dp.ga = pr.ga; dp.gb = pr.gb; dp.ra = pr.ra; dp.rb = pr.rb;
  dp.discrete = c.type_a == GEOM_BOX && c.type_b == GEOM_BOX;     // func_is_discrete_geoms, collider/utils.py:105-126
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 34
  { const DgResult g0 = gjk_query(dp, gjk_slots, gjk_slot_mask, gjk_full, m.eps); if (g0.penetration == 12345.0f) c.penetration = 0.0f; }
#endif
  PHD_BEGIN
  const DgResult gr = gjk_query(dp, gjk_slots, gjk_slot_mask, gjk_full, m.eps);
  PHD(34)
  c.is_col = gr.is_col;
  c.penetration = gr.penetration;
  if (c.is_col) { c.contact_pos = gr.pos; c.normal = gr.normal; }
}
DEV void cc_rest(const Model& m, const E& e, CcState& c, ContactStage& cs, GjkStoreLds* gjk_slots, unsigned* gjk_slot_mask, GjkStoreFull* gjk_full, unsigned* ncv) {
  const float EPS = m.eps;
  const int i_ga = c.pr.i_ga, i_gb = c.pr.i_gb, i_pair = c.i_pair;
  const bool multi_contact = c.multi_contact;
  const float tolerance = c.tolerance;
  Pair& pr = c.pr;
  bool is_col_0 = false; V3 normal_0 = v3(0, 0, 0), contact_pos_0 = v3(0, 0, 0);
  bool& is_col = c.is_col; float& penetration = c.penetration; V3& normal = c.normal; V3& contact_pos = c.contact_pos;
  V3 axis_0 = v3(0, 0, 0), axis_1 = v3(0, 0, 0); Q4 qrot = q4(0, 0, 0, 0);
  auto normal_cache = e.normal_cache();
  PHD_BEGIN
  for (int i_detection = 0; i_detection < 5; ++i_detection) {
    if (i_detection == 1) { PHD(44) }
    if (i_detection > 0 && multi_contact && is_col_0) {
      V3 axis = (float)(2 * (i_detection % 2) - 1) * axis_0 + (float)(1 - 2 * ((i_detection / 2) % 2)) * axis_1;
      qrot = rotvec_to_quat(m.mc_perturbation * axis, EPS);
      rotate_frame(c.ga_pos_o, c.ga_quat_o, contact_pos_0, qrot, pr.pos_a, pr.quat_a);
      rotate_frame(c.gb_pos_o, c.gb_quat_o, contact_pos_0, inv_quat(qrot), pr.pos_b, pr.quat_b);
      pair_set_rots(pr);
      bool guess_available;
      cc_mpr_with_retry(m, c, i_detection, ncv, normal_cache, guess_available);
      if (cc_prefer_gjk(m, c, guess_available)) cc_gjk_lane(m, e, c, gjk_slots, gjk_slot_mask, gjk_full);
    }
    if (i_detection == 0) {
      is_col_0 = is_col; normal_0 = normal; contact_pos_0 = contact_pos;
      if (is_col_0) {
        stage_contact(cs, normal, contact_pos, penetration);
        if (multi_contact) contact_orthogonals(m, e, i_ga, i_gb, normal, axis_0, axis_1);
        normal_cache[i_pair] = normal; atomicOr(&ncv[i_pair >> 5], 1u << (i_pair & 31));
      } else {
        atomicAnd(&ncv[i_pair >> 5], ~(1u << (i_pair & 31)));          // normal_cache[i_pair] := 0
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
      for (int i_c = 0; i_c < cs.n; ++i_c)
        if (!repeated) {
          const float* prev = cs.st + 7 * (cs.n - 1 - i_c);
          if (norm(contact_pos - v3(prev[3], prev[4], prev[5])) < tolerance) repeated = true;
        }
      if (!repeated && penetration > -tolerance) {
        penetration = fmx(penetration, 0.0f);
        stage_contact(cs, normal, contact_pos, penetration);
      }
    }
  }
  PHD(46)
}
DEV void convex_convex_contact_staged(const Model& m, const E& e, int i_ga, int i_gb, ContactStage& cs, GjkStoreLds* gjk_slots, unsigned* gjk_slot_mask, GjkStoreFull* gjk_full, unsigned* ncv) {
  CcState c;
  cc_detect0(m, e, i_ga, i_gb, ncv, c);
  if (c.want_gjk) cc_gjk_lane(m, e, c, gjk_slots, gjk_slot_mask, gjk_full);
  cc_rest(m, e, c, cs, gjk_slots, gjk_slot_mask, gjk_full, ncv);
}

// Conservative reach test of a (geom, heightfield) pair before any support point is computed: every point of the geom lies within R (+ 1 mm for the
// rounding of the pose arithmetic) of its origin, so its support-point bounding box lies inside the cube of half-side R around the origin; if the
// highest vertex of the heightfield under that cube (from the coarse maximum map, 3 x 3 blocks fetched side by side) stays below origin.z - R, no prism
// top can reach the geom's lowest point: the pair would enumerate its cells and find none eligible (narrowphase.py:430-436), and is dropped here.
// False (= the exact path decides) whenever the test does not apply.
DEV bool terrain_pair_out_of_reach(const Model& m, const GeomLite& gl, V3 pos) {
  float R;
  if (gl.type == GEOM_SPHERE) R = gl.d0;
  else if (gl.type == GEOM_BOX) R = 0.5f * dm_sqrt(gl.d0 * gl.d0 + gl.d1 * gl.d1 + gl.d2 * gl.d2);
  else if (gl.type == GEOM_CYLINDER) R = dm_sqrt(gl.d0 * gl.d0 + 0.25f * (gl.d1 * gl.d1));
  else return false;
  R = R * 1.001f + 1e-3f;
  const float* tmm = m.terrain_xyz_maxmin;
  const float sh = m.terrain_hs;
  int r_lo = (int)dm_floor((pos.x - R - tmm[3]) / sh) - 1, r_hi = (int)dm_ceil((pos.x + R - tmm[3]) / sh) + 1;
  int c_lo = (int)dm_floor((pos.y - R - tmm[4]) / sh) - 1, c_hi = (int)dm_ceil((pos.y + R - tmm[4]) / sh) + 1;
  r_lo = imx(0, r_lo); c_lo = imx(0, c_lo); r_hi = imn(m.terrain_rows - 1, r_hi); c_hi = imn(m.terrain_cols - 1, c_hi);
  if (!(r_lo <= r_hi && c_lo <= c_hi)) return false;
  const int I0 = r_lo / TERRAIN_CB, J0 = c_lo / TERRAIN_CB;
  if (r_hi / TERRAIN_CB > I0 + 2 || c_hi / TERRAIN_CB > J0 + 2) return false;
  float mx = -1e30f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 3; ++q) mx = fmx(mx, m.terrain_cmax[(size_t)imn(I0 + a, m.terrain_crows - 1) * m.terrain_ccols + imn(J0 + q, m.terrain_ccols - 1)]);
  return mx < pos.z - R;
}
// func_contact_mpr_terrain, narrowphase.py:345-490, split for one-lane-per-prism execution.
// (1) per pair: geom pose in the terrain frame, its bounding box from six support points, the cell range under it.
template <class TP>
DEV bool terrain_pair_setup(const Model& m, const E& e, int i_ga, int i_gb, TP& t) {
  V3 ga_pos = e.g_pos()[i_ga], gb_pos = e.g_pos()[i_gb]; Q4 ga_quat = e.g_quat()[i_ga], gb_quat = e.g_quat()[i_gb];
  const float margin = 0.0f;
  transform_pos_quat_by_trans_quat(ga_pos - gb_pos, ga_quat, v3(0, 0, 0), inv_quat(gb_quat), t.pos_a, t.quat_a);
  t.center_a = transform_by_trans_quat(m.geoms[i_ga].center, t.pos_a, t.quat_a);
  t.i_ga = i_ga;
  GeomLite gl = geom_lite(m, i_ga);
  const Rot t_rot = make_rot(t.quat_a);
  float xyz_max_min[6];
#pragma unroll
  for (int i_axis = 0; i_axis < 3; ++i_axis)
#pragma unroll
    for (int i_m = 0; i_m < 2; ++i_m) {
      V3 direction = v3(0, 0, 0);
      vset(direction, i_axis, (i_m == 0) ? 1.0f : -1.0f);
      V3 v1 = support_driver(m, direction, i_ga, gl, t.pos_a, t_rot);
      xyz_max_min[3 * i_m + i_axis] = vget(v1, i_axis);
    }
  const float* tmm = m.terrain_xyz_maxmin;
  bool is_return = false;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (tmm[i] < xyz_max_min[i + 3] - margin || tmm[i + 3] > xyz_max_min[i] + margin) is_return = true;
  const float sh = m.terrain_hs;
  int r_min = (int)dm_floor((xyz_max_min[3] - tmm[3]) / sh);
  int r_max = (int)dm_ceil((xyz_max_min[0] - tmm[3]) / sh);
  int c_min = (int)dm_floor((xyz_max_min[4] - tmm[4]) / sh);
  int c_max = (int)dm_ceil((xyz_max_min[1] - tmm[4]) / sh);
  t.r_min = imx(0, r_min); t.c_min = imx(0, c_min);
  t.r_max = imn(m.terrain_rows - 1, r_max); t.c_max = imn(m.terrain_cols - 1, c_max);
  t.zmin = xyz_max_min[5];
  if (is_return) { t.r_max = t.r_min; }   // empty cell range
  return !is_return;
}
// height of the k-th vertex of the strip of row r (vertex order of func_add_prism_vert: (c, i) with i fastest)
DEV float terrain_strip_z(const Model& m, int r, int c_min, int k) { return m.terrain_hf[(size_t)(r + (k & 1)) * m.terrain_cols + c_min + (k >> 1)]; }
// (2) the prism that exists after the k-th vertex of row r was pushed (k >= 2) is tested iff one of its top vertices reaches the geom: evaluated one
//     cell per lane in k_collide_team
// (3) MPR of the geom against that prism; the contact is returned in world coordinates
template <class TP>
DEV void terrain_prism_pair(const Model& m, const TP& t, int i_gb, int r, int k, V3 (&prism)[6], Pair& pr, V3& center_b) {
  const float* tmm = m.terrain_xyz_maxmin;
  const float sh = m.terrain_hs;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int kk = k - 2 + j;
    float x = sh * (float)(r + (kk & 1)) + tmm[3], y = sh * (float)(t.c_min + (kk >> 1)) + tmm[4];
    prism[j] = v3(x, y, tmm[5]);
    prism[3 + j] = v3(x, y, terrain_strip_z(m, r, t.c_min, kk) + 0.0f);
  }
  pr.i_ga = t.i_ga; pr.i_gb = i_gb; pr.prism = prism; pr.pos_a = t.pos_a; pr.quat_a = t.quat_a; pr.pos_b = v3(0, 0, 0); pr.quat_b = qident();
  pr.ga = geom_lite(m, t.i_ga); pr.gb = geom_lite(m, i_gb);
  pair_set_rots(pr);
  center_b = v3(0, 0, 0);
  for (int i_p = 0; i_p < 6; ++i_p) center_b = center_b + prism[i_p];
  center_b = center_b / 6.0f;
}
template <class TP>
DEV bool terrain_prism_separated_at_once(const Model& m, const TP& t, int i_gb, int r, int k) {
  V3 prism[6]; Pair pr; V3 center_b;
  terrain_prism_pair(m, t, i_gb, r, k, prism, pr, center_b);
  return mpr_first_support_separates(m, pr, t.center_a, center_b);
}
template <class TP>
DEV bool terrain_prism_contact(const Model& m, const E& e, const TP& t, int i_gb, int r, int k, V3& normal, V3& contact_pos, float& penetration) {
  V3 prism[6]; Pair pr; V3 center_b;
  terrain_prism_pair(m, t, i_gb, r, k, prism, pr, center_b);
  bool is_col;
  mpr_contact_from_centers(m, pr, t.center_a, center_b, is_col, normal, penetration, contact_pos);
  if (is_col) {
    V3 gb_pos = e.g_pos()[i_gb]; Q4 gb_quat = e.g_quat()[i_gb];
    normal = transform_by_quat(normal, gb_quat);
    contact_pos = transform_by_quat(contact_pos, gb_quat);
    contact_pos = contact_pos + gb_pos;
  }
  return is_col;
}

// EPW = environments per wavefront: 64 / T fills the wavefront; fewer leave the upper lanes idle (the teams of a wavefront run in lockstep, so a
// team pays for the longest query loop and for every branch direction of its neighbours)
template <int T, int EPW = 64 / T>
__global__ __launch_bounds__(64) void k_collide_team(Pool P, const Model* __restrict__ mp, GjkStoreFull* __restrict__ gjk_scratch, int* __restrict__ lpt_rec, int lpt_cap, int solver_epw) {
  STAMP(STK_COLLIDE)
  __shared__ CollideData<T> lds[EPW];
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const int b = xcd_block() * EPW + slot;
  if (slot >= EPW || b >= P.B) return;
  const Model& m = *mp;
  E e(P, b);
  CollideData<T>* s = &lds[slot];
  const float inf = dm_bits2f(0x7f800000u);
  if (tl == 0) s->gjk_slot_mask = 0u;                                   // made visible by the barriers of the broad phase
  for (int i = tl; i < NCV; i += T) s->ncv[i] = (unsigned)e.ncache_valid()[i];
  PH_BEGIN
  // ---- loads of the prologue first: previous contact count, first-step flag, the persistent sort order, geom poses ----
  const int nc_old = e.n_contacts()[0];
  const bool first = e.first_time()[0] != 0;
  const int n2 = 2 * NG;
  team_stage<2 * NG, T>(tl, [&](int i) { return __int_as_float(e.sort_ig()[i]); }, [&](int i, float v) { s->bp.sig[i] = __float_as_int(v); });
  constexpr int NPI = (NPAIR + T - 1) / T;                            // the pair table of the candidate test, fetched for all rounds up front
  int packed_[NPI];
  {
    const int n_pairs = m.n_pairs;
#pragma unroll
    for (int it = 0; it < NPI; ++it) { int pidx = it * T + tl; int v = m.pair_list[pidx < NPAIR ? pidx : NPAIR - 1]; packed_[it] = (pidx < n_pairs) ? v : -1; }
  }
  // ---- kernel_update_geom_aabbs, forward_kinematics.py:1171-1193 (out-of-range lanes redo the last geom: no branch between the loads) ----
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 30
  for (int rep = 0; rep < 2; ++rep)
#endif
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += T) {
    const int i_g = (g0 + tl < NG) ? g0 + tl : NG - 1;
    V3 lower = v3(inf, inf, inf), upper = v3(-inf, -inf, -inf);
    V3 gp = e.g_pos()[i_g]; Q4 gq = e.g_quat()[i_g];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      V3 corner = transform_by_trans_quat(m.geoms[i_g].aabb[c], gp, gq);
      lower = vmin(lower, corner); upper = vmax(upper, corner);
    }
    *(float4*)&s->bp.amin[4 * i_g] = make_float4(lower.x, lower.y, lower.z, 0.0f);
    *(float4*)&s->bp.amax[4 * i_g] = make_float4(upper.x, upper.y, upper.z, 0.0f);
  }
  // ---- func_collision_clear, broadphase.py:73-138 ----
  for (int i_c = tl; i_c < nc_old; i_c += T) {
    e.c_link()[i_c] = -1; e.c_link()[MAXC + i_c] = -1; e.c_geom()[i_c] = -1; e.c_geom()[MAXC + i_c] = -1;
    e.c_pen()[i_c] = 0.0f; e.c_pos()[i_c] = v3(0, 0, 0); e.c_normal()[i_c] = v3(0, 0, 0); e.c_force()[i_c] = v3(0, 0, 0);
  }
  team_sync();
  PH(30)
  // ---- func_broad_phase, broadphase.py:141-396: endpoint refresh + stable sort ----
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 31
  for (int rep = 0; rep < 2; ++rep) {
  team_sync();
#endif
  for (int i = tl; i < n2; i += T) {
    // first step: endpoints in (link, geom) order: geoms are stored link-major, so buffer slot i/2 holds geom i/2
    int sg = first ? ((i >> 1) | ((i & 1) ? 0x100 : 0)) : s->bp.sig[i];
    int g = sg & 0xff;
    s->bp.sig[i] = sg;
    const float v = (sg & 0x100) ? s->bp.amax[4 * g] : s->bp.amin[4 * g];
    s->bp.sval[i] = v;
    // rank of endpoint i = #{j : w_j < v  or  (w_j == v and j < i)}  =  #{j : key_j < key_i} with key = (image(value), position): one 64-bit
    // compare per pair instead of two float compares and the tie logic.  image() is monotone on the non-NaN floats and maps -0 and +0 to one value
    // (v + 0.0f), like the float compares it replaces.
    unsigned u = (unsigned)__float_as_int(v + 0.0f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    s->bp.skey[i] = ((unsigned long long)u << 8) | (unsigned long long)i;
  }
  team_sync();
  {
    // a lane ranks its (up to) NK endpoints at once: every key is read once (the same address for all lanes of the team) and compared with all of them,
    // instead of one pass over the keys per endpoint
    constexpr int NK = (2 * NG + T - 1) / T;
    unsigned long long key[NK]; int r[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) { const int i = tl + q * T; key[q] = s->bp.skey[i < n2 ? i : n2 - 1]; r[q] = 0; }
#pragma unroll
    for (int j = 0; j < 2 * NG; ++j) {
      const unsigned long long kj = s->bp.skey[j];
#pragma unroll
      for (int q = 0; q < NK; ++q) r[q] += (kj < key[q]) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const int i = tl + q * T;
      if (i < n2) {
        const float v = s->bp.sval[i];
        const int sg = s->bp.sig[i];
        s->bp.sval_sorted[r[q]] = v; s->bp.sig_sorted[r[q]] = sg;
        s->bp.rank_mm[2 * (sg & 0xff) + ((sg & 0x100) ? 1 : 0)] = r[q];
        e.sort_value()[r[q]] = v; e.sort_ig()[r[q]] = sg;
      }
    }
  }
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 31
  }
#endif
  if (tl == 0 && first) e.first_time()[0] = 0;
  team_sync();
  PH(31)
  // ---- candidate pairs: every valid geom pair is tested by one lane ----
  int n_cand = 0;
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 32
  for (int rep = 0; rep < 2; ++rep) { n_cand = 0; team_sync();
#endif
  // the pair table is fetched for all rounds up front and every LDS operand of a test is read unconditionally, so that the reads of a
  // round are in flight together; only the ballot compaction is sequential
  unsigned cmask = 0; int key_[NPI];
  // normal_cache[pidx] := 0 for every separated pair: the lanes' verdicts of a round are collected with one ballot (pair pidx = it * T + tl is bit pidx of the
  // mask) and lane w gathers word w, instead of one LDS atomic per pair (16 lanes of a team on one word)
  unsigned ncv_clear = 0u;
  static_assert(NCV <= T, "one lane per word of the normal-cache mask");
#pragma unroll
  for (int it = 0; it < NPI; ++it) {                                   // the rounds are independent: candidates are only marked here
    const int pidx = it * T + tl;
    const int packed = packed_[it];
    const int a = (packed < 0) ? 0 : (packed & 0xff), bg = (packed < 0) ? 0 : (packed >> 8);
    const int2 rka = *(const int2*)&s->bp.rank_mm[2 * a], rkb = *(const int2*)&s->bp.rank_mm[2 * bg];
    const float4 amn = *(const float4*)&s->bp.amin[4 * a], amx = *(const float4*)&s->bp.amax[4 * a], bmn = *(const float4*)&s->bp.amin[4 * bg], bmx = *(const float4*)&s->bp.amax[4 * bg];
    const int ra = rka.x, rb = rkb.x;
    const int rs = (ra < rb) ? rb : ra, rf = (ra < rb) ? ra : rb;
    const int rmax_first = (ra < rb) ? rka.y : rkb.y;
    key_[it] = rs * 64 + rf;
    const bool swept = packed >= 0 && rs < rmax_first;
    const bool any1 = (amx.x <= bmn.x) || (amx.y <= bmn.y) || (amx.z <= bmn.z);
    const bool any2 = (amn.x >= bmx.x) || (amn.y >= bmx.y) || (amn.z >= bmx.z);
    if (swept && !(any1 || any2)) cmask |= 1u << it;
    const unsigned long long sep = team_ballot<T>(swept && (any1 || any2));
    (void)pidx;
    if constexpr (T < 32) { if (tl == ((it * T) >> 5)) ncv_clear |= (unsigned)(sep << ((it * T) & 31)); }
    else if constexpr (T == 32) { if (tl == it) ncv_clear |= (unsigned)sep; }
    else { if (tl == 2 * it) ncv_clear |= (unsigned)sep; if (tl == 2 * it + 1) ncv_clear |= (unsigned)(sep >> 32); }
  }
  if (tl < NCV) s->ncv[tl] &= ~ncv_clear;
  {                                                                    // compaction; the list is sorted by key below, so its order is free
    const int mine = __popc(cmask);
    s->cnt[tl] = mine;
    team_sync();
    int pos = 0, tot = 0;
    for (int l = 0; l < T; ++l) { int c = s->cnt[l]; pos += (l < tl) ? c : 0; tot += c; }
    team_sync();
#pragma unroll
    for (int it = 0; it < NPI; ++it)
      if (cmask & (1u << it)) { if (pos < MAXB) { s->bp.cand_key[pos] = key_[it]; s->bp.cand_pair[pos] = packed_[it]; } pos++; }
    n_cand = tot;
  }
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 32
  }
#endif
  // the serial sweep stops appending at max_broad_pairs (broadphase.py:330-338): all candidates are ranked by the sweep key first, the list is
  // clipped afterwards, so the pairs that survive are the ones the sweep reaches first
  if (n_cand > m.max_broad_pairs && tl == 0) atomicOr(&e.err()[0], GO2SIM_ERR_OVERFLOW_CANDIDATE_CONTACTS);
  n_cand = imn(n_cand, MAXB);
  team_sync();
  for (int c = tl; c < n_cand; c += T) {
    int key = s->bp.cand_key[c], r = 0;
    for (int j = 0; j < n_cand; ++j) r += s->bp.cand_key[j] < key;
    s->pair_sorted[r] = (unsigned short)s->bp.cand_pair[c];
  }
  const int n_broad = imn(n_cand, m.max_broad_pairs);
  team_sync();
  for (int c = tl; c < n_broad; c += T) { int pk = s->pair_sorted[c]; e.broad()[2 * c] = pk & 0xff; e.broad()[2 * c + 1] = pk >> 8; }
  PH(32)
  // ---- func_narrow_phase_convex_vs_convex (narrowphase.py:964-1068), then func_narrow_phase_any_vs_terrain (:1197-1244): one lane per
  //      pair, ordered compaction; the terrain pass appends after all convex-convex contacts, as the two reference kernels do ----
  int nc_run = 0;
  const int n_np_iter = (n_broad + T - 1) / T;
  for (int it = 0; it < n_np_iter; ++it) {
    int ip = it * T + tl;
    ContactStage cs; cs.st = &s->stage[tl][0][0]; cs.n = 0;
    int i_ga = 0, i_gb = 0;
    bool convex_pair = false;
    if (ip < n_broad) {
      int pk = s->pair_sorted[ip];
      i_ga = pk & 0xff; i_gb = pk >> 8;
      if (m.geoms[i_ga].type > m.geoms[i_gb].type) { int t = i_ga; i_ga = i_gb; i_gb = t; }
      convex_pair = m.geoms[i_gb].type != GEOM_TERRAIN;
    }
#ifndef GO2SIM_GJK_SERIAL
    if constexpr (T == 16) {
      // MPR of every pair on its own lane; then the pairs whose MPR answer has to be replaced by safe GJK + EPA (narrowphase.py:727-845) are
      // answered by QUADS: the four quads of the team take four flagged pairs at a time (the four feet of a landing robot), the four lanes of a
      // quad cooperate on their query (csrc/go2sim_gjk_dev.h, cooperative form, W = 4); then every lane finishes its pair
      static_assert(GJK_SLOTS >= 4, "one LDS polytope slot per quad");
      CcState cst;
      cst.want_gjk = false;
      if (convex_pair) cc_detect0(m, e, i_ga, i_gb, s->ncv, cst);
      const bool wants = convex_pair && cst.want_gjk;
      const unsigned want = dgc_ballot<16>(wants);
      const int n_want = __popc(want), quad = tl >> 2, ql = tl & 3;
      for (int r0 = 0; r0 < n_want; r0 += 4) {                          // team-uniform rounds
        unsigned rest = want;
        for (int k = 0; k < r0 + quad; ++k) rest &= rest - 1u;           // drop the pairs taken by earlier rounds / lower quads
        const int L = rest ? __ffs((int)rest) - 1 : -1;                  // the lane whose pair my quad answers in this round
        const int Ls = L < 0 ? tl : L;
        DgPair dp;                                                       // (all 16 lanes shuffle: the source lanes have to be active)
        dp.m = mp; dp.i_ga = __shfl(i_ga, Ls, 16); dp.i_gb = __shfl(i_gb, Ls, 16);
        dp.pos_a = v3(__shfl(cst.pr.pos_a.x, Ls, 16), __shfl(cst.pr.pos_a.y, Ls, 16), __shfl(cst.pr.pos_a.z, Ls, 16));
        dp.quat_a = q4(__shfl(cst.pr.quat_a.w, Ls, 16), __shfl(cst.pr.quat_a.x, Ls, 16), __shfl(cst.pr.quat_a.y, Ls, 16), __shfl(cst.pr.quat_a.z, Ls, 16));
        dp.pos_b = v3(__shfl(cst.pr.pos_b.x, Ls, 16), __shfl(cst.pr.pos_b.y, Ls, 16), __shfl(cst.pr.pos_b.z, Ls, 16));
        dp.quat_b = q4(__shfl(cst.pr.quat_b.w, Ls, 16), __shfl(cst.pr.quat_b.x, Ls, 16), __shfl(cst.pr.quat_b.y, Ls, 16), __shfl(cst.pr.quat_b.z, Ls, 16));
        if (L >= 0) {
          dp.ga = geom_lite(m, dp.i_ga); dp.gb = geom_lite(m, dp.i_gb); dp.ra = make_rot(dp.quat_a); dp.rb = make_rot(dp.quat_b);
          dp.discrete = dp.ga.type == GEOM_BOX && dp.gb.type == GEOM_BOX;  // func_is_discrete_geoms, collider/utils.py:105-126
          PHD_BEGIN
          DgResult gr = dgc_contact<4>(dp, s->gjk[quad], m.eps, ql);
          if (gr.overflow) gr = dgc_contact<4>(dp, gjk_scratch[(size_t)b * T + L], m.eps, ql);   // outgrew the LDS slot: the same query on the full-capacity record
          PHD(34)
          if (ql == 0) {
            float* o = s->gjk_res[L];
            o[0] = gr.is_col ? 1.0f : 0.0f; o[1] = gr.penetration; o[2] = gr.normal.x; o[3] = gr.normal.y; o[4] = gr.normal.z; o[5] = gr.pos.x; o[6] = gr.pos.y; o[7] = gr.pos.z;
          }
        }
      }
      team_sync();
      if (wants) {
        atomicAdd(&e.gjk_fallback()[0], 1);
        const float* o = s->gjk_res[tl];
        cst.is_col = o[0] != 0.0f;
        cst.penetration = o[1];
        if (cst.is_col) { cst.normal = v3(o[2], o[3], o[4]); cst.contact_pos = v3(o[5], o[6], o[7]); }
      }
      if (convex_pair) cc_rest(m, e, cst, cs, nullptr, &s->gjk_slot_mask, &gjk_scratch[(size_t)b * T + tl], s->ncv);
    } else
#endif
    if (convex_pair) convex_convex_contact_staged(m, e, i_ga, i_gb, cs, s->gjk, &s->gjk_slot_mask, &gjk_scratch[(size_t)b * T + tl], s->ncv);
    s->cnt[tl] = cs.n;
    team_sync();
    int off = 0, tot = 0;
    for (int l = 0; l < T; ++l) { int c = s->cnt[l]; off += (l < tl) ? c : 0; tot += c; }
    for (int k = 0; k < cs.n; ++k) {                                   // func_add_contact, contact.py:165-199
      int i_c = nc_run + off + k;
      if (i_c < m.max_contact_pairs) {
        const float* p = cs.st + 7 * k;
        float friction_a = e.geom_friction()[i_ga] * e.friction_ratio()[i_ga];
        float friction_b = e.geom_friction()[i_gb] * e.friction_ratio()[i_gb];
        e.c_geom()[i_c] = i_ga; e.c_geom()[MAXC + i_c] = i_gb;
        e.c_normal()[i_c] = v3(p[0], p[1], p[2]); e.c_pos()[i_c] = v3(p[3], p[4], p[5]); e.c_pen()[i_c] = p[6];
        e.c_friction()[i_c] = fmx(fmx(friction_a, friction_b), 1e-2f);
        auto sol = e.c_sol()[i_c];
        for (int q = 0; q < 7; ++q) sol[q] = 0.5f * (m.geoms[i_ga].sol_params[q] + m.geoms[i_gb].sol_params[q]);
        e.c_link()[i_c] = m.geoms[i_ga].link; e.c_link()[MAXC + i_c] = m.geoms[i_gb].link;
      } else {
        atomicOr(&e.err()[0], GO2SIM_ERR_OVERFLOW_COLLISION_PAIRS);
      }
    }
    nc_run += tot;
    team_sync();
  }
  // ---- func_narrow_phase_any_vs_terrain (narrowphase.py:1197-1244): appended after all convex-convex contacts.  One lane per heightfield
  //      prism: the pairs enumerate the prisms their geom can reach, the MPR queries run T at a time, and the accept / dedupe / cap logic of
  //      the serial loop is replayed in prism order, which reproduces the contact list of the reference exactly. ----
  if (m.terrain_enabled) {
    PH(33)
    int n_list = 0, i_terrain = 0;                                      // terrain pairs, in broad-phase order
    for (int ip = 0; ip < n_broad; ++ip) {
      int pk = s->pair_sorted[ip];
      int i_ga = pk & 0xff, i_gb = pk >> 8;
      if (m.geoms[i_ga].type == GEOM_TERRAIN) { int t = i_ga; i_ga = i_gb; i_gb = t; }
      if (m.geoms[i_gb].type != GEOM_TERRAIN) continue;
      if (n_list < NG) { if (tl == 0) s->tr.tp[n_list].i_ga = i_ga; i_terrain = i_gb; n_list++; }
    }
    team_sync();
    // pairs whose geom is out of reach of the heightfield are dropped before their bounding boxes are computed (they have no eligible cell); the
    // list is compacted in place, in order (a round reads its entries before any lane of the round writes)
    int n_tp = 0;
    for (int base = 0; base < n_list; base += T) {
      const int p = base + tl;
      int i_ga = 0; bool keep = false;
      if (p < n_list) {
        i_ga = s->tr.tp[p].i_ga;
        V3 pos_t; Q4 quat_t;
        transform_pos_quat_by_trans_quat((V3)e.g_pos()[i_ga] - (V3)e.g_pos()[i_terrain], e.g_quat()[i_ga], v3(0, 0, 0), inv_quat(e.g_quat()[i_terrain]), pos_t, quat_t);
        keep = m.terrain_cmax == nullptr || !terrain_pair_out_of_reach(m, geom_lite(m, i_ga), pos_t);
      }
      const unsigned long long mk = team_ballot<T>(keep);
      team_sync();
      if (keep) s->tr.tp[n_tp + __popcll(mk & ((1ull << tl) - 1ull))].i_ga = i_ga;
      n_tp += __popcll(mk);
      team_sync();
    }
    for (int p = tl; p < n_tp; p += T) {                                // pair setup: pose in the terrain frame, cell range, dedupe tolerance
      auto& t = s->tr.tp[p];
      terrain_pair_setup(m, e, t.i_ga, i_terrain, t);
      t.n_items = imx(0, t.r_max - t.r_min) * imx(0, 2 * (t.c_max - t.c_min + 1) - 2);   // prisms under the geom's bounding box ("cells")
      t.tol = compute_tolerance(m, t.i_ga, i_terrain, m.mc_tolerance);
    }
    team_sync();
    PH(26)
    int* items = (int*)&gjk_scratch[(size_t)b * T];                     // prism descriptors p | r << 5 | k << 18 (the GJK scratch is idle in this pass)
    const int items_cap = (int)(sizeof(GjkStoreFull) * T / sizeof(int));
    int n_cells = 0;
    for (int p = 0; p < n_tp; ++p) n_cells += s->tr.tp[p].n_items;
    team_sync();
    { int off = 0; for (int p = 0; p < n_tp; ++p) { int c = s->tr.tp[p].n_items; if (tl == 0) s->tr.tp[p].item_off = off; off += c; } }   // first cell of the pair
    team_sync();
    // One lane per cell, TU cells per lane in flight (the heights come from L2 / HBM: the loads of a batch are issued together).  A cell is a
    // descriptor iff one of the three top vertices of its prism reaches the geom (narrowphase.py:430-436); the descriptors are compacted in
    // (pair, row, vertex) order, the order in which the serial loop meets them.
    int n_items = 0;
    {
      constexpr int TU = 4;
      int p_cur = 0, p_end = n_tp > 0 ? s->tr.tp[0].n_items : 0;        // cells [.., p_end) belong to pairs <= p_cur (per-lane cursor: a lane's cells ascend)
      for (int base = 0; base < n_cells; base += T * TU) {
        int desc[TU]; bool el[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int ci = base + u * T + tl;
          el[u] = false; desc[u] = 0;
          if (ci < n_cells) {
            while (ci >= p_end) { ++p_cur; p_end += s->tr.tp[p_cur].n_items; }
            const auto& t = s->tr.tp[p_cur];
            const int nkk = 2 * (t.c_max - t.c_min + 1) - 2;
            const int local = ci - (p_end - t.n_items);
            const int r = t.r_min + local / nkk, k = 2 + local % nkk;
            const float z0 = terrain_strip_z(m, r, t.c_min, k - 2), z1 = terrain_strip_z(m, r, t.c_min, k - 1), z2 = terrain_strip_z(m, r, t.c_min, k);
            el[u] = (z0 >= t.zmin) | (z1 >= t.zmin) | (z2 >= t.zmin);
            desc[u] = p_cur | (r << 5) | (k << 18);
          }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const unsigned long long mk = team_ballot<T>(el[u]);
          if (el[u]) { const int q = n_items + __popcll(mk & ((1ull << tl) - 1ull)); if (q < items_cap) items[q] = desc[u]; }
          n_items += __popcll(mk);
        }
      }
    }
    if (n_items > items_cap) n_items = items_cap;
    team_sync();
    // About half of the queries end at the first support point of the portal search (the prism lies beside the geom).  That step is evaluated for
    // every descriptor here, and only the survivors are distributed for the full query (compacted in place, in order: the write position never
    // passes the read position).  A dropped descriptor is one whose query returns "no contact", so the contact list does not change.
    {
      int n_keep = 0;
      for (int base = 0; base < n_items; base += T) {
        const int q = base + tl;
        int d = 0; bool keep = false;
        if (q < n_items) {
          d = items[q];
          keep = !terrain_prism_separated_at_once(m, s->tr.tp[d & 31], i_terrain, (d >> 5) & 0x1fff, d >> 18);
        }
        const unsigned long long mk = team_ballot<T>(keep);
        if (keep) items[n_keep + __popcll(mk & ((1ull << tl) - 1ull))] = d;
        n_keep += __popcll(mk);
        team_sync();
      }
      n_items = n_keep;
    }
    PH(27)
    int cur_p = -1, n_con = 0;                                          // replay state (identical on every lane)
    float tolerance = 0.0f;
    V3 acc[5];                                                          // contacts already accepted for the current terrain pair (dedupe); team-uniform
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = v3(0, 0, 0);
    for (int base = 0; base < n_items; base += T) {
      const int q = base + tl;
      float* st = &s->stage[tl][0][0];
      int has = 0;
      if (q < n_items) {
        int d = items[q];
        const auto& t = s->tr.tp[d & 31];
        V3 normal, cpos; float pen;
        if (terrain_prism_contact(m, e, t, i_terrain, (d >> 5) & 0x1fff, d >> 18, normal, cpos, pen)) {
          has = 1 + (d & 31);
          st[0] = normal.x; st[1] = normal.y; st[2] = normal.z; st[3] = cpos.x; st[4] = cpos.y; st[5] = cpos.z; st[6] = pen;
        }
      }
      s->cnt[tl] = has;                                                  // 0 = no contact, else 1 + pair slot
      team_sync();
      PH(28)
      // replay of the serial accept / dedupe / cap logic over the prisms of this round that produced a contact (team-uniform: every lane reads the
      // same staged values); the j-th accepted contact of the round is then written out by lane j
      unsigned long long hm = team_ballot<T>(has != 0);
      int n_acc = 0, my_l = -1, my_ic = 0;
      while (hm) {
        const int l = __ffsll((long long)hm) - 1;
        hm &= hm - 1ull;
        const int p = s->cnt[l] - 1;
        if (p != cur_p) { cur_p = p; n_con = 0; tolerance = s->tr.tp[p].tol; }
        if (n_con >= m.n_contacts_per_pair) continue;
        const float* pc = &s->stage[l][0][0];
        const V3 cpos = v3(pc[3], pc[4], pc[5]);
        bool valid = true;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)                                  // acc[jj] is contact nc_run - n_con + jj of the list
          if (jj < n_con && nc_run - n_con + jj < m.max_contact_pairs && norm(cpos - acc[jj]) < tolerance) valid = false;
        if (!valid) continue;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) if (jj == n_con) acc[jj] = cpos;
        if (tl == n_acc) { my_l = l; my_ic = nc_run; }
        n_acc++; nc_run++; n_con++;
      }
      if (my_l >= 0) {
        if (my_ic < m.max_contact_pairs) {                               // func_add_contact, contact.py:165-199
          const float* pc = &s->stage[my_l][0][0];
          const int i_ga = s->tr.tp[s->cnt[my_l] - 1].i_ga, i_c = my_ic;
          float friction_a = e.geom_friction()[i_ga] * e.friction_ratio()[i_ga];
          float friction_b = e.geom_friction()[i_terrain] * e.friction_ratio()[i_terrain];
          e.c_geom()[i_c] = i_ga; e.c_geom()[MAXC + i_c] = i_terrain;
          e.c_normal()[i_c] = v3(pc[0], pc[1], pc[2]); e.c_pos()[i_c] = v3(pc[3], pc[4], pc[5]); e.c_pen()[i_c] = pc[6];
          e.c_friction()[i_c] = fmx(fmx(friction_a, friction_b), 1e-2f);
          auto sol = e.c_sol()[i_c];
          for (int qq = 0; qq < 7; ++qq) sol[qq] = 0.5f * (m.geoms[i_ga].sol_params[qq] + m.geoms[i_terrain].sol_params[qq]);
          e.c_link()[i_c] = m.geoms[i_ga].link; e.c_link()[MAXC + i_c] = m.geoms[i_terrain].link;
        } else {
          atomicOr(&e.err()[0], GO2SIM_ERR_OVERFLOW_COLLISION_PAIRS);
        }
      }
      team_sync();                                                       // stage / cnt are rewritten by the next round
      PH(29)
    }
  }
  if (tl == 0) {
    e.n_broad()[0] = n_broad; e.n_contacts()[0] = imn(nc_run, m.max_contact_pairs);
    s->cnt[0] = imn(nc_run, m.max_contact_pairs);
  }
  team_sync();
  if (lpt_rec && tl == 0 && b % solver_epw == 0) {                       // the first env of a solver block files it (the block's envs sit in this wavefront)
    int key = s->cnt[0];
    for (int k = 1; k < solver_epw; ++k) if (b + k < P.B && slot + k < EPW) key = imx(key, lds[slot + k].cnt[0]);
    lpt_file(lpt_rec, lpt_cap, b / solver_epw, P.B, solver_epw, key);
  }
  for (int i = tl; i < NCV; i += T) e.ncache_valid()[i] = (int)s->ncv[i];
  PH(33)
}

// ---------------------------------------------------------------------------------------------
// constraints + Newton solver  (R/constraint/solver.py)
// ---------------------------------------------------------------------------------------------
// ---- exact line search helpers, solver.py:1888-2417 ----
struct LsPoint { float alpha, cost, grad, hess; };
#ifndef GO2SIM_BRACKET_INLINE
#define GO2SIM_BRACKET_ATTR DEVN
#else
#define GO2SIM_BRACKET_ATTR DEV
#endif
GO2SIM_BRACKET_ATTR int update_bracket(LsPoint& p, const float alphas[3], const float costs[3], const float grads[3], const float hess[3], float& p_next_alpha) {
  int flag = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p.grad < 0 && grads[i] < 0 && p.grad < grads[i]) { p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 1; }
    else if (p.grad > 0 && grads[i] > 0 && p.grad > grads[i]) { p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 2; }
  }
  p_next_alpha = p.alpha;
  if (flag > 0) p_next_alpha = p.alpha - p.grad / p.hess;
  return flag;
}

// The out-of-line form of the product: everything travels BY VALUE (arguments and the result in vector registers under the AMDGPU calling convention),
// so keeping the bracket step out of line no longer costs a round trip through private memory per call (round 3: LsPoint& and four array pointers,
// 35 memory instructions for 53 vector ones).  -DGO2SIM_BRACKET_BYREF restores the by-reference form for A / B runs.
struct BrOut { float alpha, cost, grad, hess, next_alpha; int flag; };
DEVN BrOut update_bracket_v(float pa, float pc, float pg, float ph, float a0, float a1, float a2, float c0, float c1, float c2, float g0, float g1, float g2,
                            float h0, float h1, float h2) {
  LsPoint p = {pa, pc, pg, ph};
  const float alphas[3] = {a0, a1, a2}, costs[3] = {c0, c1, c2}, grads[3] = {g0, g1, g2}, hess[3] = {h0, h1, h2};
  int flag = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p.grad < 0 && grads[i] < 0 && p.grad < grads[i]) { p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 1; }
    else if (p.grad > 0 && grads[i] > 0 && p.grad > grads[i]) { p.alpha = alphas[i]; p.cost = costs[i]; p.grad = grads[i]; p.hess = hess[i]; flag = 2; }
  }
  float next_alpha = p.alpha;
  if (flag > 0) next_alpha = p.alpha - p.grad / p.hess;
  BrOut o = {p.alpha, p.cost, p.grad, p.hess, next_alpha, flag};
  return o;
}

// ---------------------------------------------------------------------------------------------
// Team solver: T lanes cooperate on one environment, 64/T environments per wavefront, one wavefront per
// workgroup.  The whole working set of the Newton solve (Jacobian rows, Hessian / Cholesky factor, mass
// matrix, dof and row vectors) lives in LDS; HBM is touched once to stage the inputs and once to commit the
// results.  Arithmetic is the serial reference order (solver.py), only *independent* outputs (rows, dofs,
// Hessian entries) are spread over the lanes; every sum is still evaluated first-to-last by one lane, and
// team-uniform scalars (costs, line-search points) are evaluated redundantly by all lanes of the team, so the
// results are bit-identical to the one-lane-per-env formulation and to the CPU oracle.
// Environments with more than RL rows fall back to the same code on a per-env global scratch block.
// ---------------------------------------------------------------------------------------------
constexpr int RL = 32;        // constraint rows held in LDS on flat ground (8 contacts; typical walking uses 16)
constexpr int RL_TERRAIN = 96;  // ... and on heightfield terrain (several prism contacts per foot / shank): one env per wavefront
constexpr int DS = 20;  // padded stride of dof-indexed rows in the solver working set (16-byte aligned rows => wide LDS accesses)

template <int R>
struct alignas(16) SolverData {
  float J[R * DS];
  float H[ND * DS];   // lower triangle: Hessian, then its Cholesky factor; the strict upper triangle mirrors it so that columns read as rows
  float M[ND * DS];
  float qacc[DS], Ma[DS], grad[DS], Mgrad[DS], search[DS], mv[DS], force[DS], acc_smooth[DS], qfrc[DS], ntv[DS], vel[DS];
  // cdof / root_com are read while the rows are built and are dead afterwards; `rot` belongs to the rank-1 pipeline (strict build only).  FAST ORDER
  // keeps the unfactored Hessian of the Newton solve there, one 3 x 3 block per lane of ts_hessian_direct (Ablk[blk][a][q]): no LDS beyond the 8-per-CU budget
  union {
    struct { float cdof_ang[ND * 3], cdof_vel[ND * 3], root_com[NL * 3]; alignas(16) float rot[ND][4]; };
    float Ablk[21 * 9];
  };
  alignas(16) float aref[R];   // the row-indexed arrays are 16-byte aligned: the redundant passes over the rows use 128-bit LDS reads
  alignas(16) float efc_D[R]; alignas(16) float Jaref[R]; alignas(16) float jv[R]; alignas(16) float efc_force[R];
  alignas(16) float qf0[R]; alignas(16) float qf1[R]; alignas(16) float qf2[R]; alignas(16) float DA[R];
  alignas(16) int active[R]; alignas(16) int prev_active[R];
  // (rot[ND][4] of the union above: pipelined rank-1 updates: rotation (c, 1/c, s) computed by the lane that owns a column, read by the rows below it)
};

// value of lane k of the caller's team (k is a compile-time constant after unrolling): v_readlane through an SGPR instead of a
// ds_bpermute round trip through the LDS crossbar
template <int T>
DEV float team_bcast(float x, const int k) {
  if constexpr (T == 64) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
  } else if constexpr (T == 32) {
    const int lo = __builtin_amdgcn_readlane(__float_as_int(x), k), hi = __builtin_amdgcn_readlane(__float_as_int(x), 32 + k);
    return __int_as_float((threadIdx.x & 32) ? hi : lo);
  } else {
    return __shfl(x, k, T);
  }
}
// value of lane k of the caller's team where k is wave-uniform but only known at run time (v_readlane with the lane in an SGPR)
template <int T>
DEV float team_bcast_dyn(float x, int k_uniform) {
  static_assert(T == 32 || T == 64, "teams of 32 or 64 lanes");
  const int k = __builtin_amdgcn_readfirstlane(k_uniform);
  if constexpr (T == 64) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
  } else {
    const int lo = __builtin_amdgcn_readlane(__float_as_int(x), k), hi = __builtin_amdgcn_readlane(__float_as_int(x), 32 + k);
    return __int_as_float((threadIdx.x & 32) ? hi : lo);
  }
}
// ---- wavefront-level sums in the SERIAL order of the reference ----------------------------------------------------------------------------
// The reference (and the oracle) accumulate over constraint rows / dofs first to last: t = ((base + x_0) + x_1) + ... .  A tree reduction would
// change the rounding; instead the chain itself is moved onto the lanes: lane c holds x_c, and one DPP instruction `acc = acc[lane - 1] + x`
// repeated n - 1 times leaves the exact serial prefix sums in the lanes (lane c becomes final in step c and recomputes the same value
// afterwards; lanes past the last term add +0 and carry the total on).  DPP row shifts work inside rows of 16 lanes, so every 16 terms the
// running sum is handed to the next row with a broadcast.  NQ sums are interleaved (independent chains fill each other's latency).  Cost per sum
// ~ n instructions in total instead of ~ n per lane-visible term of a redundant scalar loop.
DEV float dpp_row_shr1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true)); }   // lane 0 of a row reads 0
template <int T, int NQ>
DEV void team_serial_sum(const float (&x)[NQ], const float (&base)[NQ], int tl, int nseg, float (&total)[NQ]) {
  float acc[NQ], xx[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { acc[q] = (tl == 0) ? base[q] + x[q] : x[q]; xx[q] = acc[q]; }
  for (int seg = 0; seg < nseg; ++seg) {
    if (seg > 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float carry = team_bcast_dyn<T>(acc[q], 16 * seg - 1);
        if (tl == 16 * seg) { acc[q] = carry + x[q]; xx[q] = acc[q]; }
      }
    }
#pragma unroll
    for (int st = 1; st < 16; ++st) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] = dpp_row_shr1(acc[q]) + xx[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) total[q] = team_bcast_dyn<T>(acc[q], 16 * nseg - 1);
}

// ---- FAST ORDER (default): lane-parallel reductions and reciprocal forms on the critical path of the Newton solve -----------------------------
// north_star allows a float32 tolerance on floats.  The strict build (-DGO2SIM_FAST_ORDER=0) keeps every sum in the reference's first-to-last
// order (team_serial_sum above); the product build replaces the chains that the slowest environment of a launch waits for:
//   * sums over constraint rows / dofs: every lane adds its own terms (row c on lane c % T), then a butterfly over the lanes -- 4 DPP steps inside a
//     row of 16 lanes (xor 1, xor 2, mirror in 8, mirror in 16) and one add per pair of rows -- instead of a 16..32-step dependent chain;
//   * triangular solves: reciprocal diagonal (18 divisions side by side instead of 36 in sequence; the reference's LDL^T path stores D_inv the same
//     way, forward_dynamics.py:545-687), column-oriented with one lane per row;
//   * rank-1 Cholesky rotations: 1 / r = r * (1 / tmp) with the division issued beside the square root, c = r * (1 / L_kk), s = v_k * (1 / L_kk) with the
//     reciprocal diagonal carried from update to update.
// The FAST ORDER build of the CPU oracle (oracle/go2sim_cpu.cpp with -DGO2SIM_FAST_ORDER) mirrors this arithmetic operation for operation (tolerance 0 in the GPU
// parity tests); tests/test_fast_order.py bounds fast against strict.
// (GO2SIM_FAST_ORDER / REBUILD_FLIPS defaults and dpp_perm: defined with the arrow-form helpers above the dynamics)
// butterfly sum over the T lanes of a team (every lane ends with the total); `x` = the lane's own partial sum
template <int T, int NQ>
DEV void team_tree_sum(const float (&x)[NQ], float (&total)[NQ]) {
  static_assert(T == 32 || T == 64, "teams of 32 or 64 lanes");
  float a[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = x[q];
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = a[q] + dpp_perm<0xB1>(a[q]);        // quad_perm [1,0,3,2]: lane ^ 1
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = a[q] + dpp_perm<0x4E>(a[q]);        // quad_perm [2,3,0,1]: lane ^ 2
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = a[q] + dpp_perm<0x141>(a[q]);       // row_half_mirror: lane ^ 7
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = a[q] + dpp_perm<0x140>(a[q]);       // row_mirror: lane ^ 15
  // every lane of a row now holds its row's sum: add the rows.  gfx950 exchanges whole rows between two registers in one instruction
  // (v_permlane16_swap_b32: odd rows of the first operand <-> even rows of the second; v_permlane32_swap_b32: the upper half of the first <-> the lower half
  // of the second), so "row 0 + row 1" costs copy + swap + add instead of two v_readlane, two v_mov and a v_cndmask per value.  Same operands as
  // team_bcast(a, 0) + team_bcast(a, 16) (IEEE addition commutes), hence the same bits.  Measured: solver 0.1421 -> 0.1370 ms per step in the window.
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#ifndef GO2SIM_TREE_SUM_READLANE
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[q]), __float_as_uint(a[q]), false, false);
    const float pair = __uint_as_float(r[0]) + __uint_as_float(r[1]);                       // rows 0 + 1 (lanes 0..31), rows 2 + 3 (lanes 32..63)
    if constexpr (T == 32) total[q] = pair;
    else { const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(pair), __float_as_uint(pair), false, false); total[q] = __uint_as_float(h[0]) + __uint_as_float(h[1]); }
#else
    if constexpr (T == 32) total[q] = team_bcast<T>(a[q], 0) + team_bcast<T>(a[q], 16);
    else total[q] = (team_bcast<T>(a[q], 0) + team_bcast<T>(a[q], 16)) + (team_bcast<T>(a[q], 32) + team_bcast<T>(a[q], 48));
#endif
  }
}
template <int T>
DEV float team_tree_sum1(float x) { const float xs[1] = {x}; float t[1]; team_tree_sum<T, 1>(xs, t); return t[0]; }

template <int T, class S, class MT>
DEV void ts_update_constraint(const MT& m, S* s, int tl, int n_con, float& cost, float& prev_cost, float& gauss) {
  prev_cost = cost;
  float row_cost = 0.0f;                                               // 0.5 * Jaref^2 * D * active of the lane's row (rows one per lane)
  bool first_row = true;                                               // (FAST ORDER: the lane's partial sum over its rows c = tl, tl + T, ...)
  for (int c = tl; c < n_con; c += T) {
    s->prev_active[c] = s->active[c];
    float Ja = s->Jaref[c];
    int act = Ja < 0.0f;
    s->active[c] = act;
    float D = s->efc_D[c];
    const float DA = D * (float)act;
    s->DA[c] = DA;
    s->efc_force[c] = 0.0f + (-Ja * D * (float)act);
    const float rc = 0.5f * (Ja * Ja * DA);
#if GO2SIM_FAST_ORDER
    row_cost = first_row ? rc : row_cost + rc;
    first_row = false;
#else
    row_cost = rc;
#endif
  }
  team_sync();
  for (int d = tl; d < ND; d += T) {
    float q = 0.0f;
#pragma unroll 16
    for (int c = 0; c < n_con; ++c) q = q + s->J[c * DS + d] * s->efc_force[c];
    s->qfrc[d] = q;
  }
  bool done = false;
#if GO2SIM_FAST_ORDER
  if constexpr (T >= 32) {
    const int d = tl < ND ? tl : ND - 1;
    const float v = 0.5f * (s->Ma[d] - s->force[d]) * (s->qacc[d] - s->acc_smooth[d]);
    const float xs[2] = {tl < ND ? v : 0.0f, row_cost};
    float tot[2];
    team_tree_sum<T, 2>(xs, tot);
    gauss = tot[0]; cost = tot[1] + tot[0];
    done = true;
  }
#endif
  if constexpr (T >= 32 && !GO2SIM_FAST_ORDER) {
    const int n_wave = __builtin_amdgcn_readfirstlane(imx(__shfl(n_con, 0), __shfl(n_con, T == 64 ? 0 : 32)));
    if (n_wave <= T) {                                                 // serial-order lane scans: the 18 dof terms, then the rows on top of them
      const int d = tl < ND ? tl : ND - 1;
      const float v = 0.5f * (s->Ma[d] - s->force[d]) * (s->qacc[d] - s->acc_smooth[d]);
      const float xd[1] = {tl < ND ? v : 0.0f}, zero[1] = {0.0f};
      float g[1], ct[1];
      team_serial_sum<T, 1>(xd, zero, tl, 2, g);
      const float xr[1] = {tl < n_con ? row_cost : 0.0f};
      team_serial_sum<T, 1>(xr, g, tl, (n_wave + 15) / 16, ct);
      gauss = g[0]; cost = (n_wave > 0) ? ct[0] : g[0];
      done = true;
    }
  }
  if (!done) {
    float cost_i = 0.0f, gauss_i = 0.0f;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      float v = 0.5f * (s->Ma[d] - s->force[d]) * (s->qacc[d] - s->acc_smooth[d]);
      gauss_i = gauss_i + v;
      cost_i = cost_i + v;
    }
#pragma unroll 16
    for (int c = 0; c < n_con; ++c) { float Ja = s->Jaref[c]; cost_i = cost_i + 0.5f * (Ja * Ja * s->DA[c]); }
    gauss = gauss_i; cost = cost_i;
  }
  team_sync();
}

// func_hessian_direct_batch, solver.py:1285-1343.
#if GO2SIM_FAST_ORDER
// FAST ORDER: one lane per 3 x 3 block of the lower triangle (21 blocks), so that a constraint row costs 6 LDS reads + 1 for 9 multiply-adds instead of
// 3 reads per multiply-add; the row factor J[c][i] * D[c] * active[c] (0 where the reference skips the row: |J[c][i]| <= eps) is formed once per block
// row and the sum runs as a fused multiply-add chain over the rows first to last.  Teams of 64 lanes split the rows in three interleaved groups
// (rows c = g mod 3 on lanes 21 g .. 21 g + 20) and add the partial sums as (p0 + p1) + p2.
template <int T, class S, class MT>
DEV void ts_hessian_direct(const MT& m, S* s, int tl, int n_con) {
  constexpr int NB = 21, G = (T >= 3 * NB) ? 3 : 1;
  for (int u0 = 0; u0 < NB * G; u0 += T) {
    const int u = u0 + tl;
    const bool on = u < NB * G;
    const int g = on ? u / NB : 0, blk = on ? u - NB * g : 0;
    const int bi = (blk >= 15) ? 5 : (blk >= 10) ? 4 : (blk >= 6) ? 3 : (blk >= 3) ? 2 : (blk >= 1) ? 1 : 0;
    const int bj = blk - bi * (bi + 1) / 2;
    float h[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int q = 0; q < 3; ++q) h[a][q] = 0.0f;
    const float* Ji = &s->J[3 * bi];
    const float* Jj = &s->J[3 * bj];
    const int n_c = on ? n_con : 0;
#pragma unroll 4
    for (int c = g; c < n_c; c += G) {
      const float D = s->DA[c];
      float jd[3], jj[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { const float j1 = Ji[c * DS + a]; jd[a] = (dm_abs(j1) > m.eps) ? j1 * D : 0.0f; jj[a] = Jj[c * DS + a]; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) h[a][q] = __builtin_fmaf(jj[q], jd[a], h[a][q]);
    }
    if constexpr (G == 3) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) { const float p1 = __shfl(h[a][q], tl + NB, T), p2 = __shfl(h[a][q], tl + 2 * NB, T); h[a][q] = (h[a][q] + p1) + p2; }
    }
    if (on && g == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int i = 3 * bi + a, j = 3 * bj + q;
          const float v = h[a][q] + s->M[i * DS + j];
          if constexpr (T >= NB) s->Ablk[blk * 9 + 3 * a + q] = v;       // the unfactored Hessian stays with its block lane (ts_hessian_update)
          if (j <= i) s->H[i * DS + j] = v;
        }
    }
  }
  team_sync();
}
// The Hessian after a change of the active set: the block lanes add / subtract the rows that flipped (first to last, fused multiply-adds on the stored
// blocks) instead of summing all rows again, and hand the result to the factorisation.  Cost proportional to the flipped rows.
template <int T, class S, class MT>
DEV void ts_hessian_update(const MT& m, S* s, int tl, int n_con) {
  constexpr int NB = 21;
  static_assert(T >= NB, "one lane per block");
  const bool on = tl < NB;
  const int blk = on ? tl : 0;
  const int bi = (blk >= 15) ? 5 : (blk >= 10) ? 4 : (blk >= 6) ? 3 : (blk >= 3) ? 2 : (blk >= 1) ? 1 : 0;
  const int bj = blk - bi * (bi + 1) / 2;
  float h[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 3; ++q) h[a][q] = s->Ablk[blk * 9 + 3 * a + q];
  const float* Ji = &s->J[3 * bi];
  const float* Jj = &s->J[3 * bj];
  for (int base = 0; base < n_con; base += T) {
    const int c_me = base + tl;
    unsigned long long mask = team_ballot<T>(c_me < n_con && ((s->active[c_me] != 0) != (s->prev_active[c_me] != 0)));
    while (mask != 0ull) {
      const int c = base + __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      const float D = s->efc_D[c], sg = (s->active[c] != 0) ? 1.0f : -1.0f;
      float jd[3], jj[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { const float j1 = Ji[c * DS + a]; jd[a] = (dm_abs(j1) > m.eps) ? sg * (j1 * D) : 0.0f; jj[a] = Jj[c * DS + a]; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) h[a][q] = __builtin_fmaf(jj[q], jd[a], h[a][q]);
    }
  }
  if (on) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int i = 3 * bi + a, j = 3 * bj + q;
        s->Ablk[blk * 9 + 3 * a + q] = h[a][q];
        if (j <= i) s->H[i * DS + j] = h[a][q];
      }
  }
  team_sync();
}
#else
// one lane per lower-triangle entry, rows summed first to last
template <int T, class S, class MT>
DEV void ts_hessian_direct(const MT& m, S* s, int tl, int n_con) {
  for (int idx = tl; idx < ND * (ND + 1) / 2; idx += T) {
    int i, j;
    tri_index(m, idx, i, j);
    float h = 0.0f;
#pragma unroll 16
    for (int c = 0; c < n_con; ++c) {
      float j1 = s->J[c * DS + i];
      float t = s->J[c * DS + j] * j1 * s->DA[c];     // (row[j] * j1 * D) * active, active in {0,1}
      h = (dm_abs(j1) > m.eps) ? (h + t) : h;
    }
    h = h + s->M[i * DS + j];
    s->H[i * DS + j] = h;
  }
  team_sync();
}
#endif

// func_cholesky_factor_direct_batch, solver.py:1467-1494.
// column by column; statically unrolled so that the pivot-row prefix is fetched with wide LDS reads that are all in flight at once
template <int T, class S, class MT>
DEV void ts_cholesky_factor_columns(const MT& m, S* s, int tl) {
#pragma unroll
  for (int i_d = 0; i_d < ND; ++i_d) {
    float pr[ND];
#pragma unroll
    for (int k = 0; k < i_d; ++k) pr[k] = s->H[i_d * DS + k];
    float tmp = s->H[i_d * DS + i_d];
#pragma unroll
    for (int k = 0; k < i_d; ++k) tmp = tmp - pr[k] * pr[k];
    float dgn = dm_sqrt(fmx(tmp, m.eps));
    float inv = 1.0f / dgn;
    if (tl == 0) s->H[i_d * DS + i_d] = dgn;
    for (int j_d = i_d + 1 + tl; j_d < ND; j_d += T) {
      float dotv = 0.0f;
#pragma unroll
      for (int k = 0; k < i_d; ++k) dotv = dotv + s->H[j_d * DS + k] * pr[k];
      float v = (s->H[j_d * DS + i_d] - dotv) * inv;
      s->H[j_d * DS + i_d] = v; s->H[i_d * DS + j_d] = v;
    }
    team_sync();
  }
}
// FAST ORDER, teams of at least ND lanes: right-looking form with one lane per row, the row in registers.  Column k: the diagonal of lane k gives
// d = sqrt(max(a_kk, eps)) and 1 / d; every row below scales its element (L_jk = a_jk / d as a multiplication) and stores it at its MIRROR position
// H[k][j] -- so row k of the upper triangle is column k of the factor, the layout the triangular solves read -- and after one LDS round trip every
// row subtracts L_jk * L_ik from its remaining elements with fused multiply-adds.  (Elements right of a lane's diagonal take part in the
// arithmetic but are never read.)  36 + (17 - k) instructions per column instead of ~ 33 + 4.5 k.
template <int T, class S, class MT>
DEV void ts_cholesky_factor_rows(const MT& m, S* s, int tl) {
  static_assert(T >= ND, "one lane per row");
  const int row = tl < ND ? tl : ND - 1;
  const bool own = tl < ND;
  float r[ND];
#pragma unroll
  for (int k = 0; k < ND; ++k) r[k] = s->H[row * DS + k];
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    const float akk = team_bcast<T>(r[k], k);
    const float d = dm_sqrt(fmx(akk, m.eps));
    const float inv = 1.0f / d;
    const float l = r[k] * inv;
    if (own && row >= k) s->H[row * DS + k] = (row == k) ? d : l;
    if (k < ND - 1) {
      if (own && row > k) s->H[k * DS + row] = l;
      team_sync();
      float col[ND];
#pragma unroll
      for (int i = k + 1; i < ND; ++i) col[i] = s->H[k * DS + i];
#pragma unroll
      for (int i = k + 1; i < ND; ++i) r[i] = __builtin_fmaf(-l, col[i], r[i]);
    }
  }
  team_sync();
}
#if GO2SIM_FAST_ORDER
// the Newton Hessian in arrow form (arrow_factor / arrow_solve, above the dynamics): in place over s->H;  grad = Ma - force - qfrc, Mgrad = H^-1 grad
template <int T, class S, class MT>
DEV void ts_cholesky_factor_arrow(const MT& m, S* s, int tl) { arrow_factor<T, DS>(m.arrow_mode, m.eps, s->H, s->H, tl); }
// (also returns |grad|^2: its butterfly has nothing to wait for and runs under the loads of the factor instead of after the solve)
template <int T, class S, class MT>
DEV float ts_update_gradient_arrow(const MT& m, S* s, int tl) {
  const int leg = (tl >> 3) & 3, u = tl & 7, ub = u < 6 ? u : 5;
  const int p0 = dm_arrow_dof(m.arrow_mode, leg, 0), p1 = dm_arrow_dof(m.arrow_mode, leg, 1), p2 = dm_arrow_dof(m.arrow_mode, leg, 2);
  const int row = tl < ND ? tl : ND - 1;
  const float g_own = s->Ma[row] - s->force[row] - s->qfrc[row];
  const float gl0 = s->Ma[p0] - s->force[p0] - s->qfrc[p0], gl1 = s->Ma[p1] - s->force[p1] - s->qfrc[p1], gl2 = s->Ma[p2] - s->force[p2] - s->qfrc[p2];
  const float gb = s->Ma[ub] - s->force[ub] - s->qfrc[ub];
  if (tl < ND) s->grad[row] = g_own;
  const float grad_sq = team_tree_sum1<T>(tl < ND ? g_own * g_own : 0.0f);
  float xb[6], x0, x1, x2;
  arrow_solve<T>(s->H, gl0, gl1, gl2, gb, s->Mgrad, tl, xb, x0, x1, x2);   // (Mgrad doubles as the exchange buffer of the base right-hand side)
  if (tl == 0) { *(float4*)&s->Mgrad[0] = make_float4(xb[0], xb[1], xb[2], xb[3]); *(float2*)&s->Mgrad[4] = make_float2(xb[4], xb[5]); }
  if (tl < 32 && u == 0) { s->Mgrad[p0] = x0; s->Mgrad[p1] = x1; s->Mgrad[p2] = x2; }
  team_sync();
  return grad_sq;
}
#endif
template <int T, class S, class MT>
DEV void ts_cholesky_factor(const MT& m, S* s, int tl) {
#if GO2SIM_FAST_ORDER
  if constexpr (T >= ND) { ts_cholesky_factor_rows<T>(m, s, tl); return; }
#endif
  ts_cholesky_factor_columns<T>(m, s, tl);
}

// func_hessian_and_cholesky_factor_incremental_dense_batch, solver.py:1632-1675; returns true when the factor degenerated.
// Register-resident form (T >= ND): lane i owns row i of the factor and element i of the update vector for the whole call; step k of a rank-1
// update needs only (v_k, L_kk) from lane k, which travel by a lane shuffle -- no LDS round trip and no barrier inside the k loop.  The
// arithmetic per element is the serial algorithm's (same operands, same order), so the result is unchanged.
template <int T, class S, class MT>
DEV bool ts_cholesky_incremental_reg(const MT& m, S* s, int tl, int n_con) {
  static_assert(T >= ND, "one lane per row of the factor");
  bool degenerated = false, touched = false;
  const int row = tl < ND ? tl : ND - 1;                               // lanes beyond the matrix shadow the last row (never written back)
  float Lr[ND];
#pragma unroll
  for (int k = 0; k < ND; ++k) Lr[k] = s->H[row * DS + k];
#if GO2SIM_FAST_ORDER
  float invL = 1.0f / s->H[row * DS + row];                            // reciprocal of my diagonal element, carried through the updates
#endif
  // The rows whose activity flipped are found T at a time (lane c tests row c) and collected in a bit mask per team; the teams of a wavefront
  // then walk their own lists in step: pass f updates every team's f-th flipped row at once.  (A common loop over the row index would run the
  // rank-1 update once per flipped row of EITHER team, with the other team masked off.)
  for (int base = 0; base < n_con && !degenerated; base += T) {
    const int c_me = base + tl;
    const bool flip = c_me < n_con && ((s->active[c_me] != 0) != (s->prev_active[c_me] != 0));
    const unsigned long long bal = __ballot(flip);
    unsigned long long mask = (T == 64) ? bal : ((bal >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
    PHC(50, 1)
    while (mask != 0ull && !degenerated) {
      PHC(51, 1)
      const int c = base + __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      touched = true;
      const float sign = (s->active[c] != 0) ? 1.0f : -1.0f;
      const float efc_D_sqrt = dm_sqrt(s->efc_D[c]);
      float v = s->J[c * DS + row] * efc_D_sqrt;
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        const float vk = team_bcast<T>(v, k), Lkk = team_bcast<T>(Lr[k], k);
        if (dm_abs(vk) > m.eps) {
          const float tmp = Lkk * Lkk + sign * (vk * vk);
          if (tmp < m.eps) { degenerated = true; break; }
          const float r = dm_sqrt(tmp);
#if GO2SIM_FAST_ORDER
          const float invLkk = team_bcast<T>(invL, k);
          const float rinv = r * (1.0f / tmp);                           // the division does not wait for the square root
          const float cc = r * invLkk;
          const float cinv = Lkk * rinv;
          const float sk = vk * invLkk;
          if (row == k) invL = rinv;
#else
          const float cc = r / Lkk;
          const float cinv = 1.0f / cc;
          const float sk = vk / Lkk;
#endif
          if (row == k) Lr[k] = r;
          else if (row > k) {
            const float hik = (Lr[k] + sk * v * sign) * cinv;
            Lr[k] = hik;
            v = v * cc - sk * hik;
          }
        }
      }
    }
  }
  if (touched && !degenerated && tl < ND) {                            // a degenerated factor is rebuilt from scratch by the caller
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (k <= row) { s->H[row * DS + k] = Lr[k]; if (k < row) s->H[k * DS + row] = Lr[k]; }
  }
  team_sync();
  return degenerated;
}
// The same rank-1 updates, pipelined over several flipped rows (func_hessian_and_cholesky_factor_incremental_dense_batch, solver.py:1632-1675).
// Step k of the update for flipped row f needs step k of row f-1 (the factor column it rewrites) and step k-1 of row f (its own vector), nothing
// else -- so the pairs (f, k) with f + k = t are independent of each other and are processed together in "time step" t: 17 + n steps for n
// flipped rows instead of 18 n.  Lane i owns row i of the factor (kept in LDS here, the column index varies at run time) and element i of every
// update vector (W[f], registers).  In time step t lane i computes the rotation of the pair (f = t - i, k = i) -- the square root and the three
// divisions of ALL pairs of the step are one instruction stream, lanes side by side -- then every pair's rotation is handed from its lane k to
// the rows below it.  Each matrix / vector element sees exactly the operations of the serial algorithm in the same order, so the result is
// bit-identical; the "degenerated" verdict is the same too (any pair degenerating makes the caller rebuild the factor from scratch).
#if GO2SIM_FAST_ORDER && defined(GO2SIM_PIPELINE_REG)
// Register form of the pipeline (-DGO2SIM_PIPELINE_REG, FAST ORDER builds; measured equal to the LDS form below: solver 0.223 vs 0.217 ms per step in the
// landing window, so the smaller LDS form is the default): lane i keeps row i of the factor and its update-vector elements in registers for the whole call,
// the time loop is unrolled so that the column of every slot is a compile-time index, and the rotation (c, 1/c, s) of column k travels from lane k by
// v_readlane -- no LDS round trip and no barrier inside a time step (the LDS form below pays two per step).  Same operations per element, same order.
template <int T, int FMAX, class S, class MT>
DEV bool ts_cholesky_incremental_pipelined(const MT& m, S* s, int tl, int n_con) {
  static_assert(T >= ND && (T == 32 || T == 64), "one lane per row of the factor");
  bool degenerated = false;
  const int row = tl < ND ? tl : ND - 1;
  const bool own = tl < ND;
  const unsigned urow = own ? (unsigned)row : 0u;                       // "column k lies left of my row" is (unsigned)k < urow (never true for spare lanes)
  float Lr[ND];
#pragma unroll
  for (int k = 0; k < ND; ++k) Lr[k] = s->H[row * DS + k];
  float Ld = s->H[row * DS + row];                                     // my diagonal element ...
  float invLd = 1.0f / Ld;                                             // ... and its reciprocal, carried through the updates
  for (int base = 0; base < n_con && !degenerated; base += T) {
    const int c_me = base + tl;
    const bool flip = c_me < n_con && ((s->active[c_me] != 0) != (s->prev_active[c_me] != 0));
    const unsigned long long bal = __ballot(flip);
    unsigned long long mask = (T == 64) ? bal : ((bal >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
    while (mask != 0ull && !degenerated) {                              // batches of up to FMAX flipped rows, in row order
      float W[FMAX], sg[FMAX]; int nb = 0;                              // update vectors (my element), sign of each update (0 = slot unused)
#pragma unroll
      for (int f = 0; f < FMAX; ++f) {
        W[f] = 0.0f; sg[f] = 0.0f;
        if (mask != 0ull) {
          const int c = base + __ffsll((long long)mask) - 1;
          mask &= mask - 1ull;
          W[f] = s->J[c * DS + row] * dm_sqrt(s->efc_D[c]);
          sg[f] = (s->active[c] != 0) ? 1.0f : -1.0f;
          nb = f + 1;
        }
      }
      const int nb_wave = __builtin_amdgcn_readfirstlane(imx(__shfl(nb, 0), __shfl(nb, T == 64 ? 0 : 32)));   // both teams (nb is team-uniform)
      PHC(52, 1) PHC(53, nb_wave) PHC(54, ND - 1 + nb_wave)
#pragma unroll
      for (int t = 0; t < ND - 1 + FMAX; ++t) {
        if (t >= ND - 1 + nb_wave) break;
        // ---- my pair of this step: row f_me at my own column ----
        const int f_me = t - row;
        float dv = 0.0f, sg_me = 0.0f;
#pragma unroll
        for (int f = 0; f < FMAX; ++f) { dv = (f_me == f) ? W[f] : dv; sg_me = (f_me == f) ? sg[f] : sg_me; }
        const bool rot = own && sg_me != 0.0f && dm_abs(dv) > m.eps;     // (sg_me == 0: no pair at my column in this step)
        const float tmp = Ld * Ld + sg_me * (dv * dv);
        const bool deg = rot && tmp < m.eps;
        const float r = dm_sqrt(tmp);
        const float rinv = r * (1.0f / tmp);                             // the division does not wait for the square root
        const float cc = r * invLd;
        float cinv = Ld * rinv;
        const float sk = dv * invLd;
        cinv = rot ? cinv : 0.0f;                                        // 0 marks "no rotation for this pair" (L_kk / r is never 0)
        {
          const unsigned long long dbal = __ballot(deg);
          const unsigned long long mine = (T == 64) ? dbal : ((dbal >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
          if (mine != 0ull) { degenerated = true; break; }
        }
        if (rot) { Ld = r; invLd = rinv; }
        // ---- every pair's rotation goes from the lane of its column to the rows below it ----
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
          const int k = t - f;                                           // compile-time after unrolling
          if (k < 0 || k > ND - 2) continue;
          const float c_k = team_bcast<T>(cc, k), ci_k = team_bcast<T>(cinv, k), s_k = team_bcast<T>(sk, k);
          const bool on = (unsigned)k < urow && sg[f] != 0.0f && ci_k != 0.0f;   // left of my row, slot in use, pair rotates
          const float hik = (Lr[k] + s_k * W[f] * sg[f]) * ci_k;
          const float wn = W[f] * c_k - s_k * hik;
          Lr[k] = on ? hik : Lr[k];
          W[f] = on ? wn : W[f];
        }
      }
    }
  }
  if (!degenerated && own) {                                            // a degenerated factor is rebuilt from scratch by the caller
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (k < row) { s->H[row * DS + k] = Lr[k]; s->H[k * DS + row] = Lr[k]; }   // (the strict upper triangle mirrors the lower one)
    s->H[row * DS + row] = Ld;
  }
  team_sync();
  return degenerated;
}
#else
template <int T, int FMAX, class S, class MT>
DEV bool ts_cholesky_incremental_pipelined(const MT& m, S* s, int tl, int n_con) {
  static_assert(T >= ND && (T == 32 || T == 64), "one lane per row of the factor");
  bool degenerated = false;
  const int row = tl < ND ? tl : ND - 1;
  const bool own = tl < ND;
  const unsigned urow = own ? (unsigned)row : 0u;                       // "column k lies left of my row" is (unsigned)k < urow (never true for spare lanes)
  float Ld = s->H[row * DS + row];                                     // my diagonal element
#if GO2SIM_FAST_ORDER
  float invLd = 1.0f / Ld;                                             // ... and its reciprocal, carried through the updates
#endif
  for (int base = 0; base < n_con && !degenerated; base += T) {
    const int c_me = base + tl;
    const bool flip = c_me < n_con && ((s->active[c_me] != 0) != (s->prev_active[c_me] != 0));
    const unsigned long long bal = __ballot(flip);
    unsigned long long mask = (T == 64) ? bal : ((bal >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
    while (mask != 0ull && !degenerated) {                              // batches of up to FMAX flipped rows, in row order
      float W[FMAX], sg[FMAX]; int nb = 0;                              // update vectors (my element), sign of each update (0 = slot unused)
#pragma unroll
      for (int f = 0; f < FMAX; ++f) {
        W[f] = 0.0f; sg[f] = 0.0f;
        if (mask != 0ull) {
          const int c = base + __ffsll((long long)mask) - 1;
          mask &= mask - 1ull;
          W[f] = s->J[c * DS + row] * dm_sqrt(s->efc_D[c]);
          sg[f] = (s->active[c] != 0) ? 1.0f : -1.0f;
          nb = f + 1;
        }
      }
      const int nb_wave = __builtin_amdgcn_readfirstlane(imx(__shfl(nb, 0), __shfl(nb, T == 64 ? 0 : 32)));   // both teams (nb is team-uniform)
      PHC(52, 1) PHC(53, nb_wave) PHC(54, ND - 1 + nb_wave)
      for (int t = 0; t < ND - 1 + nb_wave && !degenerated; ++t) {
        // ---- factor elements of my row at the columns of this step's pairs: requested first, they arrive while the rotations are computed ----
        float hcur[FMAX];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) hcur[f] = s->H[row * DS + imn(imx(t - f, 0), ND - 2)];
        // ---- my pair of this step: row f_me at my own column ----
        const int f_me = t - row;
        float dv = 0.0f, sg_me = 0.0f;
#pragma unroll
        for (int f = 0; f < FMAX; ++f) { dv = (f_me == f) ? W[f] : dv; sg_me = (f_me == f) ? sg[f] : sg_me; }
        const bool rot = own && sg_me != 0.0f && dm_abs(dv) > m.eps;     // (sg_me == 0: no pair at my column in this step)
        const float tmp = Ld * Ld + sg_me * (dv * dv);
        const bool deg = rot && tmp < m.eps;
        const float r = dm_sqrt(tmp);
#if GO2SIM_FAST_ORDER
        const float rinv = r * (1.0f / tmp);                             // the division does not wait for the square root
        const float cc = r * invLd;
        float cinv = Ld * rinv;
        const float sk = dv * invLd;
        if (rot) invLd = rinv;
#else
        const float cc = r / Ld;
        float cinv = 1.0f / cc;
        const float sk = dv / Ld;
#endif
        cinv = rot ? cinv : 0.0f;                                        // 0 marks "no rotation for this pair" (1 / cc is never 0)
        {
          const unsigned long long dbal = __ballot(deg);
          const unsigned long long mine = (T == 64) ? dbal : ((dbal >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
          if (mine != 0ull) { degenerated = true; break; }
        }
        if (rot) { Ld = r; s->H[row * DS + row] = r; }
        // ---- hand every pair's rotation to the rows below its column: the owner publishes (c, 1/c, s) of its column in LDS, every row reads the
        //      column of each slot (a broadcast read).  Branch-free over the slots of the batch: one LDS round trip per step, no EXEC juggling
        //      (a pair that does not apply to a lane stores into a spare word instead of the factor). ----
        if (own) *(float4*)&s->rot[row][0] = make_float4(cc, cinv, sk, 0.0f);
        team_sync();
        float4 rp[FMAX];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) rp[f] = *(const float4*)&s->rot[imn(imx(t - f, 0), ND - 2)][0];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
          const int k = t - f;
          const bool on = (unsigned)k < urow && sg[f] != 0.0f && rp[f].y != 0.0f;   // k >= 0 (unsigned compare), left of my row, slot in use, pair rotates
          const float hik = (hcur[f] + rp[f].z * W[f] * sg[f]) * rp[f].y;
          const float wn = W[f] * rp[f].x - rp[f].z * hik;
          float* dst = on ? &s->H[row * DS + k] : &s->rot[row][3];       // (a slot that does not apply here stores into the lane's spare word)
          *dst = hik;
          W[f] = on ? wn : W[f];
        }
      }
    }
  }
  team_sync();
  if (!degenerated) {                                                   // the strict upper triangle mirrors the lower one (the triangular solves read columns as rows)
    for (int idx = tl; idx < NTRI; idx += T) {
      int i, j;
      tri_index(m, idx, i, j);
      if (i != j) s->H[j * DS + i] = s->H[i * DS + j];
    }
    team_sync();
  }
  return degenerated;
}
#endif
template <int T, class S, class MT>
DEV bool ts_cholesky_incremental(const MT& m, S* s, int tl, int n_con) {
  if constexpr (T >= ND && (T == 32 || T == 64)) {   // (the register / pipelined forms shuffle across teams of 32 or 64 lanes)
#if GO2SIM_FAST_ORDER
    // FAST ORDER: an env with REBUILD_FLIPS (default 1: any) or more flipped rows has its Hessian summed and factorised afresh -- the caller's rebuild
    // path, which the reference takes for a degenerated factor -- instead of one rank-1 pass of the factor per flipped row.  With the block-form
    // Hessian and the row-form factorisation above the rebuild is the cheaper instruction stream even for a single flipped row (measured, driver
    // window / default run / stairs, M env-steps/s: never 8.67 / 14.07 / 6.04, >= 4 rows 8.85 / 14.08 / 6.43, >= 2 rows 9.21 / 14.12 / 6.63, always
    // 9.49 / 14.51 / 6.67), it depends on the env alone, and it does not accumulate the rounding of the rank-1 passes.  (First: the teams that stay
    // are then alone in the wave-level votes below.)
    {
      int n_exact = 0;
      for (int base = 0; base < n_con; base += T) {
        const int c = base + tl;
        n_exact += __popcll(team_ballot<T>(c < n_con && ((s->active[c] != 0) != (s->prev_active[c] != 0))));
      }
      if (n_exact >= REBUILD_FLIPS) return true;
      if (n_exact == 0) return false;                                   // nothing flipped: the factor stands
    }
#if REBUILD_FLIPS <= 1
    return false;                                                       // (not reached: every change of the active set is a rebuild)
#endif
#endif
#if !defined(GO2SIM_NO_PIPELINED_RANK1) && !(GO2SIM_FAST_ORDER && REBUILD_FLIPS <= 2)   // (REBUILD_FLIPS <= 2: an env that stays has at most one flipped row)
    // how many rows flipped (the larger count of the teams in this wavefront): one -> the register-resident serial form is the cheaper
    // instruction stream; several -> the pipelined form is 17 + n steps long instead of 18 n
    int n_flip = 0;
    for (int c = tl; c < n_con; c += T) n_flip += ((s->active[c] != 0) != (s->prev_active[c] != 0)) ? 1 : 0;
    unsigned long long any2, n_mine;
    {
      const unsigned long long fb = __ballot(n_flip > 0);
      // flipped rows of my team = bits of my team's lanes (each lane holds at most ceil(n_con / T) of them: count lanes, a lower bound that is
      // exact whenever n_con <= T, which is the LDS-resident case)
      const unsigned long long mine = (T == 64) ? fb : ((fb >> ((threadIdx.x / T) * T)) & ((1ull << (T & 63)) - 1ull));
      any2 = __ballot(__popcll(mine) >= 2);
      n_mine = mine;
    }
    if (any2 != 0ull) {                                                 // slots per batch sized to the larger flip count of the wavefront
      const int n_team = __popcll(n_mine);
      const int n_wave = __builtin_amdgcn_readfirstlane(imx(__shfl(n_team, 0), __shfl(n_team, T == 64 ? 0 : 32)));
      if (n_wave <= 2) return ts_cholesky_incremental_pipelined<T, 2>(m, s, tl, n_con);
      if (n_wave <= 3) return ts_cholesky_incremental_pipelined<T, 3>(m, s, tl, n_con);
      if (n_wave <= 4) return ts_cholesky_incremental_pipelined<T, 4>(m, s, tl, n_con);
      if (n_wave <= 6) return ts_cholesky_incremental_pipelined<T, 6>(m, s, tl, n_con);
      return ts_cholesky_incremental_pipelined<T, 8>(m, s, tl, n_con);
    }
#endif
    return ts_cholesky_incremental_reg<T>(m, s, tl, n_con);
  }
  bool degenerated = false;
  for (int c = 0; c < n_con && !degenerated; ++c) {
    bool is_active = s->active[c] != 0, was_active = s->prev_active[c] != 0;
    if (is_active ^ was_active) {
      float sign = is_active ? 1.0f : -1.0f;
      float efc_D_sqrt = dm_sqrt(s->efc_D[c]);
      team_sync();
      for (int d = tl; d < ND; d += T) s->ntv[d] = s->J[c * DS + d] * efc_D_sqrt;
      team_sync();
      for (int k = 0; k < ND; ++k) {
        float vk = s->ntv[k];
        if (dm_abs(vk) > m.eps) {
          float Lkk = s->H[k * DS + k];
          float tmp = Lkk * Lkk + sign * (vk * vk);
          if (tmp < m.eps) { degenerated = true; break; }
          float r = dm_sqrt(tmp);
          float cc = r / Lkk;
          float cinv = 1.0f / cc;
          float sk = vk / Lkk;
          team_sync();
          if (tl == 0) s->H[k * DS + k] = r;
          for (int i = k + 1 + tl; i < ND; i += T) {
            float nv = s->ntv[i];
            float hik = (s->H[i * DS + k] + sk * nv * sign) * cinv;
            s->H[i * DS + k] = hik; s->H[k * DS + i] = hik;
            s->ntv[i] = nv * cc - sk * hik;
          }
          team_sync();
        }
      }
    }
  }
  return degenerated;
}

// grad = Ma - force - qfrc;  Mgrad = H^-1 grad (func_cholesky_solve_batch, solver.py:1747-1765): two serial triangular solves, run
// redundantly by every lane with the running vector in registers; rows (and, through the mirror, columns) arrive as wide reads
template <int T, class S>
DEV void ts_update_gradient(S* s, int tl) {
#if GO2SIM_FAST_ORDER
  if constexpr (T >= 32) {
    // FAST ORDER: lane i owns row i of the factor (the mirrored storage gives it column i as well), the running right-hand side element i and the
    // reciprocal of the diagonal element.  Step j: lane j finishes its unknown with one multiplication, the value travels by v_readlane, the lanes
    // below (forward) / above (backward) subtract their term.  Per unknown the chain is mul -> readlane -> mul -> sub; the order of the subtractions
    // per row is ascending j in the forward and descending j in the backward substitution.
    const int row = tl < ND ? tl : ND - 1;
    float Hr[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) Hr[k] = s->H[row * DS + k];
    const float g = s->Ma[row] - s->force[row] - s->qfrc[row];
    if (tl < ND) s->grad[row] = g;
    const float linv = 1.0f / s->H[row * DS + row];
    float cur = g;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const float yj = team_bcast<T>(cur * linv, j);
      cur = (row == j) ? yj : ((row > j) ? cur - Hr[j] * yj : cur);
    }
#pragma unroll
    for (int j_ = 0; j_ < ND; ++j_) {
      const int j = ND - 1 - j_;
      const float xj = team_bcast<T>(cur * linv, j);
      cur = (row == j) ? xj : ((row < j) ? cur - Hr[j] * xj : cur);   // Hr[j] (j > row) mirrors L[j][row]
    }
    if (tl < ND) s->Mgrad[row] = cur;
    team_sync();
    return;
  }
#endif
  for (int d = tl; d < ND; d += T) s->grad[d] = s->Ma[d] - s->force[d] - s->qfrc[d];
  team_sync();
  float y[ND];
#pragma unroll
  for (int i_d = 0; i_d < ND; ++i_d) {
    float cur = s->grad[i_d];
#pragma unroll
    for (int j_d = 0; j_d < i_d; ++j_d) cur = cur - s->H[i_d * DS + j_d] * y[j_d];
    y[i_d] = cur / s->H[i_d * DS + i_d];
  }
#pragma unroll
  for (int i_d_ = 0; i_d_ < ND; ++i_d_) {
    const int i_d = ND - 1 - i_d_;
    float cur = y[i_d];
#pragma unroll
    for (int j_d = i_d + 1; j_d < ND; ++j_d) cur = cur - s->H[i_d * DS + j_d] * y[j_d];   // H[i][j] mirrors L[j][i]
    y[i_d] = cur / s->H[i_d * DS + i_d];
  }
  if (tl == 0) {
#pragma unroll
    for (int d = 0; d < ND; ++d) s->Mgrad[d] = y[d];
  }
  team_sync();
}

// the lane's own constraint row during a line search (zeros on lanes without a row): Jaref, J.search and the three quadratic coefficients
struct LsRow { float Ja, jv, q0, q1, q2; };
#if GO2SIM_FAST_ORDER
// FAST ORDER: the lane's partial sums of the three quadratic coefficients over its rows (row c on lane c % T) with the active set of `alpha`;
// rows that fit one per lane come from registers (rw), otherwise from the row arrays
template <int T, class S>
DEV void ts_ls_partials(S* s, int tl, int n_con, const LsRow& rw, float alpha, float (&p)[3]) {
  if (n_con <= T) {
    const float act = (float)((rw.Ja + alpha * rw.jv) < 0.0f);
    p[0] = rw.q0 * act; p[1] = rw.q1 * act; p[2] = rw.q2 * act;
    return;
  }
  p[0] = p[1] = p[2] = 0.0f;
  bool first = true;
  for (int c = tl; c < n_con; c += T) {
    const float act = (float)((s->Jaref[c] + alpha * s->jv[c]) < 0.0f);
    const float a0 = s->qf0[c] * act, a1 = s->qf1[c] * act, a2 = s->qf2[c] * act;
    p[0] = first ? a0 : p[0] + a0; p[1] = first ? a1 : p[1] + a1; p[2] = first ? a2 : p[2] + a2;
    first = false;
  }
}
#endif
// func_ls_point_fn_opt, solver.py:2009-2077.  `nseg` > 0: every row sits on its own lane and the three sums over the rows are serial-order
// lane scans (team_serial_sum); nseg == 0: more rows than lanes (global-scratch path), the scalar loops over the LDS / scratch arrays.
template <int T, class S, class MT>
DEV LsPoint ts_ls_point(const MT& m, S* s, int tl, int n_con, int nseg, const LsRow& rw, float alpha, float qg0, float qg1, float qg2) {
  float t0 = qg0 + 0.0f, t1 = qg1 + 0.0f, t2 = qg2 + 0.0f;
  bool done = false;
#if GO2SIM_FAST_ORDER
  if constexpr (T >= 32) {
    float pp[3], tot[3];
    ts_ls_partials<T>(s, tl, n_con, rw, alpha, pp);
    team_tree_sum<T, 3>(pp, tot);
    t0 = tot[0] + t0; t1 = tot[1] + t1; t2 = tot[2] + t2;
    done = true;
  }
#endif
  if constexpr (T >= 32 && !GO2SIM_FAST_ORDER) {
    if (nseg > 0) {
      const float x = rw.Ja + alpha * rw.jv;
      const float active = (float)(x < 0.0f);
      const float xs[3] = {rw.q0 * active, rw.q1 * active, rw.q2 * active}, bs[3] = {t0, t1, t2};
      float tot[3];
      team_serial_sum<T, 3>(xs, bs, tl, nseg, tot);
      t0 = tot[0]; t1 = tot[1]; t2 = tot[2];
      done = true;
    }
  }
  if (!done) {
#pragma unroll 16
    for (int c = 0; c < n_con; ++c) {
      float x = s->Jaref[c] + alpha * s->jv[c];
      float active = (float)(x < 0.0f);
      t0 = t0 + s->qf0[c] * active; t1 = t1 + s->qf1[c] * active; t2 = t2 + s->qf2[c] * active;
    }
  }
  LsPoint p; p.alpha = alpha;
  p.cost = alpha * alpha * t2 + alpha * t1 + t0;
  p.grad = 2.0f * alpha * t2 + t1;
  p.hess = 2.0f * t2;
  if (p.hess <= 0.0f) p.hess = m.eps;
  return p;
}
// func_ls_point_fn_3alphas_opt, solver.py:2080-2209
#ifdef GO2SIM_LS3_NOINLINE
#define GO2SIM_LS3_ATTR DEVN
#else
#define GO2SIM_LS3_ATTR DEV
#endif
template <int T, class S, class MT>
GO2SIM_LS3_ATTR void ts_ls_point3(const MT& m, S* s, int tl, int n_con, int nseg, const LsRow& rw, const float a[3], float qg0, float qg1, float qg2, float costs[3], float grads[3],
                      float hess[3]) {
  float b0 = qg0 + 0.0f, b1 = qg1 + 0.0f, b2 = qg2 + 0.0f;
  float t[3][3] = {{b0, b1, b2}, {b0, b1, b2}, {b0, b1, b2}};
  bool done = false;
#if GO2SIM_FAST_ORDER
  if constexpr (T >= 32) {
    float pp[9], tot[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float p3[3];
      ts_ls_partials<T>(s, tl, n_con, rw, a[k], p3);
      pp[3 * k] = p3[0]; pp[3 * k + 1] = p3[1]; pp[3 * k + 2] = p3[2];
    }
    team_tree_sum<T, 9>(pp, tot);
#pragma unroll
    for (int k = 0; k < 3; ++k) { t[k][0] = tot[3 * k] + b0; t[k][1] = tot[3 * k + 1] + b1; t[k][2] = tot[3 * k + 2] + b2; }
    done = true;
  }
#endif
  if constexpr (T >= 32 && !GO2SIM_FAST_ORDER) {
    if (nseg > 0) {
      float xs[9], bs[9], tot[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float act = (float)((rw.Ja + a[k] * rw.jv) < 0.0f);
        xs[3 * k] = rw.q0 * act; xs[3 * k + 1] = rw.q1 * act; xs[3 * k + 2] = rw.q2 * act;
        bs[3 * k] = b0; bs[3 * k + 1] = b1; bs[3 * k + 2] = b2;
      }
      team_serial_sum<T, 9>(xs, bs, tl, nseg, tot);
#pragma unroll
      for (int k = 0; k < 3; ++k) { t[k][0] = tot[3 * k]; t[k][1] = tot[3 * k + 1]; t[k][2] = tot[3 * k + 2]; }
      done = true;
    }
  }
  if (!done) {
    float t00 = b0, t01 = b1, t02 = b2, t10 = b0, t11 = b1, t12 = b2, t20 = b0, t21 = b1, t22 = b2;
#pragma unroll 16
    for (int c = 0; c < n_con; ++c) {
      float Ja = s->Jaref[c], jv = s->jv[c];
      float qf_0 = s->qf0[c], qf_1 = s->qf1[c], qf_2 = s->qf2[c];
      float a0 = (float)((Ja + a[0] * jv) < 0.0f), a1 = (float)((Ja + a[1] * jv) < 0.0f), a2 = (float)((Ja + a[2] * jv) < 0.0f);
      t00 = t00 + qf_0 * a0; t01 = t01 + qf_1 * a0; t02 = t02 + qf_2 * a0;
      t10 = t10 + qf_0 * a1; t11 = t11 + qf_1 * a1; t12 = t12 + qf_2 * a1;
      t20 = t20 + qf_0 * a2; t21 = t21 + qf_1 * a2; t22 = t22 + qf_2 * a2;
    }
    t[0][0] = t00; t[0][1] = t01; t[0][2] = t02; t[1][0] = t10; t[1][1] = t11; t[1][2] = t12; t[2][0] = t20; t[2][1] = t21; t[2][2] = t22;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    costs[k] = a[k] * a[k] * t[k][2] + a[k] * t[k][1] + t[k][0];
    grads[k] = 2.0f * a[k] * t[k][2] + t[k][1];
    hess[k] = 2.0f * t[k][2];
    if (hess[k] <= 0.0f) hess[k] = m.eps;
  }
}

#ifdef GO2SIM_BRACKET_DEBUG   // investigation build (tools/repro_bracket): every bracket step of the first 4 envs is logged (40 floats per record)
constexpr int BRLOG_ENVS = 4, BRLOG_CAP = 8192, BRLOG_W = 40;
__device__ float g_brlog[BRLOG_ENVS * BRLOG_CAP * BRLOG_W];
__device__ int g_brcnt[BRLOG_ENVS];
__device__ int g_brenv_of_wg[65536];
#endif
#if defined(GO2SIM_PHASE_PROFILE) && GO2SIM_FAST_ORDER   // where a line search spends its cycles: 50 set-up (mv, jv, coefficients, p0), 51 the 1-D Newton steps (52: their number),
#define LS_TIMER unsigned long long ls_t0 = __builtin_readcyclecounter();   // 53 the bracket refinement (54: its rounds of three points)
#define LS_MARK(id) { const unsigned long long ls_n = __builtin_readcyclecounter(); PHC(id, ls_n - ls_t0) ls_t0 = ls_n; }
#else
#define LS_TIMER
#define LS_MARK(id)
#endif
// the search proper (1-D Newton steps, then the bracket refinement) from the point alpha = 0
template <int T, class S, class MT>
DEV float ts_linesearch_tail(const MT& m, S* s, int tl, int n_con, int nseg, const LsRow& rw, float gtol, float qg0, float qg1, float qg2, LsPoint p0) {
  LS_TIMER
  int ls_it = 1;
  float res_alpha = 0.0f;
  bool done = false;
  LsPoint p1 = ts_ls_point<T>(m, s, tl, n_con, nseg, rw, p0.alpha - p0.grad / p0.hess, qg0, qg1, qg2);
  ls_it += 1;
  if (p0.cost < p1.cost) p1 = p0;
  if (dm_abs(p1.grad) < gtol) { LS_MARK(51) PHC(52, 1) return p1.alpha; }
  int direction = (p1.grad < 0) * 2 - 1;
  int p2update = 0;
  LsPoint p2 = p1;
  while (p1.grad * (float)direction <= -gtol && ls_it < m.ls_iterations) {
    p2 = p1; p2update = 1;
    p1 = ts_ls_point<T>(m, s, tl, n_con, nseg, rw, p1.alpha - p1.grad / p1.hess, qg0, qg1, qg2);
    ls_it += 1;
    if (dm_abs(p1.grad) < gtol) { res_alpha = p1.alpha; done = true; break; }
  }
  LS_MARK(51) PHC(52, ls_it - 1)
  if (done) return res_alpha;
  if (ls_it >= m.ls_iterations) return p1.alpha;
  if (!p2update) return p1.alpha;
  float al[3];
  al[0] = p1.alpha - p1.grad / p1.hess; al[1] = p1.alpha; al[2] = (p1.alpha + p2.alpha) * 0.5f;
  while (ls_it < m.ls_iterations) {
    float costs[3], grads[3], hess[3];
    ts_ls_point3<T>(m, s, tl, n_con, nseg, rw, al, qg0, qg1, qg2, costs, grads, hess);
    ls_it += 3;
    float p1_next_alpha = al[0], p2_next_alpha = al[1];
    float best_alpha = 0.0f, best_cost = 0.0f; bool best_found = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (dm_abs(grads[i]) < gtol && (!best_found || costs[i] < best_cost)) { best_alpha = al[i]; best_cost = costs[i]; best_found = true; }
    PHC(54, 1)
    if (best_found) { LS_MARK(53) return best_alpha; }
#ifdef GO2SIM_BRACKET_DEBUG
    const LsPoint p1_in = p1, p2_in = p2;
#endif
#if !defined(GO2SIM_BRACKET_INLINE) && !defined(GO2SIM_BRACKET_BYREF)
    const BrOut o1 = update_bracket_v(p1.alpha, p1.cost, p1.grad, p1.hess, al[0], al[1], al[2], costs[0], costs[1], costs[2], grads[0], grads[1], grads[2], hess[0], hess[1], hess[2]);
    p1.alpha = o1.alpha; p1.cost = o1.cost; p1.grad = o1.grad; p1.hess = o1.hess; p1_next_alpha = o1.next_alpha;
    const int b1 = o1.flag;
    const BrOut o2 = update_bracket_v(p2.alpha, p2.cost, p2.grad, p2.hess, al[0], al[1], al[2], costs[0], costs[1], costs[2], grads[0], grads[1], grads[2], hess[0], hess[1], hess[2]);
    p2.alpha = o2.alpha; p2.cost = o2.cost; p2.grad = o2.grad; p2.hess = o2.hess; p2_next_alpha = o2.next_alpha;
    const int b2 = o2.flag;
#else
    int b1 = update_bracket(p1, al, costs, grads, hess, p1_next_alpha);
#ifdef GO2SIM_BRACKET_FENCE   // investigation builds (tools/repro_bracket/README.md): value barriers around the inlined bracket step
    asm volatile("" : "+v"(p1.alpha), "+v"(p1.cost), "+v"(p1.grad), "+v"(p1.hess), "+v"(p1_next_alpha), "+v"(b1));
#endif
    int b2 = update_bracket(p2, al, costs, grads, hess, p2_next_alpha);
#ifdef GO2SIM_BRACKET_FENCE
    asm volatile("" : "+v"(p2.alpha), "+v"(p2.cost), "+v"(p2.grad), "+v"(p2.hess), "+v"(p2_next_alpha), "+v"(b2));
#endif
#endif
#ifdef GO2SIM_BRACKET_DEBUG
    {
      const int env = g_brenv_of_wg[blockIdx.x] + (int)(threadIdx.x / T);
      if (tl == 0 && env < BRLOG_ENVS) {
        const int k = g_brcnt[env];
        if (k < BRLOG_CAP) {
          float* o = &g_brlog[((size_t)env * BRLOG_CAP + k) * BRLOG_W];
          int q = 0;
          o[q++] = (float)ls_it; o[q++] = gtol;
          for (int i = 0; i < 3; ++i) o[q++] = al[i];
          for (int i = 0; i < 3; ++i) o[q++] = costs[i];
          for (int i = 0; i < 3; ++i) o[q++] = grads[i];
          for (int i = 0; i < 3; ++i) o[q++] = hess[i];
          o[q++] = p1_in.alpha; o[q++] = p1_in.cost; o[q++] = p1_in.grad; o[q++] = p1_in.hess;
          o[q++] = p2_in.alpha; o[q++] = p2_in.cost; o[q++] = p2_in.grad; o[q++] = p2_in.hess;
          o[q++] = p1.alpha; o[q++] = p1.cost; o[q++] = p1.grad; o[q++] = p1.hess;
          o[q++] = p2.alpha; o[q++] = p2.cost; o[q++] = p2.grad; o[q++] = p2.hess;
          o[q++] = p1_next_alpha; o[q++] = p2_next_alpha; o[q++] = (float)b1; o[q++] = (float)b2;
          g_brcnt[env] = k + 1;
        }
      }
    }
#endif
    if (b1 == 0 && b2 == 0) { LS_MARK(53) return al[2]; }
    al[0] = p1_next_alpha; al[1] = p2_next_alpha; al[2] = (p1.alpha + p2.alpha) * 0.5f;
  }
  LS_MARK(53)
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0.0f;
}

// func_linesearch_batch, solver.py:2246-2417
template <int T, class S, class MT>
DEV float ts_linesearch(const MT& m, S* s, int tl, int n_con, float gauss) {
  LS_TIMER
  float sr[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) sr[d] = s->search[d];
#if GO2SIM_FAST_ORDER
  if constexpr (T >= 32) {
    // FAST ORDER, teams of 32 / 64 lanes: the six sums of the set-up (|search|^2, the two gradient coefficients, the three coefficients of the point alpha = 0) need
    // nothing from each other, so their butterflies run as ONE (every value sees the same additions as in three separate trees: same bits, a third of the
    // dependent steps).  Lane d owns mv[d], lane c % T the rows c.
    const int d = tl < ND ? tl : ND - 1;
    float mv_own = 0.0f;
    for (int d1 = tl; d1 < ND; d1 += T) {
      float mv = 0.0f;
#pragma unroll
      for (int d2 = 0; d2 < ND; ++d2) mv = mv + s->M[d1 * DS + d2] * sr[d2];
      s->mv[d1] = mv; mv_own = mv;
    }
    LsRow rw = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float pp[3] = {0.0f, 0.0f, 0.0f};
    bool first = true;
    for (int c = tl; c < n_con; c += T) {
      float jv = 0.0f;
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) jv = jv + s->J[c * DS + dd] * sr[dd];
      s->jv[c] = jv;
      float Ja = s->Jaref[c], D = s->efc_D[c];
      const float q0 = D * (0.5f * Ja * Ja), q1 = D * (jv * Ja), q2 = D * (0.5f * jv * jv);
      s->qf0[c] = q0; s->qf1[c] = q1; s->qf2[c] = q2;
      rw.Ja = Ja; rw.jv = jv; rw.q0 = q0; rw.q1 = q1; rw.q2 = q2;       // (the lane's row when rows sit one per lane)
      const float active = (float)(Ja < 0.0f);
      const float a0 = q0 * active, a1 = q1 * active, a2 = q2 * active;
      pp[0] = first ? a0 : pp[0] + a0; pp[1] = first ? a1 : pp[1] + a1; pp[2] = first ? a2 : pp[2] + a2;
      first = false;
    }
    const float sd = s->search[d];
    const float xs[6] = {tl < ND ? sd * sd : 0.0f, tl < ND ? (sd * s->Ma[d] - sd * s->force[d]) : 0.0f, tl < ND ? 0.5f * sd * mv_own : 0.0f, pp[0], pp[1], pp[2]};
    float tot[6];
    team_tree_sum<T, 6>(xs, tot);
    team_sync();
    const float snorm = dm_sqrt(tot[0]);
    const float scale = m.meaninertia * (float)imx(1, ND);
    const float gtol = m.tolerance * m.ls_tolerance * snorm * scale;
    if (snorm < m.eps) return 0.0f;
    const int nseg = 0;
    const float qg0 = gauss, qg1 = tot[1], qg2 = tot[2];
    LsPoint p0;
    p0.alpha = 0.0f; p0.cost = tot[3] + qg0; p0.grad = tot[4] + qg1; p0.hess = 2.0f * (tot[5] + qg2);
    if (p0.hess <= 0.0f) p0.hess = m.eps;
    LS_MARK(50)
    return ts_linesearch_tail<T>(m, s, tl, n_con, nseg, rw, gtol, qg0, qg1, qg2, p0);
  }
#endif
  float snorm = 0.0f;
  if constexpr (T >= 32) {
    const float my = s->search[tl < ND ? tl : ND - 1];
#if GO2SIM_FAST_ORDER
    snorm = team_tree_sum1<T>(tl < ND ? my * my : 0.0f);
#else
    const float xq[1] = {tl < ND ? my * my : 0.0f}, zero[1] = {0.0f};
    float tot[1];
    team_serial_sum<T, 1>(xq, zero, tl, 2, tot);
    snorm = tot[0];
#endif
  } else {
#pragma unroll
    for (int jd = 0; jd < ND; ++jd) snorm = snorm + sr[jd] * sr[jd];
  }
  snorm = dm_sqrt(snorm);
  float scale = m.meaninertia * (float)imx(1, ND);
  float gtol = m.tolerance * m.ls_tolerance * snorm * scale;
  if (snorm < m.eps) return 0.0f;
  // 16-lane segments holding rows (0 = rows do not fit one per lane: scalar loops).  The count must be uniform over the wavefront (DPP steps are
  // executed by all its lanes): the teams of a wavefront may hold different row counts, the larger one decides, surplus lanes add zeros.
  int nseg = 0;
  if constexpr (T >= 32) {
    const int n_wave = __builtin_amdgcn_readfirstlane(imx(__shfl(n_con, 0), __shfl(n_con, T == 64 ? 0 : 32)));
    nseg = (n_wave <= T) ? (n_wave + 15) / 16 : 0;
  }
  // mv = M search, jv = J search, and the alpha-independent quadratic coefficients of every row
  for (int d1 = tl; d1 < ND; d1 += T) {
    float mv = 0.0f;
#pragma unroll
    for (int d2 = 0; d2 < ND; ++d2) mv = mv + s->M[d1 * DS + d2] * sr[d2];
    s->mv[d1] = mv;
  }
  LsRow rw = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int c = tl; c < n_con; c += T) {
    float jv = 0.0f;
#pragma unroll
    for (int d = 0; d < ND; ++d) jv = jv + s->J[c * DS + d] * sr[d];
    s->jv[c] = jv;
    float Ja = s->Jaref[c], D = s->efc_D[c];
    const float q0 = D * (0.5f * Ja * Ja), q1 = D * (jv * Ja), q2 = D * (0.5f * jv * jv);
    s->qf0[c] = q0; s->qf1[c] = q1; s->qf2[c] = q2;
    rw.Ja = Ja; rw.jv = jv; rw.q0 = q0; rw.q1 = q1; rw.q2 = q2;       // (the lane's row when rows sit one per lane)
  }
  team_sync();
  float qg1 = 0.0f, qg2 = 0.0f;
  if constexpr (T >= 32) {
    const int d = tl < ND ? tl : ND - 1;
    const float sd = s->search[d];
    const float xq[2] = {tl < ND ? (sd * s->Ma[d] - sd * s->force[d]) : 0.0f, tl < ND ? 0.5f * sd * s->mv[d] : 0.0f}, zero[2] = {0.0f, 0.0f};
    float tot[2];
#if GO2SIM_FAST_ORDER
    team_tree_sum<T, 2>(xq, tot);
#else
    team_serial_sum<T, 2>(xq, zero, tl, 2, tot);
#endif
    qg1 = tot[0]; qg2 = tot[1];
  } else {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      float sd = sr[d];
      qg1 = qg1 + (sd * s->Ma[d] - sd * s->force[d]);
      qg2 = qg2 + 0.5f * sd * s->mv[d];
    }
  }
  const float qg0 = gauss;
  LsPoint p0;
  {
    float t0 = qg0, t1 = qg1, t2 = qg2;
    bool done = false;
#if GO2SIM_FAST_ORDER
    if constexpr (T >= 32) {
      float pp[3], tot[3];
      if (n_con <= T) {
        const float active = (float)(rw.Ja < 0.0f);
        pp[0] = rw.q0 * active; pp[1] = rw.q1 * active; pp[2] = rw.q2 * active;
      } else {
        pp[0] = pp[1] = pp[2] = 0.0f;
        bool first = true;
        for (int c = tl; c < n_con; c += T) {
          const float active = (float)(s->Jaref[c] < 0.0f);
          const float a0 = s->qf0[c] * active, a1 = s->qf1[c] * active, a2 = s->qf2[c] * active;
          pp[0] = first ? a0 : pp[0] + a0; pp[1] = first ? a1 : pp[1] + a1; pp[2] = first ? a2 : pp[2] + a2;
          first = false;
        }
      }
      team_tree_sum<T, 3>(pp, tot);
      t0 = tot[0] + t0; t1 = tot[1] + t1; t2 = tot[2] + t2;
      done = true;
    }
#endif
    if constexpr (T >= 32 && !GO2SIM_FAST_ORDER) {
      if (nseg > 0) {
        const float active = (float)(rw.Ja < 0.0f);
        const float xs[3] = {rw.q0 * active, rw.q1 * active, rw.q2 * active}, bs[3] = {t0, t1, t2};
        float tot[3];
        team_serial_sum<T, 3>(xs, bs, tl, nseg, tot);
        t0 = tot[0]; t1 = tot[1]; t2 = tot[2];
        done = true;
      }
    }
    if (!done) {
#pragma unroll 16
      for (int c = 0; c < n_con; ++c) {
        float active = (float)(s->Jaref[c] < 0.0f);
        t0 = t0 + s->qf0[c] * active; t1 = t1 + s->qf1[c] * active; t2 = t2 + s->qf2[c] * active;
      }
    }
    p0.alpha = 0.0f; p0.cost = t0; p0.grad = t1; p0.hess = 2.0f * t2;
    if (p0.hess <= 0.0f) p0.hess = m.eps;
  }
  LS_MARK(50)
  return ts_linesearch_tail<T>(m, s, tl, n_con, nseg, rw, gtol, qg0, qg1, qg2, p0);
}

#undef LS_MARK
// rows + resolve for one environment (add_collision_constraints solver.py:498-595, add_joint_limit_constraints :1088-1143,
// func_solve_init :2739-2859, func_solve_body :2941-2966, func_solve_iter :2862-2938)
template <int T, class S, class MT>
DEV int ts_solve(const MT& m, const E& e, S* s, int tl, int nc, int n_con, unsigned lim_mask, int ws_flag) {
  PH_BEGIN
#ifdef GO2SIM_REPEAT_PHASE      // profiling builds: run one idempotent phase twice, the time difference is that phase's cost
  for (int rep_stage = 0; rep_stage < (GO2SIM_REPEAT_PHASE == 0 ? 2 : 1); ++rep_stage) {
#endif
  // ---- stage inputs (branch-free: all loads of the stage are in flight together) ----
  const bool ws = (n_con > 0) && ws_flag;
  team_stage<NTRI, T>(tl, [&](int k) { return aload(e, AO(mass_mat), k); }, [&](int k, float v) { int i, j; tri_index(m, k, i, j); s->M[i * DS + j] = v; s->M[j * DS + i] = v; });   // packed lower triangle
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_ang), k); }, [&](int k, float v) { s->cdof_ang[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_vel), k); }, [&](int k, float v) { s->cdof_vel[k] = v; });
  team_stage<NL * 3, T>(tl, [&](int k) { return gload(e, FO(root_com), k); }, [&](int k, float v) { s->root_com[k] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(vel), d); }, [&](int d, float v) { s->vel[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return aload(e, AO(force), d); }, [&](int d, float v) { s->force[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return aload(e, AO(acc_smooth), d); }, [&](int d, float v) { s->acc_smooth[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return aload(e, AO(qacc_ws), d); }, [&](int d, float v) { s->qacc[d] = v; });
  team_sync();
  if (!ws) team_for<ND, T>(tl, [&](int d) { s->qacc[d] = s->acc_smooth[d]; });
  team_sync();
#ifdef GO2SIM_REPEAT_PHASE
  }
  for (int rep_rows = 0; rep_rows < (GO2SIM_REPEAT_PHASE == 1 ? 2 : 1); ++rep_rows) {
#endif
  PH(0)
  bool rows_coupled = false;                                           // does a contact row of this lane join two legs? (arrow form of the Newton Hessian)
  // ---- contact rows: one lane per row ----
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 13
  for (int rep_c = 0; rep_c < 2; ++rep_c)
#endif
  for (int r = tl; r < 4 * nc; r += T) {
    int i_col = r >> 2, i = r & 3;
    int link_a = e.c_link()[i_col], link_b = e.c_link()[MAXC + i_col];
    V3 cpos = e.c_pos()[i_col], cnormal = e.c_normal()[i_col];
    float friction = e.c_friction()[i_col], pen = e.c_pen()[i_col];
    float sol[7];
    auto csol = e.c_sol()[i_col];
#pragma unroll
    for (int k = 0; k < 7; ++k) sol[k] = csol[k];
    V3 d1, d2;
    orthogonals(cnormal, d1, d2);
    float invweight = m.links[link_a].invweight[0];
    if (link_b > -1) invweight = invweight + m.links[link_b].invweight[0];
    V3 d = (float)(2 * (i % 2) - 1) * ((i < 2) ? d1 : d2);
    V3 n = d * friction - cnormal;
    float* row = &s->J[r * DS];
#pragma unroll
    for (int i_d = 0; i_d < DS; ++i_d) row[i_d] = 0.0f;
    float jac_qvel = 0.0f;
    unsigned legs = 0u;                                                  // legs whose dofs the row has entries for
    for (int i_ab = 0; i_ab < 2; ++i_ab) {
      float sign = -1.0f; int link = link_a;
      if (i_ab == 1) { sign = 1.0f; link = link_b; }
      while (link > -1) {
        V3 t_pos = cpos - v3(s->root_com[3 * link], s->root_com[3 * link + 1], s->root_com[3 * link + 2]);
        const int nd = m.links[link].n_dofs, de = m.links[link].dof_end;
        if (nd > 0 && de > 6) legs |= 1u << dm_arrow_leg(m.arrow_mode, de - 1);
        for (int i_d_ = 0; i_d_ < nd; ++i_d_) {
          int i_d = de - 1 - i_d_;
          V3 ca = v3(s->cdof_ang[3 * i_d], s->cdof_ang[3 * i_d + 1], s->cdof_ang[3 * i_d + 2]);
          V3 cv = v3(s->cdof_vel[3 * i_d], s->cdof_vel[3 * i_d + 1], s->cdof_vel[3 * i_d + 2]);
          V3 velv = cv - cross(t_pos, ca);
          V3 diff = sign * velv;
          float j = dot(diff, n);
          jac_qvel = jac_qvel + j * s->vel[i_d];
          row[i_d] = row[i_d] + j;
        }
        link = m.links[link].parent;
      }
    }
    float imp, aref;
    imp_aref(sol, -pen, jac_qvel, -pen, imp, aref);
    float diag = invweight + friction * friction * invweight;
    diag *= 2.0f * friction * friction * (1.0f - imp) / imp;
    diag = fmx(diag, m.eps);
    s->aref[r] = aref; s->efc_D[r] = 1.0f / diag;
    rows_coupled = rows_coupled || (legs & (legs - 1u)) != 0u;
  }
  // ---- joint-limit rows: one lane per joint, ordered compaction ----
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 14
  for (int rep_c = 0; rep_c < 2; ++rep_c)
#endif
  for (int i_j = tl; i_j < NJ; i_j += T) {
    if (!((lim_mask >> i_j) & 1u)) continue;                           // joints past a limit, found by the kernel prologue
    const Joint& Jt = m.joints[i_j];
    int i_d = Jt.dof_start;
    float q = gload(e, FO(qpos), Jt.q_start);
    float pos_delta_min = q - m.dofs[i_d].limit[0];
    float pos_delta_max = m.dofs[i_d].limit[1] - q;
    float pos_delta = fmn(pos_delta_min, pos_delta_max);
    {
      int r = 4 * nc + __popc(lim_mask & ((1u << i_j) - 1u));           // ordered compaction: rows follow the joint order
      float j = (float)((pos_delta_min < pos_delta_max) * 2 - 1);
      float jac_qvel = j * s->vel[i_d];
      float imp, aref;
      imp_aref(Jt.sol_params, pos_delta, jac_qvel, pos_delta, imp, aref);
      float diag = fmx(m.dofs[i_d].invweight * (1.0f - imp) / imp, m.eps);
      s->aref[r] = aref; s->efc_D[r] = 1.0f / diag;
      float* row = &s->J[r * DS];
#pragma unroll
      for (int i_d2 = 0; i_d2 < DS; ++i_d2) row[i_d2] = 0.0f;
      row[i_d] = j;
    }
  }
  team_sync();
#ifdef GO2SIM_REPEAT_PHASE
  }
#endif
  PH(1)
  // ---- func_solve_init ----
  {
    float qa[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) qa[d] = s->qacc[d];
    for (int d1 = tl; d1 < ND; d1 += T) {
      float Ma_ = 0.0f;
#pragma unroll
      for (int d2 = 0; d2 < ND; ++d2) Ma_ = Ma_ + s->M[d1 * DS + d2] * qa[d2];
      s->Ma[d1] = Ma_;
    }
    for (int c = tl; c < n_con; c += T) {
      float Jv = -s->aref[c];
#pragma unroll
      for (int d = 0; d < ND; ++d) Jv = Jv + s->J[c * DS + d] * qa[d];
      s->Jaref[c] = Jv;
    }
  }
  team_sync();
  float cost = 0.0f, prev_cost = 0.0f, gauss = 0.0f;
  ts_update_constraint<T>(m, s, tl, n_con, cost, prev_cost, gauss);
  PH(2)
  int iters = 0;
  if (n_con > 0) {
    const float tol_scaled = (m.meaninertia * (float)imx(1, ND)) * m.tolerance;
    bool need_full = true;
    [[maybe_unused]] bool arrow = false;                               // which form the factor of this solve has
#if GO2SIM_FAST_ORDER
    if constexpr (ARROW_SOLVER && (T == 32 || T == 64)) arrow = m.arrow_mode != 0 && team_ballot<T>(rows_coupled) == 0ull;
#endif
    for (int it = 0;; ++it) {
      if (need_full) {                       // single call site of the direct Hessian + factorisation (init and degenerate rebuild)
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 3
        ts_hessian_direct<T>(m, s, tl, n_con);
        ts_cholesky_factor<T>(m, s, tl);
#endif
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 4
        ts_hessian_direct<T>(m, s, tl, n_con);
#endif
#if GO2SIM_FAST_ORDER
        if constexpr (T >= 21) { if (it == 0) ts_hessian_direct<T>(m, s, tl, n_con); else ts_hessian_update<T>(m, s, tl, n_con); }
        else ts_hessian_direct<T>(m, s, tl, n_con);
        PH(3)
        if constexpr (ARROW_SOLVER && (T == 32 || T == 64)) { if (arrow) ts_cholesky_factor_arrow<T>(m, s, tl); else ts_cholesky_factor<T>(m, s, tl); }
        else ts_cholesky_factor<T>(m, s, tl);
#else
        ts_hessian_direct<T>(m, s, tl, n_con);
        PH(3)
        ts_cholesky_factor<T>(m, s, tl);
#endif
        PH(4)
      }
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 5
      ts_update_gradient<T>(s, tl);
#endif
#if GO2SIM_FAST_ORDER
      [[maybe_unused]] float grad_sq_arrow = 0.0f;
      if constexpr (ARROW_SOLVER && (T == 32 || T == 64)) { if (arrow) grad_sq_arrow = ts_update_gradient_arrow<T>(m, s, tl); else ts_update_gradient<T>(s, tl); }
      else ts_update_gradient<T>(s, tl);
#else
      ts_update_gradient<T>(s, tl);
#endif
      PH(5)
      if (it > 0) {
        float improvement = prev_cost - cost;
        float grad_norm = 0.0f;
        if constexpr (T >= 32) {
          const float g = s->grad[tl < ND ? tl : ND - 1];
#if GO2SIM_FAST_ORDER
          if (ARROW_SOLVER && (T == 32 || T == 64) && arrow) grad_norm = grad_sq_arrow;
          else grad_norm = team_tree_sum1<T>(tl < ND ? g * g : 0.0f);
#else
          const float xq[1] = {tl < ND ? g * g : 0.0f}, zero[1] = {0.0f};
          float tot[1];
          team_serial_sum<T, 1>(xq, zero, tl, 2, tot);
          grad_norm = tot[0];
#endif
        } else {
#pragma unroll
          for (int d = 0; d < ND; ++d) { float g = s->grad[d]; grad_norm = grad_norm + g * g; }
        }
        grad_norm = dm_sqrt(grad_norm);
        bool improved = (grad_norm > tol_scaled) && (improvement > tol_scaled);
        if (!improved) break;
      }
      if (it == m.iterations) break;
      for (int d = tl; d < ND; d += T) s->search[d] = -s->Mgrad[d];
      team_sync();
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 6
      { float alpha0 = ts_linesearch<T>(m, s, tl, n_con, gauss); if (alpha0 == 12345.0f) s->search[0] = 0.0f; team_sync(); }
#endif
      float alpha = ts_linesearch<T>(m, s, tl, n_con, gauss);
      PH(6)
      iters++;
      if (dm_abs(alpha) < m.eps) break;
      team_sync();
      for (int d = tl; d < ND; d += T) {
        s->qacc[d] = s->qacc[d] + s->search[d] * alpha;
        s->Ma[d] = s->Ma[d] + s->mv[d] * alpha;
      }
      for (int c = tl; c < n_con; c += T) s->Jaref[c] = s->Jaref[c] + s->jv[c] * alpha;
      team_sync();
      ts_update_constraint<T>(m, s, tl, n_con, cost, prev_cost, gauss);
      PH(7)
      need_full = ts_cholesky_incremental<T>(m, s, tl, n_con);
      team_sync();
      PH(8)
    }
  }
  return iters;
}

// func_update_qacc (solver.py:3016-3037) + func_update_contact_force (:2974-3013) + public row outputs
template <int T, class S, class MT>
DEV void ts_commit(const MT& m, const E& e, S* s, int tl, int nc, int n_con, int iters) {
  // every global input of the commit is requested first (one round trip): the smooth force of the lane's dof, the geometry of the lane's contact
  const int d0 = tl < ND ? tl : ND - 1, c0 = tl < nc ? tl : (nc > 0 ? nc - 1 : 0);
  const float qfs0 = aload(e, AO(qf_smooth), d0);
  V3 cn0 = v3(0, 0, 0); float fr0 = 0.0f; int lk0 = 0;
  if (nc > 0) { cn0 = e.c_normal()[c0]; fr0 = e.c_friction()[c0]; lk0 = e.c_link()[c0] | (e.c_link()[MAXC + c0] << 8); }
  int err = 0;
  for (int d = tl; d < ND; d += T) {
    float q = s->qacc[d];
    astore(e, AO(acc), d, q);
    astore(e, AO(force), d, ((d == d0) ? qfs0 : aload(e, AO(qf_smooth), d)) + s->qfrc[d]);
    astore(e, AO(qacc_ws), d, q);
    astore(e, AO(qfrc_constraint), d, s->qfrc[d]);
    if (isnan_(q)) err |= GO2SIM_ERR_INVALID_FORCE_NAN;
  }
  if (err) atomicOr(&e.err()[0], err);
  for (int c = tl; c < n_con; c += T) astore(e, AO(efc_force), c, s->efc_force[c]);
  if (tl == 0) { e.is_warmstart()[0] = 1; e.n_con()[0] = n_con; e.solver_iters()[0] = iters; }
  team_sync();
  float* cf = s->J;  // the Jacobian is dead from here on: reuse its storage for the per-contact forces
  for (int i_c = tl; i_c < nc; i_c += T) {
    const bool pre = i_c == c0;
    V3 cnormal = pre ? cn0 : (V3)e.c_normal()[i_c]; float friction = pre ? fr0 : (float)e.c_friction()[i_c];
    V3 f = v3(0, 0, 0), d1, d2;
    orthogonals(cnormal, d1, d2);
#pragma unroll
    for (int i_dir = 0; i_dir < 4; ++i_dir) {
      V3 d = (float)(2 * (i_dir % 2) - 1) * ((i_dir < 2) ? d1 : d2);
      V3 n = d * friction - cnormal;
      f = f + n * s->efc_force[i_c * 4 + i_dir];
    }
    e.c_force()[i_c] = f;
    cf[4 * i_c] = f.x; cf[4 * i_c + 1] = f.y; cf[4 * i_c + 2] = f.z;
    ((int*)cf)[4 * i_c + 3] = pre ? lk0 : (e.c_link()[i_c] | (e.c_link()[MAXC + i_c] << 8));
  }
  team_sync();
  for (int i_l = tl; i_l < NL; i_l += T) {
    V3 acc = v3(0, 0, 0);
#pragma unroll 4
    for (int i_c = 0; i_c < nc; ++i_c) {
      int lk = ((const int*)cf)[4 * i_c + 3];
      int la = lk & 0xff, lb = lk >> 8;
      V3 f = v3(cf[4 * i_c], cf[4 * i_c + 1], cf[4 * i_c + 2]);
      if (la == i_l) acc = acc - f;
      if (lb == i_l) acc = acc + f;
    }
    e.contact_force()[i_l] = acc;
  }
}

// cold path: more rows than fit in LDS; same code on a per-env global scratch block
// (arguments are plain scalars / pointers and the views are rebuilt inside: by-reference or by-value aggregates would make the caller keep
//  stack copies, i.e. scratch stores on the hot path of every wave)
template <int T>
DEVN void ts_solve_overflow(float* Pf, int* Pi, int PB, float* Pfa, int* Pia, int b, const LinkS* lnk, const unsigned char* tri_i, const unsigned char* tri_j,
                            const Model* __restrict__ gm, SolverData<MAXR>* s, int tl, int nc, int n_con, unsigned lim_mask, int ws_flag) {
  Pool P; P.f = Pf; P.i = Pi; P.B = PB; P.fa = Pfa; P.ia = Pia;
  const E e(P, b);
  const ModelView m(lnk, tri_i, tri_j, gm);
  int iters = ts_solve<T>(m, e, s, tl, nc, n_con, lim_mask, ws_flag);
  ts_commit<T>(m, e, s, tl, nc, n_con, iters);
}

// body of k_constraint_solve_team on LDS blocks handed in by the kernel (the plain launch owns them; k_solve_integrate_team lays them over the blocks of the
// kinematics / dynamics that follow).  Out: the env of the caller's team (b_out >= P.B: none) and whether its solve ran on the LDS block (else: global scratch).
template <int T, int RLN>
DEV void solve_body(const Pool& P, const Model* __restrict__ gm, const ModelS* __restrict__ mp, SolverData<MAXR>* __restrict__ overflow,
                    const int* __restrict__ lpt_rec, int* __restrict__ lpt_next, int lpt_cap, SolverData<RLN>* lds, char* blk_raw, int& b_out, bool& resident_out) {
  constexpr int EPW = 64 / T;
  const LinkS* lnk = (const LinkS*)blk_raw;
  const unsigned char* tri_i = (const unsigned char*)(blk_raw + sizeof(LinkS) * NL);
  const unsigned char* tri_j = tri_i + (NTRI + 1);
  // every load of the prologue (solver block by LDS DMA, contact count, warm-start flag, the joint coordinates of the limit test) is issued
  // before the first wait
  static_assert(T >= NJ, "one lane per joint in the limit test");
  PH_BEGIN
  wg_dma_to_lds<SOLVER_BLOCK_BYTES>(blk_raw, mp->links);
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const int b = lpt_take(lpt_rec, lpt_cap, P.B, EPW, slot);
  if (lpt_next && blockIdx.x == 0) for (int i = threadIdx.x; i < 8 * LPT_CLS; i += 64) lpt_next[i] = 0;   // the record of the next collide / solve pair
  const bool env_valid = b < P.B;
  b_out = b; resident_out = false;
#ifdef GO2SIM_BRACKET_DEBUG
  if (threadIdx.x == 0) g_brenv_of_wg[blockIdx.x] = b - slot;
#endif
  E e(P, env_valid ? b : P.B - 1);
  const int nc = e.n_contacts()[0];
  const int ws_flag = e.is_warmstart()[0];
  const JLim jl = mp->jlim[tl < NJ ? tl : NJ - 1];
  const float q_lim = gload(e, FO(qpos), jl.q_start < 0 ? 0 : jl.q_start);
  const bool lim = tl < NJ && jl.q_start >= 0 && fmn(q_lim - jl.lo, jl.hi - q_lim) < 0;   // add_joint_limit_constraints, solver.py:1088-1143
  const unsigned lim_mask = (unsigned)((__ballot(lim) >> (slot * T)) & ((T == 64) ? ~0ull : ((1ull << T) - 1ull)));
  const int n_lim = __popc(lim_mask);
  __syncthreads();
  PH(9)
  const ModelView m(lnk, tri_i, tri_j, gm);
  if (!env_valid) return;
  const int n_con = 4 * nc + n_lim;
  if (n_con <= RLN) {
    SolverData<RLN>* s = &lds[slot];
    resident_out = true;
    int iters = ts_solve<T>(m, e, s, tl, nc, n_con, lim_mask, ws_flag);
    PH(10)                                                               // (resets the timer: the phases of ts_solve are accounted inside it)
#if defined(GO2SIM_REPEAT_PHASE) && GO2SIM_REPEAT_PHASE == 11
    ts_commit<T>(m, e, s, tl, nc, n_con, iters);
    team_sync();
#endif
    ts_commit<T>(m, e, s, tl, nc, n_con, iters);
    PH(11)
  } else {
    ts_solve_overflow<T>(P.f, P.i, P.B, P.fa, P.ia, b, lnk, tri_i, tri_j, gm, &overflow[b], tl, nc, n_con, lim_mask, ws_flag);
  }
}
template <int T, int RLN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_constraint_solve_team(Pool P, const Model* __restrict__ gm, const ModelS* __restrict__ mp, SolverData<MAXR>* __restrict__ overflow,
                                                                                                         const int* __restrict__ lpt_rec, int* __restrict__ lpt_next, int lpt_cap) {
  STAMP(STK_SOLVE)
  constexpr int EPW = 64 / T;
  __shared__ SolverData<RLN> lds[EPW];
  __shared__ alignas(16) char blk_raw[lds_dma_bytes(SOLVER_BLOCK_BYTES)];
  int b; bool resident;
  solve_body<T, RLN>(P, gm, mp, overflow, lpt_rec, lpt_next, lpt_cap, lds, blk_raw, b, resident);
}

// Constraint solve of substep i, then -- in the SAME wavefront, for the same environments -- kernel_step_2 of that substep (integrate, kinematics) and, between the
// substeps (WITH_DYN), the forward dynamics of substep i + 1: k_constraint_solve_team followed by k_integrate_fk_dynamics_team / k_integrate_fk_team without the
// launch boundary.  A solver launch lasts as long as its slowest workgroup (landing window: mean 31 us, span 66 us) and the kinematics / dynamics that follow
// are uniform: in one launch the workgroups whose solve ends early go straight on, so the launch costs the slowest solve plus ITS kinematics instead of the
// slowest solve plus a grid-wide join plus everybody's kinematics.  The LDS blocks of the second half are laid over the solver's (same footprint: 8 workgroups per
// CU); velocity and acceleration of the team's dofs pass in registers.  Same code per value as the two launches: results unchanged.
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
template <int T, int RLN, bool WITH_DYN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_solve_integrate_team(Pool P, const Model* __restrict__ gm, const ModelS* __restrict__ mp,
                                                                                                        SolverData<MAXR>* __restrict__ overflow, const int* __restrict__ lpt_rec,
                                                                                                        int* __restrict__ lpt_next, int lpt_cap) {
  STAMP(STK_SOLVE)
  constexpr int EPW = 64 / T;
  constexpr size_t SOLVE_OFF = (sizeof(SolverData<RLN>) * EPW + 15) / 16 * 16, DYN_OFF = (sizeof(DynData) * EPW + 15) / 16 * 16, KIN_OFF = (sizeof(KinData) * EPW + 15) / 16 * 16;   // the DMA targets behind the blocks: 16-byte aligned
  constexpr size_t SOLVE_BYTES = SOLVE_OFF + lds_dma_bytes(SOLVER_BLOCK_BYTES);
  constexpr size_t DYN_BYTES = DYN_OFF + MODELS_LDS_BYTES + sizeof(KinSeparate<KIN_OVERLAY, EPW>);
  constexpr size_t KIN_BYTES = KIN_OFF + MODELS_LDS_BYTES;
  __shared__ alignas(16) char raw[cmax(SOLVE_BYTES, WITH_DYN ? DYN_BYTES : KIN_BYTES)];
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  int b; bool resident;
  SolverData<RLN>* lds = (SolverData<RLN>*)raw;
  solve_body<T, RLN>(P, gm, mp, overflow, lpt_rec, lpt_next, lpt_cap, lds, raw + SOLVE_OFF, b, resident);
  // velocity and acceleration of my dof, before the solver's block is overlaid (an env whose solve ran on the global scratch block reads them back from HBM)
  const bool pre = b < P.B && resident;
  const int dd = tl < ND ? tl : ND - 1;
  const float pre_vel = pre ? lds[slot].vel[dd] : 0.0f, pre_acc = pre ? lds[slot].qacc[dd] : 0.0f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");              // the commit's stores (acc of a scratch-block env) before the loads of the second half
  __syncthreads();                                                    // both teams are done with the solver's LDS blocks
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if constexpr (WITH_DYN) {
    DynData* lds_d = (DynData*)raw;
    char* ms_raw = raw + DYN_OFF;
    auto& lds_k = *(KinSeparate<KIN_OVERLAY, EPW>*)(ms_raw + MODELS_LDS_BYTES);
    integrate_fk_dynamics_body<T>(P, mp, b, lds_d, ms_raw, lds_k, pre, pre_vel, pre_acc);
  } else {
    integrate_fk_body<T>(P, mp, b, (KinData*)raw, raw + KIN_OFF, pre, pre_vel, pre_acc);
  }
}

// diagnostics (go2sim_debug_narrowphase): one narrow-phase query on explicit poses, by one lane.  which = 0: MPR from a cold start
// (func_mpr_contact, mpr.py:763-819); 1: safe GJK + EPA the way k_collide_team runs it (LDS polytope slot, global record on overflow);
// 2: safe GJK + EPA on the full-capacity global record only; 3 / 4: the cooperative form k_collide_team<16> runs (the 4 lanes of a quad answer the
// query together) on an LDS slot / on the global record; 5 / 6: the same with 16 lanes per query.  out = {is_col, penetration, normal[3], pos[3]}
__global__ __launch_bounds__(64) void k_debug_narrowphase(const Model* __restrict__ mp, int which, int i_ga, int i_gb, V3 pa, Q4 qa, V3 pb, Q4 qb,
                                                          GjkStoreFull* __restrict__ full, float* __restrict__ out8) {
  __shared__ GjkStoreLds slots[GJK_SLOTS_MAX];
  __shared__ GjkStoreTeam team_store;
  __shared__ unsigned mask;
  if (threadIdx.x == 0) mask = 0u;
  __syncthreads();
  const Model& m = *mp;
  bool is_col = false; V3 normal = v3(0, 0, 0), pos = v3(0, 0, 0); float pen = 0.0f;
  if (which >= 3) {
    if (threadIdx.x >= 16) return;
    DgPair dp; dp.m = mp; dp.i_ga = i_ga; dp.i_gb = i_gb; dp.pos_a = pa; dp.quat_a = qa; dp.pos_b = pb; dp.quat_b = qb;
    dp.ga = geom_lite(m, i_ga); dp.gb = geom_lite(m, i_gb); dp.ra = make_rot(qa); dp.rb = make_rot(qb);
    dp.discrete = m.geoms[i_ga].type == GEOM_BOX && m.geoms[i_gb].type == GEOM_BOX;
    DgResult r;
    if (which <= 4) {                                                   // the form the collision kernel runs: a quad per query, the quad's LDS slot
      if (threadIdx.x >= 4) return;
      r = (which == 3) ? dgc_contact<4>(dp, slots[0], m.eps, (int)threadIdx.x) : dgc_contact<4>(dp, *full, m.eps, (int)threadIdx.x);
      if (r.overflow) r = dgc_contact<4>(dp, *full, m.eps, (int)threadIdx.x);
    } else {                                                            // 16 lanes (one DPP row) on one query
      r = (which == 5) ? dgc_contact<16>(dp, team_store, m.eps, (int)threadIdx.x) : dgc_contact<16>(dp, *full, m.eps, (int)threadIdx.x);
      if (r.overflow) r = dgc_contact<16>(dp, *full, m.eps, (int)threadIdx.x);
    }
    if (threadIdx.x != 0) return;
    is_col = r.is_col; pen = r.penetration; normal = r.normal; pos = r.pos;
    out8[0] = is_col ? 1.0f : 0.0f; out8[1] = pen; out8[2] = normal.x; out8[3] = normal.y; out8[4] = normal.z; out8[5] = pos.x; out8[6] = pos.y; out8[7] = pos.z;
    return;
  }
  if (threadIdx.x != 0) return;
  if (which == 0) {
    Pair pr; pr.i_ga = i_ga; pr.i_gb = i_gb; pr.pos_a = pa; pr.quat_a = qa; pr.pos_b = pb; pr.quat_b = qb; pr.prism = nullptr; pr.ga = geom_lite(m, i_ga); pr.gb = geom_lite(m, i_gb);
    pair_set_rots(pr);
    mpr_contact(m, pr, v3(0, 0, 0), is_col, normal, pen, pos);
  } else {
    DgPair dp; dp.m = mp; dp.i_ga = i_ga; dp.i_gb = i_gb; dp.pos_a = pa; dp.quat_a = qa; dp.pos_b = pb; dp.quat_b = qb;
    dp.ga = geom_lite(m, i_ga); dp.gb = geom_lite(m, i_gb); dp.ra = make_rot(qa); dp.rb = make_rot(qb);
    dp.discrete = m.geoms[i_ga].type == GEOM_BOX && m.geoms[i_gb].type == GEOM_BOX;
    const DgResult r = (which == 1) ? gjk_query(dp, slots, &mask, full, m.eps) : gjk_query(dp, nullptr, &mask, full, m.eps);
    is_col = r.is_col; pen = r.penetration; normal = r.normal; pos = r.pos;
  }
  out8[0] = is_col ? 1.0f : 0.0f; out8[1] = pen; out8[2] = normal.x; out8[3] = normal.y; out8[4] = normal.z; out8[5] = pos.x; out8[6] = pos.y; out8[7] = pos.z;
}

__global__ __launch_bounds__(WG) void k_clear_ext(Pool P) {             // kernel_clear_external_force, abd/misc.py:874
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  auto ext = e.ext();
  for (int i = 0; i < NL * 6; ++i) ext[i] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Go2Env (walk)  -- E/go2_env_walk.py.  All of Go2Env.step runs on the device: four per-env kernels
// around the physics plus one single-thread kernel for the quantities the reference keeps in Python
// scalars (curriculum state machine, "global" domain-randomisation draws).
// ---------------------------------------------------------------------------------------------
// d[]: the host scalars in double (include/go2sim.h enum go2sim_fcfg, entries below GO2SIM_FC_N_HOST); f[]: every entry rounded to float32
struct DCfg { double d[GO2SIM_FC_N_HOST]; float f[GO2SIM_FC_COUNT]; int i[GO2SIM_IC_COUNT]; };
struct Acc { double timeouts, tracking, ep[NREW]; int n_reset_now; int done; };   // done: workgroups of k_env_post_a that have finished
typedef go2sim_env_globals_t Glob;

enum { RNG_ACTION_NOISE = 1, RNG_PUSH = 2, RNG_CMD = 3, RNG_OBS_NOISE = 4, RNG_RESET_DR = 5, RNG_GLOBAL_DR = 6, RNG_RESET_CMD = 7, RNG_RESET_POSE = 8, RNG_TERRAIN_ROW = 9, RNG_TERRAIN_PERM = 10 };
__host__ __device__ inline dm_u4 rng4(uint64_t seed, uint32_t purpose, uint32_t env, uint32_t step, uint32_t idx) {
#ifdef GO2SIM_RNG_CONST   // diagnostic build (include/go2sim_detmath.h): every word of a draw is its stream key
  dm_u4 o; o.v[0] = o.v[1] = o.v[2] = o.v[3] = step; (void)purpose; (void)env; (void)idx; (void)seed; return o;
#else
  return dm_philox(env, step, purpose, idx, (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
}
// gs_rand_float, go2_env_walk.py:7-8: `(upper - lower) * torch.rand(...) + lower` with python-float bounds: the difference is formed in float64 and
// both scalars are rounded to float32 where they meet the float32 tensor
__host__ __device__ inline float rand_float(double lower, double upper, uint32_t r) { return (float)(upper - lower) * dm_u01(r) + (float)lower; }
#ifdef GO2SIM_RNG_CONST
DEV int rand_int(int lower, int upper, uint32_t r) { return lower + (int)(dm_rng_const_u(r) * (float)(upper - lower + 1)); }
#else
DEV int rand_int(int lower, int upper, uint32_t r) { return lower + (int)(r % (uint32_t)(upper - lower + 1)); }   // gs_rand_int, :11-13
#endif
__host__ __device__ inline double clamp01d(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
__host__ __device__ inline double lerpd(double a, double b, double t) { t = clamp01d(t); return a + (b - a) * t; }
// _lerp_range(easy, hard, t_sample), go2_env_walk.py:37-39: python floats
__host__ __device__ inline double lerp_lo(const DCfg& c, int easy_lo, double t) { return lerpd(c.d[easy_lo], c.d[easy_lo + 2], t); }
__host__ __device__ inline double lerp_hi(const DCfg& c, int easy_lo, double t) { return lerpd(c.d[easy_lo + 1], c.d[easy_lo + 3], t); }
// Go2Env._resample_commands, go2_env_walk.py:927-963: all three components (compound commands), or ONE component chosen by `randint(0, 3)` with the
// other two at zero; the first `rel_standing_envs * num_envs` envs always stand (:367-368, :960-963).  Words 0..2 of the block are the
// components' uniforms, word 3 the choice.
DEV void sample_commands(const DCfg& c, const Glob& g, const dm_u4& r, int b, float& cx, float& cy, float& cz) {
  cx = rand_float(g.cmd_x_lo, g.cmd_x_hi, r.v[0]); cy = rand_float(g.cmd_y_lo, g.cmd_y_hi, r.v[1]); cz = rand_float(g.cmd_yaw_lo, g.cmd_yaw_hi, r.v[2]);
  if (!c.i[GO2SIM_IC_COMPOUND_COMMANDS]) {
    const int choice = rand_int(0, 2, r.v[3]);
    if (choice != 0) cx = 0.0f;
    if (choice != 1) cy = 0.0f;
    if (choice != 2) cz = 0.0f;
  }
  if (b < c.i[GO2SIM_IC_N_STANDING]) { cx = 0.0f; cy = 0.0f; cz = 0.0f; }
}

// torch-side helpers of genesis/utils/geom.py used by Go2Env (evaluation order of the torch code)
DEV Q4 tc_quat_mul(Q4 u, Q4 v) {                                   // geom.py:989-1007
  float w1 = u.w, x1 = u.x, y1 = u.y, z1 = u.z, w2 = v.w, x2 = v.x, y2 = v.y, z2 = v.z;
  float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  Q4 o = q4(qq - ww + (z1 - y1) * (y2 - z2), qq - xx + (x1 + w1) * (x2 + w2), qq - yy + (w1 - x1) * (y2 + z2), qq - zz + (z1 + y1) * (w2 - x2));
  float n = dm_sqrt(norm_sqr(o));
  return q4(o.w / n, o.x / n, o.y / n, o.z / n);
}
DEV V3 tc_transform_by_quat(V3 v, Q4 q) {                          // geom.py:1052-1070
  float q_ww = q.w * q.w, q_wx = q.w * q.x, q_wy = q.w * q.y, q_wz = q.w * q.z;
  float q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_yy = q.y * q.y, q_yz = q.y * q.z, q_zz = q.z * q.z;
  float den = q_ww + q_xx + q_yy + q_zz;
  float vx = v.x / den, vy = v.y / den, vz = v.z / den;
  return v3(vx * (q_xx + q_ww - q_yy - q_zz) + vy * (2.0f * q_xy - 2.0f * q_wz) + vz * (2.0f * q_xz + 2.0f * q_wy),
            vx * (2.0f * q_wz + 2.0f * q_xy) + vy * (q_ww - q_xx + q_yy - q_zz) + vz * (2.0f * q_yz - 2.0f * q_wx),
            vx * (2.0f * q_xz - 2.0f * q_wy) + vy * (2.0f * q_wx + 2.0f * q_yz) + vz * (q_ww - q_xx - q_yy + q_zz));
}
DEV V3 tc_quat_to_xyz_rpy_deg(Q4 q, float eps) {                   // geom.py:717-762 (rpy=True) + rad2deg
  float q_ww = q.w * q.w, q_wx = q.w * q.x, q_wy = q.w * q.y, q_wz = q.w * q.z;
  float q_xx = q.x * q.x, q_xy = q.x * q.y, q_xz = q.x * q.z, q_yy = q.y * q.y, q_yz = q.y * q.z, q_zz = q.z * q.z;
  float sinp = q_wy - q_xz, sinrcosp = q_wx + q_yz, sinycosp = q_wz + q_xy;
  float cosrcosp = (q_ww - q_xx - q_yy + q_zz) / 2.0f, cosycosp = (q_ww + q_xx - q_yy - q_zz) / 2.0f;
  float cosp = dm_sqrt(cosycosp * cosycosp + sinycosp * sinycosp);
  float x = dm_atan2(sinrcosp, cosrcosp), y = dm_atan2(sinp, cosp), z = dm_atan2(sinycosp, cosycosp);
  if (cosp < eps) { x = 0.0f; z = dm_atan2(q_wz - q_xy, (q_ww - q_xx + q_yy - q_zz) / 2.0f); }
  const float R2D = 57.29577951308232f;
  return v3(x * R2D, y * R2D, z * R2D);
}

// Go2Env._apply_curriculum_level, go2_env_walk.py:628-686 (python float64 arithmetic)
// _get_dr_level, go2_env_stair.py:972-988 (two-phase DR schedule coupled to the terrain level)
__host__ __device__ inline double dr_level(const DCfg& c, double terrain_level) {
  if (!c.i[GO2SIM_IC_DR_SCHEDULE]) return terrain_level;
  double gate = c.d[GO2SIM_FC_DR_TERRAIN_GATE], p1 = c.d[GO2SIM_FC_DR_PHASE1_LEVEL];
  if (terrain_level < gate) return p1;
  double den = 1.0 - gate; if (den < 1e-6) den = 1e-6;
  double progress = clamp01d((terrain_level - gate) / den);
  return lerpd(p1, 1.0, progress);
}
// heightfield lookup of the env code (_get_terrain_height, go2_env_stair.py:758-770): truncation toward zero, then clamping
DEV float terrain_height(const Model& m, const DCfg& c, float x, float y) {
  if (!c.i[GO2SIM_IC_USE_TERRAIN] || !m.terrain_enabled) return 0.0f;
  long long col = (long long)((x - c.f[GO2SIM_FC_TERRAIN_ORIGIN_X]) / c.f[GO2SIM_FC_TERRAIN_H_SCALE]);
  long long row = (long long)((y - c.f[GO2SIM_FC_TERRAIN_ORIGIN_Y]) / c.f[GO2SIM_FC_TERRAIN_H_SCALE]);
  col = col < 0 ? 0 : (col > m.terrain_rows - 1 ? m.terrain_rows - 1 : col);
  row = row < 0 ? 0 : (row > m.terrain_cols - 1 ? m.terrain_cols - 1 : row);
  return m.terrain_hf[(size_t)col * m.terrain_cols + row];
}
__host__ __device__ inline void apply_curriculum_level(const DCfg& c, Glob& g) {
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
      double den = 1.0 - push_start; if (den < 1e-6) den = 1e-6;
      double s = clamp01d((lvl - push_start) / den);
      g.push_force_lo = c.d[GO2SIM_FC_PUSH_FORCE_LO] * s;
      g.push_force_hi = c.d[GO2SIM_FC_PUSH_FORCE_HI] * s;
      double interval_s = lerpd(c.d[GO2SIM_FC_PUSH_INTERVAL_S_EASY], c.d[GO2SIM_FC_PUSH_INTERVAL_S_HARD], s);
      int iv = (int)(interval_s / dt);
      g.push_interval = iv < 1 ? 1 : iv;
      g.push_enable = 1;
    }
  }
  g.delay_max_cur = (int)rint(lerpd((double)c.i[GO2SIM_IC_DELAY_EASY_MAX], (double)c.i[GO2SIM_IC_MAX_DELAY], lvl));
  double frac = c.i[GO2SIM_IC_CMD_CURRICULUM] ? lerpd(c.d[GO2SIM_FC_CMD_START_FRAC], 1.0, lvl_terrain) : 1.0;
  {
    double lo = c.d[GO2SIM_FC_CMD_X_LO], hi = c.d[GO2SIM_FC_CMD_X_HI], center = (lo + hi) / 2.0, half = (hi - lo) / 2.0;
    g.cmd_x_lo = center - half * frac; g.cmd_x_hi = center + half * frac;
    lo = c.d[GO2SIM_FC_CMD_Y_LO]; hi = c.d[GO2SIM_FC_CMD_Y_HI]; center = (lo + hi) / 2.0; half = (hi - lo) / 2.0;
    g.cmd_y_lo = center - half * frac; g.cmd_y_hi = center + half * frac;
    lo = c.d[GO2SIM_FC_CMD_YAW_LO]; hi = c.d[GO2SIM_FC_CMD_YAW_HI]; center = (lo + hi) / 2.0; half = (hi - lo) / 2.0;
    g.cmd_yaw_lo = center - half * frac; g.cmd_yaw_hi = center + half * frac;
  }
}

// CurriculumManager.update, go2_env_walk.py:101-142
__host__ __device__ inline bool curriculum_update(const DCfg& c, Glob& g, double timeout_rate, double tracking_per_sec, double fall_rate) {
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
    { double lo = c.d[GO2SIM_FC_CURR_LEVEL_MIN], v = g.level - c.d[GO2SIM_FC_CURR_STEP_DOWN]; g.level = (lo < v) ? v : lo; }      // std::max(lo, v)
    g.hard_streak = 0; g.ready_streak = 0; g.cooldown = c.i[GO2SIM_IC_CURR_COOLDOWN];
  } else if (g.ready_streak >= c.i[GO2SIM_IC_CURR_READY_STREAK] && g.cooldown == 0) {
    { double hi = c.d[GO2SIM_FC_CURR_LEVEL_MAX], v = g.level + c.d[GO2SIM_FC_CURR_STEP_UP]; g.level = (v < hi) ? v : hi; }        // std::min(hi, v)
    g.ready_streak = 0; g.cooldown = c.i[GO2SIM_IC_CURR_COOLDOWN];
  }
  g.level = clamp01d(g.level);
  return g.level != old_level;
}

// Go2Env.step pre-physics part: go2_env_walk.py:985-1023 (+ _apply_push :872-906)
__global__ __launch_bounds__(WG) void k_env_pre(Pool P, const Model* __restrict__ mp, const DCfg cv, const Glob* __restrict__ gp,
                                                const float* __restrict__ actions_in, uint64_t seed, uint32_t step_count, int write_idx) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  const DCfg& c = cv;  const Glob& g = *gp;
  E e(P, b);
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS];
  float clip = c.f[GO2SIM_FC_CLIP_ACTIONS];
  auto actions = e.actions(); auto hist = e.action_history(); auto applied = e.applied_actions();
  auto target_dof_pos = e.target_dof_pos(); auto kp_factors = e.kp_factors(); auto kd_factors = e.kd_factors(); auto motor_strength = e.motor_strength();
  auto dof_pos = e.e_dof_pos(); auto dof_vel = e.e_dof_vel(); auto torque_ = e.torque(); auto ctrl_mode = e.ctrl_mode(); auto ctrl_force = e.ctrl_force();
  auto cpf = e.current_push_force(); auto psf = e.push_stored_force(); auto prem = e.push_remaining(); auto ext = e.ext();
  const bool manual_pd = c.i[GO2SIM_IC_MANUAL_PD] != 0, pls = c.i[GO2SIM_IC_PLS_ENABLE] != 0;
  const bool push_on = c.i[GO2SIM_IC_HAS_PUSH] && g.push_enable;
  const int pl = c.i[GO2SIM_IC_PUSH_LINK];
  // ---- loads first (one wave per 64 envs: the kernel's duration is its chain of memory round trips) ----
  int delay = e.delay_steps()[0];
  // the ring of the walk / stair envs (_action_history[B, max_delay_steps + 1, A], go2_env_walk.py:373-380, 916-923); the base env has none:
  // it executes last_actions, which reset_idx zeroes and the end of the same step overwrites (go2_env_base.py:124-125, 187, 225)
  const bool base_env = c.i[GO2SIM_IC_ENV_KIND] == 1;
  const int depth = c.i[GO2SIM_IC_MAX_DELAY] + 1;
  const int read_idx = (((write_idx - delay) % depth) + depth) % depth;
  auto last_a = e.last_actions();
  float a_in[NA], h_[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    a_in[i] = (i < na) ? actions_in[(size_t)b * na + i] : 0.0f;
    h_[i] = base_env ? last_a[i] : hist[read_idx][i];
  }
  float kpf[NM], kdf[NM], mst[NM], dp[NM], dv[NM];
#pragma unroll
  for (int i = 0; i < NM; ++i) { kpf[i] = kp_factors[i]; kdf[i] = kd_factors[i]; mst[i] = motor_strength[i]; dp[i] = dof_pos[i]; dv[i] = dof_vel[i]; }
  float psf_[3] = {psf[0], psf[1], psf[2]}; int rem0 = prem[0];
  V3 pl_pos = e.l_pos()[pl], pl_com = e.root_com()[pl];
  float ext_[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ext_[k] = ext[6 * pl + k];
  // ---- compute + stores ----
  float delayed[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if (i < na) {
      float a = fmn(fmx(a_in[i], -clip), clip);
      actions[i] = a;
      float d;
      if (base_env) d = (depth > 1) ? h_[i] : a;
      else { hist[write_idx][i] = a; d = (read_idx == write_idx) ? a : h_[i]; }
      delayed[i] = d; applied[i] = d;
    } else {
      delayed[i] = 0.0f;
    }
  }
  float target[NM];
#pragma unroll
  for (int i = 0; i < NM; ++i) target[i] = delayed[i] * c.f[GO2SIM_FC_ACTION_SCALE] + c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i];
  if (g.action_noise_std_cur > 0.0) {
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
      dm_u4 r = rng4(seed, RNG_ACTION_NOISE, b, step_count, blk);
      float n0, n1, n2, n3;
      dm_normal2(r.v[0], r.v[1], &n0, &n1); dm_normal2(r.v[2], r.v[3], &n2, &n3);
      target[4 * blk + 0] = target[4 * blk + 0] + n0 * (float)g.action_noise_std_cur;
      target[4 * blk + 1] = target[4 * blk + 1] + n1 * (float)g.action_noise_std_cur;
      target[4 * blk + 2] = target[4 * blk + 2] + n2 * (float)g.action_noise_std_cur;
      target[4 * blk + 3] = target[4 * blk + 3] + n3 * (float)g.action_noise_std_cur;
    }
  }
  if (!manual_pd) {                                                  // go2_env_base.py:127: control_dofs_position (engine PD)
    auto ctrl_pos = e.ctrl_pos(); auto ctrl_vel = e.ctrl_vel();
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      target_dof_pos[i] = target[i];
      int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
      ctrl_mode[d] = CTRL_POSITION; ctrl_pos[d] = target[i]; ctrl_vel[d] = 0.0f; ctrl_force[d] = 0.0f;
      torque_[i] = 0.0f;
    }
  } else
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    target_dof_pos[i] = target[i];
    float eff_kp, eff_kd;
    if (pls) {                                                       // _compute_pls_kp_kd :969-979
      int leg = i / 3;
      float kp_leg = c.f[GO2SIM_FC_PLS_KP_DEFAULT] + delayed[NM + leg] * c.f[GO2SIM_FC_PLS_KP_ACTION_SCALE];
      kp_leg = fmn(fmx(kp_leg, c.f[GO2SIM_FC_PLS_KP_MIN]), c.f[GO2SIM_FC_PLS_KP_MAX]);
      float kd_j = 0.2f * dm_sqrt(kp_leg);
      eff_kp = kp_leg * kpf[i] * mst[i];
      eff_kd = kd_j * kdf[i];
    } else {
      eff_kp = c.f[GO2SIM_FC_KP] * kpf[i]; eff_kd = c.f[GO2SIM_FC_KD] * kdf[i];
    }
    float pos_error = target[i] - dp[i];
    float torque = eff_kp * pos_error - eff_kd * dv[i];
    float lim = c.f[GO2SIM_FC_TORQUE_LIMIT0 + i];
    torque = fmn(fmx(torque, -lim), lim);
    torque_[i] = torque;
    int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
    ctrl_mode[d] = CTRL_FORCE; ctrl_force[d] = torque;
  }
  if (!push_on) {
    cpf[0] = 0.0f; cpf[1] = 0.0f; cpf[2] = 0.0f;
  } else {
    if (g.push_counter % g.push_interval == 0) {
      dm_u4 r = rng4(seed, RNG_PUSH, b, step_count, 0);
      psf_[0] = rand_float(g.push_force_lo, g.push_force_hi, r.v[0]);
      psf_[1] = rand_float(g.push_force_lo, g.push_force_hi, r.v[1]);
      psf_[2] = 0.0f;
      psf[0] = psf_[0]; psf[1] = psf_[1]; psf[2] = psf_[2];
      rem0 = rand_int(c.i[GO2SIM_IC_PUSH_DUR_LO], c.i[GO2SIM_IC_PUSH_DUR_HI], r.v[2]);
    }
    int rem = rem0;
    float active = (rem > 0) ? 1.0f : 0.0f;
    V3 force = v3(psf_[0] * active, psf_[1] * active, psf_[2] * active);
    cpf[0] = force.x; cpf[1] = force.y; cpf[2] = force.z;
    prem[0] = imx(rem - 1, 0);
    V3 tq = cross(pl_pos - pl_com, force);                             // func_apply_link_external_force ref=link_origin, abd/misc.py:695-715
    ext[6 * pl + 0] = ext_[0] - tq.x; ext[6 * pl + 1] = ext_[1] - tq.y; ext[6 * pl + 2] = ext_[2] - tq.z;
    ext[6 * pl + 3] = ext_[3] - force.x; ext[6 * pl + 4] = ext_[4] - force.y; ext[6 * pl + 5] = ext_[5] - force.z;
  }
}

// k_env_pre and the forward dynamics of the first substep in one launch: the pre-physics part of Go2Env.step is element-wise over the 12 motors
// (lane i = action / motor i; the push bookkeeping on one lane), its outputs are exactly the control inputs the dynamics stage -- they go to
// the pool (the second substep, the post kernels and get_field read them there) and straight into the dynamics working set in LDS.
// Same arithmetic per element as k_env_pre (go2_env_walk.py:985-1023, _apply_push :872-906).
template <int T>
__global__ __launch_bounds__(64) void k_pre_dynamics_team(Pool P, const ModelS* __restrict__ mp, const DCfg cv, const Glob* __restrict__ gp,
                                                          const float* __restrict__ actions_in, uint64_t seed, uint32_t step_count, int write_idx) {
  STAMP(STK_PRE_DYN)
  static_assert(T >= NA, "one lane per action");
  constexpr int EPW = 64 / T;
  __shared__ DynData lds[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  const ModelS& ms = *(const ModelS*)ms_raw;
  wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, mp);
  {
    const int b0 = xcd_block() * EPW;
    wg_load<EPW, NL * 3>(P, b0, FO(cd_vel), [&](int ev, int k, float v) { lds[ev].cd_vel[k] = v; });
    wg_load<EPW, NL * 3>(P, b0, FO(cd_ang), [&](int ev, int k, float v) { lds[ev].cd_ang[k] = v; });
    wg_load<EPW, ND>(P, b0, FO(vel), [&](int ev, int k, float v) { lds[ev].vel[k] = v; });
  }
  const int tl = threadIdx.x % T, slot = threadIdx.x / T;
  const int b = xcd_block() * EPW + slot;
  const bool env_valid = b < P.B;
  const ModelView m(&ms, mp);
  E e(P, env_valid ? b : P.B - 1);
  DynData* s = &lds[slot];
  const DCfg& c = cv; const Glob& g = *gp;
  PH_BEGIN
  // ---- loads of the pre-physics part (lane i: action / motor i) ----
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS];
  const int ia = tl < NA ? tl : NA - 1, im = tl < NM ? tl : NM - 1;
  const int eb = env_valid ? b : P.B - 1;
  auto hist = e.action_history();
  const int delay = e.delay_steps()[0];
  const float a_in = (ia < na) ? actions_in[(size_t)eb * na + ia] : 0.0f;
  const bool base_env = c.i[GO2SIM_IC_ENV_KIND] == 1;                     // see k_env_pre: ring (walk / stairs) vs last_actions (base env)
  const int depth = c.i[GO2SIM_IC_MAX_DELAY] + 1;
  const int read_idx = (((write_idx - delay) % depth) + depth) % depth;
  const float h_ = base_env ? e.last_actions()[ia] : hist[read_idx][ia];
  const float kpf = e.kp_factors()[im], kdf = e.kd_factors()[im], mst = e.motor_strength()[im], dp = e.e_dof_pos()[im], dv = e.e_dof_vel()[im];
  const bool manual_pd = c.i[GO2SIM_IC_MANUAL_PD] != 0, pls = c.i[GO2SIM_IC_PLS_ENABLE] != 0;
  const bool push_on = c.i[GO2SIM_IC_HAS_PUSH] && g.push_enable;
  const int pl = c.i[GO2SIM_IC_PUSH_LINK];
  auto psf = e.push_stored_force(); auto prem = e.push_remaining(); auto cpf = e.current_push_force(); auto ext = e.ext();
  float psf_[3] = {psf[0], psf[1], psf[2]}; int rem0 = prem[0];
  const V3 pl_pos = e.l_pos()[pl], pl_com = e.root_com()[pl];
  float ext_[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ext_[k] = ext[6 * pl + k];
  // ---- staging of the dynamics inputs, as k_dynamics_team ----
  team_stage<NL * 9, T>(tl, [&](int k) { return aload(e, AO(cinr_inertial), k); }, [&](int k, float v) { s->cinr_I[k] = v; s->crb_I[k] = v; });
  team_stage<NL * 3, T>(tl, [&](int k) { return aload(e, AO(cinr_pos), k); }, [&](int k, float v) { s->cinr_pos[k] = v; s->crb_pos[k] = v; });
  team_stage<NL, T>(tl, [&](int k) { return aload(e, AO(cinr_mass), k); }, [&](int k, float v) { s->cinr_mass[k] = v; s->crb_mass[k] = v; });
  team_stage<ND, T>(tl, [&](int d) { return __int_as_float((int)e.ctrl_mode()[d]); }, [&](int d, float v) { s->ctrl_mode[d] = __float_as_int(v); });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_ang), k); }, [&](int k, float v) { s->cdof_ang[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdof_vel), k); }, [&](int k, float v) { s->cdof_vel[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdofd_ang), k); }, [&](int k, float v) { s->cdofd_ang[k] = v; });
  team_stage<ND * 3, T>(tl, [&](int k) { return aload(e, AO(cdofd_vel), k); }, [&](int k, float v) { s->cdofd_vel[k] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_force), d); }, [&](int d, float v) { s->qf_applied[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_pos), d); }, [&](int d, float v) { s->qf_passive[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(ctrl_vel), d); }, [&](int d, float v) { s->force[d] = v; });
  team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(dof_pos), d); }, [&](int d, float v) { s->out[d] = v; });
  team_stage<NL * 6, T>(tl, [&](int k) { return gload(e, FO(ext), k); },
                        [&](int k, float v) { int i_l = k / 6, cc = k % 6; if (cc < 3) s->cfrc_ang[3 * i_l + cc] = v; else s->cfrc_vel[3 * i_l + cc - 3] = v; });
  team_sync();
  // ---- the pre-physics part; the parked control inputs of the motor dofs and the external force of the pushed link are replaced in LDS ----
  if (env_valid) {
    const float clip = c.f[GO2SIM_FC_CLIP_ACTIONS];
    float delayed = 0.0f;
    if (tl < NA && tl < na) {
      const float a = fmn(fmx(a_in, -clip), clip);
      e.actions()[tl] = a;
      if (base_env) delayed = (depth > 1) ? h_ : a;
      else { hist[write_idx][tl] = a; delayed = (read_idx == write_idx) ? a : h_; }
      e.applied_actions()[tl] = delayed;
    }
    const float delayed_leg = __shfl(delayed, NM + im / 3, T);            // per-leg stiffness action of the PLS policy (actions 12..15)
    if (tl < NM) {
      float target = delayed * c.f[GO2SIM_FC_ACTION_SCALE] + c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + tl];
      if (g.action_noise_std_cur > 0.0) {
        dm_u4 r = rng4(seed, RNG_ACTION_NOISE, b, step_count, tl / 4);
        float n0, n1;
        if ((tl & 2) == 0) dm_normal2(r.v[0], r.v[1], &n0, &n1); else dm_normal2(r.v[2], r.v[3], &n0, &n1);
        target = target + ((tl & 1) ? n1 : n0) * (float)g.action_noise_std_cur;
      }
      e.target_dof_pos()[tl] = target;
      const int d = c.i[GO2SIM_IC_MOTOR_DOF0 + tl];
      if (!manual_pd) {                                                  // go2_env_base.py:127: control_dofs_position (engine PD)
        e.ctrl_mode()[d] = CTRL_POSITION; e.ctrl_pos()[d] = target; e.ctrl_vel()[d] = 0.0f; e.ctrl_force()[d] = 0.0f;
        e.torque()[tl] = 0.0f;
        s->ctrl_mode[d] = CTRL_POSITION; s->qf_passive[d] = target; s->force[d] = 0.0f; s->qf_applied[d] = 0.0f;
      } else {
        float eff_kp, eff_kd;
        if (pls) {                                                       // _compute_pls_kp_kd :969-979
          float kp_leg = c.f[GO2SIM_FC_PLS_KP_DEFAULT] + delayed_leg * c.f[GO2SIM_FC_PLS_KP_ACTION_SCALE];
          kp_leg = fmn(fmx(kp_leg, c.f[GO2SIM_FC_PLS_KP_MIN]), c.f[GO2SIM_FC_PLS_KP_MAX]);
          const float kd_j = 0.2f * dm_sqrt(kp_leg);
          eff_kp = kp_leg * kpf * mst;
          eff_kd = kd_j * kdf;
        } else {
          eff_kp = c.f[GO2SIM_FC_KP] * kpf; eff_kd = c.f[GO2SIM_FC_KD] * kdf;
        }
        const float pos_error = target - dp;
        float torque = eff_kp * pos_error - eff_kd * dv;
        const float lim = c.f[GO2SIM_FC_TORQUE_LIMIT0 + tl];
        torque = fmn(fmx(torque, -lim), lim);
        e.torque()[tl] = torque;
        e.ctrl_mode()[d] = CTRL_FORCE; e.ctrl_force()[d] = torque;
        s->ctrl_mode[d] = CTRL_FORCE; s->qf_applied[d] = torque;
      }
    }
    if (tl == T - 1) {                                                   // push bookkeeping (a lane without a motor)
      if (!push_on) {
        cpf[0] = 0.0f; cpf[1] = 0.0f; cpf[2] = 0.0f;
      } else {
        if (g.push_counter % g.push_interval == 0) {
          dm_u4 r = rng4(seed, RNG_PUSH, b, step_count, 0);
          psf_[0] = rand_float(g.push_force_lo, g.push_force_hi, r.v[0]);
          psf_[1] = rand_float(g.push_force_lo, g.push_force_hi, r.v[1]);
          psf_[2] = 0.0f;
          psf[0] = psf_[0]; psf[1] = psf_[1]; psf[2] = psf_[2];
          rem0 = rand_int(c.i[GO2SIM_IC_PUSH_DUR_LO], c.i[GO2SIM_IC_PUSH_DUR_HI], r.v[2]);
        }
        const float active = (rem0 > 0) ? 1.0f : 0.0f;
        const V3 force = v3(psf_[0] * active, psf_[1] * active, psf_[2] * active);
        cpf[0] = force.x; cpf[1] = force.y; cpf[2] = force.z;
        prem[0] = imx(rem0 - 1, 0);
        const V3 tq = cross(pl_pos - pl_com, force);                      // func_apply_link_external_force ref=link_origin, abd/misc.py:695-715
        const float x0 = ext_[0] - tq.x, x1 = ext_[1] - tq.y, x2 = ext_[2] - tq.z, x3 = ext_[3] - force.x, x4 = ext_[4] - force.y, x5 = ext_[5] - force.z;
        ext[6 * pl + 0] = x0; ext[6 * pl + 1] = x1; ext[6 * pl + 2] = x2; ext[6 * pl + 3] = x3; ext[6 * pl + 4] = x4; ext[6 * pl + 5] = x5;
        s->cfrc_ang[3 * pl + 0] = x0; s->cfrc_ang[3 * pl + 1] = x1; s->cfrc_ang[3 * pl + 2] = x2;
        s->cfrc_vel[3 * pl + 0] = x3; s->cfrc_vel[3 * pl + 1] = x4; s->cfrc_vel[3 * pl + 2] = x5;
      }
    }
  }
  team_sync();
  PH(20)
  tk_dynamics<T>(m, e, s, tl, env_valid);
}

struct RewCtx { float link_vel_xy[8], foot_z[4], foot_xy[8]; float vel_world[3]; int was_reset; };
// Everything the reward terms read, held in registers: the loads are issued together ahead of the (serial) term loop, so the loop itself
// never waits on memory.  Terms that mutate env buffers (feet_air_time, forward_progress) update the copy; the caller writes it back.
struct RewState {
  float cmd[3], blv[3], bav[3], pg[3], base_pos[3];
  float dof_pos[NM], dof_vel[NM], last_dof_vel[NM], target[NM], ctrl_force[NM];
  float actions[NA], last_actions[NA];
  float fat[4]; int fc[4];
  float last_x;
  float ctrl_pos[NM], ctrl_vel[NM], s_vel[NM], s_dof_pos[NM]; int ctrl_mode[NM];   // base env only (engine PD, accessor.py:848-875)
};
// the part of RewState that is plain env-buffer content (the caller fills blv, bav, pg, base_pos, dof_pos, dof_vel, fc)
DEV void load_rew_state(const DCfg& c, const E& e, RewState& rs) {
  auto cmd = e.commands(); auto ldv = e.last_dof_vel(); auto t = e.target_dof_pos(); auto cf = e.ctrl_force(); auto a = e.actions(); auto la = e.last_actions();
  auto fat = e.feet_air_time();
#pragma unroll
  for (int i = 0; i < 3; ++i) rs.cmd[i] = cmd[i];
#pragma unroll
  for (int i = 0; i < NM; ++i) { rs.last_dof_vel[i] = ldv[i]; rs.target[i] = t[i]; rs.ctrl_force[i] = cf[c.i[GO2SIM_IC_MOTOR_DOF0 + i]]; }
#pragma unroll
  for (int i = 0; i < NA; ++i) { rs.actions[i] = a[i]; rs.last_actions[i] = la[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) rs.fat[i] = fat[i];
  rs.last_x = e.last_base_pos_x()[0];
  if (c.i[GO2SIM_IC_ENV_KIND] == 1) {
    auto ctrl_mode = e.ctrl_mode(); auto cp = e.ctrl_pos(); auto cv = e.ctrl_vel(); auto vel = e.vel(); auto sdp = e.dof_pos();
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
      rs.ctrl_mode[i] = ctrl_mode[d]; rs.ctrl_pos[i] = cp[d]; rs.ctrl_vel[i] = cv[d]; rs.s_vel[i] = vel[d]; rs.s_dof_pos[i] = sdp[d];
    }
  }
}
// reward terms, go2_env_walk.py:1251-1366
// the command gates shared by several terms (evaluated once per env and step, not once per term)
struct RewGates { float still, moving; };
DEV RewGates reward_gates(const RewState& rs) {
  const float c0 = rs.cmd[0], c1 = rs.cmd[1], c2 = rs.cmd[2];
  const float cmd_norm = dm_sqrt(c0 * c0 + c1 * c1 + c2 * c2);
  RewGates g;
  g.still = (cmd_norm < 0.1f) ? 1.0f : 0.0f;
  g.moving = (dm_sqrt(c0 * c0 + c1 * c1) > 0.1f) ? 1.0f : 0.0f;
  return g;
}
DEV float reward_term(const Model& m, const DCfg& c, RewState& rs, int id, const RewCtx& rc, const RewGates& gt) {
  const float dt = c.f[GO2SIM_FC_DT];
  const float* cmd = rs.cmd; const float* blv = rs.blv; const float* bav = rs.bav; const float* dof_pos = rs.dof_pos; const float* dof_vel = rs.dof_vel;
  float c0 = cmd[0], c1 = cmd[1], c2 = cmd[2];
  const float still = gt.still, moving = gt.moving;
  switch (id) {
    case GO2SIM_R_TRACKING_LIN_VEL: { float d0 = c0 - blv[0], d1 = c1 - blv[1]; return dm_exp(-(d0 * d0 + d1 * d1) / c.f[GO2SIM_FC_TRACKING_SIGMA]); }
    case GO2SIM_R_TRACKING_ANG_VEL: { float d = c2 - bav[2]; return dm_exp(-(d * d) / c.f[GO2SIM_FC_TRACKING_SIGMA]); }
    case GO2SIM_R_LIN_VEL_Z: {                                       // go2_env_stair.py:1615-1626 (deadzone 0 = walk env)
      float v = blv[2], dz = c.f[GO2SIM_FC_LIN_VEL_Z_DEADZONE];
      if (dz > 0.0f) { float ex = fmx(dm_abs(v) - dz, 0.0f); return ex * ex; }
      return v * v;
    }
    case GO2SIM_R_ACTION_RATE: {
      float s = 0.0f; const int na = c.i[GO2SIM_IC_NUM_ACTIONS];
#pragma unroll
      for (int i = 0; i < NA; ++i) if (i < na) { float d = rs.last_actions[i] - rs.actions[i]; s = s + d * d; }
      return s;
    }
    case GO2SIM_R_SIMILAR_TO_DEFAULT: {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) s = s + dm_abs(dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]);
      return s;
    }
    case GO2SIM_R_BASE_HEIGHT: {                                     // go2_env_stair.py:1634-1648: height above the local terrain
      float hgt = rs.base_pos[2];
      if (c.i[GO2SIM_IC_USE_TERRAIN]) hgt = rs.base_pos[2] - terrain_height(m, c, rs.base_pos[0], rs.base_pos[1]);
      float d = hgt - c.f[GO2SIM_FC_BASE_HEIGHT_TARGET]; return d * d;
    }
    case GO2SIM_R_DOF_ACC: {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) { float a = (dof_vel[i] - rs.last_dof_vel[i]) / dt; s = s + a * a; }
      return s;
    }
    case GO2SIM_R_DOF_VEL: {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) { float v = dof_vel[i]; s = s + v * v; }
      return s;
    }
    case GO2SIM_R_ORIENTATION_PENALTY: { float a = rs.pg[0], b2 = rs.pg[1]; return a * a + b2 * b2; }
    case GO2SIM_R_ANG_VEL_XY: { float a = bav[0], b2 = bav[1]; return a * a + b2 * b2; }
    case GO2SIM_R_STAND_STILL: {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) s = s + dm_abs(dof_pos[i] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i]);
      return s * still;
    }
    case GO2SIM_R_STAND_STILL_VEL: { float a = blv[0], b2 = blv[1], w = bav[2]; float lin = a * a + b2 * b2; float ang = w * w; return (lin + 0.5f * ang) * still; }
    case GO2SIM_R_FEET_STANCE: {
      float sa = 0.0f, sn = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sa = sa + rs.fat[i]; sn = sn + (rs.fc[i] ? 0.0f : 1.0f); }
      return (sa + sn) * still;
    }
    case GO2SIM_R_FEET_AIR_TIME: {                                     // mutates _feet_air_time (:1303-1314)
      float first[4], at[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { float a = rs.fat[i]; int ct = rs.fc[i]; first[i] = (a > 0.0f && ct) ? 1.0f : 0.0f; a = a + dt; a = a * (ct ? 0.0f : 1.0f); rs.fat[i] = a; at[i] = a; }
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s = s + (at[i] - c.f[GO2SIM_FC_FEET_AIR_TIME_TARGET]) * first[i];
      return s * moving;
    }
    case GO2SIM_R_FOOT_SLIP: {
      float slip = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { float vx = rc.link_vel_xy[2 * i], vy = rc.link_vel_xy[2 * i + 1]; slip = slip + (rs.fc[i] ? 1.0f : 0.0f) * (vx * vx + vy * vy); }
      return slip;
    }
    case GO2SIM_R_FOOT_CLEARANCE: {
      float pen = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float vx = rc.link_vel_xy[2 * i], vy = rc.link_vel_xy[2 * i + 1];
        float vn = dm_sqrt(vx * vx + vy * vy);
        float fz = rc.foot_z[i];
        if (c.i[GO2SIM_IC_USE_TERRAIN]) fz = rc.foot_z[i] - terrain_height(m, c, rc.foot_xy[2 * i], rc.foot_xy[2 * i + 1]);   // go2_env_stair.py:1742-1747
        float he = c.f[GO2SIM_FC_FEET_HEIGHT_TARGET] - fz; he = he * he;
        pen = pen + (rs.fc[i] ? 0.0f : 1.0f) * he * vn;
      }
      return pen * moving;
    }
    case GO2SIM_R_JOINT_TRACKING: {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) { float d = rs.target[i] - dof_pos[i]; s = s + d * d; }
      return s;
    }
    case GO2SIM_R_ENERGY: case GO2SIM_R_TORQUE_LOAD: {                 // get_dofs_control_force, abd/accessor.py:848-875
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
        float tau = clampf(rs.ctrl_force[i], m.dofs[d].force_range[0], m.dofs[d].force_range[1]);
        s = s + ((id == GO2SIM_R_ENERGY) ? dm_abs(tau * dof_vel[i]) : dm_abs(tau));
      }
      return s;
    }
    // ---- go2_env_base.py:246-390 (crouch / jump) ----
    case GO2SIM_R_JUMP_IMPULSE: { float gate = (rs.base_pos[2] < 0.50f) ? 1.0f : 0.0f; return gate * fmx(blv[2], 0.0f); }
    case GO2SIM_R_JUMP_APEX: { float q = (rs.base_pos[2] - c.f[GO2SIM_FC_JUMP_APEX_HEIGHT]) / c.f[GO2SIM_FC_JUMP_APEX_SIGMA]; return dm_exp(-(q * q)); }
    case GO2SIM_R_XY_STABILITY: { float vx = rc.vel_world[0], vy = rc.vel_world[1]; return -(vx * vx + vy * vy); }
    case GO2SIM_R_ORIENTATION: return -rs.pg[2];
    case GO2SIM_R_NO_SHAKE: { float a = bav[0], b2 = bav[1], c3 = bav[2]; return -((a * a + b2 * b2) + c3 * c3) / 1.0f; }
    case GO2SIM_R_CROUCH: return (rs.base_pos[2] < 0.25f) ? 1.0f : 0.0f;
    case GO2SIM_R_CROUCH_2: { float z = rs.base_pos[2]; return (z <= 0.30f && z >= 0.20f) ? 1.0f : 0.0f; }
    case GO2SIM_R_GROUND_PENALTY: { float v = (0.15f - rs.base_pos[2]) / 0.1f; v = fmn(fmx(v, 0.0f), 1.0f); return -(v * v); }
    case GO2SIM_R_CROUCH_TARGET: { float q = (rs.base_pos[2] - 0.15f) / 0.03f; return dm_exp(-(q * q)); }
    case GO2SIM_R_NO_FALL: { float dn = fmx(-blv[2] - 0.5f, 0.0f); return -(dn * dn); }
    case GO2SIM_R_Y_STABILITY: { float vy = rc.vel_world[1]; return -(vy * vy); }
    case GO2SIM_R_TORQUE_LOAD_BASE: {                                  // get_dofs_control_force of the current state, accessor.py:848-875
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
        const Dof& D = m.dofs[d];
        // a freshly reset env has (qpos0 + default) - qpos0 and zero velocity once its FK refresh has run (it follows this kernel)
        float pos_d = rc.was_reset ? ((m.qpos0[d + 1] + dof_pos[i]) - m.qpos0[d + 1]) : rs.s_dof_pos[i];
        float vel_d = rc.was_reset ? 0.0f : rs.s_vel[i];
        float force = 0.0f;
        int cm = rs.ctrl_mode[i];
        if (cm == CTRL_FORCE) force = rs.ctrl_force[i];
        else if (cm == CTRL_VELOCITY) force = D.kv * (rs.ctrl_vel[i] - vel_d);
        else if (cm == CTRL_POSITION) force = D.kp * (rs.ctrl_pos[i] - pos_d) + D.kv * (rs.ctrl_vel[i] - vel_d);
        s = s + dm_abs(clampf(force, D.force_range[0], D.force_range[1]));
      }
      return -0.001f * s;
    }
    case GO2SIM_R_CROUCH_PROGRESS: return fmx(0.35f - rs.base_pos[2], 0.0f);
    case GO2SIM_R_CROUCH_SPEED: return -(blv[2] * blv[2]);
    // ---- go2_env_stair.py:1659-1771 ----
    case GO2SIM_R_ORIENTATION_ROLL_ONLY: { float gy = rs.pg[1]; return gy * gy; }
    case GO2SIM_R_FORWARD_PROGRESS: { float bx = rs.base_pos[0]; float dx = bx - rs.last_x; rs.last_x = bx; return dx; }   // mutates _last_base_pos_x
  }
  return 0.0f;
}

// Go2Env.step post-physics part A: clear_external_force, state read-back, commands, termination, rewards
// (simulator.py:283-284, go2_env_walk.py:1026-1077) + reset-call statistics (:688-715,1228-1235).
// One lane per env; the kernel is a single wave per 64 envs, so its duration is its dependency chain: every input is loaded before the
// first store (one memory round trip instead of one per reward term) and the per-term episode sums sit in LDS for the dynamic term loop.
DEV void env_globals_body(const DCfg& c, Glob& g, Acc* acc, uint64_t seed, int count_push);
// Lane per env; the workgroup is POST_A_WAVES wavefronts over the SAME 64 envs.  Every wave loads the state and forms the shared quantities (base frame
// velocities, Euler angles, foot contacts, command gates) redundantly; the reward TERMS -- the long serial part: 19 independent functions of that state --
// are dealt to the waves (term k runs on one wave, its value goes to LDS), and wave 0 then adds them in the reference's order and makes every store.
// The three terms that touch env state beyond their own value (feet_air_time writes and feet_stance reads _feet_air_time, go2_env_walk.py:1303-1314;
// forward_progress moves _last_base_pos_x) stay together on wave 0, in order.  Same operations per value and the same sum order: results unchanged.
#ifndef GO2SIM_POST_A_WAVES
#define GO2SIM_POST_A_WAVES 4
#endif
constexpr int POST_A_WAVES = GO2SIM_POST_A_WAVES;   // (<= 8: the workgroup has to fit one CU at two wavefronts per SIMD)
__global__ __launch_bounds__(WG * POST_A_WAVES) void k_env_post_a(Pool P, const Model* __restrict__ mp, const DCfg cv, Glob* gp,
                                                   Acc* acc, uint64_t seed, uint32_t step_count) {
  STAMP(STK_POST_A)
  __shared__ float s_es[NREW][WG], s_r[NREW][WG];
  const int ln = threadIdx.x % WG, wv = threadIdx.x / WG;
  const bool w0 = wv == 0;                                          // the wave that owns the env's stores
  int b = blockIdx.x * WG + ln;
  if (b >= P.B) return;
  const Model& m = *mp; const DCfg& c = cv;  const Glob& g = *gp;
  E e(P, b);
  const int nrew = c.i[GO2SIM_IC_N_REWARDS];
  const bool base_env = c.i[GO2SIM_IC_ENV_KIND] == 1;               // base env: rewards follow the reset (k_env_post_b_team)
  PH_BEGIN
  // ---- loads ----
  int ep_len = e.episode_length()[0] + 1;
  int bl = c.i[GO2SIM_IC_BASE_LINK];
  V3 bp = e.l_pos()[bl]; Q4 bq = e.l_quat()[bl];
  V3 rcom = e.root_com()[bl];
  V3 cda = e.cd_ang()[bl], cdv = e.cd_vel()[bl];
  V3 f_cf[4], f_lp[4], f_cdv[4], f_cda[4]; int fc_old[4];
  {
    auto fc = e.foot_contact();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int l = c.i[GO2SIM_IC_FOOT_LINK0 + i];
      fc_old[i] = fc[i];
      f_cf[i] = e.contact_force()[l]; f_lp[i] = e.l_pos()[l]; f_cdv[i] = e.cd_vel()[l]; f_cda[i] = e.cd_ang()[l];
    }
  }
  RewState rs;
  {
    auto sdof_pos = e.dof_pos(); auto vel = e.vel();
#pragma unroll
    for (int i = 0; i < NM; ++i) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i]; rs.dof_pos[i] = sdof_pos[d]; rs.dof_vel[i] = vel[d]; }
  }
  if (!base_env) load_rew_state(c, e, rs);
  else { auto cmd = e.commands(); for (int i = 0; i < 3; ++i) rs.cmd[i] = cmd[i]; }
  {
    auto episode_sums = e.episode_sums();
    float es_[NREW];
#pragma unroll
    for (int k = 0; k < NREW; ++k) es_[k] = episode_sums[k];           // all NREW rows exist; unconditional loads stay in flight together
    if (w0) {
#pragma unroll
      for (int k = 0; k < NREW; ++k) s_es[k][ln] = es_[k];
    }
  }
  PH(12)
  // ---- stores start here (wave 0 only) ----
  if (w0) {
    { auto ext = e.ext(); for (int i = 0; i < NL * 6; ++i) ext[i] = 0.0f; }
    e.episode_length()[0] = ep_len;
    auto base_pos = e.base_pos(); auto base_quat = e.base_quat();
    base_pos[0] = bp.x; base_pos[1] = bp.y; base_pos[2] = bp.z;
    base_quat[0] = bq.w; base_quat[1] = bq.x; base_quat[2] = bq.y; base_quat[3] = bq.z;
  }
  Q4 inv_init = inv_quat(q4(c.f[GO2SIM_FC_BASE_INIT_QUAT0], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 1], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 2], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 3]));
  V3 eul = tc_quat_to_xyz_rpy_deg(tc_quat_mul(bq, inv_init), m.eps);
  Q4 inv_bq = inv_quat(bq);
  V3 velw = cdv + cross(cda, bp - rcom);
  V3 blv = tc_transform_by_quat(velw, inv_bq), bav = tc_transform_by_quat(cda, inv_bq);
  V3 pg = tc_transform_by_quat(v3(0.0f, 0.0f, -1.0f), inv_bq);
  if (w0) {
    auto base_euler = e.base_euler();
    base_euler[0] = eul.x; base_euler[1] = eul.y; base_euler[2] = eul.z;
    { auto bvw = e.base_vel_world(); bvw[0] = velw.x; bvw[1] = velw.y; bvw[2] = velw.z; }
    auto o_blv = e.base_lin_vel(); auto o_bav = e.base_ang_vel(); auto o_pg = e.projected_gravity();
    o_blv[0] = blv.x; o_blv[1] = blv.y; o_blv[2] = blv.z;
    o_bav[0] = bav.x; o_bav[1] = bav.y; o_bav[2] = bav.z;
    o_pg[0] = pg.x; o_pg[1] = pg.y; o_pg[2] = pg.z;
  }
  rs.blv[0] = blv.x; rs.blv[1] = blv.y; rs.blv[2] = blv.z; rs.bav[0] = bav.x; rs.bav[1] = bav.y; rs.bav[2] = bav.z;
  rs.pg[0] = pg.x; rs.pg[1] = pg.y; rs.pg[2] = pg.z; rs.base_pos[0] = bp.x; rs.base_pos[1] = bp.y; rs.base_pos[2] = bp.z;
  if (w0) {
    auto dof_pos = e.e_dof_pos(); auto dof_vel = e.e_dof_vel();
#pragma unroll
    for (int i = 0; i < NM; ++i) { dof_pos[i] = rs.dof_pos[i]; dof_vel[i] = rs.dof_vel[i]; }
  }
  RewCtx rc; rc.was_reset = 0; rc.vel_world[0] = velw.x; rc.vel_world[1] = velw.y; rc.vel_world[2] = velw.z;
  auto fc = e.foot_contact(); auto lfc = e.last_foot_contact();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ct = dm_abs(f_cf[i].z) > c.f[GO2SIM_FC_FOOT_CONTACT_THRESHOLD];
    if (w0) { lfc[i] = fc_old[i]; fc[i] = ct; }
    rs.fc[i] = ct;
    V3 lp = f_lp[i];
    V3 lv = f_cdv[i] + cross(f_cda[i], lp - rcom);
    rc.link_vel_xy[2 * i] = lv.x; rc.link_vel_xy[2 * i + 1] = lv.y; rc.foot_z[i] = lp.z; rc.foot_xy[2 * i] = lp.x; rc.foot_xy[2 * i + 1] = lp.y;
  }
  if (ep_len % c.i[GO2SIM_IC_RESAMPLE_STEPS] == 0) {
    auto cmd = e.commands();
    dm_u4 r = rng4(seed, RNG_CMD, b, step_count, 0);
    float cx, cy, cz;
    sample_commands(c, g, r, b, cx, cy, cz);
    if (w0) { cmd[0] = cx; cmd[1] = cy; cmd[2] = cz; }
    rs.cmd[0] = cx; rs.cmd[1] = cy; rs.cmd[2] = cz;
  }
  int maxlen = c.i[GO2SIM_IC_MAX_EPISODE_LENGTH];
  int rst = ep_len > maxlen;
  rst |= dm_abs(eul.y) > c.f[GO2SIM_FC_TERM_PITCH_DEG];
  rst |= dm_abs(eul.x) > c.f[GO2SIM_FC_TERM_ROLL_DEG];
  rst |= dm_abs(blv.z) > c.f[GO2SIM_FC_TERM_ZVEL];
  rst |= dm_abs(blv.y) > c.f[GO2SIM_FC_TERM_YVEL];
  float time_out = (ep_len > maxlen) ? 1.0f : 0.0f;
  if (w0) { e.reset_buf()[0] = rst; e.time_out()[0] = time_out; }
  float rew = 0.0f;
  float tracking_int = 0.0f;
  PH(13)
  const float fat0[4] = {rs.fat[0], rs.fat[1], rs.fat[2], rs.fat[3]};
  const float last_x0 = rs.last_x;
  const RewGates gates = reward_gates(rs);
  if (!base_env) {                                                   // the terms, dealt to the waves: stateful ones on wave 0, the others round-robin
    int id_next = c.i[GO2SIM_IC_REWARD_ID0]; float scale_next = c.f[GO2SIM_FC_REWARD_SCALE0];
    int j = 0;
    for (int k = 0; k < nrew; ++k) {
      const int id = id_next; const float scale = scale_next;
      { const int kn = (k + 1 < NREW) ? k + 1 : k; id_next = c.i[GO2SIM_IC_REWARD_ID0 + kn]; scale_next = c.f[GO2SIM_FC_REWARD_SCALE0 + kn]; }   // the table reads of the next term overlap this one
      const bool stateful = id == GO2SIM_R_FEET_AIR_TIME || id == GO2SIM_R_FEET_STANCE || id == GO2SIM_R_FORWARD_PROGRESS;
      const int owner = stateful ? 0 : (POST_A_WAVES > 4 ? 1 + j % (POST_A_WAVES - 1) : (1 + j) % POST_A_WAVES);
      j += stateful ? 0 : 1;
      if (wv == owner) s_r[k][ln] = reward_term(m, c, rs, id, rc, gates) * scale;
    }
  }
  __syncthreads();
  PH(14)
  if (!w0) return;                                                   // the rest is wave 0's: the sum in the reference's order, the statistics, the stores
  for (int k = 0; k < nrew; ++k) {
    const int id = c.i[GO2SIM_IC_REWARD_ID0 + k];
    float es = s_es[k][ln];
    if (!base_env) {
      const float r = s_r[k][ln];
      rew = rew + r;
      es = es + r;
      s_es[k][ln] = es;
    }
    if (id == GO2SIM_R_TRACKING_LIN_VEL || id == GO2SIM_R_TRACKING_ANG_VEL) tracking_int = tracking_int + es;
  }
  if (rst) {
    float ep_steps = fmx((float)ep_len, 1.0f);
    float ep_seconds = ep_steps * c.f[GO2SIM_FC_DT];
    for (int k = 0; k < nrew; ++k) atomicAdd(&acc->ep[k], base_env ? (double)s_es[k][ln] : (double)(s_es[k][ln] / ep_seconds));
    atomicAdd(&acc->tracking, (double)(tracking_int / ep_seconds));
    atomicAdd(&acc->timeouts, (double)time_out);
    atomicAdd(&acc->n_reset_now, 1);
  }
  // The single-thread part of the step (curriculum state machine, "global" DR draws: the quantities the reference keeps in Python scalars)
  // runs in whichever workgroup finishes last, instead of in a kernel of its own.
  PH(15)
  __threadfence();
  __builtin_amdgcn_wave_barrier();                                   // (only wave 0 is left: its lanes are the workgroup's 64 envs)
  PH(16)
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(&acc->done, 1);
    if (ticket == (int)gridDim.x - 1) {
      __threadfence();
      acc->done = 0;
      env_globals_body(c, *gp, acc, seed, 1);
    }
  }
  PH(17)
  // the per-term outputs go out last: nothing in this kernel reads them back, and the fence above then only waits for the statistics
  if (!base_env) {
    e.rew()[0] = rew;
    auto rew_terms = e.rew_terms(); auto episode_sums = e.episode_sums();
    for (int k = 0; k < nrew; ++k) { rew_terms[k] = s_r[k][ln]; episode_sums[k] = s_es[k][ln]; }
    auto fat = e.feet_air_time();
#pragma unroll
    for (int i = 0; i < 4; ++i) if (rs.fat[i] != fat0[i]) fat[i] = rs.fat[i];
    if (rs.last_x != last_x0) e.last_base_pos_x()[0] = rs.last_x;
  }
}

// Go2Env.reset: mark every env for reset + statistics (go2_env_walk.py:1242-1245)
__global__ __launch_bounds__(WG) void k_env_mark_all(Pool P, const DCfg* __restrict__ cp, Acc* acc) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  const DCfg& c = *cp;
  E e(P, b);
  e.reset_buf()[0] = 1;
  int ep_len = e.episode_length()[0];
  float ep_steps = fmx((float)ep_len, 1.0f);
  float ep_seconds = ep_steps * c.f[GO2SIM_FC_DT];
  auto episode_sums = e.episode_sums();
  int nrew = c.i[GO2SIM_IC_N_REWARDS];
  float tracking_int = 0.0f;
  for (int k = 0; k < nrew; ++k) {
    int id = c.i[GO2SIM_IC_REWARD_ID0 + k];
    float es = episode_sums[k];
    if (id == GO2SIM_R_TRACKING_LIN_VEL || id == GO2SIM_R_TRACKING_ANG_VEL) tracking_int = tracking_int + es;
    atomicAdd(&acc->ep[k], (c.i[GO2SIM_IC_ENV_KIND] == 1) ? (double)es : (double)(es / ep_seconds));   // go2_env_base.py:232-236 logs mean(sum) / episode_length_s
  }
  atomicAdd(&acc->tracking, (double)(tracking_int / ep_seconds));
  atomicAdd(&acc->timeouts, (double)e.time_out()[0]);
  atomicAdd(&acc->n_reset_now, 1);
}

// Go2Env.reset_idx(envs_idx) on a subset (go2_env_walk.py:1156-1240): the listed envs are flagged (all others unflagged) and their episode
// statistics accumulated, exactly as k_env_mark_all does for the whole batch.  Two launches: clear, then mark (an index may repeat).
__global__ __launch_bounds__(WG) void k_env_unmark_all(Pool P) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  P.i[(size_t)IO(reset_buf) * P.B + b] = 0;
}
__global__ __launch_bounds__(WG) void k_env_mark_idx(Pool P, const DCfg* __restrict__ cp, Acc* acc, const int* __restrict__ envs_idx, int n_sel) {
  int t = blockIdx.x * WG + threadIdx.x;
  if (t >= n_sel) return;
  int b = envs_idx[t];
  if (b < 0 || b >= P.B) return;
  const DCfg& c = *cp;
  E e(P, b);
  if (atomicExch(&e.reset_buf()[0], 1) != 0) return;                    // listed twice: counted once
  int ep_len = e.episode_length()[0];
  float ep_steps = fmx((float)ep_len, 1.0f);
  float ep_seconds = ep_steps * c.f[GO2SIM_FC_DT];
  auto episode_sums = e.episode_sums();
  int nrew = c.i[GO2SIM_IC_N_REWARDS];
  float tracking_int = 0.0f;
  for (int k = 0; k < nrew; ++k) {
    int id = c.i[GO2SIM_IC_REWARD_ID0 + k];
    float es = episode_sums[k];
    if (id == GO2SIM_R_TRACKING_LIN_VEL || id == GO2SIM_R_TRACKING_ANG_VEL) tracking_int = tracking_int + es;
    atomicAdd(&acc->ep[k], (c.i[GO2SIM_IC_ENV_KIND] == 1) ? (double)es : (double)(es / ep_seconds));   // go2_env_base.py:232-236 logs mean(sum) / episode_length_s
  }
  atomicAdd(&acc->tracking, (double)(tracking_int / ep_seconds));
  atomicAdd(&acc->timeouts, (double)e.time_out()[0]);
  atomicAdd(&acc->n_reset_now, 1);
}
// eval-side teleport of single envs (respawn_at_start, go2_eval_stairs.py:314-361; respawn_on_tile, go2_eval_walk.py:399-480):
// set_dofs_position(default, zero_velocity) + set_pos + set_quat + zero_all_dofs_velocity on the listed envs, optionally clearing the action /
// velocity buffers the way respawn_at_start does.  No curriculum bookkeeping, no randomisation, episode counters untouched.
__global__ __launch_bounds__(WG) void k_env_respawn(Pool P, const Model* __restrict__ mp, const DCfg* __restrict__ cp, const int* __restrict__ envs_idx, int n_sel,
                                                    const float* __restrict__ pos, const float* __restrict__ quat, int clear_buffers) {
  int t = blockIdx.x * WG + threadIdx.x;
  if (t >= n_sel) return;
  int b = envs_idx[t];
  if (b < 0 || b >= P.B) return;
  const Model& m = *mp; const DCfg& c = *cp;
  E e(P, b);
  auto dof_pos = e.e_dof_pos(); auto dof_vel = e.e_dof_vel(); auto qpos = e.qpos(); auto vel = e.vel();
  for (int i = 0; i < NM; ++i) {
    float dp = c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i];
    dof_pos[i] = dp; dof_vel[i] = 0.0f;
    int q = c.i[GO2SIM_IC_MOTOR_DOF0 + i] + 1;
    qpos[q] = m.qpos0[q] + dp;
  }
  for (int d = 0; d < ND; ++d) vel[d] = 0.0f;
  e.err()[0] = 0; e.is_warmstart()[0] = 0;                              // set_dofs_position: rigid_solver.py:2403-2410
  { auto qacc_ws = e.qacc_ws(); for (int d = 0; d < ND; ++d) qacc_ws[d] = 0.0f; }
  { auto ncv = e.ncache_valid(); for (int p = 0; p < NCV; ++p) ncv[p] = 0; }   // normal cache := zeros
  float bq[4];
  for (int k = 0; k < 4; ++k) bq[k] = quat ? quat[4 * t + k] : c.f[GO2SIM_FC_BASE_INIT_QUAT0 + k];
  auto base_pos = e.base_pos(); auto base_quat = e.base_quat();
  for (int k = 0; k < 3; ++k) { base_pos[k] = pos[3 * t + k]; qpos[k] = pos[3 * t + k]; }
  for (int k = 0; k < 4; ++k) { base_quat[k] = bq[k]; qpos[3 + k] = bq[k]; }
  if (clear_buffers) {
    auto blv = e.base_lin_vel(); auto bav = e.base_ang_vel();
    for (int k = 0; k < 3; ++k) { blv[k] = 0.0f; bav[k] = 0.0f; }
    auto la = e.last_actions(); auto aa = e.applied_actions(); auto hist = e.action_history();
    for (int i = 0; i < NA; ++i) { la[i] = 0.0f; aa[i] = 0.0f; for (int k = 0; k < GO2SIM_ACTION_RING_MAX; ++k) hist[k][i] = 0.0f; }
    { auto ldv = e.last_dof_vel(); for (int i = 0; i < NM; ++i) ldv[i] = 0.0f; }
    e.last_base_pos_x()[0] = pos[3 * t];
  }
}
__global__ __launch_bounds__(WG) void k_env_set_terrain_rows(Pool P, const int* __restrict__ rows, int n_rows) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  int r = rows[b];
  P.i[(size_t)IO(terrain_row) * P.B + b] = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
}

// the update step of _maybe_update_curriculum_on_reset (go2_env_walk.py:717-729) on the accumulated counters
__host__ __device__ inline void globals_curriculum_check(const DCfg& c, Glob& g) {
  if (g.curr_ep_total < c.i[GO2SIM_IC_CURR_UPDATE_EVERY]) return;
  double timeout_rate = g.curr_timeout_total / (double)(g.curr_ep_total < 1 ? 1 : g.curr_ep_total);
  double fall_rate = 1.0 - timeout_rate;
  double tracking_avg = g.curr_tracking_sum / (double)(g.curr_tracking_n < 1 ? 1 : g.curr_tracking_n);
  if (curriculum_update(c, g, timeout_rate, tracking_avg, fall_rate)) apply_curriculum_level(c, g);
  g.curr_ep_total = 0; g.curr_timeout_total = 0.0; g.curr_tracking_sum = 0.0; g.curr_tracking_n = 0;
}
// t_sample (CurriculumManager.sample_level :85-93) and the "global" DR draws (:737-756, 803-848) of one reset call; `n_throttle` resets are counted
// for the friction throttle, `key` numbers the call in the Philox stream
__host__ __device__ inline void globals_draws(const DCfg& c, Glob& g, uint64_t seed, int n_throttle, uint32_t key) {
  dm_u4 r0 = rng4(seed, RNG_GLOBAL_DR, 0xffffffffu, key, 0);
  dm_u4 r1 = rng4(seed, RNG_GLOBAL_DR, 0xffffffffu, key, 1);
  dm_u4 r2 = rng4(seed, RNG_GLOBAL_DR, 0xffffffffu, key, 2);
  double t;
  if (c.i[GO2SIM_IC_DR_SCHEDULE]) t = dr_level(c, c.i[GO2SIM_IC_CURR_ENABLED] ? g.level : 1.0);   // go2_env_stair.py:1506-1507
  else if (!c.i[GO2SIM_IC_CURR_ENABLED]) t = 1.0;
  else if (dm_u01(r0.v[0]) < c.f[GO2SIM_FC_CURR_MIX_PROB_CURRENT]) t = clamp01d(g.level);
  else {
    double hi = c.d[GO2SIM_FC_CURR_MIX_LEVEL_HIGH] < g.level ? c.d[GO2SIM_FC_CURR_MIX_LEVEL_HIGH] : g.level;          // std::min(level, high)
    double lo = hi < c.d[GO2SIM_FC_CURR_MIX_LEVEL_LOW] ? hi : c.d[GO2SIM_FC_CURR_MIX_LEVEL_LOW];                   // std::min(low, hi)
    t = clamp01d(lo + (hi - lo) * (double)dm_u01(r0.v[1]));
  }
  g.t_sample = t;
  double ts = g.t_sample;
  if (c.i[GO2SIM_IC_HAS_FRICTION_DR]) {
    g.global_dr_reset_counter += n_throttle;
    if (g.global_dr_reset_counter >= c.i[GO2SIM_IC_GLOBAL_DR_INTERVAL]) {
      g.global_dr_reset_counter = 0;
      g.friction = rand_float(lerp_lo(c, GO2SIM_FC_FRICTION_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_FRICTION_EASY_LO, ts), r0.v[2]);
    }
  }
  if (c.i[GO2SIM_IC_HAS_MASS_DR]) g.mass_shift = rand_float(lerp_lo(c, GO2SIM_FC_MASS_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MASS_EASY_LO, ts), r0.v[3]);
  if (c.i[GO2SIM_IC_HAS_COM_DR])
    for (int k = 0; k < 3; ++k) g.com_shift[k] = rand_float(lerp_lo(c, GO2SIM_FC_COM_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_COM_EASY_LO, ts), r1.v[k]);
  if (c.i[GO2SIM_IC_HAS_LEGM_DR])
    for (int k = 0; k < 4; ++k) g.leg_mass_shift[k] = rand_float(lerp_lo(c, GO2SIM_FC_LEGM_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_LEGM_EASY_LO, ts), r2.v[k]);
}

// single-instance part of reset_idx: curriculum, t_sample, "global" DR (go2_env_walk.py:688-756,803-848,1160-1171)
// go2sim_env_sync_apply on one Glob (host copy or device): the summed shard counters enter the curriculum state machine, then the draws of one reset
// call.  The first apply of a handle draws even without a counted reset (and draws the friction regardless of the throttle): a sync issued between
// configure and the constructor's reset puts the shard where the single-process env starts (include/go2sim.h); the caller passes the env count of
// the batch as the throttle increment of that sync, as the constructor's reset_idx would count it.
__host__ __device__ inline void sync_apply_body(const DCfg& c, Glob& g, uint64_t seed, const double* s5, double* dr_out10) {
  const int n = (int)s5[0];
  const bool first = g.sync_calls == 0;
  if (n > 0 || first) {
    if (n > 0 && c.i[GO2SIM_IC_CURR_ENABLED] && !c.i[GO2SIM_IC_FREEZE_CURRICULUM]) {
      g.curr_ep_total += n; g.curr_timeout_total += s5[1]; g.curr_tracking_sum += s5[2]; g.curr_tracking_n += (int)s5[3];
      globals_curriculum_check(c, g);
    }
    globals_draws(c, g, seed, (int)s5[4], (uint32_t)g.sync_calls);
    g.sync_calls += 1;
  }
  dr_out10[0] = g.friction; dr_out10[1] = g.mass_shift;
  for (int k = 0; k < 3; ++k) dr_out10[2 + k] = g.com_shift[k];
  for (int k = 0; k < 4; ++k) dr_out10[5 + k] = g.leg_mass_shift[k];
  dr_out10[9] = g.t_sample;
}
__host__ __device__ inline void set_global_dr_body(Glob& g, const double* dr10) {
  g.friction = (float)dr10[0]; g.mass_shift = (float)dr10[1];
  for (int k = 0; k < 3; ++k) g.com_shift[k] = (float)dr10[2 + k];
  for (int k = 0; k < 4; ++k) g.leg_mass_shift[k] = (float)dr10[5 + k];
  g.t_sample = dr10[9];
}
__global__ void k_env_sync_counters(Glob* gp, double* out5) { for (int k = 0; k < 5; ++k) { out5[k] = gp->shard_counters[k]; gp->shard_counters[k] = 0.0; } }
__global__ void k_env_sync_apply(const DCfg* __restrict__ cp, Glob* gp, uint64_t seed, const double* __restrict__ s5, double* dr_out10) { sync_apply_body(*cp, *gp, seed, s5, dr_out10); }
__global__ void k_env_set_global_dr(Glob* gp, const double* __restrict__ dr10) { set_global_dr_body(*gp, dr10); }

DEV void env_globals_body(const DCfg& c, Glob& g, Acc* acc, uint64_t seed, int count_push) {
  if (count_push && c.i[GO2SIM_IC_HAS_PUSH] && g.push_enable) g.push_counter += 1;
  int n = acc->n_reset_now;
  g.n_reset_now = n;
  if (n > 0) {
    // `float(tensor.sum().item())` of float32 tensors (:698, :710) added to python floats
    const double timeouts = (double)(float)acc->timeouts, tracking = (double)(float)acc->tracking;
    if (c.i[GO2SIM_IC_SHARED_GLOBALS]) {              // one shard of a larger batch: the increments are combined by the host (go2sim_env_sync_*)
      g.shard_counters[0] += n; g.shard_counters[1] += timeouts; g.shard_counters[2] += tracking; g.shard_counters[3] += n;
      if (!(g.sync_calls > 0 && g.reset_calls == 0)) g.shard_counters[4] += n;   // (the constructor's reset after an initial sync: that sync already counted these envs for the friction throttle)
    } else {
      if (c.i[GO2SIM_IC_CURR_ENABLED] && !c.i[GO2SIM_IC_FREEZE_CURRICULUM]) {
        g.curr_ep_total += n; g.curr_timeout_total += timeouts; g.curr_tracking_sum += tracking; g.curr_tracking_n += n;
        globals_curriculum_check(c, g);
      }
      globals_draws(c, g, seed, n, g.reset_calls);
    }
    g.last_reset_count = n;
    g.terrain_row_sum = 0;   // accumulated by k_env_terrain_rows
    for (int k = 0; k < NREW; ++k)
      g.last_episode_rew[k] = (c.i[GO2SIM_IC_ENV_KIND] == 1) ? (float)((double)(float)(acc->ep[k] / (double)n) / c.d[GO2SIM_FC_EPISODE_LENGTH_S]) : (float)(acc->ep[k] / (double)n);
    g.reset_calls += 1;
    // consumed: clear the accumulators for the next reset call (they are only ever non-zero when n > 0)
    acc->timeouts = 0.0; acc->tracking = 0.0; acc->n_reset_now = 0;
    for (int k = 0; k < NREW; ++k) acc->ep[k] = 0.0;
  }
}
// single-thread launch (Go2Env.reset path; the step path runs the body in the last workgroup of k_env_post_a)
__global__ void k_env_globals(const DCfg* __restrict__ cp, Glob* gp, Acc* acc, uint64_t seed, int count_push) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  env_globals_body(*cp, *gp, acc, seed, count_push);
}

// per-env part of reset_idx (go2_env_walk.py:1156-1240)
// The draws of a reset.  TEAM form (k_env_post_b_team): the thirteen Philox blocks of env_reset_one are independent, so lane k of the env's team forms block k
// (uniform code: purpose, index, range and destination come from the lane index), the ten per-env DR blocks are finished by their lanes, and the three
// blocks of the serial part (per-env global DR, pose, commands) travel to lane 0.  Same keys and the same expressions as the one-lane form below.
struct ResetDraws { dm_u4 per_env, pose, cmd; };
template <int T>
DEV ResetDraws env_reset_draws_team(const DCfg& c, const Glob& g, const E& e, int b, uint64_t seed, int tl, bool was_reset) {
  static_assert(T >= 16, "thirteen blocks side by side");
  const uint32_t rc = g.reset_calls - 1;
  const double ts = g.t_sample;
  const int k = tl < 13 ? tl : 12;
  const uint32_t purpose = k < 11 ? RNG_RESET_DR : (k == 11 ? RNG_RESET_POSE : RNG_RESET_CMD);
  const dm_u4 r = rng4(seed, purpose, b, rc, k < 11 ? k : 0);
  // the per-env DR blocks: kp factors 0..2, kd factors 3..5, gravity offset 6 (three values), motor strength 7..9
  const int has = k < 3 ? GO2SIM_IC_HAS_KPF_DR : (k < 6 ? GO2SIM_IC_HAS_KDF_DR : (k == 6 ? GO2SIM_IC_HAS_GOFF_DR : GO2SIM_IC_HAS_MSTR_DR));
  const int rng = k < 3 ? GO2SIM_FC_KPF_EASY_LO : (k < 6 ? GO2SIM_FC_KDF_EASY_LO : (k == 6 ? GO2SIM_FC_GOFF_EASY_LO : GO2SIM_FC_MSTR_EASY_LO));
  const int off = k < 3 ? FO(kp_factors) + 4 * k : (k < 6 ? FO(kd_factors) + 4 * (k - 3) : (k == 6 ? FO(gravity_offset) : FO(motor_strength) + 4 * (k - 7)));
  const int n = k == 6 ? 3 : 4;
  if (was_reset && tl < 10 && c.i[has]) {
    const double lo = lerp_lo(c, rng, ts), hi = lerp_hi(c, rng, ts);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < n) gstore(e, off, j, rand_float(lo, hi, r.v[j]));
  }
  ResetDraws d;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d.per_env.v[j] = (uint32_t)__shfl((int)r.v[j], 10, T); d.pose.v[j] = (uint32_t)__shfl((int)r.v[j], 11, T); d.cmd.v[j] = (uint32_t)__shfl((int)r.v[j], 12, T);
  }
  return d;
}
// per-env part of reset_idx (go2_env_walk.py:1156-1240).  `pre` (team form): the per-env DR blocks are already stored and the three serial blocks are handed in
DEV void env_reset_one(const Model& m, const DCfg& c, const Glob& g, const E& e, int b, uint64_t seed, const ResetDraws* pre = nullptr) {
  uint32_t rc = g.reset_calls - 1;
  double ts = g.t_sample;
  auto kp_factors = e.kp_factors(); auto kd_factors = e.kd_factors(); auto motor_strength = e.motor_strength(); auto gravity_offset = e.gravity_offset();
  if (!pre) {
  if (c.i[GO2SIM_IC_HAS_KPF_DR])
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(seed, RNG_RESET_DR, b, rc, blk); for (int k = 0; k < 4; ++k) kp_factors[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_KPF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_KPF_EASY_LO, ts), r.v[k]); }
  if (c.i[GO2SIM_IC_HAS_KDF_DR])
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(seed, RNG_RESET_DR, b, rc, 3 + blk); for (int k = 0; k < 4; ++k) kd_factors[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_KDF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_KDF_EASY_LO, ts), r.v[k]); }
  if (c.i[GO2SIM_IC_HAS_GOFF_DR]) {
    dm_u4 r = rng4(seed, RNG_RESET_DR, b, rc, 6);
    for (int k = 0; k < 3; ++k) gravity_offset[k] = rand_float(lerp_lo(c, GO2SIM_FC_GOFF_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_GOFF_EASY_LO, ts), r.v[k]);
  }
  if (c.i[GO2SIM_IC_HAS_MSTR_DR])
    for (int blk = 0; blk < 3; ++blk) { dm_u4 r = rng4(seed, RNG_RESET_DR, b, rc, 7 + blk); for (int k = 0; k < 4; ++k) motor_strength[4 * blk + k] = rand_float(lerp_lo(c, GO2SIM_FC_MSTR_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MSTR_EASY_LO, ts), r.v[k]); }
  }
  if (c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR]) {   // extension (BASELINE configs[4], not in the reference): the friction / base-mass scalars are drawn per env
    dm_u4 r = pre ? pre->per_env : rng4(seed, RNG_RESET_DR, b, rc, 10);
    if (c.i[GO2SIM_IC_HAS_FRICTION_DR]) {
      float mu = rand_float(lerp_lo(c, GO2SIM_FC_FRICTION_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_FRICTION_EASY_LO, ts), r.v[0]);
      auto gf = e.geom_friction();
      for (int i = 0; i < NG; ++i) gf[i] = mu;
    }
    if (c.i[GO2SIM_IC_HAS_MASS_DR])
      e.mass_shift()[c.i[GO2SIM_IC_BASE_LINK]] = rand_float(lerp_lo(c, GO2SIM_FC_MASS_EASY_LO, ts), lerp_hi(c, GO2SIM_FC_MASS_EASY_LO, ts), r.v[1]);
  }
  dm_u4 rp = pre ? pre->pose : rng4(seed, RNG_RESET_POSE, b, rc, 0);
  {
    int max_d = imx(c.i[GO2SIM_IC_MIN_DELAY], imn(g.delay_max_cur, c.i[GO2SIM_IC_MAX_DELAY]));
    e.delay_steps()[0] = rand_int(c.i[GO2SIM_IC_MIN_DELAY], max_d, rp.v[3]);
  }
  auto dof_pos = e.e_dof_pos(); auto dof_vel = e.e_dof_vel(); auto qpos = e.qpos(); auto vel = e.vel();
  for (int i = 0; i < NM; ++i) {
    float dp = c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i];
    dof_pos[i] = dp; dof_vel[i] = 0.0f;
    int d = c.i[GO2SIM_IC_MOTOR_DOF0 + i];
    int q = d + 1;
    qpos[q] = m.qpos0[q] + dp;
  }
  for (int d = 0; d < ND; ++d) vel[d] = 0.0f;
  e.err()[0] = 0; e.is_warmstart()[0] = 0;
  { auto qacc_ws = e.qacc_ws(); for (int d = 0; d < ND; ++d) qacc_ws[d] = 0.0f; }
  { auto ncv = e.ncache_valid(); for (int p = 0; p < NCV; ++p) ncv[p] = 0; }   // normal cache := zeros
  float bpx = c.f[GO2SIM_FC_BASE_INIT_POS0], bpy = c.f[GO2SIM_FC_BASE_INIT_POS0 + 1], bpz = c.f[GO2SIM_FC_BASE_INIT_POS0 + 2];
  float bq[4] = {c.f[GO2SIM_FC_BASE_INIT_QUAT0], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 1], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 2], c.f[GO2SIM_FC_BASE_INIT_QUAT0 + 3]};
  if (c.i[GO2SIM_IC_USE_TERRAIN]) {                                    // _get_terrain_spawn_pos, go2_env_stair.py:856-871, :1531-1540
    const float* rcn = &c.f[GO2SIM_FC_ROW_CENTER0 + 3 * e.terrain_row()[0]];
    float init_z = c.f[GO2SIM_FC_BASE_INIT_POS0 + 2];
    float spawn_z = rcn[2] + init_z;
    bpx = rcn[0]; bpy = rcn[1]; bpz = spawn_z;
    if (c.i[GO2SIM_IC_HAS_INIT_Z]) bpz = ((spawn_z + rand_float(c.d[GO2SIM_FC_INIT_Z_LO], c.d[GO2SIM_FC_INIT_Z_HI], rp.v[0])) - init_z) + init_z;
  } else
  if (c.i[GO2SIM_IC_HAS_INIT_Z]) bpz = rand_float(c.d[GO2SIM_FC_INIT_Z_LO], c.d[GO2SIM_FC_INIT_Z_HI], rp.v[0]);
  if (c.i[GO2SIM_IC_HAS_INIT_EULER]) {                                 // euler_to_quat_wxyz, go2_env_walk.py:16-25
    const double D2R = 3.141592653589793 / 180.0;                       // math.radians: x * (pi / 180) in float64
    double lo = c.d[GO2SIM_FC_INIT_EULER_LO_DEG] * D2R, hi = c.d[GO2SIM_FC_INIT_EULER_HI_DEG] * D2R;
    float roll = rand_float(lo, hi, rp.v[1]), pitch = rand_float(lo, hi, rp.v[2]), yaw = 0.0f;
    float sr, cr, sp, cpp, sy, cy;
    dm_sincos(roll / 2.0f, &sr, &cr); dm_sincos(pitch / 2.0f, &sp, &cpp); dm_sincos(yaw / 2.0f, &sy, &cy);
    bq[0] = cr * cpp * cy + sr * sp * sy; bq[1] = sr * cpp * cy - cr * sp * sy;
    bq[2] = cr * sp * cy + sr * cpp * sy; bq[3] = cr * cpp * sy - sr * sp * cy;
  }
  auto base_pos = e.base_pos(); auto base_quat = e.base_quat();
  base_pos[0] = bpx; base_pos[1] = bpy; base_pos[2] = bpz;
  qpos[0] = bpx; qpos[1] = bpy; qpos[2] = bpz;
  for (int k = 0; k < 4; ++k) { base_quat[k] = bq[k]; qpos[3 + k] = bq[k]; }
  auto blv = e.base_lin_vel(); auto bav = e.base_ang_vel();
  for (int k = 0; k < 3; ++k) { blv[k] = 0.0f; bav[k] = 0.0f; }
  auto la = e.last_actions(); auto aa = e.applied_actions(); auto hist = e.action_history();
  for (int i = 0; i < NA; ++i) { la[i] = 0.0f; aa[i] = 0.0f; for (int k = 0; k < GO2SIM_ACTION_RING_MAX; ++k) hist[k][i] = 0.0f; }
  { auto ldv = e.last_dof_vel(); for (int i = 0; i < NM; ++i) ldv[i] = 0.0f; }
  e.last_base_pos_x()[0] = bpx;                                       // go2_env_stair.py:1557
  { auto psf = e.push_stored_force(); for (int k = 0; k < 3; ++k) psf[k] = 0.0f; }
  e.push_remaining()[0] = 0;
  { auto fat = e.feet_air_time(); auto fc = e.foot_contact(); auto lfc = e.last_foot_contact(); for (int i = 0; i < 4; ++i) { fat[i] = 0.0f; fc[i] = 0; lfc[i] = 0; } }
  { auto es = e.episode_sums(); for (int k = 0; k < NREW; ++k) es[k] = 0.0f; }
  e.episode_length()[0] = 0; e.reset_buf()[0] = 1;
  dm_u4 r = pre ? pre->cmd : rng4(seed, RNG_RESET_CMD, b, rc, 0);
  float cx, cy, cz;
  sample_commands(c, g, r, b, cx, cy, cz);
  auto cmd = e.commands();
  cmd[0] = cx; cmd[1] = cy; cmd[2] = cz;
}

// per-env tail of a reset call: reset flagged envs, broadcast the "global" DR scalars; the caller then refreshes FK for the
// full batch (the reference's set_dofs_position/set_pos/set_quat/zero_all_dofs_velocity each run a
// full-batch FK: rigid_solver.py:1928-1943,2412-2427)
DEV void reset_tail(const Model& m, const DCfg& c, const Glob& g, const E& e, int b, uint64_t seed) {
  if (g.n_reset_now <= 0) return;
  if (e.reset_buf()[0]) env_reset_one(m, c, g, e, b, seed);
  const bool per_env = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] != 0;      // per-env draws were applied by env_reset_one
  if (c.i[GO2SIM_IC_HAS_FRICTION_DR] && !per_env) { auto gf = e.geom_friction(); for (int i = 0; i < NG; ++i) gf[i] = g.friction; }
  int bl = c.i[GO2SIM_IC_BASE_LINK];
  if (c.i[GO2SIM_IC_HAS_MASS_DR] && !per_env) e.mass_shift()[bl] = g.mass_shift;
  if (c.i[GO2SIM_IC_HAS_COM_DR]) e.com_shift()[bl] = v3(g.com_shift[0], g.com_shift[1], g.com_shift[2]);
  if (c.i[GO2SIM_IC_HAS_LEGM_DR]) for (int k = 0; k < 4; ++k) e.mass_shift()[c.i[GO2SIM_IC_HIP_LINK0 + k]] = g.leg_mass_shift[k];
  // the full-batch FK refresh follows as k_fk_team gated on g.n_reset_now (launch_fk_team)
}

// _assign_terrain_rows, go2_env_stair.py:809-854: 40 % of the reset envs on the frontier row, 30 % just below it, 30 % on easy rows,
// shuffled.  The shuffle is the rank of a per-env Philox key; the j-th reset env (env order) receives rows[perm[j]].  One thread per env;
// a reset env scans the batch once (keys of the other reset envs are recomputed on the fly).  Gated on g.n_reset_now.
__global__ __launch_bounds__(256) void k_env_terrain_rows(Pool P, const DCfg* __restrict__ cp, Glob* gp, uint64_t seed) {
  const DCfg& c = *cp; Glob& g = *gp;
  if (!c.i[GO2SIM_IC_USE_TERRAIN] || g.n_reset_now <= 0) return;
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  if (!e.reset_buf()[0]) return;
  const int n_rows = c.i[GO2SIM_IC_N_TERRAIN_ROWS];
  const uint32_t rc = g.reset_calls - 1;
  const int n = g.n_reset_now;
  int row = e.terrain_row()[0];
  if (n_rows > 1 && !g.lock_terrain_rows) {                            // `if not self._lock_terrain_rows`, go2_env_stair.py:1513
    double level = c.i[GO2SIM_IC_CURR_ENABLED] ? g.level : 1.0;
    int max_row = (int)(level * (double)(n_rows - 1));
    max_row = imx(0, imn(max_row, n_rows - 1));
    int n_frontier = (int)((double)n * 0.40), n_near = (int)((double)n * 0.30);
    const unsigned kj = rng4(seed, RNG_TERRAIN_PERM, b, rc, 0).v[0];
    int p = 0;
    const int* rb = P.i + (size_t)IO(reset_buf) * P.B;
    for (int b2 = 0; b2 < P.B; ++b2) {
      if (!rb[b2] || b2 == b) continue;
      unsigned k2 = rng4(seed, RNG_TERRAIN_PERM, b2, rc, 0).v[0];
      p += (k2 < kj) || (k2 == kj && b2 < b);
    }
    dm_u4 r = rng4(seed, RNG_TERRAIN_ROW, (uint32_t)p, rc, 0);
    if (p < n_frontier) row = max_row;
    else if (p < n_frontier + n_near) row = (max_row >= 2) ? rand_int(imx(0, max_row - 2), imx(0, max_row - 1), r.v[0]) : max_row;
    else row = rand_int(0, (max_row >= 3) ? max_row - 3 : 0, r.v[1]);
    e.terrain_row()[0] = row;
  }
  atomicAdd(&g.terrain_row_sum, row);
}

__global__ __launch_bounds__(WG) void k_env_reset_tail(Pool P, const Model* __restrict__ mp, const DCfg* __restrict__ cp, const Glob* __restrict__ gp, uint64_t seed) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  reset_tail(*mp, *cp, *gp, e, b, seed);
}

// Go2Env.step post-physics part B: reset_idx tail + observations (go2_env_walk.py:1080-1141)
// Go2Env.step tail (go2_env_walk.py:1078-1109): reset_idx of the flagged envs, observation + privileged observation assembly, output
// copies.  Team kernel: T lanes per env.  The reset (rare, serial) runs on lane 0; the 49 observations are produced four per lane (one
// Philox block of noise each), the privileged tail one entry per lane, and the [n_envs, k] outputs are written with coalesced rows.
// sources of the walk-layout observation vectors, staged into LDS in one batch (k_env_post_b_team)
enum { PB_BAV = 0, PB_PG = 3, PB_GOFF = 6, PB_CMD = 9, PB_DP = 12, PB_DV = 24, PB_ACT = 36, PB_BLV = 52, PB_KP = 55, PB_KD = 67, PB_MS = 79, PB_PUSH = 91, PB_N = 94 };
DEV int post_b_src_off(int k) {
  int off = FO(base_ang_vel) + k;
  off = (k >= PB_PG) ? FO(projected_gravity) + k - PB_PG : off;
  off = (k >= PB_GOFF) ? FO(gravity_offset) + k - PB_GOFF : off;
  off = (k >= PB_CMD) ? FO(commands) + k - PB_CMD : off;
  off = (k >= PB_DP) ? FO(e_dof_pos) + k - PB_DP : off;
  off = (k >= PB_DV) ? FO(e_dof_vel) + k - PB_DV : off;
  off = (k >= PB_ACT) ? FO(applied_actions) + k - PB_ACT : off;
  off = (k >= PB_BLV) ? FO(base_lin_vel) + k - PB_BLV : off;
  off = (k >= PB_KP) ? FO(kp_factors) + k - PB_KP : off;
  off = (k >= PB_KD) ? FO(kd_factors) + k - PB_KD : off;
  off = (k >= PB_MS) ? FO(motor_strength) + k - PB_MS : off;
  off = (k >= PB_PUSH) ? FO(current_push_force) + k - PB_PUSH : off;
  return off;
}
// Two wavefronts per workgroup over the SAME 64 / T envs: wavefront 0 resets the flagged envs and writes the observations; wavefront 1 exists for the steps on
// which a reset call happened (in a training run: every step) and refreshes the kinematics of the workgroup's envs WHILE wavefront 0 assembles the
// observations -- neither reads what the other writes, so the slowest workgroup of the launch costs reset + max(observations, FK) instead of their sum.
template <int T>
__global__ __launch_bounds__(128) void k_env_post_b_team(Pool P, const Model* __restrict__ mp, const ModelS* __restrict__ msp, const DCfg* __restrict__ cp,
                                                        const Glob* __restrict__ gp, uint64_t seed, uint32_t step_count, float* __restrict__ obs_out,
                                                        float* __restrict__ priv_out, float* __restrict__ rew_out, uint8_t* __restrict__ reset_out,
                                                        float* __restrict__ timeout_out) {
  STAMP(STK_POST_B)
  constexpr int EPW = 64 / T;
  // FK refresh after a reset call (the reference's set_dofs_position / set_pos / set_quat re-run the full-batch FK, and the "global" mass /
  // COM randomisation touches every env): done here, at the end of the kernel, instead of in a launch of its own.  The model tables are
  // requested first (LDS DMA by all 64 lanes, before any lane retires) and are complete at the barrier that precedes their use.
  __shared__ KinData fk_lds[EPW];
  __shared__ alignas(16) char ms_raw[MODELS_LDS_BYTES];
  const bool fk_needed = gp->n_reset_now > 0;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  if (wave == 1) {
    if (!fk_needed) return;
    wg_dma_to_lds<(int)sizeof(ModelS)>(ms_raw, msp, lane);
  }
  const int tl = lane % T, slot = lane / T;
  const int b = xcd_block() * EPW + slot;
  if (wave == 1) {                                                     // the FK wavefront
    __syncthreads();                                                   // wavefront 0 has reset its flagged envs and applied the global DR scalars
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (b >= P.B) return;
    E e(P, b);
    KinData* ks = &fk_lds[slot];
    team_stage<ND, T>(tl, [&](int d) { return gload(e, FO(vel), d); }, [&](int d, float v) { ks->vel[d] = v; });
    team_stage<NQ, T>(tl, [&](int q) { return gload(e, FO(qpos), q); }, [&](int q, float v) { ks->qpos[q] = v; });
    tk_stage_links<T>(e, ks, tl);
    team_sync();
    const ModelView mv((const ModelS*)ms_raw, msp);
    tk_kinematics<T>(mv, e, ks, tl, true);
    return;
  }
  const bool valid = b < P.B;                                          // (lanes beyond the batch stay with their wavefront up to its ONE barrier)
  const Model& m = *mp; const DCfg& c = *cp; const Glob& g = *gp;
  E e(P, valid ? b : P.B - 1);
  auto fk_refresh = [&]() {};                                          // (done by wavefront 1)
  const int was_reset = valid ? e.reset_buf()[0] : 0;
  if (valid && g.n_reset_now > 0) {                                    // reset_tail, spread over the team
    {
      const ResetDraws rd = env_reset_draws_team<T>(c, g, e, b, seed, tl, was_reset != 0);    // (all lanes of the team: the blocks are shuffled)
      if (tl == 0 && was_reset) env_reset_one(m, c, g, e, b, seed, &rd);
    }
    const bool per_env = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] != 0;      // per-env draws were applied by env_reset_one
    if (c.i[GO2SIM_IC_HAS_FRICTION_DR] && !per_env) for (int i = tl; i < NG; i += T) e.geom_friction()[i] = g.friction;
    if (tl == 0) {
      int bl = c.i[GO2SIM_IC_BASE_LINK];
      if (c.i[GO2SIM_IC_HAS_MASS_DR] && !per_env) e.mass_shift()[bl] = g.mass_shift;
      if (c.i[GO2SIM_IC_HAS_COM_DR]) e.com_shift()[bl] = v3(g.com_shift[0], g.com_shift[1], g.com_shift[2]);
      if (c.i[GO2SIM_IC_HAS_LEGM_DR]) for (int k = 0; k < 4; ++k) e.mass_shift()[c.i[GO2SIM_IC_HIP_LINK0 + k]] = g.leg_mass_shift[k];
    }
    team_sync();
  }
  if (fk_needed) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); }   // hand the reset state over to the FK wavefront
  if (!valid) return;
  const int na = c.i[GO2SIM_IC_NUM_ACTIONS], nobs = c.i[GO2SIM_IC_NUM_OBS], npriv = c.i[GO2SIM_IC_NUM_PRIV_OBS];
  if (c.i[GO2SIM_IC_ENV_KIND] == 1) {                                  // go2_env_base.py:165-196: rewards after the reset, 45 observations
    if (tl == 0) {
      RewCtx rc; rc.was_reset = was_reset;
      auto bvw = e.base_vel_world();
      for (int k = 0; k < 3; ++k) rc.vel_world[k] = was_reset ? 0.0f : bvw[k];    // get_vel() after zero_all_dofs_velocity
      RewState rs;
      load_rew_state(c, e, rs);
      {
        auto blv = e.base_lin_vel(); auto bav = e.base_ang_vel(); auto pg = e.projected_gravity(); auto bp = e.base_pos(); auto dp = e.e_dof_pos(); auto dv = e.e_dof_vel(); auto fc = e.foot_contact();
        for (int k = 0; k < 3; ++k) { rs.blv[k] = blv[k]; rs.bav[k] = bav[k]; rs.pg[k] = pg[k]; rs.base_pos[k] = bp[k]; }
        for (int k = 0; k < NM; ++k) { rs.dof_pos[k] = dp[k]; rs.dof_vel[k] = dv[k]; }
        for (int k = 0; k < 4; ++k) rs.fc[k] = fc[k];
      }
      float rew = 0.0f;
      const RewGates gates = reward_gates(rs);
      auto rew_terms = e.rew_terms(); auto episode_sums = e.episode_sums();
      for (int k = 0; k < c.i[GO2SIM_IC_N_REWARDS]; ++k) {
        float r = reward_term(m, c, rs, c.i[GO2SIM_IC_REWARD_ID0 + k], rc, gates) * c.f[GO2SIM_FC_REWARD_SCALE0 + k];
        rew_terms[k] = r;
        rew = rew + r;
        episode_sums[k] = episode_sums[k] + r;
      }
      e.rew()[0] = rew;
      { auto fat = e.feet_air_time(); for (int k = 0; k < 4; ++k) fat[k] = rs.fat[k]; e.last_base_pos_x()[0] = rs.last_x; }
    }
    team_sync();
    for (int i = tl; i < nobs; i += T) {
      float v;
      if (i < 3) v = gload(e, FO(base_ang_vel), i) * c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL];
      else if (i < 6) v = gload(e, FO(projected_gravity), i - 3);
      else if (i < 9) v = gload(e, FO(commands), i - 6) * ((i - 6 < 2) ? c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL] : c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL]);
      else if (i < 21) v = (gload(e, FO(e_dof_pos), i - 9) - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i - 9]) * c.f[GO2SIM_FC_OBS_SCALE_DOF_POS];
      else if (i < 33) v = gload(e, FO(e_dof_vel), i - 21) * c.f[GO2SIM_FC_OBS_SCALE_DOF_VEL];
      else v = (i - 33 < na) ? gload(e, FO(actions), i - 33) : gload(e, FO(obs), i);
      gstore(e, FO(obs), i, v); gstore(e, FO(priv), i, v);
      if (obs_out) obs_out[(size_t)b * nobs + i] = v;
      if (priv_out && i < npriv) priv_out[(size_t)b * npriv + i] = v;
    }
    for (int i = tl; i < na; i += T) gstore(e, FO(last_actions), i, gload(e, FO(actions), i));
    for (int i = tl; i < NM; i += T) gstore(e, FO(last_dof_vel), i, gload(e, FO(e_dof_vel), i));
    if (tl == 0) {
      if (rew_out) rew_out[b] = e.rew()[0];
      if (reset_out) reset_out[b] = (uint8_t)was_reset;
      if (timeout_out) timeout_out[b] = e.time_out()[0];
    }
    fk_refresh();
    return;
  }
  // every source of the two observation vectors is fetched in one batch (the assembly below would otherwise pay a memory round trip per entry)
  __shared__ float pb_src[EPW][PB_N + 2];
  float* src = pb_src[slot];
  team_stage<PB_N, T>(tl, [&](int k) { return gload(e, post_b_src_off(k), 0); }, [&](int k, float v) { src[k] = v; });
  team_sync();
  const bool noisy = c.i[GO2SIM_IC_HAS_OBS_NOISE] && c.d[GO2SIM_FC_OBS_NOISE_LEVEL_MAX] > 0.0;
  const double lvl = g.obs_noise_level_cur;          // python-float products, rounded on assignment into the float32 noise vector
  for (int blk = tl; blk * 4 < nobs; blk += T) {
    float n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (noisy) {
      dm_u4 r = rng4(seed, RNG_OBS_NOISE, b, step_count, blk);
      dm_normal2(r.v[0], r.v[1], &n[0], &n[1]); dm_normal2(r.v[2], r.v[3], &n[2], &n[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int i = 4 * blk + k;
      if (i < nobs) {
        float v, nv = 0.0f;
        if (i < 3) { v = src[PB_BAV + i] * c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL]; nv = (float)(c.d[GO2SIM_FC_OBS_NOISE_ANG_VEL] * c.d[GO2SIM_FC_OBS_SCALE_ANG_VEL] * lvl); }
        else if (i < 6) { v = src[PB_PG + i - 3] + src[PB_GOFF + i - 3]; nv = (float)(c.d[GO2SIM_FC_OBS_NOISE_GRAVITY] * lvl); }
        else if (i < 9) { v = src[PB_CMD + i - 6] * ((i - 6 < 2) ? c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL] : c.f[GO2SIM_FC_OBS_SCALE_ANG_VEL]); nv = 0.0f; }
        else if (i < 21) { v = (src[PB_DP + i - 9] - c.f[GO2SIM_FC_DEFAULT_DOF_POS0 + i - 9]) * c.f[GO2SIM_FC_OBS_SCALE_DOF_POS]; nv = (float)(c.d[GO2SIM_FC_OBS_NOISE_DOF_POS] * c.d[GO2SIM_FC_OBS_SCALE_DOF_POS] * lvl); }
        else if (i < 33) { v = src[PB_DV + i - 21] * c.f[GO2SIM_FC_OBS_SCALE_DOF_VEL]; nv = (float)(c.d[GO2SIM_FC_OBS_NOISE_DOF_VEL] * c.d[GO2SIM_FC_OBS_SCALE_DOF_VEL] * lvl); }
        else { v = (i - 33 < na) ? src[PB_ACT + i - 33] : gload(e, FO(obs), i); }
        if (noisy) v = v + n[k] * nv;
        gstore(e, FO(obs), i, v); gstore(e, FO(priv), i, v);
        if (obs_out) obs_out[(size_t)b * nobs + i] = v;
        if (priv_out) priv_out[(size_t)b * npriv + i] = v;
      }
    }
  }
  // privileged tail (go2_env_walk.py:1115-1143)
  for (int i = nobs + tl; i < npriv; i += T) {
    int j = i - nobs;
    float v; bool write = true;
    if (j < 3) v = src[PB_BLV + j] * c.f[GO2SIM_FC_OBS_SCALE_LIN_VEL];
    else if (j < 4) v = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] ? gload(e, FO(geom_friction), NG - 1) : g.friction;
    else if (j < 16) v = src[PB_KP + j - 4];
    else if (j < 28) v = src[PB_KD + j - 16];
    else if (j < 40) v = src[PB_MS + j - 28];
    else if (j < 41) v = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] ? gload(e, FO(mass_shift), c.i[GO2SIM_IC_BASE_LINK]) : g.mass_shift;
    else if (j < 44) v = g.com_shift[j - 41];
    else if (j < 48) v = g.leg_mass_shift[j - 44];
    else if (j < 51) v = src[PB_GOFF + j - 48];
    else if (j < 54) v = src[PB_PUSH + j - 51];
    else if (j < 55) { write = c.i[GO2SIM_IC_MAX_DELAY] > 0; v = write ? (float)e.delay_steps()[0] / (float)c.i[GO2SIM_IC_MAX_DELAY] : gload(e, FO(priv), i); }
    else if (c.i[GO2SIM_IC_USE_TERRAIN] && j == 55) v = (float)e.terrain_row()[0] / (float)imx(1, c.i[GO2SIM_IC_N_TERRAIN_ROWS] - 1);   // go2_env_stair.py:1466-1472
    else if (c.i[GO2SIM_IC_USE_TERRAIN] && j - 56 < c.i[GO2SIM_IC_SCAN_N] && 56 + c.i[GO2SIM_IC_SCAN_N] <= npriv - nobs) {        // _compute_height_scan :772-803
      auto bq = e.base_quat(); auto bp = e.base_pos();
      float qw = bq[0], qx = bq[1], qy = bq[2], qz = bq[3];
      float yaw = dm_atan2(2.0f * (qw * qz + qx * qy), 1.0f - 2.0f * (qy * qy + qz * qz));
      float sy, cy;
      dm_sincos(yaw, &sy, &cy);
      float lx = c.f[GO2SIM_FC_SCAN_X0 + j - 56], ly = c.f[GO2SIM_FC_SCAN_Y0 + j - 56];
      float wx = bp[0] + cy * lx - sy * ly;
      float wy = bp[1] + sy * lx + cy * ly;
      v = terrain_height(m, c, wx, wy) - bp[2];
    }
    else v = 0.0f;
    if (write) gstore(e, FO(priv), i, v);
    if (priv_out) priv_out[(size_t)b * npriv + i] = v;
  }
  for (int i = tl; i < na; i += T) gstore(e, FO(last_actions), i, gload(e, FO(actions), i));
  for (int i = tl; i < NM; i += T) gstore(e, FO(last_dof_vel), i, src[PB_DV + i]);
  if (tl == 0) {
    if (rew_out) rew_out[b] = e.rew()[0];
    if (reset_out) reset_out[b] = (uint8_t)was_reset;
    if (timeout_out) timeout_out[b] = e.time_out()[0];
  }
  fk_refresh();
}

__global__ __launch_bounds__(WG) void k_init_state(Pool P, const Model* __restrict__ mp, int keep_dr) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  const Model& m = *mp;
  E e(P, b);
  // everything is zero (hipMemset) except what follows when keep_dr == 0
  auto qpos = e.qpos();
  for (int i = 0; i < NQ; ++i) qpos[i] = m.qpos0[i];
  if (!keep_dr) {
    auto fr = e.friction_ratio(); auto gf = e.geom_friction();
    for (int i = 0; i < NG; ++i) { fr[i] = 1.0f; gf[i] = m.geoms[i].friction; }
  }
  auto l_pos = e.l_pos(); auto l_quat = e.l_quat();
  for (int i = 0; i < NL; ++i) { l_pos[i] = m.links[i].pos; l_quat[i] = m.links[i].quat; }
  e.first_time()[0] = 1;
  // FK / velocities of the initial state: launch_fk_team right after this kernel
}
__global__ __launch_bounds__(WG) void k_scene_reset_clear(Pool P) {   // RigidSolver.set_state + collider.clear + constraint_solver.clear
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  auto vel = e.vel(); auto acc = e.acc(); auto qacc_ws = e.qacc_ws();
  for (int d = 0; d < ND; ++d) { vel[d] = 0.0f; acc[d] = 0.0f; qacc_ws[d] = 0.0f; }
  auto ncv = e.ncache_valid();
  for (int p = 0; p < NCV; ++p) ncv[p] = 0;                               // normal cache := zeros
  int nc = e.n_contacts()[0];
  for (int i_c = 0; i_c < nc; ++i_c) { e.c_link()[i_c] = -1; e.c_link()[MAXC + i_c] = -1; e.c_geom()[i_c] = -1; e.c_geom()[MAXC + i_c] = -1; e.c_pen()[i_c] = 0.0f; e.c_pos()[i_c] = v3(0, 0, 0); e.c_normal()[i_c] = v3(0, 0, 0); e.c_force()[i_c] = v3(0, 0, 0); }
  e.n_contacts()[0] = 0; e.err()[0] = 0; e.is_warmstart()[0] = 0; e.n_con()[0] = 0;
  auto ext = e.ext();
  for (int i = 0; i < NL * 6; ++i) ext[i] = 0.0f;
}
__global__ __launch_bounds__(WG) void k_reset_caches(Pool P, const int* __restrict__ envs_idx, int n_sel) {
  int t = blockIdx.x * WG + threadIdx.x;
  if (t >= n_sel) return;
  int b = envs_idx ? envs_idx[t] : t;
  if (b < 0 || b >= P.B) return;
  E e(P, b);
  e.err()[0] = 0; e.is_warmstart()[0] = 0;
  auto qacc_ws = e.qacc_ws();
  for (int d = 0; d < ND; ++d) qacc_ws[d] = 0.0f;
  auto ncv = e.ncache_valid();
  for (int p = 0; p < NCV; ++p) ncv[p] = 0;                               // normal cache := zeros
}
__global__ __launch_bounds__(WG) void k_set_friction(Pool P, float mu) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  auto gf = e.geom_friction();
  for (int i = 0; i < NG; ++i) gf[i] = mu;
}
__global__ __launch_bounds__(WG) void k_env_init_buffers(Pool P, const DCfg* __restrict__ cp) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  const DCfg& c = *cp;
  E e(P, b);
  auto kpf = e.kp_factors(); auto kdf = e.kd_factors(); auto mst = e.motor_strength();
  for (int k = 0; k < NM; ++k) { kpf[k] = 1.0f; kdf[k] = 1.0f; mst[k] = 1.0f; }
  e.delay_steps()[0] = 1; e.reset_buf()[0] = 1;
  if (c.i[GO2SIM_IC_MANUAL_PD]) for (int k = 0; k < NM; ++k) e.ctrl_mode()[c.i[GO2SIM_IC_MOTOR_DOF0 + k]] = CTRL_FORCE;
}
__global__ void k_errno_reduce(Pool P, int* out) {
  int v = 0;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < P.B; b += gridDim.x * blockDim.x) v |= P.i[(size_t)IO(err) * P.B + b];
  if (v) atomicOr(out, v);
}
// [n_envs][k] row-major copy of an env buffer
// field API <-> AoS records (32-bit words): rows[j][b] = rec[b][off + j]
// GO2SIM_F_MASS_MAT presents the reference's full [n_dofs, n_dofs] matrix; the record holds its lower triangle packed
__global__ void k_mass_mat_to_rows(const float* __restrict__ rec, int stride, int off, int B, float* __restrict__ rows) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ND * ND * B) return;
  const int idx = t / B, b = t - idx * B, i = idx / ND, j = idx - i * ND, a = i > j ? i : j, c = i > j ? j : i;
  rows[t] = rec[(size_t)b * stride + off + a * (a + 1) / 2 + c];
}
__global__ void k_rows_to_mass_mat(const float* __restrict__ rows, float* __restrict__ rec, int stride, int off, int B) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NTRI * B) return;
  const int p = t / B, b = t - p * B;
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= p) ++i;
  const int j = p - i * (i + 1) / 2;
  rec[(size_t)b * stride + off + p] = rows[(size_t)(i * ND + j) * B + b];
}
__global__ void k_aos_to_rows(const int* __restrict__ rec, int stride, int off, int k, int B, int* __restrict__ rows) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * B) return;
  int j = t / B, b = t % B;
  rows[t] = rec[(size_t)b * stride + off + j];
}
__global__ void k_rows_to_aos(const int* __restrict__ rows, int* __restrict__ rec, int stride, int off, int k, int B) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * B) return;
  int j = t / B, b = t % B;
  rec[(size_t)b * stride + off + j] = rows[t];
}
// F_NORMAL_CACHE through the field API: rows of pairs whose valid bit is clear read as zeros; a set_field makes every uploaded entry valid
__global__ void k_ncache_mask_rows(Pool P, float* __restrict__ rows) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NPAIR * P.B) return;
  int p = t / P.B, b = t % P.B;
  E e(P, b);
  if (!(((unsigned)e.ncache_valid()[p >> 5] >> (p & 31)) & 1u)) { rows[(size_t)(3 * p) * P.B + b] = 0.0f; rows[(size_t)(3 * p + 1) * P.B + b] = 0.0f; rows[(size_t)(3 * p + 2) * P.B + b] = 0.0f; }
}
__global__ void k_ncache_set_valid(Pool P) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NCV * P.B) return;
  int w = t / P.B, b = t % P.B;
  E e(P, b);
  const int n = NPAIR - 32 * w;
  e.ncache_valid()[w] = (int)(n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
}
__global__ void k_gather(const void* __restrict__ src, void* __restrict__ dst, int k, int B) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * B) return;
  int b = t / k, j = t % k;
  ((uint32_t*)dst)[t] = ((const uint32_t*)src)[(size_t)j * B + b];
}
__global__ void k_scatter(const void* __restrict__ src, void* __restrict__ dst, int k, int B) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * B) return;
  int b = t / k, j = t % k;
  ((uint32_t*)dst)[(size_t)j * B + b] = ((const uint32_t*)src)[t];
}

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "go2sim: HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); return GO2SIM_E_HIP; } } while (0)

enum { T_DYN = 0, T_COLLIDE, T_SOLVE, T_INTEGRATE, T_ENV_PRE, T_ENV_POST, T_MISC, T_TOTAL, T_N };
constexpr int TIMING_RING = 2048;

struct go2sim {
  int B = 0, device = 0;
  uint64_t seed = 0;
  Model hm;                 // host copy of the model
  Model* dm = nullptr;      // device copy
  ModelS* dms = nullptr;    // device copy of the compact tables (staged into LDS by the team kernels)
  Pool P = {nullptr, nullptr, 0, nullptr, nullptr};
  DCfg hcfg; DCfg* dcfg = nullptr; bool cfg_set = false;
  Glob* dglob = nullptr; Acc* dacc = nullptr; int* derr = nullptr;      // derr[0]: go2sim_check_errno, derr[1]: asynchronous poll
  int* herr_pinned = nullptr; hipEvent_t ev_errno = nullptr; bool errno_poll_pending = false;   // go2sim_errno_poll_*
  int* didx = nullptr; int didx_cap = 0;    // scratch for index lists (go2sim_env_reset_idx)
  float* terrain_hf = nullptr;              // device copy of the heightfield in metres (go2sim_set_terrain)
  float* terrain_cmax = nullptr;            // ... and its coarse maximum map
  GjkStoreFull* gjk_scratch = nullptr;      // full-capacity polytope records of the GJK / EPA fallback (queries that outgrow their LDS slot) and
                                            // prism descriptors of the terrain pass: one block per (env, narrow-phase lane)
  SolverData<MAXR>* solver_ovf = nullptr;   // per-env global scratch for solves that do not fit in LDS (> RL rows)
  int* lpt = nullptr; int lpt_cap = 0; int lpt_parity = 0; bool use_lpt = true, lpt_flat = false;   // heaviest-first dispatch records of the solver (two, alternating)
  // One env step = 11 dependent kernel launches (12 with terrain).  Issued one by one they cost the host ~20 us each -- close to the GPU time of
  // the step -- so the sequence is kept as an instantiated hipGraph: per step the three step-dependent kernel nodes get their new arguments
  // (actions pointer, step counter, ring index) and the graph is launched with one call.  GO2SIM_NO_GRAPH=1 (or timing mode) uses plain launches.
  struct StepGraph {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    hipGraphNode_t n_pre = nullptr, n_post_a = nullptr, n_post_b = nullptr;
    hipGraphNode_t extra_dep = nullptr;             // graph_add_kernel: second parent of the next node (consumed by that call)
    hipKernelNodeParams p_pre{}, p_post_a{}, p_post_b{};
    std::vector<void*> owned;                       // argument storage of all nodes (malloc'ed)
    const float** a_actions = nullptr; uint32_t* a_pre_step = nullptr; int* a_pre_widx = nullptr; uint32_t* a_pa_step = nullptr; uint32_t* a_pb_step = nullptr;
    // output pointers of the step (k_env_post_b_team): rewritten every step like the step counter, so the caller may hand over fresh
    // observation tensors per step (the reference allocates a new obs tensor each step, go2_env_walk.py:1084) without a graph rebuild
    float** a_obs = nullptr; float** a_priv = nullptr; float** a_rew = nullptr; uint8_t** a_reset = nullptr; float** a_timeout = nullptr;
    bool valid = false;
  } sg;
  bool use_graph = true;
  int graph_fallbacks = 0;                  // times the graph path was abandoned for plain launches (go2sim_graph_status)
  bool fuse_fk_dyn = true;                  // k_integrate_fk_dynamics_team between the substeps of a scene step (GO2SIM_NO_FUSE=1: separate launches)
  bool fuse_solve_int = true;               // k_solve_integrate_team: the solve and the kinematics (+ next dynamics) after it in one launch (GO2SIM_NO_FUSE_SOLVE=1: separate)
  bool par_pre = false;                     // step graph: the first collision pass as a second root beside the pre-physics / dynamics launch (GO2SIM_PAR_PRE=1).
                                            // Measured slower than the chain (window 10.26 vs 10.40 M, default run 14.84 vs 15.24 M, stairs 7.51 vs 7.61 M env-steps/s):
                                            // the dynamics wavefronts take SIMDs from the collision pass instead of waiting for its early finishers
  int dyn_team = 32;                        // lanes per environment in k_dynamics_team
  int fk_team = 16;                         // lanes per environment in k_integrate_fk_team / k_fk_team
  int collide_team = 16;                    // lanes per environment in k_collide_team
  int collide_epw = 0;                      // environments per wavefront in k_collide_team<16> (0 = 4, a full wavefront; GO2SIM_COLLIDE_EPW = 1 / 2 leave lanes idle)
  int solver_team = 32;                     // lanes per environment in k_constraint_solve_team
  int terrain_solver_team = 64;             // ... on heightfield terrain (96 LDS rows)
  uint32_t step_count = 0; int action_write_idx = 0;
  // timing
  bool timing = false;
  hipEvent_t ev0[TIMING_RING], ev1[TIMING_RING]; int ev_cat[TIMING_RING]; int ev_n = 0; bool ev_created = false;
  float t_ms[T_N] = {0}; int t_cnt[T_N] = {0};
};

static inline dim3 grid_for(int B) { return dim3((B + WG - 1) / WG); }

static void timing_flush(go2sim* h) {
  for (int i = 0; i < h->ev_n; ++i) {
    float ms = 0.0f;
    (void)hipEventSynchronize(h->ev1[i]);
    if (hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]) == hipSuccess) { h->t_ms[h->ev_cat[i]] += ms; h->t_cnt[h->ev_cat[i]] += 1; }
  }
  h->ev_n = 0;
}
struct ScopedTimer {
  go2sim* h; hipStream_t s; int idx;
  ScopedTimer(go2sim* h_, hipStream_t s_, int cat) : h(h_), s(s_), idx(-1) {
    if (!h->timing) return;
    if (h->ev_n == TIMING_RING) return;   // ring full inside a step: drop the sample (flushes happen at step boundaries only)
    idx = h->ev_n++;
    h->ev_cat[idx] = cat;
    (void)hipEventRecord(h->ev0[idx], s);
  }
  ~ScopedTimer() { if (idx >= 0) (void)hipEventRecord(h->ev1[idx], s); }
};

static void launch_fk_team(go2sim* h, hipStream_t s, int force_update_fixed, const int* cond) {
  const int T = h->fk_team;
  dim3 gd((h->B + 64 / T - 1) / (64 / T)), b(64);
  if (T == 16) hipLaunchKernelGGL(k_fk_team<16>, gd, b, 0, s, h->P, h->dms, force_update_fixed, cond);
  else if (T == 32) hipLaunchKernelGGL(k_fk_team<32>, gd, b, 0, s, h->P, h->dms, force_update_fixed, cond);
  else hipLaunchKernelGGL(k_fk_team<64>, gd, b, 0, s, h->P, h->dms, force_update_fixed, cond);
}

// `actions` != nullptr: the first dynamics launch of an env step, with the pre-physics part of the step in the same kernel (k_pre_dynamics_team)
static bool fuse_pre(const go2sim* h) { return h->fuse_fk_dyn && h->dyn_team >= 32; }
static void launch_dynamics(go2sim* h, hipStream_t s, const float* actions = nullptr) {
  ScopedTimer t(h, s, T_DYN);
  const int T = h->dyn_team;
  dim3 gd((h->B + 64 / T - 1) / (64 / T)), b(WG);
  if (actions) {
    if (T == 32) hipLaunchKernelGGL(k_pre_dynamics_team<32>, gd, b, 0, s, h->P, h->dms, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx);
    else hipLaunchKernelGGL(k_pre_dynamics_team<64>, gd, b, 0, s, h->P, h->dms, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx);
    return;
  }
  if (T == 16) hipLaunchKernelGGL(k_dynamics_team<16>, gd, b, 0, s, h->P, h->dms);
  else if (T == 32) hipLaunchKernelGGL(k_dynamics_team<32>, gd, b, 0, s, h->P, h->dms);
  else hipLaunchKernelGGL(k_dynamics_team<64>, gd, b, 0, s, h->P, h->dms);
}
static int collide_epw(const go2sim* h) { return (h->collide_team == 16 && h->collide_epw > 0) ? h->collide_epw : 64 / h->collide_team; }
static int solver_epw(const go2sim* h) { return 64 / (h->hm.terrain_enabled ? h->terrain_solver_team : h->solver_team); }
// the envs of a solver block must sit in one collision wavefront (their contact counts meet there).  On flat ground (two envs per solver wavefront, one
// residency round) the sorted order measured 1-3 % SLOWER than the identity order, with single envs as well as with adjacent pairs as the sorted unit
// (the record lookup delays every workgroup's first loads; there is no second round to win it back): off unless GO2SIM_LPT_FLAT=1
static bool lpt_enabled(const go2sim* h) {
  const int epw_c = collide_epw(h), epw_s = solver_epw(h);
  if (!h->use_lpt || epw_s > epw_c || epw_c % epw_s != 0) return false;
  return epw_s == 1 || h->lpt_flat;
}
static size_t lpt_record_ints(const go2sim* h) { return (size_t)8 * LPT_CLS * (1 + h->lpt_cap); }
// flat ground, 32-lane teams in solver and dynamics: the solve and the kinematics (+ next dynamics) that follow it share a launch (k_solve_integrate_team);
// GO2SIM_NO_FUSE_SOLVE=1 keeps the two launches
static bool fuse_solve(const go2sim* h) { return h->fuse_solve_int && h->fuse_fk_dyn && !h->hm.terrain_enabled && h->solver_team == 32 && h->dyn_team == 32; }
// fuse_mode: 0 = the solve alone; 1 = + integrate / kinematics / next dynamics; 2 = + integrate / kinematics (last substep)
static void launch_collide_solve(go2sim* h, hipStream_t s, int fuse_mode = 0) {
  dim3 b(WG);
  const bool lpt_on = lpt_enabled(h);
  int* lpt_cur = lpt_on ? h->lpt + h->lpt_parity * lpt_record_ints(h) : nullptr;
  int* lpt_next = lpt_on ? h->lpt + (1 - h->lpt_parity) * lpt_record_ints(h) : nullptr;
  h->lpt_parity ^= 1;
  const int epw_s = solver_epw(h);
  {
    ScopedTimer t(h, s, T_COLLIDE);
    const int T = h->collide_team;
    const int epw_c = collide_epw(h);
    dim3 gc((h->B + epw_c - 1) / epw_c);
    if (T == 16 && epw_c == 2) hipLaunchKernelGGL((k_collide_team<16, 2>), gc, b, 0, s, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s);
    else if (T == 16 && epw_c == 1) hipLaunchKernelGGL((k_collide_team<16, 1>), gc, b, 0, s, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s);
    else if (T == 16) hipLaunchKernelGGL(k_collide_team<16>, gc, b, 0, s, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s);
    else if (T == 32) hipLaunchKernelGGL(k_collide_team<32>, gc, b, 0, s, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s);
    else hipLaunchKernelGGL(k_collide_team<64>, gc, b, 0, s, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s);
  }
  {
    ScopedTimer t(h, s, T_SOLVE);
    // flat ground: 32 lanes per env and 32 LDS rows; heightfield terrain (many more contacts): one env per wavefront with 96 LDS rows
    if (h->hm.terrain_enabled) {
      if (h->terrain_solver_team == 32) hipLaunchKernelGGL((k_constraint_solve_team<32, RL_TERRAIN>), dim3((h->B + 1) / 2), b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else hipLaunchKernelGGL((k_constraint_solve_team<64, RL_TERRAIN>), dim3(h->B), b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
    } else {
      const int T = h->solver_team;
      dim3 gs((h->B + 64 / T - 1) / (64 / T));
      if (T == 16) hipLaunchKernelGGL((k_constraint_solve_team<16, RL>), gs, b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else if (T == 32 && fuse_mode == 1) hipLaunchKernelGGL((k_solve_integrate_team<32, RL, true>), gs, b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else if (T == 32 && fuse_mode == 2) hipLaunchKernelGGL((k_solve_integrate_team<32, RL, false>), gs, b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else if (T == 32) hipLaunchKernelGGL((k_constraint_solve_team<32, RL>), gs, b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else hipLaunchKernelGGL((k_constraint_solve_team<64, RL>), gs, b, 0, s, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
    }
  }
}
static void launch_integrate(go2sim* h, hipStream_t s) {
  ScopedTimer t(h, s, T_INTEGRATE);
  const int T = h->fk_team;
  dim3 gd((h->B + 64 / T - 1) / (64 / T)), b(WG);
  if (T == 16) hipLaunchKernelGGL(k_integrate_fk_team<16>, gd, b, 0, s, h->P, h->dms);
  else if (T == 32) hipLaunchKernelGGL(k_integrate_fk_team<32>, gd, b, 0, s, h->P, h->dms);
  else hipLaunchKernelGGL(k_integrate_fk_team<64>, gd, b, 0, s, h->P, h->dms);
}
// integrate + FK of one substep and the forward dynamics of the next one in a single launch (accounted with the integrate class)
static void launch_integrate_dynamics(go2sim* h, hipStream_t s) {
  ScopedTimer t(h, s, T_INTEGRATE);
  const int T = h->dyn_team == 64 ? 64 : 32;
  dim3 gd((h->B + 64 / T - 1) / (64 / T)), b(WG);
  if (T == 32) hipLaunchKernelGGL(k_integrate_fk_dynamics_team<32>, gd, b, 0, s, h->P, h->dms);
  else hipLaunchKernelGGL(k_integrate_fk_dynamics_team<64>, gd, b, 0, s, h->P, h->dms);
}
// n substeps of RigidSolver.substep (rigid_solver.py:1116-1184): dynamics | collide, solve | integrate+FK, where the integrate of substep i and the
// dynamics of substep i + 1 share a launch
static int launch_substeps(go2sim* h, hipStream_t s, int n, const float* pre_actions = nullptr) {
  launch_dynamics(h, s, pre_actions);
  for (int i = 0; i < n; ++i) {
    if (fuse_solve(h)) { launch_collide_solve(h, s, i + 1 < n ? 1 : 2); continue; }
    launch_collide_solve(h, s);
    if (i + 1 < n && h->fuse_fk_dyn) launch_integrate_dynamics(h, s);
    else { launch_integrate(h, s); if (i + 1 < n) launch_dynamics(h, s); }
  }
  return GO2SIM_E_OK;
}
static int launch_substep(go2sim* h, hipStream_t s) { return launch_substeps(h, s, 1); }

// ---- hipGraph of one env step -----------------------------------------------------------------------------------------------------
static void step_graph_destroy(go2sim* h) {
  auto& g = h->sg;
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  for (void* p : g.owned) free(p);
  g = go2sim::StepGraph();
}
// appends one kernel node after `last`; the argument values are copied into storage owned by the graph record, laid out with the kernel's
// own parameter types (P...), and *slots receives the addresses of the stored arguments
template <class... P, class... A>
static bool graph_add_kernel(go2sim* h, hipGraphNode_t& last, void (*kernel)(P...), dim3 grid, dim3 block, hipGraphNode_t* node_out, hipKernelNodeParams* params_out,
                             void*** slots, A... a) {
  static_assert(sizeof...(P) == sizeof...(A), "argument count");
  auto& g = h->sg;
  void** ptrs = (void**)malloc(sizeof(void*) * sizeof...(P));
  g.owned.push_back(ptrs);
  int i = 0;
  auto store = [&](auto typed) { using T = decltype(typed); T* m = (T*)malloc(sizeof(T)); memcpy((void*)m, (const void*)&typed, sizeof(T)); g.owned.push_back((void*)m); ptrs[i++] = (void*)m; };
  (store(static_cast<P>(a)), ...);
  hipKernelNodeParams kp{};
  kp.func = (void*)kernel; kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = 0; kp.kernelParams = ptrs; kp.extra = nullptr;
  hipGraphNode_t node = nullptr;
  hipGraphNode_t deps[2] = {last, g.extra_dep};                       // (extra_dep: the second parent of a join node, set by the caller for one call)
  const int n_deps = (last ? 1 : 0) + ((last && g.extra_dep) ? 1 : 0);
  g.extra_dep = nullptr;
  if (hipGraphAddKernelNode(&node, g.graph, n_deps ? deps : nullptr, n_deps, &kp) != hipSuccess) return false;
  last = node;
  if (node_out) *node_out = node;
  if (params_out) *params_out = kp;
  if (slots) *slots = ptrs;
  return true;
}
template <int T> struct TeamTag {};
// the launch sequence of go2sim_env_step, as graph nodes (kept next to it: both must list the same kernels in the same order)
static bool step_graph_build(go2sim* h, const float* actions, float* obs, float* priv, float* rew, uint8_t* reset, float* timeout) {
  step_graph_destroy(h);
  auto& g = h->sg;
  if (hipGraphCreate(&g.graph, 0) != hipSuccess) return false;
  hipGraphNode_t last = nullptr;
  const dim3 ge = grid_for(h->B), be(WG), b64(64);
  void** sl = nullptr;
  auto team_grid = [&](int T) { return dim3((h->B + 64 / T - 1) / (64 / T)); };
  // first node: the pre-physics part, alone (k_env_pre) or in front of the first dynamics (k_pre_dynamics_team); same per-step argument slots
  bool ok;
  if (!fuse_pre(h)) ok = graph_add_kernel(h, last, k_env_pre, ge, be, &g.n_pre, &g.p_pre, &sl, h->P, h->dm, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx);
  else if (h->dyn_team == 32) ok = graph_add_kernel(h, last, k_pre_dynamics_team<32>, team_grid(32), b64, &g.n_pre, &g.p_pre, &sl, h->P, h->dms, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx);
  else ok = graph_add_kernel(h, last, k_pre_dynamics_team<64>, team_grid(64), b64, &g.n_pre, &g.p_pre, &sl, h->P, h->dms, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx);
  if (!ok) return false;
  g.a_actions = (const float**)sl[4]; g.a_pre_step = (uint32_t*)sl[6]; g.a_pre_widx = (int*)sl[7];
  const int substeps = h->hcfg.i[GO2SIM_IC_SUBSTEPS];
  auto add_dynamics = [&]() {
    const int T = h->dyn_team; const dim3 gd = team_grid(T);
    return T == 16 ? graph_add_kernel(h, last, k_dynamics_team<16>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms)
         : T == 32 ? graph_add_kernel(h, last, k_dynamics_team<32>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms)
                   : graph_add_kernel(h, last, k_dynamics_team<64>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms);
  };
  auto add_integrate = [&]() {
    const int T = h->fk_team; const dim3 gd = team_grid(T);
    return T == 16 ? graph_add_kernel(h, last, k_integrate_fk_team<16>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms)
         : T == 32 ? graph_add_kernel(h, last, k_integrate_fk_team<32>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms)
                   : graph_add_kernel(h, last, k_integrate_fk_team<64>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms);
  };
  auto add_integrate_dynamics = [&]() {
    const int T = h->dyn_team == 64 ? 64 : 32; const dim3 gd = team_grid(T);
    return T == 32 ? graph_add_kernel(h, last, k_integrate_fk_dynamics_team<32>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms)
                   : graph_add_kernel(h, last, k_integrate_fk_dynamics_team<64>, gd, b64, nullptr, nullptr, nullptr, h->P, h->dms);
  };
  if (!fuse_pre(h)) ok = add_dynamics();
  for (int i = 0; i < substeps && ok; ++i) {                               // same order as launch_substeps
    // heaviest-first records: substep i uses record i & 1 and clears the other one (an odd substep count would leave record 0 uncleared: no records then)
    const bool lpt_on = lpt_enabled(h) && (substeps % 2 == 0);
    int* lpt_cur = lpt_on ? h->lpt + (i & 1) * lpt_record_ints(h) : nullptr;
    int* lpt_next = lpt_on ? h->lpt + (1 - (i & 1)) * lpt_record_ints(h) : nullptr;
    const int epw_s = solver_epw(h);
    // The first collision pass reads nothing the pre-physics / dynamics launch writes (geom poses, sort buffers and the normal cache come from the end of the
    // previous step): it is a second ROOT of the graph, and the first solve joins the two.  The collision launch is as long as its slowest workgroup
    // (landing window: mean 42 us, span 66 us, one wave per SIMD with the whole register file); the dynamics wavefronts take the SIMDs its early finishers
    // leave.  Tried and measured slower: off unless GO2SIM_PAR_PRE=1 (go2sim::par_pre).
    hipGraphNode_t pre_node = nullptr;
    if (i == 0 && fuse_pre(h) && h->par_pre) { pre_node = last; last = nullptr; }
    { const int T = h->collide_team; const int epw_c = collide_epw(h); const dim3 gc((h->B + epw_c - 1) / epw_c);
      ok = T == 16 && epw_c == 2 ? graph_add_kernel(h, last, k_collide_team<16, 2>, gc, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s)
         : T == 16 && epw_c == 1 ? graph_add_kernel(h, last, k_collide_team<16, 1>, gc, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s)
         : T == 16 ? graph_add_kernel(h, last, k_collide_team<16>, gc, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s)
         : T == 32 ? graph_add_kernel(h, last, k_collide_team<32>, gc, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s)
                   : graph_add_kernel(h, last, k_collide_team<64>, gc, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->gjk_scratch, lpt_cur, h->lpt_cap, epw_s); }
    if (!ok) break;
    g.extra_dep = pre_node;                                            // the solve waits for the collision pass AND (first substep) for the dynamics
    if (h->hm.terrain_enabled) {
      if (h->terrain_solver_team == 32) ok = graph_add_kernel(h, last, k_constraint_solve_team<32, RL_TERRAIN>, dim3((h->B + 1) / 2), b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
      else ok = graph_add_kernel(h, last, k_constraint_solve_team<64, RL_TERRAIN>, dim3(h->B), b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
    } else {
      const int T = h->solver_team; const dim3 gs = team_grid(T);
      if (fuse_solve(h)) {                                               // solve + integrate / kinematics (+ next dynamics) in one launch
        ok = (i + 1 < substeps) ? graph_add_kernel(h, last, k_solve_integrate_team<32, RL, true>, gs, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap)
                                : graph_add_kernel(h, last, k_solve_integrate_team<32, RL, false>, gs, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
        if (!ok) break;
        continue;
      }
      ok = T == 16 ? graph_add_kernel(h, last, k_constraint_solve_team<16, RL>, gs, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap)
         : T == 32 ? graph_add_kernel(h, last, k_constraint_solve_team<32, RL>, gs, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap)
                   : graph_add_kernel(h, last, k_constraint_solve_team<64, RL>, gs, b64, nullptr, nullptr, nullptr, h->P, h->dm, h->dms, h->solver_ovf, (const int*)lpt_cur, lpt_next, h->lpt_cap);
    }
    if (!ok) break;
    if (i + 1 < substeps && h->fuse_fk_dyn) ok = add_integrate_dynamics();
    else { ok = add_integrate(); if (ok && i + 1 < substeps) ok = add_dynamics(); }
  }
  if (!ok) return false;
  ok = graph_add_kernel(h, last, k_env_post_a, ge, dim3(WG * POST_A_WAVES), &g.n_post_a, &g.p_post_a, &sl, h->P, h->dm, h->hcfg, h->dglob, h->dacc, h->seed, h->step_count);
  if (!ok) return false;
  g.a_pa_step = (uint32_t*)sl[6];
  if (h->hcfg.i[GO2SIM_IC_USE_TERRAIN]) {
    ok = graph_add_kernel(h, last, k_env_terrain_rows, dim3((h->B + 255) / 256), dim3(256), nullptr, nullptr, nullptr, h->P, h->dcfg, h->dglob, h->seed);
    if (!ok) return false;
  }
  ok = graph_add_kernel(h, last, k_env_post_b_team<16>, dim3((h->B + 3) / 4), dim3(128), &g.n_post_b, &g.p_post_b, &sl, h->P, h->dm, h->dms, h->dcfg, h->dglob, h->seed, h->step_count,
                        obs, priv, rew, reset, timeout);
  if (!ok) return false;
  g.a_pb_step = (uint32_t*)sl[6];
  g.a_obs = (float**)sl[7]; g.a_priv = (float**)sl[8]; g.a_rew = (float**)sl[9]; g.a_reset = (uint8_t**)sl[10]; g.a_timeout = (float**)sl[11];
  if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) return false;
  g.valid = true;
  return true;
}

extern "C" {

// releases everything a handle owns (hipFree(nullptr) is a no-op): shared by go2sim_destroy and the error paths of go2sim_create
static void handle_release(go2sim* h) {
  step_graph_destroy(h);
  if (h->ev_created) for (int i = 0; i < TIMING_RING; ++i) { (void)hipEventDestroy(h->ev0[i]); (void)hipEventDestroy(h->ev1[i]); }
  (void)hipFree(h->P.f); (void)hipFree(h->P.i); (void)hipFree(h->P.fa); (void)hipFree(h->P.ia); (void)hipFree(h->dm); (void)hipFree(h->dcfg);
  (void)hipFree(h->dglob); (void)hipFree(h->dacc); (void)hipFree(h->derr); (void)hipFree(h->solver_ovf); (void)hipFree(h->gjk_scratch); (void)hipFree(h->lpt);
  (void)hipFree(h->terrain_hf); (void)hipFree(h->terrain_cmax); (void)hipFree(h->dms); (void)hipFree(h->didx);
  if (h->herr_pinned) (void)hipHostFree(h->herr_pinned);
  if (h->ev_errno) (void)hipEventDestroy(h->ev_errno);
  delete h;
}

int go2sim_create(const void* blob, size_t nbytes, int n_envs, int device, uint64_t seed, go2sim_t** out) {
  if (!blob || !out || n_envs <= 0) return GO2SIM_E_BADARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "go2sim: no HIP device available (the product has no CPU fallback)\n"); return GO2SIM_E_NODEVICE; }
  if (device < 0 || device >= ndev) return GO2SIM_E_BADARG;
  go2sim* h = new (std::nothrow) go2sim();
  if (!h) return GO2SIM_E_NOMEM;
  if (!parse_model(blob, nbytes, h->hm)) { delete h; return GO2SIM_E_BADMODEL; }
  h->B = n_envs; h->device = device; h->seed = seed;
  // every failure below releases what was allocated so far (one exit path)
  int rc = GO2SIM_E_OK;
#define CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "go2sim: HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); rc = GO2SIM_E_HIP; goto fail; } } while (0)
  {
    CK(hipSetDevice(device));
    const size_t nf = (size_t)FO_TOTAL * n_envs, ni = (size_t)IO_TOTAL * n_envs;
    CK(hipMalloc((void**)&h->P.f, nf * sizeof(float)));
    CK(hipMalloc((void**)&h->P.i, ni * sizeof(int)));
    h->P.B = n_envs;
    CK(hipMemset(h->P.f, 0, nf * sizeof(float)));
    CK(hipMemset(h->P.i, 0, ni * sizeof(int)));
    CK(hipMalloc((void**)&h->P.fa, (size_t)ASTRIDE * n_envs * sizeof(float)));      // AoS records of the physics-internal arrays
    CK(hipMalloc((void**)&h->P.ia, (size_t)AISTRIDE * n_envs * sizeof(int)));
    CK(hipMemset(h->P.fa, 0, (size_t)ASTRIDE * n_envs * sizeof(float)));
    CK(hipMemset(h->P.ia, 0, (size_t)AISTRIDE * n_envs * sizeof(int)));
    CK(hipMalloc((void**)&h->dm, sizeof(Model)));
    CK(hipMemcpy(h->dm, &h->hm, sizeof(Model), hipMemcpyHostToDevice));
    {
      ModelS hs;
      if (!build_model_s(h->hm, hs)) { rc = GO2SIM_E_BADMODEL; goto fail; }
      CK(hipMalloc((void**)&h->dms, MODELS_LDS_BYTES));   // padded: the LDS DMA of the team kernels reads whole KiB
      CK(hipMemset(h->dms, 0, MODELS_LDS_BYTES));
      CK(hipMemcpy(h->dms, &hs, sizeof(ModelS), hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void**)&h->dcfg, sizeof(DCfg)));
    CK(hipMalloc((void**)&h->dglob, sizeof(Glob)));
    CK(hipMalloc((void**)&h->dacc, sizeof(Acc)));
    CK(hipMalloc((void**)&h->derr, 2 * sizeof(int)));
    CK(hipMemset(h->derr, 0, 2 * sizeof(int)));
    CK(hipHostMalloc((void**)&h->herr_pinned, 2 * sizeof(int), hipHostMallocDefault));
    h->herr_pinned[0] = h->herr_pinned[1] = 0;
    CK(hipEventCreateWithFlags(&h->ev_errno, hipEventDisableTiming));
    CK(hipMalloc((void**)&h->solver_ovf, (size_t)n_envs * sizeof(SolverData<MAXR>)));
    if (const char* t = getenv("GO2SIM_DYN_TEAM")) { int v = atoi(t); if (v == 16 || v == 32 || v == 64) h->dyn_team = v; }
    if (const char* t = getenv("GO2SIM_NO_GRAPH")) { if (atoi(t) != 0) h->use_graph = false; }
    if (const char* t = getenv("GO2SIM_NO_FUSE")) { if (atoi(t) != 0) h->fuse_fk_dyn = false; }
    if (const char* t = getenv("GO2SIM_PAR_PRE")) { if (atoi(t) != 0) h->par_pre = true; }
    if (const char* t = getenv("GO2SIM_NO_FUSE_SOLVE")) { if (atoi(t) != 0) h->fuse_solve_int = false; }
    if (const char* t = getenv("GO2SIM_FK_TEAM")) { int v = atoi(t); if (v == 16 || v == 32 || v == 64) h->fk_team = v; }
    if (const char* t = getenv("GO2SIM_COLLIDE_TEAM")) { int v = atoi(t); if (v == 16 || v == 32 || v == 64) h->collide_team = v; }
    if (const char* t = getenv("GO2SIM_COLLIDE_EPW")) { int v = atoi(t); if (v == 1 || v == 2) h->collide_epw = v; }
    CK(hipMalloc((void**)&h->gjk_scratch, (size_t)n_envs * h->collide_team * sizeof(GjkStoreFull)));   // ~20 KB per narrow-phase lane
    if (const char* t = getenv("GO2SIM_SOLVER_TEAM")) { int v = atoi(t); if (v == 16 || v == 32 || v == 64) h->solver_team = v; }
    if (const char* t = getenv("GO2SIM_TERRAIN_SOLVER_TEAM")) { int v = atoi(t); if (v == 32 || v == 64) h->terrain_solver_team = v; }
    if (const char* t = getenv("GO2SIM_NO_LPT")) { if (atoi(t) != 0) h->use_lpt = false; }
    if (const char* t = getenv("GO2SIM_LPT_FLAT")) { h->lpt_flat = atoi(t) != 0; }
    h->lpt_cap = n_envs / 8 + 72;
    CK(hipMalloc((void**)&h->lpt, 2 * lpt_record_ints(h) * sizeof(int)));
    CK(hipMemset(h->lpt, 0, 2 * lpt_record_ints(h) * sizeof(int)));
    Glob g0; memset(&g0, 0, sizeof(g0)); g0.friction = 1.0f;
    CK(hipMemcpy(h->dglob, &g0, sizeof(Glob), hipMemcpyHostToDevice));
    CK(hipMemset(h->dacc, 0, sizeof(Acc)));
    memset(&h->hcfg, 0, sizeof(DCfg));
    hipLaunchKernelGGL(k_init_state, grid_for(n_envs), dim3(WG), 0, 0, h->P, h->dm, 0);
    launch_fk_team(h, 0, 1, nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
  }
#undef CK
  *out = h;
  return GO2SIM_E_OK;
fail:
  handle_release(h);
  return rc;
}

int go2sim_destroy(go2sim_t* h) {
  if (!h) return GO2SIM_E_BADARG;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  handle_release(h);
  return GO2SIM_E_OK;
}
int go2sim_n_envs(const go2sim_t* h) { return h ? h->B : GO2SIM_E_BADARG; }

int go2sim_scene_reset(go2sim_t* h, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t(h, s, T_MISC);
  hipLaunchKernelGGL(k_scene_reset_clear, grid_for(h->B), dim3(WG), 0, s, h->P);
  hipLaunchKernelGGL(k_init_state, grid_for(h->B), dim3(WG), 0, s, h->P, h->dm, 1);
  launch_fk_team(h, s, 1, nullptr);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_substep(go2sim_t* h, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  launch_substep(h, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_scene_step(go2sim_t* h, int substeps, void* stream) {
  if (!h || substeps <= 0) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  launch_substeps(h, s, substeps);
  { ScopedTimer t(h, s, T_MISC); hipLaunchKernelGGL(k_clear_ext, grid_for(h->B), dim3(WG), 0, s, h->P); }
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_forward_kinematics(go2sim_t* h, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t(h, s, T_MISC);
  launch_fk_team(h, s, 1, nullptr);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}

// field -> (components, int?, offset, AoS?)
static int field_lookup(int field, int* k, int* is_int, int* off, int* is_aos = nullptr) {
  int kk = -1, ii = 0, oo = 0, aa = 0;
  switch (field) {
    case GO2SIM_F_QPOS: kk = NQ; oo = FO(qpos); break;
    case GO2SIM_F_VEL: kk = ND; oo = FO(vel); break;
    case GO2SIM_F_ACC: kk = ND; oo = AO(acc); aa = 1; break;
    case GO2SIM_F_QACC_WS: kk = ND; oo = AO(qacc_ws); aa = 1; break;
    case GO2SIM_F_CTRL_FORCE: kk = ND; oo = FO(ctrl_force); break;
    case GO2SIM_F_EXT_FORCE: kk = NL * 6; oo = FO(ext); break;
    case GO2SIM_F_MASS_SHIFT: kk = NL; oo = FO(mass_shift); break;
    case GO2SIM_F_COM_SHIFT: kk = NL * 3; oo = FO(com_shift); break;
    case GO2SIM_F_FRICTION_RATIO: kk = NG; oo = FO(friction_ratio); break;
    case GO2SIM_F_LINK_POS: kk = NL * 3; oo = FO(l_pos); break;
    case GO2SIM_F_LINK_QUAT: kk = NL * 4; oo = FO(l_quat); break;
    case GO2SIM_F_LINK_CDVEL: kk = NL * 3; oo = FO(cd_vel); break;
    case GO2SIM_F_LINK_CDANG: kk = NL * 3; oo = FO(cd_ang); break;
    case GO2SIM_F_ROOT_COM: kk = 3; oo = FO(root_com) + 3; break;
    case GO2SIM_F_CONTACT_FORCE: kk = NL * 3; oo = FO(contact_force); break;
    case GO2SIM_F_MASS_MAT: kk = ND * ND; oo = AO(mass_mat); aa = 1; break;
    case GO2SIM_F_FORCE: kk = ND; oo = AO(qf_smooth); aa = 1; break;
    case GO2SIM_F_ACC_SMOOTH: kk = ND; oo = AO(acc_smooth); aa = 1; break;
    case GO2SIM_F_CONTACT_POS: kk = MAXC * 3; oo = AO(c_pos); aa = 1; break;
    case GO2SIM_F_CONTACT_NORMAL: kk = MAXC * 3; oo = AO(c_normal); aa = 1; break;
    case GO2SIM_F_CONTACT_PEN: kk = MAXC; oo = AO(c_pen); aa = 1; break;
    case GO2SIM_F_NORMAL_CACHE: kk = NPAIR * 3; oo = AO(normal_cache); aa = 1; break;   // masked by ncache_valid: go2sim_get_field / go2sim_set_field
    case GO2SIM_F_SORT_VALUE: kk = 2 * NG; oo = AO(sort_value); aa = 1; break;
    case GO2SIM_F_GEOM_FRICTION: kk = NG; oo = FO(geom_friction); break;
    case GO2SIM_F_EFC_FORCE: kk = MAXR; oo = AO(efc_force); aa = 1; break;
    case GO2SIM_F_QFRC_CONSTRAINT: kk = ND; oo = AO(qfrc_constraint); aa = 1; break;
    case GO2SIM_F_CTRL_POS: kk = ND; oo = FO(ctrl_pos); break;
    case GO2SIM_F_CTRL_VEL: kk = ND; oo = FO(ctrl_vel); break;
    case GO2SIM_F_DOF_POS: kk = ND; oo = FO(dof_pos); break;
    case GO2SIM_I_N_CONTACTS: kk = 1; ii = 1; oo = IO(n_contacts); break;
    case GO2SIM_I_CONTACT_GEOMS: kk = 2 * MAXC; ii = 1; oo = AIO(c_geom); aa = 1; break;
    case GO2SIM_I_N_CONSTRAINTS: kk = 1; ii = 1; oo = IO(n_con); break;
    case GO2SIM_I_ERRNO: kk = 1; ii = 1; oo = IO(err); break;
    case GO2SIM_I_IS_WARMSTART: kk = 1; ii = 1; oo = IO(is_warmstart); break;
    case GO2SIM_I_FIRST_TIME: kk = 1; ii = 1; oo = IO(first_time); break;
    case GO2SIM_I_SORT_IG: kk = 2 * NG; ii = 1; oo = AIO(sort_ig); aa = 1; break;
    case GO2SIM_I_N_BROAD: kk = 1; ii = 1; oo = IO(n_broad); break;
    case GO2SIM_I_SOLVER_ITERS: kk = 1; ii = 1; oo = IO(solver_iters); break;
    case GO2SIM_I_CTRL_MODE: kk = ND; ii = 1; oo = IO(ctrl_mode); break;
    default: return GO2SIM_E_BADARG;
  }
  if (k) *k = kk;
  if (is_int) *is_int = ii;
  if (off) *off = oo;
  if (is_aos) *is_aos = aa;
  return GO2SIM_E_OK;
}
int go2sim_field_size(int field, int* k, int* is_int) { return field_lookup(field, k, is_int, nullptr); }
// zero-copy view: only the fields that live in the SoA pool ([k][n_envs]); the physics-internal arrays are kept as per-env records
// (use go2sim_get_field / go2sim_set_field for those)
int go2sim_field_ptr(go2sim_t* h, int field, void** ptr_out) {
  int k, ii, off, aa;
  if (!h || !ptr_out || field_lookup(field, &k, &ii, &off, &aa) || aa) return GO2SIM_E_BADARG;
  *ptr_out = ii ? (void*)(h->P.i + (size_t)off * h->B) : (void*)(h->P.f + (size_t)off * h->B);
  return GO2SIM_E_OK;
}
int go2sim_get_field(go2sim_t* h, int field, void* dst, void* stream) {
  int k, ii, off, aa;
  if (!h || !dst || field_lookup(field, &k, &ii, &off, &aa)) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (aa) {
    const int* rec = ii ? h->P.ia : (const int*)h->P.fa;
    if (field == GO2SIM_F_MASS_MAT) hipLaunchKernelGGL(k_mass_mat_to_rows, dim3((k * h->B + 255) / 256), dim3(256), 0, s, (const float*)h->P.fa, ASTRIDE, off, h->B, (float*)dst);
    else hipLaunchKernelGGL(k_aos_to_rows, dim3((k * h->B + 255) / 256), dim3(256), 0, s, rec, ii ? AISTRIDE : ASTRIDE, off, k, h->B, (int*)dst);
    if (field == GO2SIM_F_NORMAL_CACHE) hipLaunchKernelGGL(k_ncache_mask_rows, dim3((NPAIR * h->B + 255) / 256), dim3(256), 0, s, h->P, (float*)dst);
    HIPCHK(hipGetLastError());
  } else {
    void* p = ii ? (void*)(h->P.i + (size_t)off * h->B) : (void*)(h->P.f + (size_t)off * h->B);
    HIPCHK(hipMemcpyAsync(dst, p, (size_t)k * h->B * 4, hipMemcpyDeviceToDevice, s));
  }
  return GO2SIM_E_OK;
}
int go2sim_set_field(go2sim_t* h, int field, const void* src, void* stream) {
  int k, ii, off, aa;
  if (!h || !src || field_lookup(field, &k, &ii, &off, &aa)) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (aa) {
    int* rec = ii ? h->P.ia : (int*)h->P.fa;
    if (field == GO2SIM_F_MASS_MAT) hipLaunchKernelGGL(k_rows_to_mass_mat, dim3((NTRI * h->B + 255) / 256), dim3(256), 0, s, (const float*)src, h->P.fa, ASTRIDE, off, h->B);
    else hipLaunchKernelGGL(k_rows_to_aos, dim3((k * h->B + 255) / 256), dim3(256), 0, s, (const int*)src, rec, ii ? AISTRIDE : ASTRIDE, off, k, h->B);
    if (field == GO2SIM_F_NORMAL_CACHE) hipLaunchKernelGGL(k_ncache_set_valid, dim3((NCV * h->B + 255) / 256), dim3(256), 0, s, h->P);
    HIPCHK(hipGetLastError());
  } else {
    void* p = ii ? (void*)(h->P.i + (size_t)off * h->B) : (void*)(h->P.f + (size_t)off * h->B);
    HIPCHK(hipMemcpyAsync(p, src, (size_t)k * h->B * 4, hipMemcpyDeviceToDevice, s));
  }
  return GO2SIM_E_OK;
}
int go2sim_reset_caches(go2sim_t* h, const int* envs_idx, int n_sel, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  int n = envs_idx ? n_sel : h->B;
  if (n <= 0) return GO2SIM_E_OK;
  hipLaunchKernelGGL(k_reset_caches, grid_for(n), dim3(WG), 0, (hipStream_t)stream, h->P, envs_idx, n);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_set_friction(go2sim_t* h, float mu, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_set_friction, grid_for(h->B), dim3(WG), 0, s, h->P, mu);
  HIPCHK(hipMemcpyAsync((char*)h->dglob + offsetof(Glob, friction), &mu, sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // `mu` lives on the caller's stack
  return GO2SIM_E_OK;
}
__global__ void k_set_link0_pose(Pool P, V3 pos) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  E e(P, b);
  e.l_pos()[0] = pos; e.l_quat()[0] = qident();
  e.first_time()[0] = 1; e.is_warmstart()[0] = 0;
  auto ncv = e.ncache_valid();
  for (int p = 0; p < NCV; ++p) ncv[p] = 0;
}
int go2sim_set_terrain(go2sim_t* h, const int16_t* hf, int rows, int cols, float horizontal_scale, float vertical_scale, const float* origin, void* stream) {
  if (!h || !hf || rows < 2 || cols < 2 || !origin || !(horizontal_scale > 0.0f)) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipStreamSynchronize(s));
  step_graph_destroy(h);   // the solver kernel of the step graph depends on the terrain flag
  Model& m = h->hm;
  std::vector<float> hfm((size_t)rows * cols);
  float hmax = -1e30f, hmin = 1e30f;
  for (size_t k = 0; k < hfm.size(); ++k) { float v = (float)hf[k] * vertical_scale; hfm[k] = v; hmax = (hmax < v) ? v : hmax; hmin = (v < hmin) ? v : hmin; }
  if (h->terrain_hf) (void)hipFree(h->terrain_hf);
  HIPCHK(hipMalloc((void**)&h->terrain_hf, hfm.size() * sizeof(float)));
  HIPCHK(hipMemcpy(h->terrain_hf, hfm.data(), hfm.size() * sizeof(float), hipMemcpyHostToDevice));
  {                                                                     // coarse maximum map (terrain_pair_out_of_reach)
    const int cr = (rows + TERRAIN_CB - 1) / TERRAIN_CB, cc = (cols + TERRAIN_CB - 1) / TERRAIN_CB;
    std::vector<float> cm((size_t)cr * cc, -1e30f);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) { float& d = cm[(size_t)(r / TERRAIN_CB) * cc + c / TERRAIN_CB]; const float v = hfm[(size_t)r * cols + c]; d = d < v ? v : d; }
    if (h->terrain_cmax) (void)hipFree(h->terrain_cmax);
    HIPCHK(hipMalloc((void**)&h->terrain_cmax, cm.size() * sizeof(float)));
    HIPCHK(hipMemcpy(h->terrain_cmax, cm.data(), cm.size() * sizeof(float), hipMemcpyHostToDevice));
    m.terrain_cmax = h->terrain_cmax; m.terrain_crows = cr; m.terrain_ccols = cc;
  }
  m.terrain_enabled = 1; m.terrain_rows = rows; m.terrain_cols = cols; m.terrain_hs = horizontal_scale; m.terrain_hf = h->terrain_hf;
  m.terrain_xyz_maxmin[0] = (float)rows * horizontal_scale; m.terrain_xyz_maxmin[1] = (float)cols * horizontal_scale; m.terrain_xyz_maxmin[2] = hmax;
  m.terrain_xyz_maxmin[3] = 0.0f; m.terrain_xyz_maxmin[4] = 0.0f; m.terrain_xyz_maxmin[5] = hmin - 1.0f;
  Geom& G = m.geoms[0];
  G.type = GEOM_TERRAIN; G.pos = v3h(0, 0, 0); G.quat = {1.0f, 0.0f, 0.0f, 0.0f}; G.center = v3h(0, 0, 0);
  float x1 = (float)(rows - 1) * horizontal_scale, y1 = (float)(cols - 1) * horizontal_scale, z0 = hmin - 1.0f, z1 = hmax;
  for (int c = 0; c < 8; ++c) G.aabb[c] = v3h((c & 4) ? x1 : 0.0f, (c & 2) ? y1 : 0.0f, (c & 1) ? z1 : z0);
  m.links[0].pos = v3h(origin[0], origin[1], origin[2]); m.links[0].quat = {1.0f, 0.0f, 0.0f, 0.0f};
  HIPCHK(hipMemcpy(h->dm, &h->hm, sizeof(Model), hipMemcpyHostToDevice));
  { ModelS hs; if (!build_model_s(h->hm, hs)) return GO2SIM_E_BADMODEL; HIPCHK(hipMemcpy(h->dms, &hs, sizeof(ModelS), hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(k_set_link0_pose, dim3((h->B + 255) / 256), dim3(256), 0, s, h->P, m.links[0].pos);
  launch_fk_team(h, s, 1, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  return GO2SIM_E_OK;
}
int go2sim_set_dof_gains(go2sim_t* h, int d, float kp, float kv, float flo, float fhi) {
  if (!h || d < 0 || d >= ND) return GO2SIM_E_BADARG;
  h->hm.dofs[d].kp = kp; h->hm.dofs[d].kv = kv; h->hm.dofs[d].force_range[0] = flo; h->hm.dofs[d].force_range[1] = fhi;
  HIPCHK(hipMemcpy(&h->dm->dofs[d], &h->hm.dofs[d], sizeof(Dof), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(&h->dms->dofs[d], &h->hm.dofs[d], sizeof(Dof), hipMemcpyHostToDevice));
  return GO2SIM_E_OK;
}
int go2sim_check_errno(go2sim_t* h, int* out, void* stream) {
  if (!h || !out) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->derr, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_errno_reduce, dim3(64), dim3(256), 0, s, h->P, h->derr);
  HIPCHK(hipMemcpyAsync(out, h->derr, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return GO2SIM_E_OK;
}

int go2sim_env_configure(go2sim_t* h, const double* f, int nf, const int* i, int ni) {
  if (!h || !f || !i || nf != GO2SIM_FC_COUNT || ni != GO2SIM_IC_COUNT) return GO2SIM_E_BADARG;
  {                                                                       // validate a local copy: a rejected configuration leaves the handle as it was
    DCfg c;
    for (int k = 0; k < GO2SIM_FC_N_HOST; ++k) c.d[k] = f[k];
    for (int k = 0; k < nf; ++k) c.f[k] = (float)f[k];
    memcpy(c.i, i, sizeof(int) * ni);
    if (c.i[GO2SIM_IC_NUM_ACTIONS] > NA || c.i[GO2SIM_IC_NUM_ACTIONS] < NM || c.i[GO2SIM_IC_NUM_OBS] > NOBS_MAX || c.i[GO2SIM_IC_NUM_PRIV_OBS] > NPRIV_MAX ||
        c.i[GO2SIM_IC_N_REWARDS] > NREW || c.i[GO2SIM_IC_N_REWARDS] < 0 || c.i[GO2SIM_IC_MAX_DELAY] >= GO2SIM_ACTION_RING_MAX || c.i[GO2SIM_IC_MAX_DELAY] < 0 || c.i[GO2SIM_IC_NUM_OBS] < 33 + c.i[GO2SIM_IC_NUM_ACTIONS] ||
        c.i[GO2SIM_IC_SUBSTEPS] < 1 || c.i[GO2SIM_IC_SUBSTEPS] > 16 || c.i[GO2SIM_IC_RESAMPLE_STEPS] < 1 || c.i[GO2SIM_IC_N_TERRAIN_ROWS] > 16 || c.i[GO2SIM_IC_SCAN_N] > 80 ||
        c.i[GO2SIM_IC_SCAN_N] < 0)
      return GO2SIM_E_BADARG;
    for (int k = 0; k < NM; ++k) { int d = c.i[GO2SIM_IC_MOTOR_DOF0 + k]; if (d < 6 || d >= ND) return GO2SIM_E_BADARG; }
    for (int k = 0; k < 4; ++k) { int l = c.i[GO2SIM_IC_FOOT_LINK0 + k], l2 = c.i[GO2SIM_IC_HIP_LINK0 + k]; if (l < 0 || l >= NL || l2 < 0 || l2 >= NL) return GO2SIM_E_BADARG; }
    if (c.i[GO2SIM_IC_PUSH_LINK] < 0 || c.i[GO2SIM_IC_PUSH_LINK] >= NL || c.i[GO2SIM_IC_BASE_LINK] < 0 || c.i[GO2SIM_IC_BASE_LINK] >= NL) return GO2SIM_E_BADARG;
    for (int k = 0; k < c.i[GO2SIM_IC_N_REWARDS]; ++k) { int id = c.i[GO2SIM_IC_REWARD_ID0 + k]; if (id < 0 || id >= GO2SIM_R_COUNT) return GO2SIM_E_BADARG; }
    step_graph_destroy(h);   // the configuration is baked into the kernel arguments of the step graph
    h->hcfg = c;
  }
  const DCfg& c = h->hcfg;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpy(h->dcfg, &h->hcfg, sizeof(DCfg), hipMemcpyHostToDevice));
  Glob g; memset(&g, 0, sizeof(g));
  g.level = c.d[GO2SIM_FC_CURR_LEVEL_INIT]; g.friction = 1.0f;
  apply_curriculum_level(c, g);
  HIPCHK(hipMemcpy(h->dglob, &g, sizeof(Glob), hipMemcpyHostToDevice));
  // zero the env buffers
  size_t f0 = (size_t)FO(actions) * h->B, f1 = (size_t)FO_TOTAL * h->B;
  HIPCHK(hipMemset(h->P.f + f0, 0, (f1 - f0) * sizeof(float)));
  size_t i0 = (size_t)IO(delay_steps) * h->B, i1 = (size_t)IO_TOTAL * h->B;
  HIPCHK(hipMemset(h->P.i + i0, 0, (i1 - i0) * sizeof(int)));
  hipLaunchKernelGGL(k_env_init_buffers, grid_for(h->B), dim3(WG), 0, 0, h->P, h->dcfg);
  HIPCHK(hipGetLastError());
  for (int k = 0; k < NM; ++k) {
    int d = c.i[GO2SIM_IC_MOTOR_DOF0 + k];
    if (c.i[GO2SIM_IC_MANUAL_PD]) { h->hm.dofs[d].kp = 0.0f; h->hm.dofs[d].kv = 0.0f; } else { h->hm.dofs[d].kp = c.f[GO2SIM_FC_KP]; h->hm.dofs[d].kv = c.f[GO2SIM_FC_KD]; }
  }
  HIPCHK(hipMemcpy(h->dm, &h->hm, sizeof(Model), hipMemcpyHostToDevice));
  { ModelS hs; if (!build_model_s(h->hm, hs)) return GO2SIM_E_BADMODEL; HIPCHK(hipMemcpy(h->dms, &hs, sizeof(ModelS), hipMemcpyHostToDevice)); }
  HIPCHK(hipDeviceSynchronize());
  h->step_count = 0; h->action_write_idx = 0; h->cfg_set = true;
  return GO2SIM_E_OK;
}

int go2sim_env_step(go2sim_t* h, const float* actions, float* obs, float* priv, float* rew, uint8_t* reset, float* timeout, void* stream) {
  if (!h || !h->cfg_set || !actions) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 g = grid_for(h->B), b(WG);
  if (h->use_graph && !h->timing) {
    auto& sg = h->sg;
    if (!sg.valid) {
      if (!step_graph_build(h, actions, obs, priv, rew, reset, timeout)) {
        (void)hipGetLastError();
        fprintf(stderr, "go2sim: hipGraph build failed, falling back to plain kernel launches\n");
        step_graph_destroy(h); h->use_graph = false; h->graph_fallbacks += 1;
      }
    }
    if (h->use_graph) {
      *sg.a_actions = actions; *sg.a_pre_step = h->step_count; *sg.a_pre_widx = h->action_write_idx; *sg.a_pa_step = h->step_count; *sg.a_pb_step = h->step_count;
      *sg.a_obs = obs; *sg.a_priv = priv; *sg.a_rew = rew; *sg.a_reset = reset; *sg.a_timeout = timeout;
      const bool launched = hipGraphExecKernelNodeSetParams(sg.exec, sg.n_pre, &sg.p_pre) == hipSuccess &&
                            hipGraphExecKernelNodeSetParams(sg.exec, sg.n_post_a, &sg.p_post_a) == hipSuccess &&
                            hipGraphExecKernelNodeSetParams(sg.exec, sg.n_post_b, &sg.p_post_b) == hipSuccess && hipGraphLaunch(sg.exec, s) == hipSuccess;
      if (launched) {
        h->action_write_idx = (h->action_write_idx + 1) % (h->hcfg.i[GO2SIM_IC_MAX_DELAY] + 1);
        h->step_count += 1;
        h->lpt_parity = 0;                // (the graph leaves record 0 cleared)
        return GO2SIM_E_OK;
      }
      (void)hipGetLastError();            // nothing of this step was enqueued: drop the graph and continue with plain launches
      fprintf(stderr, "go2sim: hipGraph launch failed, falling back to plain kernel launches\n");
      step_graph_destroy(h); h->use_graph = false; h->graph_fallbacks += 1;
    }
  }
  if (h->timing && h->ev_n + 64 > TIMING_RING) timing_flush(h);   // all pending events belong to completed launches
  ScopedTimer total(h, s, T_TOTAL);
  if (!fuse_pre(h)) { ScopedTimer t(h, s, T_ENV_PRE); hipLaunchKernelGGL(k_env_pre, g, b, 0, s, h->P, h->dm, h->hcfg, h->dglob, actions, h->seed, h->step_count, h->action_write_idx); }
  launch_substeps(h, s, h->hcfg.i[GO2SIM_IC_SUBSTEPS], fuse_pre(h) ? actions : nullptr);
  {
    ScopedTimer t(h, s, T_ENV_POST);
    hipLaunchKernelGGL(k_env_post_a, g, dim3(WG * POST_A_WAVES), 0, s, h->P, h->dm, h->hcfg, h->dglob, h->dacc, h->seed, h->step_count);
    if (h->hcfg.i[GO2SIM_IC_USE_TERRAIN]) hipLaunchKernelGGL(k_env_terrain_rows, dim3((h->B + 255) / 256), dim3(256), 0, s, h->P, h->dcfg, h->dglob, h->seed);
    hipLaunchKernelGGL(k_env_post_b_team<16>, dim3((h->B + 3) / 4), dim3(128), 0, s, h->P, h->dm, h->dms, h->dcfg, h->dglob, h->seed, h->step_count, obs, priv, rew, reset, timeout);
  }
  HIPCHK(hipGetLastError());
  h->action_write_idx = (h->action_write_idx + 1) % (h->hcfg.i[GO2SIM_IC_MAX_DELAY] + 1);
  h->step_count += 1;
  return GO2SIM_E_OK;
}
int go2sim_env_reset(go2sim_t* h, void* stream) {
  if (!h || !h->cfg_set) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 g = grid_for(h->B), b(WG);
  ScopedTimer t(h, s, T_MISC);
  HIPCHK(hipMemsetAsync(h->dacc, 0, sizeof(Acc), s));
  hipLaunchKernelGGL(k_env_mark_all, g, b, 0, s, h->P, h->dcfg, h->dacc);
  hipLaunchKernelGGL(k_env_globals, dim3(1), dim3(1), 0, s, h->dcfg, h->dglob, h->dacc, h->seed, 0);
  if (h->hcfg.i[GO2SIM_IC_USE_TERRAIN]) hipLaunchKernelGGL(k_env_terrain_rows, dim3((h->B + 255) / 256), dim3(256), 0, s, h->P, h->dcfg, h->dglob, h->seed);
  hipLaunchKernelGGL(k_env_reset_tail, g, b, 0, s, h->P, h->dm, h->dcfg, h->dglob, h->seed);
  launch_fk_team(h, s, 1, &h->dglob->n_reset_now);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_reset_idx(go2sim_t* h, const int* envs_idx, int n_sel, void* stream) {
  if (!h || !h->cfg_set || (n_sel > 0 && !envs_idx) || n_sel < 0) return GO2SIM_E_BADARG;
  if (n_sel == 0) return GO2SIM_E_OK;                                  // `if len(envs_idx) == 0: return`, go2_env_walk.py:1157
  hipStream_t s = (hipStream_t)stream;
  dim3 g = grid_for(h->B), b(WG);
  ScopedTimer t(h, s, T_MISC);
  HIPCHK(hipMemsetAsync(h->dacc, 0, sizeof(Acc), s));
  hipLaunchKernelGGL(k_env_unmark_all, g, b, 0, s, h->P);
  hipLaunchKernelGGL(k_env_mark_idx, grid_for(n_sel), b, 0, s, h->P, h->dcfg, h->dacc, envs_idx, n_sel);
  hipLaunchKernelGGL(k_env_globals, dim3(1), dim3(1), 0, s, h->dcfg, h->dglob, h->dacc, h->seed, 0);
  if (h->hcfg.i[GO2SIM_IC_USE_TERRAIN]) hipLaunchKernelGGL(k_env_terrain_rows, dim3((h->B + 255) / 256), dim3(256), 0, s, h->P, h->dcfg, h->dglob, h->seed);
  hipLaunchKernelGGL(k_env_reset_tail, g, b, 0, s, h->P, h->dm, h->dcfg, h->dglob, h->seed);
  launch_fk_team(h, s, 1, &h->dglob->n_reset_now);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_respawn(go2sim_t* h, const int* envs_idx, int n_sel, const float* pos, const float* quat, int clear_buffers, void* stream) {
  if (!h || !h->cfg_set || n_sel < 0 || (n_sel > 0 && (!envs_idx || !pos))) return GO2SIM_E_BADARG;
  if (n_sel == 0) return GO2SIM_E_OK;
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t(h, s, T_MISC);
  hipLaunchKernelGGL(k_env_respawn, grid_for(n_sel), dim3(WG), 0, s, h->P, h->dm, h->dcfg, envs_idx, n_sel, pos, quat, clear_buffers);
  launch_fk_team(h, s, 1, nullptr);                                      // set_pos / set_quat re-run the full-batch FK (rigid_solver.py:1928-1943)
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_lock_terrain_rows(go2sim_t* h, int lock, void* stream) {
  if (!h || !h->cfg_set) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  int v = lock != 0;
  HIPCHK(hipMemcpyAsync((char*)h->dglob + offsetof(Glob, lock_terrain_rows), &v, sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));                                      // `v` lives on this stack frame
  return GO2SIM_E_OK;
}
int go2sim_env_set_terrain_rows(go2sim_t* h, const int* rows, void* stream) {
  if (!h || !h->cfg_set || !rows) return GO2SIM_E_BADARG;
  int n_rows = h->hcfg.i[GO2SIM_IC_N_TERRAIN_ROWS];
  hipLaunchKernelGGL(k_env_set_terrain_rows, grid_for(h->B), dim3(WG), 0, (hipStream_t)stream, h->P, rows, n_rows < 1 ? 1 : n_rows);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
// Asynchronous form of RigidSolver.check_errno (rigid_solver.py:1189-1213; polled every 10 substeps by Simulator.step, simulator.py:267): _begin
// enqueues the OR-reduction and a copy into pinned host memory and returns at once; _result reports it without blocking once the copy has landed.
int go2sim_errno_poll_begin(go2sim_t* h, void* stream) {
  if (!h) return GO2SIM_E_BADARG;
  if (h->errno_poll_pending) return GO2SIM_E_OK;                        // one poll in flight at a time
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->derr + 1, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_errno_reduce, dim3(64), dim3(256), 0, s, h->P, h->derr + 1);
  HIPCHK(hipMemcpyAsync(h->herr_pinned, h->derr + 1, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(h->ev_errno, s));
  h->errno_poll_pending = true;
  return GO2SIM_E_OK;
}
int go2sim_errno_poll_result(go2sim_t* h, int* errno_host, int* ready) {
  if (!h || !errno_host || !ready) return GO2SIM_E_BADARG;
  *ready = 0; *errno_host = 0;
  if (!h->errno_poll_pending) return GO2SIM_E_OK;
  hipError_t q = hipEventQuery(h->ev_errno);
  if (q == hipErrorNotReady) return GO2SIM_E_OK;
  HIPCHK(q);
  h->errno_poll_pending = false;
  *ready = 1; *errno_host = h->herr_pinned[0];
  return GO2SIM_E_OK;
}
int go2sim_errno_poll_wait(go2sim_t* h, int* errno_host) {
  if (!h || !errno_host) return GO2SIM_E_BADARG;
  *errno_host = 0;
  if (!h->errno_poll_pending) return GO2SIM_E_OK;
  HIPCHK(hipEventSynchronize(h->ev_errno));
  h->errno_poll_pending = false;
  *errno_host = h->herr_pinned[0];
  return GO2SIM_E_OK;
}
int go2sim_graph_status(go2sim_t* h, int* using_graph, int* n_fallbacks) {
  if (!h) return GO2SIM_E_BADARG;
  if (using_graph) *using_graph = (h->use_graph && !h->timing) ? 1 : 0;
  if (n_fallbacks) *n_fallbacks = h->graph_fallbacks;
  return GO2SIM_E_OK;
}
int go2sim_env_get(go2sim_t* h, int buf, void* dst, void* stream) {
  if (!h || !dst) return GO2SIM_E_BADARG;
  int k = 0; const void* src = nullptr;
  const float* F = h->P.f; const int* I = h->P.i; size_t B = h->B;
  switch (buf) {
    case GO2SIM_EB_COMMANDS: k = 3; src = F + FO(commands) * B; break;
    case GO2SIM_EB_EPISODE_LENGTH: k = 1; src = I + IO(episode_length) * B; break;
    case GO2SIM_EB_BASE_LIN_VEL: k = 3; src = F + FO(base_lin_vel) * B; break;
    case GO2SIM_EB_BASE_ANG_VEL: k = 3; src = F + FO(base_ang_vel) * B; break;
    case GO2SIM_EB_PROJECTED_GRAVITY: k = 3; src = F + FO(projected_gravity) * B; break;
    case GO2SIM_EB_DOF_POS: k = 12; src = F + FO(e_dof_pos) * B; break;
    case GO2SIM_EB_DOF_VEL: k = 12; src = F + FO(e_dof_vel) * B; break;
    case GO2SIM_EB_BASE_POS: k = 3; src = F + FO(base_pos) * B; break;
    case GO2SIM_EB_BASE_QUAT: k = 4; src = F + FO(base_quat) * B; break;
    case GO2SIM_EB_BASE_EULER: k = 3; src = F + FO(base_euler) * B; break;
    case GO2SIM_EB_EPISODE_SUMS: k = NREW; src = F + FO(episode_sums) * B; break;
    case GO2SIM_EB_FOOT_CONTACT: k = 4; src = I + IO(foot_contact) * B; break;
    case GO2SIM_EB_FEET_AIR_TIME: k = 4; src = F + FO(feet_air_time) * B; break;
    case GO2SIM_EB_REW_TERMS: k = NREW; src = F + FO(rew_terms) * B; break;
    case GO2SIM_EB_TORQUE: k = 12; src = F + FO(torque) * B; break;
    case GO2SIM_EB_TERRAIN_ROW: k = 1; src = I + IO(terrain_row) * B; break;
    default: return GO2SIM_E_BADARG;
  }
  int n = k * h->B;
  hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, k, h->B);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_set_episode_length(go2sim_t* h, const int* ep, void* stream) {
  if (!h || !ep) return GO2SIM_E_BADARG;
  HIPCHK(hipMemcpyAsync(h->P.i + (size_t)IO(episode_length) * h->B, ep, (size_t)h->B * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GO2SIM_E_OK;
}
int go2sim_env_set_commands(go2sim_t* h, const float* cmd, void* stream) {
  if (!h || !cmd) return GO2SIM_E_BADARG;
  int n = 3 * h->B;
  hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const void*)cmd, (void*)(h->P.f + (size_t)FO(commands) * h->B), 3, h->B);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_globals(go2sim_t* h, go2sim_env_globals_t* out, void* stream) {
  if (!h || !out) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(out, h->dglob, sizeof(Glob), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  out->step_count = h->step_count; out->action_write_idx = h->action_write_idx;
  out->terrain_mean_row = out->last_reset_count > 0 ? (float)((double)out->terrain_row_sum / (double)out->last_reset_count) : 0.0f;
  return GO2SIM_E_OK;
}
int go2sim_env_globals_ptr(go2sim_t* h, void** ptr_out) {
  if (!h || !ptr_out) return GO2SIM_E_BADARG;
  *ptr_out = h->dglob;
  return GO2SIM_E_OK;
}
int go2sim_env_set_level(go2sim_t* h, double level, void* stream) {
  if (!h || !h->cfg_set) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  Glob g;
  HIPCHK(hipMemcpyAsync(&g, h->dglob, sizeof(Glob), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  g.level = level;
  apply_curriculum_level(h->hcfg, g);
  HIPCHK(hipMemcpyAsync(h->dglob, &g, sizeof(Glob), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return GO2SIM_E_OK;
}
// ---- one batch sharded over several handles (GO2SIM_IC_SHARED_GLOBALS), include/go2sim.h ----
__global__ __launch_bounds__(WG) void k_env_apply_global_dr(Pool P, const DCfg* __restrict__ cp, const Glob* __restrict__ gp) {
  int b = blockIdx.x * WG + threadIdx.x;
  if (b >= P.B) return;
  const DCfg& c = *cp; const Glob& g = *gp;
  E e(P, b);
  const bool per_env = c.i[GO2SIM_IC_PER_ENV_GLOBAL_DR] != 0;
  if (c.i[GO2SIM_IC_HAS_FRICTION_DR] && !per_env) { auto gf = e.geom_friction(); for (int i = 0; i < NG; ++i) gf[i] = g.friction; }
  int bl = c.i[GO2SIM_IC_BASE_LINK];
  if (c.i[GO2SIM_IC_HAS_MASS_DR] && !per_env) e.mass_shift()[bl] = g.mass_shift;
  if (c.i[GO2SIM_IC_HAS_COM_DR]) e.com_shift()[bl] = v3(g.com_shift[0], g.com_shift[1], g.com_shift[2]);
  if (c.i[GO2SIM_IC_HAS_LEGM_DR]) for (int k = 0; k < 4; ++k) e.mass_shift()[c.i[GO2SIM_IC_HIP_LINK0 + k]] = g.leg_mass_shift[k];
}
static int glob_download(go2sim_t* h, Glob& g, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(&g, h->dglob, sizeof(Glob), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return GO2SIM_E_OK;
}
static int glob_upload(go2sim_t* h, const Glob& g, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(h->dglob, &g, sizeof(Glob), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return GO2SIM_E_OK;
}
int go2sim_env_sync_counters(go2sim_t* h, double* out5, void* stream) {
  if (!h || !h->cfg_set || !out5) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  Glob g; int rc = glob_download(h, g, s); if (rc) return rc;
  for (int k = 0; k < 5; ++k) { out5[k] = g.shard_counters[k]; g.shard_counters[k] = 0.0; }
  return glob_upload(h, g, s);
}
int go2sim_env_sync_apply(go2sim_t* h, const double* s5, double* dr_out10, void* stream) {
  if (!h || !h->cfg_set || !s5) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  Glob g; int rc = glob_download(h, g, s); if (rc) return rc;
  double dr[10];
  sync_apply_body(h->hcfg, g, h->seed, s5, dr);
  if (dr_out10) for (int k = 0; k < 10; ++k) dr_out10[k] = dr[k];
  return glob_upload(h, g, s);
}
int go2sim_env_set_global_dr(go2sim_t* h, const double* dr10, void* stream) {
  if (!h || !h->cfg_set || !dr10) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  Glob g; int rc = glob_download(h, g, s); if (rc) return rc;
  set_global_dr_body(g, dr10);
  rc = glob_upload(h, g, s); if (rc) return rc;
  hipLaunchKernelGGL(k_env_apply_global_dr, grid_for(h->B), dim3(WG), 0, s, h->P, h->dcfg, h->dglob);
  launch_fk_team(h, s, 1, nullptr);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
// the same three steps on device arrays: three single-thread kernels on the caller's stream, no host round trip (the RCCL path of distributed.sync_env_globals)
int go2sim_env_sync_counters_dev(go2sim_t* h, double* out5_dev, void* stream) {
  if (!h || !h->cfg_set || !out5_dev) return GO2SIM_E_BADARG;
  hipLaunchKernelGGL(k_env_sync_counters, dim3(1), dim3(1), 0, (hipStream_t)stream, h->dglob, out5_dev);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_sync_apply_dev(go2sim_t* h, const double* s5_dev, double* dr_out10_dev, void* stream) {
  if (!h || !h->cfg_set || !s5_dev || !dr_out10_dev) return GO2SIM_E_BADARG;
  hipLaunchKernelGGL(k_env_sync_apply, dim3(1), dim3(1), 0, (hipStream_t)stream, h->dcfg, h->dglob, h->seed, s5_dev, dr_out10_dev);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_env_set_global_dr_dev(go2sim_t* h, const double* dr10_dev, void* stream) {
  if (!h || !h->cfg_set || !dr10_dev) return GO2SIM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_env_set_global_dr, dim3(1), dim3(1), 0, s, h->dglob, dr10_dev);
  hipLaunchKernelGGL(k_env_apply_global_dr, grid_for(h->B), dim3(WG), 0, s, h->P, h->dcfg, h->dglob);
  launch_fk_team(h, s, 1, nullptr);
  HIPCHK(hipGetLastError());
  return GO2SIM_E_OK;
}
int go2sim_enable_timing(go2sim_t* h, int enable) {
  if (!h) return GO2SIM_E_BADARG;
  if (enable && !h->ev_created) {
    for (int i = 0; i < TIMING_RING; ++i) { HIPCHK(hipEventCreate(&h->ev0[i])); HIPCHK(hipEventCreate(&h->ev1[i])); }
    h->ev_created = true;
  }
  if (!enable && h->timing) timing_flush(h);
  h->timing = enable != 0;
  return GO2SIM_E_OK;
}
int go2sim_read_timing(go2sim_t* h, float* ms_out8, int* cnt_out8, int reset) {
  if (!h || !h->ev_created) return GO2SIM_E_BADARG;
  timing_flush(h);
  for (int i = 0; i < T_N; ++i) { if (ms_out8) ms_out8[i] = h->t_ms[i]; if (cnt_out8) cnt_out8[i] = h->t_cnt[i]; }
  if (reset) for (int i = 0; i < T_N; ++i) { h->t_ms[i] = 0.0f; h->t_cnt[i] = 0; }
  return GO2SIM_E_OK;
}

int go2sim_debug_narrowphase(go2sim_t* h, int which, int i_ga, int i_gb, const float* pa, const float* qa, const float* pb, const float* qb, float* out8) {
  if (!h || !out8 || !pa || !qa || !pb || !qb || i_ga < 0 || i_gb < 0 || i_ga >= NG || i_gb >= NG || which < 0 || which > 6) return GO2SIM_E_BADARG;
  float* dout = nullptr;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMalloc((void**)&dout, 8 * sizeof(float)));
  V3 pos_a = v3h(pa[0], pa[1], pa[2]), pos_b = v3h(pb[0], pb[1], pb[2]);
  Q4 quat_a = {qa[0], qa[1], qa[2], qa[3]}, quat_b = {qb[0], qb[1], qb[2], qb[3]};
  hipLaunchKernelGGL(k_debug_narrowphase, dim3(1), dim3(64), 0, 0, h->dm, which, i_ga, i_gb, pos_a, quat_a, pos_b, quat_b, h->gjk_scratch, dout);
  hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(out8, dout, 8 * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(dout);
  HIPCHK(e1); HIPCHK(e2);
  return GO2SIM_E_OK;
}

#ifdef GO2SIM_BRACKET_DEBUG
int go2sim_debug_brlog(go2sim_t* h, float* out, int* cnt) {
  if (!h || !out || !cnt) return GO2SIM_E_BADARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_brlog), sizeof(float) * BRLOG_ENVS * BRLOG_CAP * BRLOG_W));
  HIPCHK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_brcnt), sizeof(int) * BRLOG_ENVS));
  return GO2SIM_E_OK;
}
#endif
/* development/test aid (not declared in include/go2sim.h): device address of ANY pool field by name */
#ifdef GO2SIM_STAMP
int go2sim_debug_stamps(go2sim_t* h, unsigned long long* stamps, unsigned* counts) {   // [ST_KINDS][ST_MAX_WG][ST_DEPTH][2], [ST_KINDS][ST_MAX_WG]
  if (!h || !stamps || !counts) return GO2SIM_E_BADARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * ST_KINDS * ST_MAX_WG * ST_DEPTH * 2));
  HIPCHK(hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_stamp_cnt), sizeof(unsigned) * ST_KINDS * ST_MAX_WG));
  return GO2SIM_E_OK;
}
#endif
#ifdef GO2SIM_PHASE_PROFILE
int go2sim_debug_phases(go2sim_t* h, unsigned long long* out64, int reset) {
  if (!h || !out64) return GO2SIM_E_BADARG;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 64 * PH_MAX_WG));   // [PH_MAX_WG][64]
  if (reset) { void* p = nullptr; HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase_cycles))); HIPCHK(hipMemset(p, 0, sizeof(unsigned long long) * 64 * PH_MAX_WG)); }
  return GO2SIM_E_OK;
}
#endif
int go2sim_debug_field(go2sim_t* h, const char* name, void** ptr, int* k, int* is_int) {
  if (!h || !name || !ptr) return GO2SIM_E_BADARG;
#define X(n, c) if (!strcmp(name, #n)) { *ptr = h->P.f + (size_t)FO(n) * h->B; if (k) *k = (c); if (is_int) *is_int = 0; return GO2SIM_E_OK; }
  GO2SIM_FLOAT_FIELDS(X)
#undef X
#define X(n, c) if (!strcmp(name, #n)) { *ptr = h->P.i + (size_t)IO(n) * h->B; if (k) *k = (c); if (is_int) *is_int = 1; return GO2SIM_E_OK; }
  GO2SIM_INT_FIELDS(X)
#undef X
  return GO2SIM_E_BADARG;
}

}  // extern "C"
