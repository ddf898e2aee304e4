/* gjk_epa_cpu.h -- "safe" GJK + EPA penetration query (the non-MuJoCo-compatible branch of Genesis): the ORACLE's host-side restatement.
 * Test infrastructure (oracle/): the product has its own, separately written device implementation (csrc/go2sim_gjk_dev.h), so GPU-vs-oracle
 * parity of the GJK / EPA fallback compares two implementations, not one header with itself.
 *
 * Restates, for the convex-convex narrow phase of go2sim, the reference functions
 *   func_safe_gjk                       genesis/engine/solvers/rigid/collider/gjk.py:1200-1416
 *   func_is_new_simplex_vertex_*        gjk.py:1420-1502,  func_is_colinear / func_is_coplanar  gjk.py:1505-1539
 *   func_search_valid_simplex_vertex    gjk.py:1543-1649
 *   func_safe_gjk_triangle_info         gjk.py:1703-1733
 *   func_safe_gjk_support               gjk.py:1737-1850
 *   func_safe_epa / _witness / _init    collider/epa.py:970-1295
 *   func_safe_attach_face_to_polytope   epa.py:1298-1380,  func_plane_normal  epa.py:1383-1419
 *   func_epa_horizon & helpers          epa.py:274-432,    func_epa_support   epa.py:810-883
 *   func_triangle_affine_coords / func_project_origin_to_plane   collider/gjk_utils.py:49-107,185-235
 *   the tail of func_gjk_contact (witness -> contact)            gjk.py:413-437
 *
 * Header-only, compiled by gcc into oracle/libgo2sim_cpu.so only; the geometric queries (support points, vertex enumeration) are supplied by
 * the includer through the `Sup` functor.  All arithmetic is binary32 with -ffp-contract=off.
 */
#ifndef GO2SIM_GJK_H
#define GO2SIM_GJK_H

#include "../include/go2sim_detmath.h"

#define GJK_FN inline

#define GJK_MAX_ITERATIONS 50          /* gjk.py:53 */
#define EPA_MAX_ITERATIONS 50          /* gjk.py:54 */
#define GJK_POLY_MAX_FACES (6 * EPA_MAX_ITERATIONS) /* gjk.py:56 */
#define GJK_POLY_MAX_VERTS (5 + EPA_MAX_ITERATIONS) /* array_class.py:744 */
#define GJK_FLOAT_MIN 1e-15f           /* gjk.py:87 */
#define GJK_FLOAT_MAX 1e15f            /* gjk.py:88 */
#define GJK_TOLERANCE 1e-6f            /* gjk.py:89 */
#define GJK_SIMPLEX_MAX_DEGENERACY_SQ (1e-5f * 1e-5f) /* gjk.py:93 */
#define GJK_POLY_MAX_REPROJECTION_ERROR 1e-4f /* gjk.py:98 */

struct G3 { float x, y, z; };
GJK_FN G3 g3(float x, float y, float z) { G3 r = {x, y, z}; return r; }
GJK_FN G3 operator+(G3 a, G3 b) { return g3(a.x + b.x, a.y + b.y, a.z + b.z); }
GJK_FN G3 operator-(G3 a, G3 b) { return g3(a.x - b.x, a.y - b.y, a.z - b.z); }
GJK_FN G3 operator-(G3 a) { return g3(-a.x, -a.y, -a.z); }
GJK_FN G3 operator*(G3 a, float s) { return g3(a.x * s, a.y * s, a.z * s); }
GJK_FN G3 operator/(G3 a, float s) { return g3(a.x / s, a.y / s, a.z / s); }
GJK_FN float gdot(G3 a, G3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GJK_FN G3 gcross(G3 a, G3 b) { return g3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
GJK_FN float gnorm_sqr(G3 a) { return gdot(a, a); }
GJK_FN float gnorm(G3 a) { return dm_sqrt(gnorm_sqr(a)); }
GJK_FN G3 gnormalized(G3 a) { return a / gnorm(a); }
GJK_FN float gget(G3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
GJK_FN float gmax(float a, float b) { return (a < b) ? b : a; }

struct GjkVert { G3 obj1, obj2, mink; int id1, id2; };

/* per-query working memory (the reference keeps it in gjk_state[i_b]) */
struct GjkScratch {
  GjkVert sv[4]; int s_nverts; int last_searched;
  GjkVert pv[GJK_POLY_MAX_VERTS]; int p_nverts;
  int f_verts[GJK_POLY_MAX_FACES][3], f_adj[GJK_POLY_MAX_FACES][3];
  G3 f_normal[GJK_POLY_MAX_FACES]; float f_dist2[GJK_POLY_MAX_FACES]; int f_map_idx[GJK_POLY_MAX_FACES];
  int p_nfaces, faces_map[GJK_POLY_MAX_FACES], nfaces_map;
  int hz_face[GJK_POLY_MAX_FACES * 3], hz_edge[GJK_POLY_MAX_FACES * 3], hz_n;
  int st_face[GJK_POLY_MAX_FACES * 3], st_edge[GJK_POLY_MAX_FACES * 3];
  G3 horizon_w;
  G3 w1, w2; int n_witness; float distance;
};

struct GjkResult { int is_col; float penetration; G3 normal, pos; int n_contacts; };

/* func_is_new_simplex_vertex_duplicate, gjk.py:1441-1460 */
GJK_FN bool gjk_vertex_duplicate(const GjkScratch& s, int id1, int id2) {
  bool found = false;
  for (int i = 0; i < s.s_nverts; ++i) {
    if (id1 == -1 || s.sv[i].id1 != id1) continue;
    if (id2 == -1 || s.sv[i].id2 != id2) continue;
    found = true;
    break;
  }
  return found;
}
/* func_is_colinear / func_is_coplanar, gjk.py:1505-1539 */
GJK_FN bool gjk_colinear(G3 v1, G3 v2, G3 v3) {
  G3 e1 = v2 - v1, e2 = v3 - v1;
  G3 normal = gcross(e1, e2);
  return gnorm_sqr(normal) < GJK_SIMPLEX_MAX_DEGENERACY_SQ * gnorm_sqr(e1) * gnorm_sqr(e2);
}
GJK_FN bool gjk_coplanar(G3 v1, G3 v2, G3 v3, G3 v4) {
  G3 e1 = gnormalized(v2 - v1), e2 = gnormalized(v3 - v1);
  G3 normal = gcross(e1, e2);
  G3 diff = v4 - v1;
  float nd = gdot(normal, diff);
  return (nd * nd) < GJK_SIMPLEX_MAX_DEGENERACY_SQ * gnorm_sqr(normal) * gnorm_sqr(diff);
}
/* func_is_new_simplex_vertex_degenerate, gjk.py:1463-1502 */
GJK_FN bool gjk_vertex_degenerate(const GjkScratch& s, G3 mink) {
  bool is_degenerate = false;
  int nverts = s.s_nverts;
  for (int i = 0; i < nverts; ++i)
    if (gnorm_sqr(s.sv[i].mink - mink) < GJK_SIMPLEX_MAX_DEGENERACY_SQ) { is_degenerate = true; break; }
  if (!is_degenerate) {
    if (nverts == 2) is_degenerate = gjk_colinear(s.sv[0].mink, s.sv[1].mink, mink);
    else if (nverts == 3) is_degenerate = gjk_coplanar(s.sv[0].mink, s.sv[1].mink, s.sv[2].mink, mink);
  }
  return is_degenerate;
}
GJK_FN bool gjk_vertex_valid(const GjkScratch& s, int id1, int id2, G3 mink) {
  return !gjk_vertex_duplicate(s, id1, id2) && !gjk_vertex_degenerate(s, mink);
}

/* func_safe_gjk_support, gjk.py:1737-1850 */
template <class Sup>
GJK_FN GjkVert gjk_safe_support(const Sup& sup, const GjkScratch& s, G3 dir) {
  const float EPS = sup.eps;
  GjkVert v; v.obj1 = g3(0, 0, 0); v.obj2 = g3(0, 0, 0); v.id1 = -1; v.id2 = -1; v.mink = v.obj1 - v.obj2;
  for (int i = 0; i < 9; ++i) {
    G3 n_dir = dir;
    if (i > 0) {
      int j = i - 1;
      n_dir.x += -(1.0f - 2.0f * (float)(j & 1)) * EPS;
      n_dir.y += -(1.0f - 2.0f * (float)(j & 2)) * EPS;
      n_dir.z += -(1.0f - 2.0f * (float)(j & 4)) * EPS;
    }
    n_dir = n_dir * (2.0f - gdot(n_dir, dir));
    int num_supports = sup.count(n_dir);
    if (i > 0 && num_supports > 1) continue;
    sup.support(n_dir, v.obj1, v.obj2, v.id1, v.id2);
    v.mink = v.obj1 - v.obj2;
    if (i == 0) { if (num_supports > 1) continue; else break; }
    if (i == 8) break;
    if (gjk_vertex_valid(s, v.id1, v.id2, v.mink)) break;
  }
  return v;
}

/* func_search_valid_simplex_vertex, gjk.py:1543-1649 */
template <class Sup>
GJK_FN bool gjk_search_valid_vertex(const Sup& sup, GjkScratch& s, GjkVert& out) {
  out.obj1 = g3(0, 0, 0); out.obj2 = g3(0, 0, 0); out.id1 = -1; out.id2 = -1; out.mink = g3(0, 0, 0);
  bool ok = false;
  if (sup.discrete) {
    int n0 = sup.nverts_a, n1 = sup.nverts_b;
    int num_cases = n0 * n1;
    for (int k = 0; k < num_cases; ++k) {
      int m = (k + s.last_searched) % num_cases;
      int i = m / n1, j = m % n1;
      sup.discrete_vertex(0, i, out.obj1, out.id1);
      sup.discrete_vertex(1, j, out.obj2, out.id2);
      out.mink = out.obj1 - out.obj2;
      if (gjk_vertex_valid(s, out.id1, out.id2, out.mink)) { ok = true; s.last_searched = (m + 1) % num_cases; break; }
    }
  } else {
    if (s.s_nverts == 3) {
      G3 v1 = s.sv[0].mink, v2 = s.sv[1].mink, v3 = s.sv[2].mink;
      G3 dir = gnormalized(gcross(v3 - v1, v2 - v1));
      for (int i = 0; i < 2; ++i) {
        G3 d = (i == 0) ? dir : -dir;
        out = gjk_safe_support(sup, s, d);
        if (gjk_vertex_valid(s, out.id1, out.id2, out.mink)) { ok = true; break; }
      }
    }
  }
  return ok;
}

/* func_safe_gjk_triangle_info, gjk.py:1703-1733 */
GJK_FN void gjk_triangle_info(const GjkScratch& s, int i_ta, int i_tb, int i_tc, int i_apex, G3& normal, float& sdist) {
  G3 vertex_1 = s.sv[i_ta].mink, vertex_2 = s.sv[i_tb].mink, vertex_3 = s.sv[i_tc].mink, apex = s.sv[i_apex].mink;
  normal = gnormalized(gcross(vertex_3 - vertex_1, vertex_2 - vertex_1));
  if (gdot(normal, apex - vertex_1) > 0.0f) normal = -normal;
  sdist = gdot(normal, vertex_1);
}

/* func_safe_gjk, gjk.py:1200-1416; returns true on INTERSECT */
template <class Sup>
GJK_FN bool gjk_safe_gjk(const Sup& sup, GjkScratch& s) {
  bool init_ok = true;
  s.s_nverts = 0;
  for (int i = 0; i < 4; ++i) {
    G3 dir = g3(0, 0, 0);
    float sgn = 1.0f - 2.0f * (float)(i % 2);
    int ax = 2 - i / 2;
    if (ax == 0) dir.x = sgn; else if (ax == 1) dir.y = sgn; else dir.z = sgn;
    GjkVert v = gjk_safe_support(sup, s, dir);
    if (!gjk_vertex_valid(s, v.id1, v.id2, v.mink)) {
      if (!gjk_search_valid_vertex(sup, s, v)) { init_ok = false; break; }
    }
    s.sv[i] = v;
    s.s_nverts += 1;
  }
  bool intersect = false;
  if (init_ok) {
    const int si[4] = {0, 1, 2, 3};
    for (int it = 0; it < GJK_MAX_ITERATIONS; ++it) {
      G3 normals[4]; float sdists[4];
      for (int j = 0; j < 4; ++j) {
        int s0 = si[2], s1 = si[1], s2 = si[3], ap = si[0];
        if (j == 1) { s0 = si[0]; s1 = si[2]; s2 = si[3]; ap = si[1]; }
        else if (j == 2) { s0 = si[1]; s1 = si[0]; s2 = si[3]; ap = si[2]; }
        else if (j == 3) { s0 = si[0]; s1 = si[1]; s2 = si[2]; ap = si[3]; }
        gjk_triangle_info(s, s0, s1, s2, ap, normals[j], sdists[j]);
      }
      int min_i = 0;
      for (int j = 1; j < 4; ++j) if (sdists[j] < sdists[min_i]) min_i = j;
      int min_si = si[min_i];
      G3 min_normal = normals[min_i];
      float min_sdist = sdists[min_i];
      if (min_sdist >= 0) { intersect = true; break; }
      s.s_nverts = 3;
      if (min_si != 3) s.sv[min_si] = s.sv[3];
      GjkVert v = gjk_safe_support(sup, s, min_normal);
      if (gjk_vertex_duplicate(s, v.id1, v.id2)) break;        /* SEPARATED */
      if (gjk_vertex_degenerate(s, v.mink)) break;             /* NUM_ERROR -> treated as SEPARATED */
      if (gdot(v.mink, min_normal) < 0.0f) break;              /* SEPARATED */
      s.sv[3] = v;
      s.s_nverts = 4;
    }
  }
  s.distance = intersect ? 0.0f : GJK_FLOAT_MAX;
  return intersect;
}

/* func_plane_normal, epa.py:1383-1419 */
GJK_FN bool gjk_plane_normal(G3 v1, G3 v2, G3 v3, G3& normal) {
  normal = g3(0, 0, 0);
  bool ok = false, finished = false;
  G3 d21 = v2 - v1, d31 = v3 - v1, d32 = v3 - v2;
  for (int i = 0; i < 3; ++i) {
    if (!finished) {
      G3 n = (i == 0) ? gcross(d32, d21) : ((i == 1) ? gcross(d21, d31) : gcross(d31, d32));
      float nn = gnorm(n);
      if (nn == 0) { ok = false; finished = true; }
      else if (nn > GJK_FLOAT_MIN) { normal = gnormalized(n); ok = true; finished = true; }
    }
  }
  return ok;
}

/* func_epa_insert_vertex_to_polytope, epa.py:408-432 */
GJK_FN int epa_insert_vertex(GjkScratch& s, const GjkVert& v) { int n = s.p_nverts; s.pv[n] = v; s.p_nverts += 1; return n; }

/* func_safe_attach_face_to_polytope, epa.py:1298-1380 */
GJK_FN bool epa_safe_attach_face(GjkScratch& s, int i_v1, int i_v2, int i_v3, int i_a1, int i_a2, int i_a3) {
  int n = s.p_nfaces;
  s.f_verts[n][0] = i_v1; s.f_verts[n][1] = i_v2; s.f_verts[n][2] = i_v3;
  s.f_adj[n][0] = i_a1; s.f_adj[n][1] = i_a2; s.f_adj[n][2] = i_a3;
  s.p_nfaces += 1;
  G3 normal;
  bool ok = gjk_plane_normal(s.pv[i_v3].mink, s.pv[i_v2].mink, s.pv[i_v1].mink, normal);
  if (ok) {
    G3 face_center = (s.pv[i_v1].mink + s.pv[i_v2].mink + s.pv[i_v3].mink) / 3.0f;
    float max_orient = -gdot(normal, face_center);
    float max_abs_orient = dm_abs(max_orient);
    for (int i_v = 0; i_v < s.p_nverts; ++i_v)
      if (i_v != i_v1 && i_v != i_v2 && i_v != i_v3) {
        G3 diff = s.pv[i_v].mink - face_center;
        float orient = gdot(normal, diff);
        if (dm_abs(orient) > max_abs_orient) { max_abs_orient = dm_abs(orient); max_orient = orient; }
      }
    if (max_orient > 0.0f) normal = -normal;
    s.f_normal[n] = normal;
    float min_dist2 = GJK_FLOAT_MAX;
    for (int i = 0; i < 3; ++i) {
      int i_v = (i == 0) ? i_v1 : ((i == 1) ? i_v2 : i_v3);
      float d = gdot(normal, s.pv[i_v].mink);
      float dist2 = d * d;
      if (dist2 < min_dist2) min_dist2 = dist2;
    }
    s.f_dist2[n] = min_dist2;
    s.f_map_idx[n] = -1;
  }
  return ok;
}

/* func_safe_epa_init, epa.py:1245-1295 */
GJK_FN void epa_safe_init(GjkScratch& s) {
  int vi[4];
  for (int i = 0; i < 4; ++i) vi[i] = epa_insert_vertex(s, s.sv[i]);
  for (int i = 0; i < 4; ++i) {
    int v1 = vi[0], v2 = vi[1], v3 = vi[2], a1 = 1, a2 = 3, a3 = 2;
    if (i == 1) { v1 = vi[0]; v2 = vi[3]; v3 = vi[1]; a1 = 2; a2 = 3; a3 = 0; }
    else if (i == 2) { v1 = vi[0]; v2 = vi[2]; v3 = vi[3]; a1 = 0; a2 = 3; a3 = 1; }
    else if (i == 3) { v1 = vi[3]; v2 = vi[2]; v3 = vi[1]; a1 = 2; a2 = 0; a3 = 1; }
    epa_safe_attach_face(s, v1, v2, v3, a1, a2, a3);
  }
  for (int i = 0; i < 4; ++i) { s.faces_map[i] = i; s.f_map_idx[i] = i; }
  s.nfaces_map = 4;
}

/* func_delete_face_from_polytope, epa.py:384-405 */
GJK_FN void epa_delete_face(GjkScratch& s, int i_f) {
  int face_map_idx = s.f_map_idx[i_f];
  if (face_map_idx >= 0) {
    int last_face_idx = s.faces_map[s.nfaces_map - 1];
    s.faces_map[face_map_idx] = last_face_idx;
    s.f_map_idx[last_face_idx] = face_map_idx;
    s.nfaces_map -= 1;
  }
  s.f_map_idx[i_f] = -2;
}
/* func_get_edge_idx, epa.py:362-381 */
GJK_FN int epa_edge_idx(const GjkScratch& s, int i_f, int i_v) {
  int ret = 2;
  if (s.f_verts[i_f][0] == i_v) ret = 0; else if (s.f_verts[i_f][1] == i_v) ret = 1;
  return ret;
}
/* func_epa_horizon, epa.py:274-341 (func_add_edge_to_horizon always succeeds) */
GJK_FN void epa_horizon(GjkScratch& s, int nearest_i_f) {
  G3 w = s.horizon_w;
  s.st_face[0] = nearest_i_f; s.st_edge[0] = 0;
  int top = 1;
  bool is_first = true;
  while (top > 0) {
    int i_f = s.st_face[top - 1], i_e = s.st_edge[top - 1];
    int i_v = s.f_verts[i_f][0];
    G3 v = s.pv[i_v].mink;
    top -= 1;
    bool is_deleted = s.f_map_idx[i_f] == -2;
    if (!is_first && is_deleted) continue;
    bool is_visible = gdot(s.f_normal[i_f], w - v) > GJK_FLOAT_MIN;
    if (is_visible || is_first) {
      epa_delete_face(s, i_f);
      for (int k = (is_first ? 0 : 1); k < 3; ++k) {
        int i_e2 = (i_e + k) % 3;
        int adj_face_idx = s.f_adj[i_f][i_e2];
        bool adj_deleted = s.f_map_idx[adj_face_idx] == -2;
        if (!adj_deleted) {
          int start_vert_idx = s.f_verts[i_f][(i_e2 + 1) % 3];
          int adj_edge_idx = epa_edge_idx(s, adj_face_idx, start_vert_idx);
          s.st_face[top] = adj_face_idx; s.st_edge[top] = adj_edge_idx;
          top += 1;
        }
      }
    } else {
      s.hz_edge[s.hz_n] = i_e; s.hz_face[s.hz_n] = i_f; s.hz_n += 1;
    }
    is_first = false;
  }
}

/* func_triangle_affine_coords, gjk_utils.py:49-107 */
GJK_FN G3 gjk_triangle_affine_coords(G3 point, G3 tri_v1, G3 tri_v2, G3 tri_v3) {
  float ms[3];
  for (int i = 0; i < 3; ++i) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    if (i == 1) { int t = i1; i1 = i2; i2 = t; }
    ms[i] = gget(tri_v2, i1) * gget(tri_v3, i2) - gget(tri_v2, i2) * gget(tri_v3, i1) - gget(tri_v1, i1) * gget(tri_v3, i2) +
            gget(tri_v1, i2) * gget(tri_v3, i1) + gget(tri_v1, i1) * gget(tri_v2, i2) - gget(tri_v1, i2) * gget(tri_v2, i1);
  }
  float m_max = 0.0f;
  int i_x = 0, i_y = 0;
  float absms[3] = {dm_abs(ms[0]), dm_abs(ms[1]), dm_abs(ms[2])};
  for (int i = 0; i < 3; ++i)
    if (absms[i] >= absms[(i + 1) % 3] && absms[i] >= absms[(i + 2) % 3]) {
      m_max = ms[i];
      i_x = (i + 1) % 3; i_y = (i + 2) % 3;
      if (i == 1) { int t = i_x; i_x = i_y; i_y = t; }
      break;
    }
  float cs[3];
  for (int i = 0; i < 3; ++i) {
    G3 tv1 = tri_v2, tv2 = tri_v3;
    if (i == 1) { tv1 = tri_v3; tv2 = tri_v1; } else if (i == 2) { tv1 = tri_v1; tv2 = tri_v2; }
    cs[i] = gget(point, i_x) * gget(tv1, i_y) + gget(point, i_y) * gget(tv2, i_x) + gget(tv1, i_x) * gget(tv2, i_y) -
            gget(point, i_x) * gget(tv2, i_y) - gget(point, i_y) * gget(tv1, i_x) - gget(tv2, i_x) * gget(tv1, i_y);
  }
  return g3(cs[0] / m_max, cs[1] / m_max, cs[2] / m_max);
}
/* func_project_origin_to_plane, gjk_utils.py:185-235 */
GJK_FN bool gjk_project_origin_to_plane(G3 v1, G3 v2, G3 v3, G3& point) {
  point = g3(0, 0, 0);
  bool ok = true;
  G3 d21 = v2 - v1, d31 = v3 - v1, d32 = v3 - v2;
  for (int i = 0; i < 3; ++i) {
    G3 n, v;
    if (i == 0) { n = gcross(d32, d21); v = v2; }
    else if (i == 1) { n = gcross(d21, d31); v = v1; }
    else { n = gcross(d31, d32); v = v3; }
    float nv = gdot(n, v);
    float nn = gnorm_sqr(n);
    if (nn == 0) { ok = false; break; }
    else if (nn > GJK_FLOAT_MIN) { point = n * (nv / nn); ok = true; break; }
    if (i == 2) {
      if (nn < GJK_FLOAT_MIN) ok = false;
      else { point = n * (nv / nn); ok = true; }
    }
  }
  return ok;
}
/* func_safe_epa_witness, epa.py:1184-1242 */
GJK_FN bool epa_safe_witness(GjkScratch& s, int i_f) {
  int iv1 = s.f_verts[i_f][0], iv2 = s.f_verts[i_f][1], iv3 = s.f_verts[i_f][2];
  G3 face_v1 = s.pv[iv1].mink, face_v2 = s.pv[iv2].mink, face_v3 = s.pv[iv3].mink;
  G3 proj_o;
  (void)gjk_project_origin_to_plane(face_v1, face_v2, face_v3, proj_o);
  G3 l = gjk_triangle_affine_coords(proj_o, face_v1, face_v2, face_v3);
  G3 v1 = face_v1, v2 = face_v2, v3 = face_v3;
  G3 proj_o_lambda = v1 * l.x + v2 * l.y + v3 * l.z;
  float reprojection_error = gnorm(proj_o - proj_o_lambda);
  float max_edge_len_inv = 1.0f / dm_sqrt(gmax(gmax(gmax(gnorm_sqr(v1 - v2), gnorm_sqr(v2 - v3)), gnorm_sqr(v3 - v1)), GJK_FLOAT_MIN * GJK_FLOAT_MIN));
  float rel = reprojection_error * max_edge_len_inv;
  if (rel > GJK_POLY_MAX_REPROJECTION_ERROR) return false;
  s.w1 = s.pv[iv1].obj1 * l.x + s.pv[iv2].obj1 * l.y + s.pv[iv3].obj1 * l.z;
  s.w2 = s.pv[iv1].obj2 * l.x + s.pv[iv2].obj2 * l.y + s.pv[iv3].obj2 * l.z;
  return true;
}

/* func_safe_epa, epa.py:970-1181 */
template <class Sup>
GJK_FN int epa_safe_epa(const Sup& sup, GjkScratch& s) {
  float upper = GJK_FLOAT_MAX, upper2 = GJK_FLOAT_MAX * GJK_FLOAT_MAX, lower = 0.0f;
  float tolerance = GJK_TOLERANCE;
  const float EPS = sup.eps;
  int nearest_i_f = -1, prev_nearest_i_f = -1;
  const bool discrete = sup.discrete;
  if (discrete) tolerance = EPS;
  for (int k = 0; k < EPA_MAX_ITERATIONS; ++k) {
    prev_nearest_i_f = nearest_i_f;
    float lower2 = GJK_FLOAT_MAX * GJK_FLOAT_MAX;
    for (int i = 0; i < s.nfaces_map; ++i) {
      int i_f = s.faces_map[i];
      float face_dist2 = s.f_dist2[i_f];
      if (face_dist2 < lower2) { lower2 = face_dist2; nearest_i_f = i_f; }
    }
    if (lower2 > upper2 || nearest_i_f == -1) { nearest_i_f = prev_nearest_i_f; break; }
    lower = dm_sqrt(lower2);
    G3 dir = s.f_normal[nearest_i_f];
    GjkVert nv;                                                         /* func_epa_support(dir, 1.0): d = dir / 1.0 */
    G3 d = dir / 1.0f;
    sup.support(d, nv.obj1, nv.obj2, nv.id1, nv.id2);
    nv.mink = nv.obj1 - nv.obj2;
    int wi = epa_insert_vertex(s, nv);
    G3 w = s.pv[wi].mink;
    float upper_k = gdot(w, dir);
    if (upper_k < upper) { upper = upper_k; upper2 = upper * upper; }
    if ((upper - lower) < tolerance) break;
    if (discrete) {
      bool repeated = false;
      for (int i = 0; i < s.p_nverts; ++i) {
        if (i == wi) continue;
        else if (s.pv[i].id1 == s.pv[wi].id1 && s.pv[i].id2 == s.pv[wi].id2) { repeated = true; break; }
      }
      if (repeated) break;
    }
    s.horizon_w = w;
    epa_horizon(s, nearest_i_f);
    if (s.hz_n < 3) { nearest_i_f = -1; break; }
    int nfaces = s.p_nfaces, nedges = s.hz_n;
    if (nfaces + nedges >= GJK_POLY_MAX_FACES) break;
    bool attach_ok = true;
    for (int i = 0; i < nedges; ++i) {
      int i_f0 = nfaces + i, i_f1 = nfaces + (i + 1) % nedges;
      int horizon_i_f = s.hz_face[i], horizon_i_e = s.hz_edge[i];
      int horizon_v1 = s.f_verts[horizon_i_f][horizon_i_e], horizon_v2 = s.f_verts[horizon_i_f][(horizon_i_e + 1) % 3];
      s.f_adj[horizon_i_f][horizon_i_e] = i_f0;
      int adj_i_f_0 = (i > 0) ? (i_f0 - 1) : (nfaces + nedges - 1);
      int adj_i_f_1 = horizon_i_f, adj_i_f_2 = i_f1;
      attach_ok = epa_safe_attach_face(s, wi, horizon_v2, horizon_v1, adj_i_f_2, adj_i_f_1, adj_i_f_0);
      if (!attach_ok) break;
      float dist2 = s.f_dist2[s.p_nfaces - 1];
      if ((dist2 >= lower2 - EPS) && (dist2 <= upper2 + EPS)) {
        int nm = s.nfaces_map;
        s.faces_map[nm] = i_f0; s.f_map_idx[i_f0] = nm; s.nfaces_map += 1;
      }
    }
    if (!attach_ok) { nearest_i_f = -1; break; }
    s.hz_n = 0;
    if (s.nfaces_map == 0 || nearest_i_f == -1) { nearest_i_f = -1; break; }
  }
  if (nearest_i_f != -1) {
    float dist2 = s.f_dist2[nearest_i_f];
    if (epa_safe_witness(s, nearest_i_f)) { s.n_witness = 1; s.distance = -dm_sqrt(dist2); }
    else { s.n_witness = 0; s.distance = 0.0f; nearest_i_f = -1; }
  } else {
    s.n_witness = 0; s.distance = 0.0f;
  }
  return nearest_i_f;
}

/* func_gjk_contact (non-MuJoCo branch), gjk.py:161-437 */
template <class Sup>
GJK_FN GjkResult gjk_contact(const Sup& sup, GjkScratch& s) {
  s.last_searched = 0;                                                  /* clear_cache, gjk.py:148-157 */
  s.n_witness = 0; s.distance = 0.0f;
  if (gjk_safe_gjk(sup, s)) {
    s.p_nverts = 0; s.p_nfaces = 0; s.nfaces_map = 0; s.hz_n = 0;
    epa_safe_init(s);
    (void)epa_safe_epa(sup, s);
  }
  GjkResult r; r.n_contacts = 0; r.normal = g3(0, 0, 0); r.pos = g3(0, 0, 0);
  r.is_col = s.distance < 0.0f;
  r.penetration = r.is_col ? -s.distance : 0.0f;
  if (r.is_col) {
    for (int i = 0; i < s.n_witness; ++i) {
      G3 w1 = s.w1, w2 = s.w2;
      G3 contact_pos = (w1 + w2) * 0.5f;
      G3 normal = w2 - w1;
      float normal_len = gnorm(normal);
      if (normal_len < GJK_FLOAT_MIN) continue;
      r.normal = normal / normal_len; r.pos = contact_pos;
      r.n_contacts += 1;
    }
  }
  if (r.n_contacts == 0) { r.is_col = 0; r.penetration = 0.0f; }
  return r;
}

#endif /* GO2SIM_GJK_H */
