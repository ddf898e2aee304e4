// go2sim_gjk_dev.h -- device-side "safe" GJK + EPA penetration query of the convex-convex narrow phase (gfx950).
//
// Written for the GPU against the reference functions themselves (it shares no code with the oracle's host-side restatement
// oracle/gjk_epa_cpu.h; the parity tests therefore compare two separately written implementations):
//   func_safe_gjk                  genesis/engine/solvers/rigid/collider/gjk.py:1200-1416
//   func_is_new_simplex_vertex_*   gjk.py:1420-1502,  func_is_colinear / func_is_coplanar  gjk.py:1505-1539
//   func_search_valid_simplex_vertex  gjk.py:1543-1649,  func_safe_gjk_triangle_info  gjk.py:1703-1733,  func_safe_gjk_support  gjk.py:1737-1850
//   func_safe_epa / _witness / _init  collider/epa.py:970-1295,  func_safe_attach_face_to_polytope  epa.py:1298-1380,  func_plane_normal  :1383-1419
//   func_epa_horizon and helpers   epa.py:274-432
//   func_triangle_affine_coords / func_project_origin_to_plane   collider/gjk_utils.py:49-107,185-235
//   tail of func_gjk_contact (witness -> contact)   gjk.py:413-437
//
// MI355X shape of the thing: the query is serial per pair (one lane), and what it costs is the latency of its working set.  The reference keeps
// a 31 KB per-env polytope in global memory; the queries of this workload (sphere / box / cylinder against the ground box) end after 2-5 EPA
// iterations with <= 9 vertices and <= 19 faces, so the working set is a compact record (44 B per vertex, 32 B per face, 16-bit indices) that
// fits a ~1.2 KB LDS slot.  The store is a template parameter: `GjkStore<10, 20, 10>` lives in LDS (k_collide_team hands a lane one of the
// team's slots, carved out of the broad-phase arrays that are dead by then); if a query outgrows it, or no slot is free, the same code runs on
// the full-capacity store in global memory (`GjkStoreFull`, the reference's capacities), which is deterministic and capacity-independent, so
// the answer is the same either way.  Loop counters and sizes live in registers (`GjkCtl`), not in memory.

#ifndef GO2SIM_GJK_DEV_H
#define GO2SIM_GJK_DEV_H

// Two code shapes, same arithmetic: the default inlines and unrolls (fastest measured: 107 k cycles per query at the landing peak); with
// -DGO2SIM_GJK_COMPACT the support evaluation / face attachment are single out-of-line copies and the loops stay rolled (4 k instead of 15 k
// instructions, but 164 k cycles per query: the call frames and rolled loops cost more than the instruction fetch they save).
#ifdef GO2SIM_GJK_COMPACT
#define DG_OUTLINE DEVN
#define DG_NOUNROLL _Pragma("nounroll")
#else
#define DG_OUTLINE DEV
#define DG_NOUNROLL
#endif
constexpr int DG_GJK_MAX_IT = 50, DG_EPA_MAX_IT = 50;                 // gjk.py:53-54
constexpr int DG_MAX_FACES = 6 * DG_EPA_MAX_IT;                        // gjk.py:56 (polytope_max_faces)
constexpr int DG_MAX_VERTS = 5 + DG_EPA_MAX_IT;                        // array_class.py:744
constexpr float DG_FLOAT_MIN = 1e-15f, DG_FLOAT_MAX = 1e15f;           // gjk.py:87-88
constexpr float DG_TOLERANCE = 1e-6f;                                  // gjk.py:89
constexpr float DG_DEGEN_SQ = 1e-5f * 1e-5f;                           // gjk.py:93 (simplex_max_degeneracy_sq)
constexpr float DG_MAX_REPROJ = 1e-4f;                                 // gjk.py:98

struct DgVert { V3 o1, o2, mk; int id1, id2; };                       // support points on both geoms, their difference, vertex ids
struct alignas(16) DgFace { V3 n; float d2; short v[3], adj[3]; short map_idx, pad; };
template <int MV, int MF, int MH>
struct alignas(16) GjkStore {
  static constexpr int CAP_V = MV, CAP_F = MF, CAP_H = MH;
  DgVert v[MV];                                                        // v[0..3] double as the GJK simplex
  DgFace f[MF];
  short map[MF];                                                       // candidate faces (polytope_faces_map)
  short hz_f[MH], hz_e[MH], st_f[MH], st_e[MH];                        // horizon edges, DFS stack
};
typedef GjkStore<DG_MAX_VERTS, DG_MAX_FACES, 3 * DG_MAX_FACES> GjkStoreFull;   // the reference's capacities (array_class.py:735-790)
typedef GjkStore<10, 20, 10> GjkStoreLds;
struct GjkCtl { int nv, nf, nmap, hz_n, ns, last_searched; bool overflow; };
struct DgResult { bool is_col, overflow; float penetration; V3 normal, pos; };

// ---- geometry of the pair: world-frame support points with vertex ids (gjk_support.py:62-186, support_field.py:183-306) ----
struct DgPair {
  const Model* m; int i_ga, i_gb; V3 pos_a; Q4 quat_a; V3 pos_b; Q4 quat_b; bool discrete; GeomLite ga, gb; Rot ra, rb;   // type / size of the two geoms and the rotation coefficients of their poses, set up once
  DEV V3 support_one(V3 d, int i_g, const GeomLite& gl, V3 pos, const Rot& rot, int& vid) const { return gjk_support_driver(*m, d, i_g, gl, pos, rot, vid); }
  DG_OUTLINE void support_into(V3 d, DgVert* out) const {                   // the one copy of the support code of a query
    PHD_BEGIN
    DgVert r;
    r.o1 = support_one(d, i_ga, ga, pos_a, ra, r.id1);
    r.o2 = support_one(-d, i_gb, gb, pos_b, rb, r.id2);
    r.mk = r.o1 - r.o2;
    *out = r;
    PHD(48)
  }
  DEV DgVert support(V3 d) const { DgVert r; support_into(d, &r); return r; }
  // count_support_driver, gjk.py:1854-1877: only a box can have several support points (a face / edge exactly normal to the direction)
  DEV int count_one(V3 d, const GeomLite& gl, const Rot& rot) const {
    if (gl.type != GEOM_BOX) return 1;
    V3 db = rot_apply_inv(rot, d);
    return 1 << ((db.x == 0.0f) + (db.y == 0.0f) + (db.z == 0.0f));
  }
  DG_OUTLINE int count(V3 d) const { return count_one(d, ga, ra) * count_one(-d, gb, rb); }
  // func_get_discrete_geom_vertex (BOX), gjk.py:1666-1700
  DEV void box_vertex(bool second, int i_v, V3& obj, int& id) const {
    const int i_g = second ? i_gb : i_ga;
    const GeomLite& G = second ? gb : ga;
    V3 loc = v3(((i_v & 1) ? 1.0f : -1.0f) * G.d0 * 0.5f, ((i_v & 2) ? 1.0f : -1.0f) * G.d1 * 0.5f, ((i_v & 4) ? 1.0f : -1.0f) * G.d2 * 0.5f);
    obj = rot_apply(second ? rb : ra, loc) + (second ? pos_b : pos_a);
    id = 64 * i_g + i_v;
  }
};

// ---- simplex vertex admission tests (gjk.py:1420-1539) ----
template <class S>
DEV bool dg_duplicate(const S& st, int ns, int id1, int id2) {
  for (int i = 0; i < ns; ++i)
    if (id1 != -1 && st.v[i].id1 == id1 && id2 != -1 && st.v[i].id2 == id2) return true;
  return false;
}
DEV bool dg_colinear(V3 a, V3 b, V3 c) {
  V3 e1 = b - a, e2 = c - a, nrm = cross(e1, e2);
  return norm_sqr(nrm) < DG_DEGEN_SQ * norm_sqr(e1) * norm_sqr(e2);
}
DEV bool dg_coplanar(V3 a, V3 b, V3 c, V3 d) {
  V3 ab = b - a, ac = c - a;
  V3 e1 = ab / norm(ab), e2 = ac / norm(ac);
  V3 nrm = cross(e1, e2), diff = d - a;
  float nd = dot(nrm, diff);
  return (nd * nd) < DG_DEGEN_SQ * norm_sqr(nrm) * norm_sqr(diff);
}
template <class S>
DEV bool dg_degenerate(const S& st, int ns, V3 mk) {
  for (int i = 0; i < ns; ++i)
    if (norm_sqr(st.v[i].mk - mk) < DG_DEGEN_SQ) return true;
  if (ns == 2) return dg_colinear(st.v[0].mk, st.v[1].mk, mk);
  if (ns == 3) return dg_coplanar(st.v[0].mk, st.v[1].mk, st.v[2].mk, mk);
  return false;
}
template <class S>
DEV bool dg_valid(const S& st, int ns, const DgVert& w) { return !dg_duplicate(st, ns, w.id1, w.id2) && !dg_degenerate(st, ns, w.mk); }

// func_safe_gjk_support, gjk.py:1737-1850: when the plain direction hits a face / edge of a box (several support points), the direction is
// nudged towards one of the 8 octants until the support point is unique and admissible
template <class S>
DEV DgVert dg_safe_support(const DgPair& pr, const S& st, int ns, V3 dir, float eps) {
  DgVert w; w.o1 = v3(0, 0, 0); w.o2 = v3(0, 0, 0); w.mk = v3(0, 0, 0); w.id1 = -1; w.id2 = -1;
DG_NOUNROLL
  for (int i = 0; i < 9; ++i) {
    V3 nd = dir;
    if (i > 0) {
      const int j = i - 1;
      nd.x += -(1.0f - 2.0f * (float)(j & 1)) * eps;
      nd.y += -(1.0f - 2.0f * (float)(j & 2)) * eps;
      nd.z += -(1.0f - 2.0f * (float)(j & 4)) * eps;
    }
    nd = nd * (2.0f - dot(nd, dir));                                    // first-order renormalisation
    const int n_sup = pr.count(nd);
    if (i > 0 && n_sup > 1) continue;
    w = pr.support(nd);
    if (i == 0) { if (n_sup > 1) continue; break; }
    if (i == 8) break;
    if (dg_valid(st, ns, w)) break;
  }
  return w;
}

// func_search_valid_simplex_vertex, gjk.py:1543-1649
template <class S>
DEV bool dg_search_vertex(const DgPair& pr, const S& st, GjkCtl& c, DgVert& w, float eps) {
  w.o1 = v3(0, 0, 0); w.o2 = v3(0, 0, 0); w.mk = v3(0, 0, 0); w.id1 = -1; w.id2 = -1;
  if (pr.discrete) {                                                    // box - box: walk the 8 x 8 vertex pairs from where the last search stopped
DG_NOUNROLL
    for (int k = 0; k < 64; ++k) {
      const int mth = (k + c.last_searched) % 64;
      pr.box_vertex(false, mth / 8, w.o1, w.id1);
      pr.box_vertex(true, mth % 8, w.o2, w.id2);
      w.mk = w.o1 - w.o2;
      if (dg_valid(st, c.ns, w)) { c.last_searched = (mth + 1) % 64; return true; }
    }
    return false;
  }
  if (c.ns == 3) {                                                      // both normals of the triangle
    V3 a = st.v[0].mk, b = st.v[1].mk, cc = st.v[2].mk;
    V3 nrm = cross(cc - a, b - a);
    V3 dir = nrm / norm(nrm);
DG_NOUNROLL
    for (int i = 0; i < 2; ++i) {
      w = dg_safe_support(pr, st, c.ns, (i == 0) ? dir : -dir, eps);
      if (dg_valid(st, c.ns, w)) return true;
    }
  }
  return false;
}

// func_safe_gjk, gjk.py:1200-1416: true when the tetrahedron v[0..3] contains the origin.  One loop serves both stages (steps 0..3 build
// the initial tetrahedron from the directions +-z, +-y, the later steps replace the vertex opposite to the worst face), so the safe-support
// code has a single call site.
template <class S>
DEV bool dg_gjk(const DgPair& pr, S& st, GjkCtl& c, float eps) {
  c.ns = 0;
  V3 best_n = v3(0, 0, 0);
DG_NOUNROLL
  for (int step = 0; step < 4 + DG_GJK_MAX_IT; ++step) {
    const bool init = step < 4;
    V3 dir = best_n;
    if (init) {
      dir = v3(0, 0, 0);
      const float sgn = 1.0f - 2.0f * (float)(step % 2);
      if (step < 2) dir.z = sgn; else dir.y = sgn;
    } else {
      // outward normal and signed distance (origin inside => positive) of the four faces; face j is opposite to vertex j
      float best_sd = 0.0f; int best = 0;
DG_NOUNROLL
      for (int j = 0; j < 4; ++j) {
        const int a = (j == 0) ? 2 : ((j == 1) ? 0 : ((j == 2) ? 1 : 0));
        const int b = (j == 0) ? 1 : ((j == 1) ? 2 : ((j == 2) ? 0 : 1));
        const int cidx = (j == 3) ? 2 : 3;
        V3 va = st.v[a].mk, vb = st.v[b].mk, vc = st.v[cidx].mk, apex = st.v[j].mk;
        V3 nrm = cross(vc - va, vb - va);                                // func_safe_gjk_triangle_info, gjk.py:1703-1733
        nrm = nrm / norm(nrm);
        if (dot(nrm, apex - va) > 0.0f) nrm = -nrm;
        const float sd = dot(nrm, va);
        if (j == 0 || sd < best_sd) { best_sd = sd; best_n = nrm; best = j; }
      }
      if (best_sd >= 0.0f) return true;                                  // INTERSECT
      c.ns = 3;
      if (best != 3) st.v[best] = st.v[3];                               // drop the vertex opposite to the worst face
      dir = best_n;
    }
    DgVert w = dg_safe_support(pr, st, c.ns, dir, eps);
    if (init) {
      if (!dg_valid(st, c.ns, w) && !dg_search_vertex(pr, st, c, w, eps)) return false;
      st.v[step] = w;
      c.ns += 1;
    } else {
      if (dg_duplicate(st, c.ns, w.id1, w.id2)) return false;            // SEPARATED
      if (dg_degenerate(st, c.ns, w.mk)) return false;                   // NUM_ERROR, treated as separated
      if (dot(w.mk, best_n) < 0.0f) return false;                        // the origin is outside the Minkowski difference
      st.v[3] = w;
      c.ns = 4;
    }
  }
  return false;
}

// func_plane_normal, epa.py:1383-1419
DEV bool dg_plane_normal(V3 p1, V3 p2, V3 p3, V3& nrm) {
  nrm = v3(0, 0, 0);
  V3 d21 = p2 - p1, d31 = p3 - p1, d32 = p3 - p2;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    V3 n = (i == 0) ? cross(d32, d21) : ((i == 1) ? cross(d21, d31) : cross(d31, d32));
    const float nn = norm(n);
    if (nn == 0.0f) return false;
    if (nn > DG_FLOAT_MIN) { nrm = n / nn; return true; }
  }
  return false;
}

// func_safe_attach_face_to_polytope, epa.py:1298-1380; the caller has checked the capacity
// (the face index and the vertex count are explicit so that several lanes can attach different faces of one polytope side by side)
template <class S>
DG_OUTLINE bool dg_attach_face_at(S& st, int nv, int n, int v1, int v2, int v3_, int a1, int a2, int a3) {
  DgFace& F = st.f[n];
  F.v[0] = (short)v1; F.v[1] = (short)v2; F.v[2] = (short)v3_; F.adj[0] = (short)a1; F.adj[1] = (short)a2; F.adj[2] = (short)a3;
  V3 p1 = st.v[v1].mk, p2 = st.v[v2].mk, p3 = st.v[v3_].mk, nrm;
  if (!dg_plane_normal(p3, p2, p1, nrm)) return false;
  // orientation: away from the origin and from the other vertices of the polytope, whichever speaks loudest
  V3 center = (p1 + p2 + p3) / 3.0f;
  float max_orient = -dot(nrm, center), max_abs = dm_abs(max_orient);
  for (int i = 0; i < nv; ++i)
    if (i != v1 && i != v2 && i != v3_) {
      const float o = dot(nrm, st.v[i].mk - center);
      if (dm_abs(o) > max_abs) { max_abs = dm_abs(o); max_orient = o; }
    }
  if (max_orient > 0.0f) nrm = -nrm;
  F.n = nrm;
  // safe lower bound of the depth: the smallest projection of the three vertices
  float d1 = dot(nrm, p1), d2 = dot(nrm, p2), d3 = dot(nrm, p3);
  float m2 = DG_FLOAT_MAX;
  d1 = d1 * d1; d2 = d2 * d2; d3 = d3 * d3;
  if (d1 < m2) m2 = d1;
  if (d2 < m2) m2 = d2;
  if (d3 < m2) m2 = d3;
  F.d2 = m2;
  F.map_idx = -1;
  return true;
}
template <class S>
DEV bool dg_attach_face(S& st, GjkCtl& c, int v1, int v2, int v3_, int a1, int a2, int a3) {
  const int n = c.nf;
  c.nf += 1;
  return dg_attach_face_at(st, c.nv, n, v1, v2, v3_, a1, a2, a3);
}

// func_delete_face_from_polytope, epa.py:384-405
template <class S>
DEV void dg_delete_face(S& st, GjkCtl& c, int i_f) {
  const int mi = st.f[i_f].map_idx;
  if (mi >= 0) {
    const int last = st.map[c.nmap - 1];
    st.map[mi] = (short)last;
    st.f[last].map_idx = (short)mi;
    c.nmap -= 1;
  }
  st.f[i_f].map_idx = -2;
}

// func_epa_horizon, epa.py:274-341: depth-first walk over the faces visible from w; the edges towards invisible faces form the horizon
template <class S>
DEV void dg_horizon(S& st, GjkCtl& c, int nearest, V3 w) {
  st.st_f[0] = (short)nearest; st.st_e[0] = 0;
  int top = 1;
  bool first = true;
  while (top > 0) {
    top -= 1;
    const int i_f = st.st_f[top], i_e = st.st_e[top];
    const DgFace& F = st.f[i_f];
    if (!first && F.map_idx == -2) continue;                            // deleted meanwhile
    const bool visible = dot(F.n, w - st.v[F.v[0]].mk) > DG_FLOAT_MIN;
    if (visible || first) {
      dg_delete_face(st, c, i_f);
      for (int k = first ? 0 : 1; k < 3; ++k) {
        const int e2 = (i_e + k) % 3;
        const int adj = st.f[i_f].adj[e2];
        if (st.f[adj].map_idx == -2) continue;
        const int start_v = st.f[i_f].v[(e2 + 1) % 3];                   // adjacent faces wind the other way: enter at the edge's end vertex
        const int adj_e = (st.f[adj].v[0] == start_v) ? 0 : ((st.f[adj].v[1] == start_v) ? 1 : 2);   // func_get_edge_idx, epa.py:362-381
        if (top >= S::CAP_H) { c.overflow = true; return; }
        st.st_f[top] = (short)adj; st.st_e[top] = (short)adj_e;
        top += 1;
      }
    } else {
      if (c.hz_n >= S::CAP_H) { c.overflow = true; return; }
      st.hz_f[c.hz_n] = (short)i_f; st.hz_e[c.hz_n] = (short)i_e;
      c.hz_n += 1;
    }
    first = false;
  }
}

// func_triangle_affine_coords, gjk_utils.py:49-107
DEV V3 dg_affine_coords(V3 p, V3 t1, V3 t2, V3 t3) {
  float ms[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    if (i == 1) { int t = i1; i1 = i2; i2 = t; }
    ms[i] = vget(t2, i1) * vget(t3, i2) - vget(t2, i2) * vget(t3, i1) - vget(t1, i1) * vget(t3, i2) + vget(t1, i2) * vget(t3, i1) +
            vget(t1, i1) * vget(t2, i2) - vget(t1, i2) * vget(t2, i1);
  }
  const float am0 = dm_abs(ms[0]), am1 = dm_abs(ms[1]), am2 = dm_abs(ms[2]);
  float m_max = 0.0f; int ix = 0, iy = 0;
  if (am0 >= am1 && am0 >= am2) { m_max = ms[0]; ix = 1; iy = 2; }
  else if (am1 >= am2 && am1 >= am0) { m_max = ms[1]; ix = 0; iy = 2; }
  else if (am2 >= am0 && am2 >= am1) { m_max = ms[2]; ix = 0; iy = 1; }
  float cs[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    V3 a = (i == 0) ? t2 : ((i == 1) ? t3 : t1), b = (i == 0) ? t3 : ((i == 1) ? t1 : t2);
    cs[i] = vget(p, ix) * vget(a, iy) + vget(p, iy) * vget(b, ix) + vget(a, ix) * vget(b, iy) - vget(p, ix) * vget(b, iy) - vget(p, iy) * vget(a, ix) -
            vget(b, ix) * vget(a, iy);
  }
  return v3(cs[0] / m_max, cs[1] / m_max, cs[2] / m_max);
}
// func_project_origin_to_plane, gjk_utils.py:185-235 (the point only; the witness code ignores the flag)
DEV V3 dg_project_origin(V3 p1, V3 p2, V3 p3) {
  V3 d21 = p2 - p1, d31 = p3 - p1, d32 = p3 - p2;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    V3 n = (i == 0) ? cross(d32, d21) : ((i == 1) ? cross(d21, d31) : cross(d31, d32));
    V3 q = (i == 0) ? p2 : ((i == 1) ? p1 : p3);
    const float nv = dot(n, q), nn = norm_sqr(n);
    if (nn == 0.0f) return v3(0, 0, 0);
    if (nn > DG_FLOAT_MIN) return n * (nv / nn);
    if (i == 2 && !(nn < DG_FLOAT_MIN)) return n * (nv / nn);
  }
  return v3(0, 0, 0);
}

// func_safe_epa, epa.py:970-1181 + func_safe_epa_init :1245-1295 + func_safe_epa_witness :1184-1242.
// Returns the distance (negative = penetration) and the two witness points; 0 distance = no contact found.
template <class S>
DEV float dg_epa(const DgPair& pr, S& st, GjkCtl& c, float eps, V3& w1, V3& w2, bool& has_witness) {
  has_witness = false;
  // polytope = the GJK tetrahedron (its vertices already sit in v[0..3])
  c.nv = 4; c.nf = 0; c.nmap = 0; c.hz_n = 0;
DG_NOUNROLL
  for (int i = 0; i < 4; ++i) {                                          // (vertices | neighbours) of the four faces, 3 bits per index
    const unsigned code = (i == 0) ? 0210u | (0231u << 9) : ((i == 1) ? 0130u | (0032u << 9) : ((i == 2) ? 0320u | (0130u << 9) : 0123u | (0102u << 9)));
    dg_attach_face(st, c, (int)(code & 7u), (int)((code >> 3) & 7u), (int)((code >> 6) & 7u), (int)((code >> 9) & 7u), (int)((code >> 12) & 7u), (int)((code >> 15) & 7u));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { st.map[i] = (short)i; st.f[i].map_idx = (short)i; }
  c.nmap = 4;
  float upper = DG_FLOAT_MAX, upper2 = DG_FLOAT_MAX * DG_FLOAT_MAX, lower = 0.0f;
  const float tol = pr.discrete ? eps : DG_TOLERANCE;
  int nearest = -1;
  for (int k = 0; k < DG_EPA_MAX_IT; ++k) {
    const int prev = nearest;
    float lower2 = DG_FLOAT_MAX * DG_FLOAT_MAX;
    { PHD_BEGIN
    for (int i = 0; i < c.nmap; ++i) {                                   // candidate face closest to the origin
      const int i_f = st.map[i];
      const float d2 = st.f[i_f].d2;
      if (d2 < lower2) { lower2 = d2; nearest = i_f; }
    }
    PHD(56) }
    if (lower2 > upper2 || nearest == -1) { nearest = prev; break; }
    lower = dm_sqrt(lower2);
    const V3 dir = st.f[nearest].n;
    if (c.nv >= S::CAP_V) { c.overflow = true; return 0.0f; }
    const int wi = c.nv;
    st.v[wi] = pr.support(dir / 1.0f);                                   // func_epa_support(dir, dir_norm = 1)
    c.nv += 1;
    const V3 w = st.v[wi].mk;
    const float upper_k = dot(w, dir);
    if (upper_k < upper) { upper = upper_k; upper2 = upper * upper; }
    if ((upper - lower) < tol) break;
    if (pr.discrete) {
      bool repeated = false;
      for (int i = 0; i < c.nv && !repeated; ++i) repeated = (i != wi) && st.v[i].id1 == st.v[wi].id1 && st.v[i].id2 == st.v[wi].id2;
      if (repeated) break;
    }
    { PHD_BEGIN
    dg_horizon(st, c, nearest, w);
    PHD(58) }
    if (c.overflow) return 0.0f;
    if (c.hz_n < 3) { nearest = -1; break; }
    const int nfaces = c.nf, nedges = c.hz_n;
    if (nfaces + nedges >= DG_MAX_FACES) break;                          // the reference's capacity rule (part of the algorithm)
    if (nfaces + nedges > S::CAP_F) { c.overflow = true; return 0.0f; }   // this store is too small: the caller reruns on the full one
    bool ok = true;
    { PHD_BEGIN
    for (int i = 0; i < nedges; ++i) {
      const int f0 = nfaces + i, f1 = nfaces + (i + 1) % nedges;
      const int h_f = st.hz_f[i], h_e = st.hz_e[i];
      const int hv1 = st.f[h_f].v[h_e], hv2 = st.f[h_f].v[(h_e + 1) % 3];
      st.f[h_f].adj[h_e] = (short)f0;
      const int before = (i > 0) ? f0 - 1 : nfaces + nedges - 1;
      ok = dg_attach_face(st, c, wi, hv2, hv1, f1, h_f, before);
      if (!ok) break;
      const float d2 = st.f[c.nf - 1].d2;
      if (d2 >= lower2 - eps && d2 <= upper2 + eps) { st.map[c.nmap] = (short)f0; st.f[f0].map_idx = (short)c.nmap; c.nmap += 1; }
    }
    PHD(60) }
    if (!ok) { nearest = -1; break; }
    c.hz_n = 0;
    if (c.nmap == 0 || nearest == -1) { nearest = -1; break; }
  }
  if (nearest == -1) return 0.0f;
  // witness points: barycentric coordinates of the origin's projection on the nearest face, checked by reprojection
  const DgFace& F = st.f[nearest];
  const DgVert &A = st.v[F.v[0]], &B = st.v[F.v[1]], &Cc = st.v[F.v[2]];
  const V3 proj = dg_project_origin(A.mk, B.mk, Cc.mk);
  const V3 l = dg_affine_coords(proj, A.mk, B.mk, Cc.mk);
  const V3 back = A.mk * l.x + B.mk * l.y + Cc.mk * l.z;
  const float err = norm(proj - back);
  const float e12 = norm_sqr(A.mk - B.mk), e23 = norm_sqr(B.mk - Cc.mk), e31 = norm_sqr(Cc.mk - A.mk);
  const float longest = fmx(fmx(fmx(e12, e23), e31), DG_FLOAT_MIN * DG_FLOAT_MIN);
  if (err * (1.0f / dm_sqrt(longest)) > DG_MAX_REPROJ) return 0.0f;
  w1 = A.o1 * l.x + B.o1 * l.y + Cc.o1 * l.z;
  w2 = A.o2 * l.x + B.o2 * l.y + Cc.o2 * l.z;
  has_witness = true;
  return -dm_sqrt(F.d2);
}

// func_gjk_contact (non-MuJoCo branch), gjk.py:161-437
template <class S>
DEV DgResult dg_contact(const DgPair& pr, S& st, float eps) {
  GjkCtl c; c.nv = c.nf = c.nmap = c.hz_n = c.ns = 0; c.last_searched = 0; c.overflow = false;
  DgResult r; r.is_col = false; r.overflow = false; r.penetration = 0.0f; r.normal = v3(0, 0, 0); r.pos = v3(0, 0, 0);
  PHD_BEGIN
  const bool hit = dg_gjk(pr, st, c, eps);
  PHD(38)
  if (!hit) return r;
  V3 w1 = v3(0, 0, 0), w2 = v3(0, 0, 0); bool has_w = false;
  const float dist = dg_epa(pr, st, c, eps, w1, w2, has_w);
  PHD(42)
  if (c.overflow) { r.overflow = true; return r; }
  if (!(dist < 0.0f) || !has_w) return r;
  const V3 nrm = w2 - w1;
  const float len = norm(nrm);
  if (len < DG_FLOAT_MIN) return r;
  r.is_col = true; r.penetration = -dist; r.normal = nrm / len; r.pos = (w1 + w2) * 0.5f;
  return r;
}

// =============================================================================================================================================
// Team-cooperative form (16 lanes = one DPP row work on ONE query).  The serial query above is a chain of ~10 k dependent instructions on one
// lane; its independent pieces are spread over the lanes here and everything else is executed redundantly by all of them, so the control
// flow of the team stays uniform:
//   * safe support: the plain direction and its 8 nudged variants are evaluated side by side (lane i = candidate i: support points of both
//     geoms, multiplicity, admission test), then the serial loop's choice is taken from ballots -- on the axis-aligned ground box the plain
//     direction always hits a face, and the serial loop walks the candidates one after the other;
//   * GJK: the four faces of the tetrahedron (normal, orientation, signed distance) on four lanes;
//   * EPA: the four initial faces and the faces attached to the horizon of every iteration one per lane, the visibility of all faces from the
//     new vertex one face per lane before the (serial, now arithmetic-free) horizon walk.
// Every value is computed by the same operations as in the serial form, so the answer is bit-identical (tests/test_gjk_epa.py runs both forms
// and the oracle on the same queries).  Lanes exchange through the store (LDS slot or global record) and DPP row broadcasts.
// =============================================================================================================================================
template <int CTRL>
DEV float dgc_dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true)); }
// W lanes cooperate on one query: W = 16 (one DPP row) or W = 4 (one quad: four queries of a 16-lane team side by side)
template <int W, int K> DEV float dgc_bcast(float x) {                   // lane K of my group to all its lanes
  static_assert(W == 4 || W == 16, "groups of 4 or 16 lanes");
  if constexpr (W == 16) return dgc_dpp<0x150 + K>(x);                   // row_newbcast:K
  else return dgc_dpp<K * 0x55>(x);                                      // quad_perm [K, K, K, K]
}
template <int W, int K> DEV V3 dgc_bcast3(V3 a) { return v3(dgc_bcast<W, K>(a.x), dgc_bcast<W, K>(a.y), dgc_bcast<W, K>(a.z)); }
template <int W> DEV unsigned dgc_ballot(bool p) {                         // the W lanes of my group
  if constexpr (W == 16) return (unsigned)((__ballot(p) >> (threadIdx.x & 48)) & 0xFFFFull);
  else return (unsigned)((__ballot(p) >> (threadIdx.x & 60)) & 0xFull);
}
template <int W>
DEV DgVert dgc_shfl_vert(const DgVert& w, int src) {
  DgVert r;
  r.o1 = v3(__shfl(w.o1.x, src, W), __shfl(w.o1.y, src, W), __shfl(w.o1.z, src, W));
  r.o2 = v3(__shfl(w.o2.x, src, W), __shfl(w.o2.y, src, W), __shfl(w.o2.z, src, W));
  r.mk = v3(__shfl(w.mk.x, src, W), __shfl(w.mk.y, src, W), __shfl(w.mk.z, src, W));
  r.id1 = __shfl(w.id1, src, W); r.id2 = __shfl(w.id2, src, W);
  return r;
}
static_assert(sizeof(DgVert) == 44, "DgVert is copied word by word");
template <class S>
DEV void dgc_store_vert(S& st, int idx, const DgVert& w, int tl) { if (tl == 0) st.v[idx] = w; }

// func_safe_gjk_support: the plain direction and its 8 nudged variants, W candidates at a time (lane l of round r = candidate r W + l); the rounds
// stop as soon as the serial loop's choice is known.  `valid` = dg_valid of the returned vertex
template <int W, class S>
DEV DgVert dgc_safe_support(const DgPair& pr, const S& st, int ns, V3 dir, float eps, int tl, bool& valid) {
  DgVert w_last; w_last.o1 = v3(0, 0, 0); w_last.o2 = v3(0, 0, 0); w_last.mk = v3(0, 0, 0); w_last.id1 = -1; w_last.id2 = -1;
  bool valid_last = false;
  for (int i0 = 0; i0 < 9; i0 += W) {
    const int i = (i0 + tl < 9) ? i0 + tl : 8;                           // surplus lanes repeat candidate 8
    V3 nd = dir;
    if (i > 0) {
      const int j = i - 1;
      nd.x += -(1.0f - 2.0f * (float)(j & 1)) * eps;
      nd.y += -(1.0f - 2.0f * (float)(j & 2)) * eps;
      nd.z += -(1.0f - 2.0f * (float)(j & 4)) * eps;
    }
    nd = nd * (2.0f - dot(nd, dir));
    const int n_sup = pr.count(nd);
    const DgVert w = pr.support(nd);
    const bool v = dg_valid(st, ns, w);
    const unsigned lanes = (9 - i0 < W) ? ((1u << (9 - i0)) - 1u) : ((1u << W) - 1u);   // lanes holding distinct candidates of this round
    const unsigned ok1 = dgc_ballot<W>(n_sup <= 1) & lanes, vmask = dgc_ballot<W>(v);
    if (i0 == 0 && (ok1 & 1u)) { valid = (vmask & 1u) != 0u; return dgc_shfl_vert<W>(w, 0); }   // the plain direction has a unique support point
    const unsigned last_bit = (i0 + W > 8) ? (1u << (8 - i0)) : 0u;      // candidate 8 is taken without the admission test
    const unsigned good = ok1 & (vmask | last_bit) & (i0 == 0 ? ~1u : ~0u);
    if (good) { const int sel = __ffs((int)good) - 1; valid = ((vmask >> sel) & 1u) != 0u; return dgc_shfl_vert<W>(w, sel); }
    const unsigned comp = (ok1 & (i0 == 0 ? ~1u : ~0u)) | (i0 == 0 ? 1u : 0u);   // support points the serial loop evaluated in this round
    if (comp) { const int sel = 31 - __clz((int)comp); w_last = dgc_shfl_vert<W>(w, sel); valid_last = ((vmask >> sel) & 1u) != 0u; }
  }
  valid = valid_last;                                                    // the loop ran out: the support point evaluated last
  return w_last;
}

// func_search_valid_simplex_vertex: rare (the plain safe support was not admissible); the triangle case uses the cooperative safe support
template <int W, class S>
DEV bool dgc_search_vertex(const DgPair& pr, const S& st, GjkCtl& c, DgVert& w, float eps, int tl) {
  if (pr.discrete) return dg_search_vertex(pr, st, c, w, eps);           // box - box vertex walk: executed redundantly by all lanes
  w.o1 = v3(0, 0, 0); w.o2 = v3(0, 0, 0); w.mk = v3(0, 0, 0); w.id1 = -1; w.id2 = -1;
  if (c.ns == 3) {
    V3 a = st.v[0].mk, b = st.v[1].mk, cc = st.v[2].mk;
    V3 nrm = cross(cc - a, b - a);
    V3 dir = nrm / norm(nrm);
    for (int i = 0; i < 2; ++i) {
      bool ok;
      w = dgc_safe_support<W>(pr, st, c.ns, (i == 0) ? dir : -dir, eps, tl, ok);
      if (ok) return true;
    }
  }
  return false;
}

// func_safe_gjk
template <int W, class S>
DEV bool dgc_gjk(const DgPair& pr, S& st, GjkCtl& c, float eps, int tl) {
  c.ns = 0;
  V3 best_n = v3(0, 0, 0);
  for (int step = 0; step < 4 + DG_GJK_MAX_IT; ++step) {
    const bool init = step < 4;
    V3 dir = best_n;
    if (init) {
      dir = v3(0, 0, 0);
      const float sgn = 1.0f - 2.0f * (float)(step % 2);
      if (step < 2) dir.z = sgn; else dir.y = sgn;
    } else {
      // face j = lane & 3: outward normal and signed distance (origin inside => positive); face j is opposite to vertex j
      const int j = tl & 3;
      const int a = (j == 0) ? 2 : ((j == 1) ? 0 : ((j == 2) ? 1 : 0));
      const int b = (j == 0) ? 1 : ((j == 1) ? 2 : ((j == 2) ? 0 : 1));
      const int cidx = (j == 3) ? 2 : 3;
      V3 va = st.v[a].mk, vb = st.v[b].mk, vc = st.v[cidx].mk, apex = st.v[j].mk;
      V3 nrm = cross(vc - va, vb - va);
      nrm = nrm / norm(nrm);
      if (dot(nrm, apex - va) > 0.0f) nrm = -nrm;
      const float sd = dot(nrm, va);
      const float sd0 = dgc_bcast<W, 0>(sd), sd1 = dgc_bcast<W, 1>(sd), sd2 = dgc_bcast<W, 2>(sd), sd3 = dgc_bcast<W, 3>(sd);
      const V3 n0 = dgc_bcast3<W, 0>(nrm), n1 = dgc_bcast3<W, 1>(nrm), n2 = dgc_bcast3<W, 2>(nrm), n3 = dgc_bcast3<W, 3>(nrm);
      float best_sd = sd0; int best = 0; best_n = n0;
      if (sd1 < best_sd) { best_sd = sd1; best_n = n1; best = 1; }
      if (sd2 < best_sd) { best_sd = sd2; best_n = n2; best = 2; }
      if (sd3 < best_sd) { best_sd = sd3; best_n = n3; best = 3; }
      if (best_sd >= 0.0f) return true;                                  // INTERSECT
      c.ns = 3;
      if (best != 3) {                                                   // drop the vertex opposite to the worst face: 11 words spread over the lanes
        const float* src = (const float*)&st.v[3];
        float* dst = (float*)&st.v[best];
        float word[(11 + W - 1) / W];
#pragma unroll
        for (int k = 0; k < (11 + W - 1) / W; ++k) word[k] = src[(k * W + tl < 11) ? k * W + tl : 10];
#pragma unroll
        for (int k = 0; k < (11 + W - 1) / W; ++k) if (k * W + tl < 11) dst[k * W + tl] = word[k];
      }
      team_sync();
      dir = best_n;
    }
    bool valid;
    DgVert w = dgc_safe_support<W>(pr, st, c.ns, dir, eps, tl, valid);
    if (init) {
      if (!valid && !dgc_search_vertex<W>(pr, st, c, w, eps, tl)) return false;
      dgc_store_vert(st, step, w, tl);
      c.ns += 1;
    } else {
      if (!valid) return false;                                          // SEPARATED (duplicate) or NUM_ERROR (degenerate), both end the query
      if (dot(w.mk, best_n) < 0.0f) return false;                        // the origin is outside the Minkowski difference
      dgc_store_vert(st, 3, w, tl);
      c.ns = 4;
    }
    team_sync();
  }
  return false;
}

// func_epa_horizon with the visibility of every face precomputed (DgFace::pad, written one face per lane)
template <class S>
DEV void dgc_horizon(S& st, GjkCtl& c, int nearest) {
  st.st_f[0] = (short)nearest; st.st_e[0] = 0;
  int top = 1;
  bool first = true;
  while (top > 0) {
    top -= 1;
    const int i_f = st.st_f[top], i_e = st.st_e[top];
    const DgFace& F = st.f[i_f];
    if (!first && F.map_idx == -2) continue;
    const bool visible = F.pad != 0;
    if (visible || first) {
      dg_delete_face(st, c, i_f);
      for (int k = first ? 0 : 1; k < 3; ++k) {
        const int e2 = (i_e + k) % 3;
        const int adj = st.f[i_f].adj[e2];
        if (st.f[adj].map_idx == -2) continue;
        const int start_v = st.f[i_f].v[(e2 + 1) % 3];
        const int adj_e = (st.f[adj].v[0] == start_v) ? 0 : ((st.f[adj].v[1] == start_v) ? 1 : 2);
        if (top >= S::CAP_H) { c.overflow = true; return; }
        st.st_f[top] = (short)adj; st.st_e[top] = (short)adj_e;
        top += 1;
      }
    } else {
      if (c.hz_n >= S::CAP_H) { c.overflow = true; return; }
      st.hz_f[c.hz_n] = (short)i_f; st.hz_e[c.hz_n] = (short)i_e;
      c.hz_n += 1;
    }
    first = false;
  }
}

// func_safe_epa + _init + _witness
template <int W, class S>
DEV float dgc_epa(const DgPair& pr, S& st, GjkCtl& c, float eps, V3& w1, V3& w2, bool& has_witness, int tl) {
  has_witness = false;
  c.nv = 4; c.nf = 0; c.nmap = 0; c.hz_n = 0;
  {                                                                      // the four faces of the tetrahedron, one per lane
    const int i = tl & 3;
    const unsigned code = (i == 0) ? 0210u | (0231u << 9) : ((i == 1) ? 0130u | (0032u << 9) : ((i == 2) ? 0320u | (0130u << 9) : 0123u | (0102u << 9)));
    if (tl < 4) {
      dg_attach_face_at(st, 4, i, (int)(code & 7u), (int)((code >> 3) & 7u), (int)((code >> 6) & 7u), (int)((code >> 9) & 7u), (int)((code >> 12) & 7u), (int)((code >> 15) & 7u));
      st.map[i] = (short)i; st.f[i].map_idx = (short)i;
    }
    c.nf = 4; c.nmap = 4;
  }
  team_sync();
  float upper = DG_FLOAT_MAX, upper2 = DG_FLOAT_MAX * DG_FLOAT_MAX, lower = 0.0f;
  const float tol = pr.discrete ? eps : DG_TOLERANCE;
  int nearest = -1;
  for (int k = 0; k < DG_EPA_MAX_IT; ++k) {
    const int prev = nearest;
    float lower2 = DG_FLOAT_MAX * DG_FLOAT_MAX;
    for (int i = 0; i < c.nmap; ++i) {                                   // candidate face closest to the origin
      const int i_f = st.map[i];
      const float d2 = st.f[i_f].d2;
      if (d2 < lower2) { lower2 = d2; nearest = i_f; }
    }
    if (lower2 > upper2 || nearest == -1) { nearest = prev; break; }
    lower = dm_sqrt(lower2);
    const V3 dir = st.f[nearest].n;
    if (c.nv >= S::CAP_V) { c.overflow = true; return 0.0f; }
    const int wi = c.nv;
    const DgVert wv = pr.support(dir / 1.0f);
    dgc_store_vert(st, wi, wv, tl);
    c.nv += 1;
    const V3 w = wv.mk;
    const float upper_k = dot(w, dir);
    if (upper_k < upper) { upper = upper_k; upper2 = upper * upper; }
    if ((upper - lower) < tol) { team_sync(); break; }
    if (pr.discrete) {
      bool repeated = false;
      for (int i = 0; i < wi && !repeated; ++i) repeated = st.v[i].id1 == wv.id1 && st.v[i].id2 == wv.id2;
      if (repeated) { team_sync(); break; }
    }
    for (int f0 = 0; f0 < c.nf; f0 += W) {                               // visibility of every face from w, one face per lane
      const int f = f0 + tl;
      if (f < c.nf) { DgFace& F = st.f[f]; F.pad = (short)(dot(F.n, w - st.v[F.v[0]].mk) > DG_FLOAT_MIN ? 1 : 0); }
    }
    team_sync();
    dgc_horizon(st, c, nearest);
    if (c.overflow) return 0.0f;
    if (c.hz_n < 3) { nearest = -1; break; }
    const int nfaces = c.nf, nedges = c.hz_n;
    if (nfaces + nedges >= DG_MAX_FACES) break;
    if (nfaces + nedges > S::CAP_F) { c.overflow = true; return 0.0f; }
    team_sync();
    bool ok = true;
    int nmap = c.nmap;
    for (int e0 = 0; e0 < nedges; e0 += W) {                             // one new face per horizon edge, one per lane
      const int i = e0 + tl;
      bool ok_i = true, incl = false;
      if (i < nedges) {
        const int f0 = nfaces + i, f1 = nfaces + (i + 1) % nedges;
        const int h_f = st.hz_f[i], h_e = st.hz_e[i];
        const int hv1 = st.f[h_f].v[h_e], hv2 = st.f[h_f].v[(h_e + 1) % 3];
        st.f[h_f].adj[h_e] = (short)f0;
        const int before = (i > 0) ? f0 - 1 : nfaces + nedges - 1;
        ok_i = dg_attach_face_at(st, c.nv, f0, wi, hv2, hv1, f1, h_f, before);
        if (ok_i) { const float d2 = st.f[f0].d2; incl = d2 >= lower2 - eps && d2 <= upper2 + eps; }
      }
      const unsigned okm = dgc_ballot<W>(ok_i), inm = dgc_ballot<W>(incl);
      if (okm != ((1u << W) - 1u)) {                                     // the serial loop stops at the first face without a normal: the query ends without a contact
        ok = false;
        break;
      }
      if (incl) { const int pos = nmap + __popc(inm & ((1u << tl) - 1u)); st.map[pos] = (short)(nfaces + i); st.f[nfaces + i].map_idx = (short)pos; }
      nmap += __popc(inm);
    }
    c.nf = nfaces + nedges; c.nmap = nmap;
    team_sync();
    if (!ok) { nearest = -1; break; }
    c.hz_n = 0;
    if (c.nmap == 0 || nearest == -1) { nearest = -1; break; }
  }
  if (nearest == -1) return 0.0f;
  const DgFace& F = st.f[nearest];
  const DgVert &A = st.v[F.v[0]], &B = st.v[F.v[1]], &Cc = st.v[F.v[2]];
  const V3 proj = dg_project_origin(A.mk, B.mk, Cc.mk);
  const V3 l = dg_affine_coords(proj, A.mk, B.mk, Cc.mk);
  const V3 back = A.mk * l.x + B.mk * l.y + Cc.mk * l.z;
  const float err = norm(proj - back);
  const float e12 = norm_sqr(A.mk - B.mk), e23 = norm_sqr(B.mk - Cc.mk), e31 = norm_sqr(Cc.mk - A.mk);
  const float longest = fmx(fmx(fmx(e12, e23), e31), DG_FLOAT_MIN * DG_FLOAT_MIN);
  if (err * (1.0f / dm_sqrt(longest)) > DG_MAX_REPROJ) return 0.0f;
  w1 = A.o1 * l.x + B.o1 * l.y + Cc.o1 * l.z;
  w2 = A.o2 * l.x + B.o2 * l.y + Cc.o2 * l.z;
  has_witness = true;
  return -dm_sqrt(F.d2);
}

// func_gjk_contact, executed by the W lanes of a group together (tl = lane within the group); every lane returns the same result
template <int W, class S>
DEV DgResult dgc_contact(const DgPair& pr, S& st, float eps, int tl) {
  GjkCtl c; c.nv = c.nf = c.nmap = c.hz_n = c.ns = 0; c.last_searched = 0; c.overflow = false;
  DgResult r; r.is_col = false; r.overflow = false; r.penetration = 0.0f; r.normal = v3(0, 0, 0); r.pos = v3(0, 0, 0);
  PHD_BEGIN
  const bool hit = dgc_gjk<W>(pr, st, c, eps, tl);
  team_sync();
  PHD(38)
  if (!hit) return r;
  V3 w1 = v3(0, 0, 0), w2 = v3(0, 0, 0); bool has_w = false;
  const float dist = dgc_epa<W>(pr, st, c, eps, w1, w2, has_w, tl);
  team_sync();
  PHD(42)
  if (c.overflow) { r.overflow = true; return r; }
  if (!(dist < 0.0f) || !has_w) return r;
  const V3 nrm = w2 - w1;
  const float len = norm(nrm);
  if (len < DG_FLOAT_MIN) return r;
  r.is_col = true; r.penetration = -dist; r.normal = nrm / len; r.pos = (w1 + w2) * 0.5f;
  return r;
}
typedef GjkStore<20, 40, 20> GjkStoreTeam;                              // the one polytope a cooperating team works on (2.4 KB of LDS)

#endif  // GO2SIM_GJK_DEV_H
