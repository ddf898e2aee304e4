#!/usr/bin/env python3
"""Offline model compiler: Go2 URDF + plane URDF -> constant model tables (JSON).

This is the host-side replacement for the reference's model-ingest stack on the hot path
(SURVEY.md section 7 step 0).  It re-derives, in float64 numpy, what the reference obtains from
urdfpy + MuJoCo + trimesh for exactly two assets (plane.urdf, go2.urdf):

  * fixed-link merging incl. inertia composition       (genesis/utils/urdf.py:396-537)
  * principal-axis inertial frames (MuJoCo `fullinertia` -> body_ipos/body_iquat/body_inertia;
    algorithm = MuJoCo's published `mju_eig3` Jacobi iteration, mujoco>=3.2.5, restated here;
    call sites genesis/utils/mjcf.py:238-249)                      ** parity unpinned **
  * breadth-first link ordering                        (genesis/utils/urdf.py:52-90)
  * free root joint insertion, root_idx, armature      (rigid_entity.py:646-722, mjcf.py:188-190)
  * solver-parameter sanitisation                      (rigid_solver.py:188-215)
  * collision pair table                               (collider/collider.py:220-329)
    (the voxel-based neutral-pose filter of :293-315 is replaced by an exact convex overlap
    test at qpos0 on 0.1%-shrunk primitives)                        ** parity unpinned **
  * primitive init-AABBs and the cylinder support-direction table
    (mjcf.py:509-550, support_field.py:22-89; trimesh 32-section cylinder restated)
                                                                    ** parity unpinned **
  * links/dofs invweight and meaninertia at qpos0      (rigid_solver.py:529-682, abd/misc.py:103-123)

Input : the two URDF data files under /root/reference/genesis/assets (read as data only).
Output: go2_sim2real_locomotion_rl_amd/model/go2_model.json (committed; the GPU box never sees
        /root/reference).

Run:  python tools/compile_go2_model.py [--assets DIR] [--out FILE]
"""
import argparse
import hashlib
import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

GEOM_SPHERE, GEOM_CYLINDER, GEOM_BOX = 1, 3, 5  # genesis/constants.py:18-27
JOINT_FIXED, JOINT_REVOLUTE, JOINT_FREE = 0, 1, 4  # genesis/constants.py:31-36
EPS32 = float(np.finfo(np.float32).eps)  # gs.EPS for precision="32" (genesis/__init__.py:229)


# --------------------------------------------------------------------------------------
# small math
# --------------------------------------------------------------------------------------
def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def T_from(xyz, rpy):
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(rpy)
    T[:3, 3] = xyz
    return T


def R_to_quat(R):
    # (w,x,y,z), standard branch-on-trace conversion
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def quat_to_R(q):
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ]
    )


def mju_eig3(mat):
    """Eigen-decomposition of a symmetric 3x3 matrix the way MuJoCo does it for `fullinertia`
    (quaternion-accumulated Jacobi iteration, eigenvalues sorted in DEcreasing order).
    Returns (eigval[3], quat[4]).  Restated from MuJoCo's published algorithm; parity unpinned."""
    eig_eps = 1e-12
    quat = np.array([1.0, 0.0, 0.0, 0.0])
    eigval = np.zeros(3)
    for _ in range(500):
        E = quat_to_R(quat)
        D = E.T @ mat @ E
        eigval = np.array([D[0, 0], D[1, 1], D[2, 2]])
        if abs(D[0, 1]) > abs(D[0, 2]) and abs(D[0, 1]) > abs(D[1, 2]):
            rk, ck, rotk = 0, 1, 2
        elif abs(D[0, 2]) > abs(D[1, 2]):
            rk, ck, rotk = 0, 2, 1
        else:
            rk, ck, rotk = 1, 2, 0
        if abs(D[rk, ck]) < eig_eps:
            break
        tau = (D[ck, ck] - D[rk, rk]) / (2 * D[rk, ck])
        if tau >= 0:
            t = 1.0 / (tau + math.sqrt(1 + tau * tau))
        else:
            t = -1.0 / (-tau + math.sqrt(1 + tau * tau))
        c = 1.0 / math.sqrt(1 + t * t)
        if c > 1.0 - eig_eps:
            break
        tmp = np.zeros(4)
        tmp[rotk + 1] = -math.sqrt(0.5 - 0.5 * c) if tau >= 0 else math.sqrt(0.5 - 0.5 * c)
        if rotk == 1:
            tmp[rotk + 1] = -tmp[rotk + 1]
        tmp[0] = math.sqrt(1.0 - tmp[rotk + 1] ** 2)
        tmp /= np.linalg.norm(tmp)
        quat = quat_mul(quat, tmp)
        quat /= np.linalg.norm(quat)
    for j in range(3):
        j1 = j % 2
        if eigval[j1] < eigval[j1 + 1]:
            eigval[j1], eigval[j1 + 1] = eigval[j1 + 1], eigval[j1]
            tmp = np.zeros(4)
            tmp[0] = 0.707106781186548
            tmp[(j1 + 2) % 3 + 1] = tmp[0]
            quat = quat_mul(quat, tmp)
            quat /= np.linalg.norm(quat)
    return eigval, quat


# --------------------------------------------------------------------------------------
# URDF ingest + fixed-link merge (genesis/utils/urdf.py:396-569)
# --------------------------------------------------------------------------------------
def _floats(s):
    return [float(v) for v in s.split()]


def load_urdf(path):
    root = ET.fromstring(path) if path.lstrip().startswith("<") else ET.parse(path).getroot()       # a file, or URDF text (the synthetic shape models)
    links, joints = [], []
    for l in root.findall("link"):
        L = {"name": l.get("name"), "inertial": None, "collisions": []}
        ine = l.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            xyz = _floats(o.get("xyz", "0 0 0")) if o is not None else [0, 0, 0]
            rpy = _floats(o.get("rpy", "0 0 0")) if o is not None else [0, 0, 0]
            a = ine.find("inertia").attrib
            I = np.array(
                [
                    [float(a["ixx"]), float(a["ixy"]), float(a["ixz"])],
                    [float(a["ixy"]), float(a["iyy"]), float(a["iyz"])],
                    [float(a["ixz"]), float(a["iyz"]), float(a["izz"])],
                ]
            )
            L["inertial"] = {"origin": T_from(xyz, rpy), "mass": float(ine.find("mass").get("value")), "inertia": I}
        for c in l.findall("collision"):
            o = c.find("origin")
            xyz = _floats(o.get("xyz", "0 0 0")) if o is not None else [0, 0, 0]
            rpy = _floats(o.get("rpy", "0 0 0")) if o is not None else [0, 0, 0]
            g = list(c.find("geometry"))[0]
            if g.tag == "box":
                geom = {"type": GEOM_BOX, "data": _floats(g.get("size"))}
            elif g.tag == "cylinder":
                geom = {"type": GEOM_CYLINDER, "data": [float(g.get("radius")), float(g.get("length"))]}
            elif g.tag == "sphere":
                geom = {"type": GEOM_SPHERE, "data": [float(g.get("radius"))]}
            else:
                raise ValueError(f"unsupported collision geometry {g.tag}")
            geom["origin"] = T_from(xyz, rpy)
            L["collisions"].append(geom)
        links.append(L)
    for j in root.findall("joint"):
        o = j.find("origin")
        J = {
            "name": j.get("name"),
            "type": j.get("type"),
            "parent": j.find("parent").get("link"),
            "child": j.find("child").get("link"),
            "origin": T_from(_floats(o.get("xyz", "0 0 0")), _floats(o.get("rpy", "0 0 0"))),
            "axis": _floats(j.find("axis").get("xyz")) if j.find("axis") is not None else [1, 0, 0],
            "limit": None,
            "damping": float(j.find("dynamics").get("damping", "0")) if j.find("dynamics") is not None else 0.0,      # urdf.py:326-330
            "frictionloss": float(j.find("dynamics").get("friction", "0")) if j.find("dynamics") is not None else 0.0,
        }
        lim = j.find("limit")
        if lim is not None:
            J["limit"] = {   # absent bounds = unbounded (genesis/utils/urdf.py:269-274: `lower if lower is not None else -inf`)
                "lower": float(lim.get("lower")) if lim.get("lower") is not None else -1e30,
                "upper": float(lim.get("upper")) if lim.get("upper") is not None else 1e30,
                "effort": float(lim.get("effort")) if lim.get("effort") is not None else None,
            }
        joints.append(J)
    return links, joints


def _translate_inertia(I, m, dist):
    d2 = float(np.dot(dist, dist))
    return I + m * (d2 * np.eye(3) - np.outer(dist, dist))


def merge_fixed_links(links, joints):
    links = list(links)
    joints = list(joints)
    name_to_idx = {l["name"]: i for i, l in enumerate(links)}
    merged = {}
    while True:
        found = False
        for joint in joints:
            if joint["type"] != "fixed":
                continue
            parent_name, child_name = joint["parent"], joint["child"]
            while parent_name in merged:
                parent_name = merged[parent_name]
            while child_name in merged:
                child_name = merged[child_name]
            pi, ci = name_to_idx.get(parent_name), name_to_idx.get(child_name)
            if pi is None or ci is None:
                continue
            parent, child = links[pi], links[ci]
            merged[joint["child"]] = parent_name
            for k in merged:
                if merged[k] == child_name:
                    merged[k] = parent_name
            # update_subtree
            T = joint["origin"]
            if child["inertial"] is not None:
                child["inertial"]["origin"] = T @ child["inertial"]["origin"]
            for g in child["collisions"]:
                g["origin"] = T @ g["origin"]
            for j2 in joints:
                if j2["parent"] == child["name"]:
                    j2["origin"] = T @ j2["origin"]
            # merge_inertia
            if child["inertial"] is not None:
                if parent["inertial"] is None:
                    parent["inertial"] = child["inertial"]
                else:
                    i1, i2 = parent["inertial"], child["inertial"]
                    m1, m2 = i1["mass"], i2["mass"]
                    c1, c2 = i1["origin"][:3, 3].copy(), i2["origin"][:3, 3].copy()
                    R1, R2 = i1["origin"][:3, :3], i2["origin"][:3, :3]
                    m = m1 + m2
                    com = (m1 * c1 + m2 * c2) / m if m > 0 else c1
                    I1 = _translate_inertia(R1 @ i1["inertia"] @ R1.T, m1, com - c1)
                    I2 = _translate_inertia(R2 @ i2["inertia"] @ R2.T, m2, com - c2)
                    i1["mass"] = m
                    i1["origin"] = np.eye(4)
                    i1["origin"][:3, 3] = com
                    i1["inertia"] = I1 + I2
            parent["collisions"].extend(child["collisions"])
            links.pop(ci)
            joints.remove(joint)
            name_to_idx = {l["name"]: i for i, l in enumerate(links)}
            found = True
            break
        if not found:
            break
    for joint in joints:
        if joint["parent"] in merged:
            joint["parent"] = merged[joint["parent"]]
        if joint["child"] in merged:
            joint["child"] = merged[joint["child"]]
    return links, joints


def bfs_order(links, joints):
    """Depth-first body creation (MuJoCo URDF import) then breadth-first re-ordering
    (genesis/utils/urdf.py:52-90).  Returns link names in final order + parent map."""
    parent_of = {j["child"]: j["parent"] for j in joints}
    children = {l["name"]: [] for l in links}
    for j in joints:  # joint order defines child order
        children[j["parent"]].append(j["child"])
    roots = [l["name"] for l in links if l["name"] not in parent_of]
    dfs = []

    def visit(n):
        dfs.append(n)
        for c in children[n]:
            visit(c)

    for r in roots:
        visit(r)
    order, level = [], [n for n in dfs if n not in parent_of]
    while level:
        order.extend(level)
        nxt = []
        for n in level:
            nxt.extend([c for c in dfs if parent_of.get(c) == n])
        level = nxt
    return order, parent_of


# --------------------------------------------------------------------------------------
# primitive helpers (trimesh restatement for AABBs / cylinder vertex ring)
# --------------------------------------------------------------------------------------
CYL_SECTIONS = 32  # trimesh.creation.cylinder default (revolve, 32 sections)
SUPPORT_RES = 180  # support_field.py:18


def cylinder_ring(radius):
    k = np.arange(CYL_SECTIONS)
    th = np.linspace(0.0, 2 * np.pi, CYL_SECTIONS + 1)[:-1]
    return np.stack([radius * np.cos(th), radius * np.sin(th)], axis=1).astype(np.float32), k


def support_theta_table():
    """For each of the 180 azimuth grid cells (support_field.py:22-35) the index of the cylinder
    ring vertex that maximises the dot product with the grid direction (float64 argmax as numpy does
    in support_field.py:59-62).  Independent of radius/height (vertices form a product set)."""
    theta = np.arange(SUPPORT_RES) / SUPPORT_RES * 2 * math.pi - math.pi
    ring, _ = cylinder_ring(1.0)
    ring = ring.astype(np.float64)
    dots = np.cos(theta)[:, None] * ring[None, :, 0] + np.sin(theta)[:, None] * ring[None, :, 1]
    srt = np.sort(dots, axis=1)
    assert (srt[:, -1] - srt[:, -2]).min() > 1e-6, "tie in cylinder support table"
    return np.argmax(dots, axis=1).astype(int)


def geom_half_extents(g):
    if g["type"] == GEOM_BOX:
        return np.array(g["data"][:3]) / 2
    if g["type"] == GEOM_CYLINDER:
        return np.array([g["data"][0], g["data"][0], g["data"][1] / 2])
    if g["type"] == GEOM_SPHERE:
        return np.array([g["data"][0]] * 3)
    raise ValueError


def aabb_corners(h):
    lo, up = -h, h
    return [
        [lo[0], lo[1], lo[2]],
        [lo[0], lo[1], up[2]],
        [lo[0], up[1], lo[2]],
        [lo[0], up[1], up[2]],
        [up[0], lo[1], lo[2]],
        [up[0], lo[1], up[2]],
        [up[0], up[1], lo[2]],
        [up[0], up[1], up[2]],
    ]


def support_analytic(g, R, p, d, shrink=1.0):
    dl = R.T @ d
    h = geom_half_extents(g) * shrink
    if g["type"] == GEOM_BOX:
        v = np.where(dl < 0, -h, h)
    elif g["type"] == GEOM_SPHERE:
        v = dl / (np.linalg.norm(dl) + 1e-300) * h[0]
    else:
        rxy = math.hypot(dl[0], dl[1])
        v = np.array([0.0, 0.0, h[2] if dl[2] >= 0 else -h[2]])
        if rxy > 1e-14:
            v[0], v[1] = h[0] * dl[0] / rxy, h[0] * dl[1] / rxy
    return R @ v + p


def convex_overlap(ga, Ra, pa, gb, Rb, pb, shrink):
    """Boolean GJK (float64) between two convex primitives, used only for the neutral-pose filter."""

    def sup(d):
        return support_analytic(ga, Ra, pa, d, shrink) - support_analytic(gb, Rb, pb, -d, shrink)

    d = pa - pb
    if np.linalg.norm(d) < 1e-12:
        d = np.array([1.0, 0.0, 0.0])
    simplex = [sup(d)]
    d = -simplex[0]
    for _ in range(64):
        if np.linalg.norm(d) < 1e-14:
            return True
        a = sup(d)
        if a @ d < 0:
            return False
        simplex.append(a)
        # nearest-simplex update
        if len(simplex) == 2:
            b, a = simplex
            ab, ao = b - a, -a
            if ab @ ao > 0:
                d = np.cross(np.cross(ab, ao), ab)
            else:
                simplex, d = [a], ao
        elif len(simplex) == 3:
            c, b, a = simplex
            ab, ac, ao = b - a, c - a, -a
            abc = np.cross(ab, ac)
            if np.cross(abc, ac) @ ao > 0:
                if ac @ ao > 0:
                    simplex, d = [c, a], np.cross(np.cross(ac, ao), ac)
                else:
                    simplex, d = [b, a], np.cross(np.cross(ab, ao), ab) if ab @ ao > 0 else ao
                    if not (ab @ ao > 0):
                        simplex = [a]
            elif np.cross(ab, abc) @ ao > 0:
                if ab @ ao > 0:
                    simplex, d = [b, a], np.cross(np.cross(ab, ao), ab)
                else:
                    simplex, d = [a], ao
            else:
                if abc @ ao > 0:
                    d = abc
                else:
                    simplex, d = [b, c, a], -abc
        else:
            dd, c, b, a = simplex
            ab, ac, ad, ao = b - a, c - a, dd - a, -a
            abc, acd, adb = np.cross(ab, ac), np.cross(ac, ad), np.cross(ad, ab)
            if abc @ ao > 0:
                simplex, d = [c, b, a], abc
            elif acd @ ao > 0:
                simplex, d = [dd, c, a], acd
            elif adb @ ao > 0:
                simplex, d = [b, dd, a], adb
            else:
                return True
    return True


# --------------------------------------------------------------------------------------
# float64 kinematics + CRB mass matrix at qpos0 (for invweight / meaninertia only)
# --------------------------------------------------------------------------------------
def fk_mass_matrix(model):
    L, D = model["links"], model["dofs"]
    nl, nd = len(L), len(D)
    qpos0 = np.array(model["qpos0"])
    pos, quat = [None] * nl, [None] * nl
    for i, l in enumerate(L):
        if l["parent"] < 0:
            if l["n_dofs"] == 6:
                pos[i], quat[i] = qpos0[l["q_start"] : l["q_start"] + 3], qpos0[l["q_start"] + 3 : l["q_start"] + 7]
            else:
                pos[i], quat[i] = np.array(l["pos"]), np.array(l["quat"])
        else:
            p = l["parent"]
            Rp = quat_to_R(quat[p])
            pos[i] = pos[p] + Rp @ np.array(l["pos"])
            quat[i] = quat_mul(quat[p], np.array(l["quat"]))  # joint angle 0 at qpos0
    # COM per root
    mass_sum, com = {}, {}
    ipos = []
    for i, l in enumerate(L):
        ip = pos[i] + quat_to_R(quat[i]) @ np.array(l["inertial_pos"])
        ipos.append(ip)
        r = l["root"]
        mass_sum[r] = mass_sum.get(r, 0.0) + l["inertial_mass"]
        com[r] = com.get(r, np.zeros(3)) + l["inertial_mass"] * ip
    root_com = [com[l["root"]] / mass_sum[l["root"]] for l in L]
    # cdof
    cdof_ang, cdof_vel = np.zeros((nd, 3)), np.zeros((nd, 3))
    for i, l in enumerate(L):
        if l["n_dofs"] == 0:
            continue
        ds = l["dof_start"]
        R = quat_to_R(quat[i])
        if l["n_dofs"] == 6:
            off = root_com[i] - pos[i]
            for k in range(3):
                cdof_vel[ds + k, k] = 1.0
                cdof_ang[ds + 3 + k] = R[:, k]
                cdof_vel[ds + 3 + k] = np.cross(R[:, k], off)
        else:
            axis = R @ np.array(D[ds]["motion_ang"])
            off = root_com[i] - pos[i]  # joint anchor == link origin (joint pos 0)
            cdof_ang[ds] = axis
            cdof_vel[ds] = np.cross(axis, off)
    # composite inertia about root COM, world axes
    crb_I, crb_p, crb_m = [], [], []
    for i, l in enumerate(L):
        Ri = quat_to_R(quat_mul(quat[i], np.array(l["inertial_quat"])))
        t = ipos[i] - root_com[i]
        I = Ri @ np.array(l["inertial_i"]) @ Ri.T + l["inertial_mass"] * (t @ t * np.eye(3) - np.outer(t, t))
        crb_I.append(I)
        crb_p.append(t * l["inertial_mass"])
        crb_m.append(l["inertial_mass"])
    for i in range(nl - 1, -1, -1):
        p = L[i]["parent"]
        if p >= 0:
            crb_I[p] = crb_I[p] + crb_I[i]
            crb_p[p] = crb_p[p] + crb_p[i]
            crb_m[p] = crb_m[p] + crb_m[i]
    f_ang, f_vel = np.zeros((nd, 3)), np.zeros((nd, 3))
    for i, l in enumerate(L):
        for d in range(l["dof_start"], l["dof_end"]):
            f_ang[d] = crb_I[i] @ cdof_ang[d] + np.cross(crb_p[i], cdof_vel[d])
            f_vel[d] = crb_m[i] * cdof_vel[d] - np.cross(crb_p[i], cdof_ang[d])
    mask = np.array(model["mass_parent_mask"]).reshape(nd, nd)
    M = np.zeros((nd, nd))
    for i in range(nd):
        for j in range(nd):
            M[i, j] = (f_ang[i] @ cdof_ang[j] + f_vel[i] @ cdof_vel[j]) * mask[i, j]
    for i in range(nd):
        for j in range(i + 1, nd):
            M[i, j] = M[j, i]
    for i in range(nd):
        M[i, i] += D[i]["armature"]
    offsets = [ipos[i] - root_com[i] for i in range(nl)]
    return M, cdof_ang, cdof_vel, offsets


def compute_invweight(model):
    M, cdof_ang, cdof_vel, offsets = fk_mass_matrix(model)
    L, D = model["links"], model["dofs"]
    nl, nd = len(L), len(D)
    s = model["robot_dof_start"]
    Minv = np.zeros((nd, nd))
    Minv[s:, s:] = np.linalg.inv(M[s:, s:])
    for i, l in enumerate(L):
        jacp, jacr = np.zeros((3, nd)), np.zeros((3, nd))
        j = i
        while j != -1:
            for d in range(L[j]["dof_start"], L[j]["dof_end"]):
                jacp[:, d] = cdof_vel[d] + np.cross(cdof_ang[d], offsets[i])
                jacr[:, d] = cdof_ang[d]
            j = L[j]["parent"]
        A = np.concatenate([jacp, jacr]) @ Minv @ np.concatenate([jacp, jacr]).T
        dg = np.diag(A)
        l["invweight"] = [float(dg[:3].mean()), float(dg[3:].mean())]
    for jn in model["joints"]:
        ds, n = jn["dof_start"], jn["dof_end"] - jn["dof_start"]
        dg = np.diag(Minv)[ds : ds + n]
        if jn["type"] == JOINT_FREE:
            for k in range(3):
                D[ds + k]["invweight"] = float(dg[:3].mean())
                D[ds + 3 + k]["invweight"] = float(dg[3:].mean())
        else:
            D[ds]["invweight"] = float(dg[0])
    model["meaninertia"] = float(np.trace(M[s:, s:]) / nd)
    return M


# --------------------------------------------------------------------------------------
# main assembly
# --------------------------------------------------------------------------------------
ROBOTS = {   # URDFs of the reference's benchmark set that have the kernels' compile-time shape (13 links, free base + 12 revolute joints, primitives)
    "go2": dict(urdf="urdf/go2/urdf/go2.urdf", base_init_pos=(0.0, 0.0, 0.42)),
    "anymal_c": dict(urdf="urdf/anymal_c/urdf/anymal_c.urdf", base_init_pos=(0.0, 0.0, 0.8)),   # tests/test_rigid_benchmarks.py:392-399
}
N_GEOMS_KERNEL = 28   # GO2SIM_NG: a model with fewer collision geoms is padded with inert spheres on the ground link (no collision pairs)


def synth_multi_pendulum(n):
    """URDF text of the n-segment pendulum of the reference's analytic tests (tests/test_rigid_physics.py:225-277 `_build_multi_pendulum`, re-expressed: a
    fixed base, per segment a continuous joint about x and a massless 1 m arm carrying a 1 kg point mass, inertia 1e-12).  The reference's links
    are visual-only; a small collision sphere is added to every mass so that the collision pipeline has geoms to carry (they never touch anything:
    spheres of adjacent links are filtered, the pivot stands 5 m above the ground)."""
    x = ['<robot name="multi_pendulum">', '<link name="base"/>']
    parent = "base"
    for i in range(n):
        x.append(f'<joint name="PendulumJoint_{i}" type="continuous"><origin xyz="0 0 0" rpy="0 0 0"/><axis xyz="1 0 0"/><parent link="{parent}"/>'
                 f'<child link="PendulumArm_{i}"/><limit effort="{100.0 * (n - i)}" velocity="30.0"/><dynamics damping="0.0" friction="0.0"/></joint>')
        x.append(f'<link name="PendulumArm_{i}"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="0.0"/>'
                 '<inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/></inertial></link>')
        x.append(f'<joint name="PendulumMassJoint_{i}" type="fixed"><origin xyz="0 0 1.0" rpy="0 0 0"/><parent link="PendulumArm_{i}"/><child link="PendulumMass_{i}"/></joint>')
        x.append(f'<link name="PendulumMass_{i}"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="1.0"/>'
                 '<inertia ixx="1e-12" ixy="0" ixz="0" iyy="1e-12" iyz="0" izz="1e-12"/></inertial>'
                 '<collision><origin xyz="0 0 0" rpy="0 0 0"/><geometry><sphere radius="0.06"/></geometry></collision></link>')
        parent = f"PendulumMass_{i}"
    x.append("</robot>")
    return "".join(x)


def synth_two_aligned_hinges():
    """The `two_aligned_hinges` model of the reference's tests (tests/test_rigid_physics.py:165-177, used by `test_link_velocity` :638-706), re-expressed as URDF
    text: two 0.5 m bodies along x, hinges about z at the origin and at (0.5, 0, 0), centres of mass in the middle of the bodies, equal masses.  The
    reference's capsule geoms are replaced by small spheres at the centres of mass (the test is kinematic; mass and inertia values do not enter it beyond
    the two masses being equal); same link / dof / geom counts as the double pendulum, so it runs on that shape variant of the libraries."""
    x = ['<robot name="two_aligned_hinges">', '<link name="base"/>']
    parent, origin = "base", "0 0 0"
    for i in range(2):
        x.append(f'<joint name="joint{i}" type="continuous"><origin xyz="{origin}" rpy="0 0 0"/><axis xyz="0 0 1"/><parent link="{parent}"/>'
                 f'<child link="body{i}"/><limit effort="100.0" velocity="30.0"/><dynamics damping="0.0" friction="0.0"/></joint>')
        x.append(f'<link name="body{i}"><inertial><origin xyz="0.25 0 0" rpy="0 0 0"/><mass value="1.0"/>'
                 '<inertia ixx="0.001" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.02"/></inertial>'
                 '<collision><origin xyz="0.25 0 0" rpy="0 0 0"/><geometry><sphere radius="0.05"/></geometry></collision></link>')
        parent, origin = f"body{i}", "0.5 0 0"
    x.append("</robot>")
    return "".join(x)


def synth_box(size=0.04, density=200.0):
    """A free cube (gs.morphs.Box(size=(0.04, 0.04, 0.04)) of tests/test_rigid_physics.py:1750-1800, default material density 200 kg / m^3): mass rho s^3,
    inertia m s^2 / 6."""
    m = density * size ** 3
    i = m * size * size / 6.0
    return (f'<robot name="box"><link name="box"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="{m!r}"/>'
            f'<inertia ixx="{i!r}" ixy="0" ixz="0" iyy="{i!r}" iyz="0" izz="{i!r}"/></inertial>'
            f'<collision><origin xyz="0 0 0" rpy="0 0 0"/><geometry><box size="{size} {size} {size}"/></geometry></collision></link></robot>')


# Shape variants (test infrastructure: the same library sources compiled for another link / dof / geom count, build.SHAPES): synthetic URDF text, whether
# the root is fixed, where it stands, the substep, and the armature (the reference's tests build these with default_armature=None, tests/utils.py:595)
SHAPE_ROBOTS = {
    "pendulum": dict(urdf=synth_multi_pendulum(1), fixed=True, base_init_pos=(0.0, 0.0, 5.0), substep_dt=0.002, armature=0.0),
    "double_pendulum": dict(urdf=synth_multi_pendulum(2), fixed=True, base_init_pos=(0.0, 0.0, 5.0), substep_dt=0.002, armature=0.0),
    "box": dict(urdf=synth_box(), fixed=False, base_init_pos=(0.65, 0.0, 0.02), substep_dt=0.01, armature=0.0),
    # the cube of `test_axis_aligned_bounding_boxes` (tests/test_rigid_physics.py:3853-3858: size 0.1 at (0.5, 0, 0.05)); libraries: the box shape
    "box01": dict(urdf=synth_box(size=0.1), fixed=False, base_init_pos=(0.5, 0.0, 0.05), substep_dt=0.01, armature=0.0),
    # kinematic known answers only (forward kinematics, no stepping: the mechanism lies in the ground plane like the reference's); libraries: the double_pendulum shape
    "two_aligned_hinges": dict(urdf=synth_two_aligned_hinges(), fixed=True, base_init_pos=(0.0, 0.0, 0.0), substep_dt=0.002, armature=0.0),
}


def build_model(assets_dir, base_init_pos=(0.0, 0.0, 0.42), base_init_quat=(1.0, 0.0, 0.0, 0.0), substep_dt=0.01, robot="go2"):
    shape = SHAPE_ROBOTS.get(robot)                                   # a shape variant: no padding to the Go2 geom count, no Go2 shape check
    robot_urdf = shape["urdf"] if shape else os.path.join(assets_dir, ROBOTS[robot]["urdf"])
    fixed_base = bool(shape and shape["fixed"])
    armature = shape["armature"] if shape else 0.1                    # options/morphs.py:1000 default_armature, mjcf.py:188-190
    if shape:
        base_init_pos, substep_dt = shape["base_init_pos"], shape["substep_dt"]
    sol_timeconst = max(0.01, 2.0 * substep_dt)  # rigid_solver.py:260-261 + _sanitize_sol_params
    sol_params = [sol_timeconst, 1.0, 0.9, 0.95, 0.001, 0.5, 2.0]

    links_out, joints_out, dofs_out, geoms_out, entities = [], [], [], [], []
    qpos0 = []

    # ---- entity 0: plane (fixed) ----------------------------------------------------------------
    pl_links, pl_joints = load_urdf(os.path.join(assets_dir, "urdf/plane/plane.urdf"))
    assert len(pl_links) == 1 and not pl_joints
    pl = pl_links[0]
    ev, eq = mju_eig3(pl["inertial"]["inertia"])
    links_out.append(
        dict(
            name=pl["name"], parent=-1, root=0, entity=0, is_fixed=1, joint_start=0, joint_end=0,
            dof_start=0, dof_end=0, q_start=0, q_end=0, n_dofs=0, geom_start=0, geom_end=len(pl["collisions"]),
            pos=[0.0, 0.0, 0.0], quat=[1.0, 0.0, 0.0, 0.0],
            inertial_pos=pl["inertial"]["origin"][:3, 3].tolist(), inertial_quat=eq.tolist(),
            inertial_i=np.diag(ev).tolist(), inertial_mass=pl["inertial"]["mass"], invweight=[0.0, 0.0],
        )
    )
    for g in pl["collisions"]:
        geoms_out.append(dict(g, link=0))
    # a robot with fewer collision geoms than the kernels' compile-time count is padded with inert spheres (no collision pairs, far away) that
    # belong to the fixed ground link -- inside its geom range, because geoms are stored link-major
    robot_links_raw, _ = load_urdf(robot_urdf)
    n_pad = 0 if shape else N_GEOMS_KERNEL - len(geoms_out) - sum(len(l["collisions"]) for l in robot_links_raw)
    if n_pad < 0:
        raise ValueError(f"{robot}: more than {N_GEOMS_KERNEL} collision geoms")
    for k in range(n_pad):
        geoms_out.append(dict(type=GEOM_SPHERE, data=[1e-3], origin=T_from([1000.0 + 10.0 * k, 1000.0, -1000.0], [0, 0, 0]), link=0, dummy=True))
    links_out[0]["geom_end"] = len(geoms_out)
    entities.append(dict(link_start=0, link_end=1, dof_start=0, dof_end=0, geom_start=0, geom_end=len(geoms_out)))

    # ---- entity 1: go2 ----------------------------------------------------------------------------
    links, joints = load_urdf(robot_urdf)
    for l in links:                                                   # a link without <inertial> (the pendulum's base): massless
        if l["inertial"] is None:
            l["inertial"] = {"origin": np.eye(4), "mass": 0.0, "inertia": np.zeros((3, 3))}
    links, joints = merge_fixed_links(links, joints)
    order, parent_of = bfs_order(links, joints)
    by_name = {l["name"]: l for l in links}
    joint_of = {j["child"]: j for j in joints}
    link0 = len(links_out)
    n_dofs, n_qs, n_joints = 0, 0, 0
    for li, name in enumerate(order):
        l = by_name[name]
        gi = link0 + li
        ine = l["inertial"]
        # MuJoCo: inertial frame = principal axes of the (rotated) URDF inertia
        Rin = ine["origin"][:3, :3]
        ev, eq = mju_eig3(Rin @ ine["inertia"] @ Rin.T)
        if name not in parent_of:
            jtype, n_d, n_q = (JOINT_FIXED, 0, 0) if fixed_base else (JOINT_FREE, 6, 7)      # gs.morphs.URDF(fixed=True): the root carries no joint
            lpos, lquat = list(base_init_pos), list(base_init_quat)
            parent = -1
        else:
            j = joint_of[name]
            if j["type"] not in ("revolute", "continuous"):
                raise ValueError(f"joint {j['name']}: only revolute / continuous joints below the root are compiled (type {j['type']})")
            if j["type"] == "continuous":
                j["limit"] = dict(j["limit"] or {}, lower=-1e30, upper=1e30, effort=(j["limit"] or {}).get("effort"))
            jtype, n_d, n_q = JOINT_REVOLUTE, 1, 1
            lpos, lquat = j["origin"][:3, 3].tolist(), R_to_quat(j["origin"][:3, :3]).tolist()
            parent = link0 + order.index(parent_of[name])
        has_joint = jtype != JOINT_FIXED
        rec = dict(
            name=name, parent=parent, root=link0, entity=1, is_fixed=int(not has_joint),
            joint_start=n_joints, joint_end=n_joints + int(has_joint), dof_start=n_dofs, dof_end=n_dofs + n_d,
            q_start=n_qs, q_end=n_qs + n_q, n_dofs=n_d, geom_start=len(geoms_out),
            geom_end=len(geoms_out) + len(l["collisions"]), pos=lpos, quat=lquat,
            inertial_pos=ine["origin"][:3, 3].tolist(), inertial_quat=eq.tolist(),
            inertial_i=np.diag(ev).tolist(), inertial_mass=ine["mass"], invweight=[0.0, 0.0],
        )
        links_out.append(rec)
        if has_joint:
            joints_out.append(
                dict(name="root_joint" if jtype == JOINT_FREE else joint_of[name]["name"], type=jtype, link=gi,
                     q_start=n_qs, dof_start=n_dofs, dof_end=n_dofs + n_d, pos=[0.0, 0.0, 0.0], sol_params=sol_params)
            )
        if jtype == JOINT_FREE:
            for k in range(6):
                dofs_out.append(
                    dict(motion_ang=[float(k - 3 == a) for a in range(3)], motion_vel=[float(k == a) for a in range(3)],
                         limit=[-1e30, 1e30], invweight=0.0, armature=0.0, damping=0.0, stiffness=0.0,
                         frictionloss=0.0, kp=0.0, kv=0.0, force_range=[-1e30, 1e30])
                )
            qpos0.extend(list(base_init_pos) + list(base_init_quat))
        elif has_joint:
            j = joint_of[name]
            eff = j["limit"]["effort"]
            dofs_out.append(
                dict(motion_ang=[float(a) for a in j["axis"]], motion_vel=[0.0, 0.0, 0.0],
                     limit=[j["limit"]["lower"], j["limit"]["upper"]], invweight=0.0,
                     armature=armature,
                     damping=j["damping"], stiffness=0.0, frictionloss=j["frictionloss"],
                     kp=100.0, kv=10.0,  # geom.py default_dofs_kp/kv (overwritten by the env)
                     force_range=[-eff, eff] if eff is not None else [-1e30, 1e30])
            )
            qpos0.append(0.0)
        for g in l["collisions"]:
            geoms_out.append(dict(g, link=gi))
        n_dofs += n_d
        n_qs += n_q
        n_joints += int(has_joint)
    entities.append(dict(link_start=link0, link_end=len(links_out), dof_start=0, dof_end=n_dofs,
                         geom_start=entities[0]["geom_end"], geom_end=len(geoms_out)))
    if not shape and (len(links_out) != 14 or n_dofs != 18 or len(geoms_out) > N_GEOMS_KERNEL):
        raise ValueError(f"{robot}: {len(links_out)} links / {n_dofs} dofs / {len(geoms_out)} geoms do not fit the kernels' compile-time shape (14 / 18 / <= {N_GEOMS_KERNEL})")
    n_real_geoms = len(geoms_out) - n_pad

    # ---- geoms ------------------------------------------------------------------------------------
    ring_k = support_theta_table()
    geoms_json = []
    for g in geoms_out:
        h = geom_half_extents(g)
        data = list(g["data"]) + [0.0] * (7 - len(g["data"]))
        rim = cylinder_ring(g["data"][0])[0].astype(np.float64).tolist() if g["type"] == GEOM_CYLINDER else []
        geoms_json.append(
            dict(type=g["type"], link=g["link"], pos=g["origin"][:3, 3].tolist(),
                 quat=R_to_quat(g["origin"][:3, :3]).tolist(), data=data, friction=1.0, sol_params=sol_params,
                 center=[0.0, 0.0, 0.0], init_aabb=aabb_corners(h), is_convex=1, rim=rim)
        )

    nd = n_dofs
    mask = np.zeros((nd, nd))
    for i, l in enumerate(links_out):
        j = i
        while j != -1:
            for a in range(l["dof_start"], l["dof_end"]):
                for b in range(links_out[j]["dof_start"], links_out[j]["dof_end"]):
                    mask[a, b] = 1.0
            j = links_out[j]["parent"]

    model = dict(
        format="go2sim-model-v1",
        source="genesis/assets/urdf/plane/plane.urdf + " + (f"synthetic URDF text ({robot}: tools/compile_go2_model.py SHAPE_ROBOTS)" if shape else ROBOTS[robot]["urdf"]) +
               " via tools/compile_go2_model.py", robot=robot, n_real_geoms=n_real_geoms,
        shape=dict(NL=len(links_out), ND=n_dofs, NQ=n_qs, NG=len(geoms_out), NJ=n_joints),
        substep_dt=substep_dt, gravity=[0.0, 0.0, -9.81], eps=EPS32,
        solver=dict(iterations=50, tolerance=1e-6, ls_iterations=50, ls_tolerance=1e-2),
        collider=dict(max_collision_pairs=30, n_contacts_per_pair=5, broad_multiplier=8, mc_perturbation=1e-2,
                      mc_tolerance=1e-2, mpr_to_gjk_overlap_ratio=0.25, ccd_eps=1e-9, ccd_tolerance=1e-6,
                      ccd_iterations=50),
        links=links_out, joints=joints_out, dofs=dofs_out, geoms=geoms_json, entities=entities,
        qpos0=qpos0, mass_parent_mask=mask.reshape(-1).tolist(), support_theta_to_ring=ring_k.tolist(),
        robot_dof_start=0,
    )

    # ---- collision pair table (collider.py:220-329) ----------------------------------------------
    M = compute_invweight(model)
    ng = len(geoms_json)
    # world poses at qpos0 for the neutral filter
    lp, lq = [None] * len(links_out), [None] * len(links_out)
    for i, l in enumerate(links_out):
        if l["parent"] < 0:
            lp[i], lq[i] = np.array(l["pos"]), np.array(l["quat"])
        else:
            p = l["parent"]
            lp[i] = lp[p] + quat_to_R(lq[p]) @ np.array(l["pos"])
            lq[i] = quat_mul(lq[p], np.array(l["quat"]))
    pair_idx = -np.ones((ng, ng), dtype=int)
    n_pairs, filtered = 0, []
    for a in range(ng):
        la = links_out[geoms_json[a]["link"]]
        for b in range(a + 1, ng):
            lb = links_out[geoms_json[b]["link"]]
            ia, ib = geoms_json[a]["link"], geoms_json[b]["link"]
            if ia == ib or geoms_out[a].get("dummy") or geoms_out[b].get("dummy"):
                continue
            if la["is_fixed"] and lb["is_fixed"]:
                continue
            if la["root"] == lb["root"]:
                lo, hi = (ia, ib) if ia < ib else (ib, ia)
                if links_out[hi]["parent"] == lo:  # adjacent (every robot link has a non-fixed joint)
                    continue
                Ra = quat_to_R(lq[ia]) @ quat_to_R(np.array(geoms_json[a]["quat"]))
                pa = lp[ia] + quat_to_R(lq[ia]) @ np.array(geoms_json[a]["pos"])
                Rb = quat_to_R(lq[ib]) @ quat_to_R(np.array(geoms_json[b]["quat"]))
                pb = lp[ib] + quat_to_R(lq[ib]) @ np.array(geoms_json[b]["pos"])
                if convex_overlap(geoms_out[a], Ra, pa, geoms_out[b], Rb, pb, 1.0 - 1e-3):
                    filtered.append((a, b))
                    continue
            pair_idx[a, b] = n_pairs
            n_pairs += 1
    model["collision_pair_idx"] = pair_idx.reshape(-1).tolist()
    model["n_possible_pairs"] = n_pairs
    model["neutral_filtered_pairs"] = filtered
    mcp = min(model["collider"]["max_collision_pairs"], n_pairs)
    model["collider"]["max_collision_pairs"] = mcp
    model["collider"]["max_contact_pairs"] = mcp * model["collider"]["n_contacts_per_pair"]
    model["collider"]["max_broad_pairs"] = mcp * model["collider"]["broad_multiplier"]
    for g in geoms_json:
        g.pop("origin", None)
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="/root/reference/genesis/assets")
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--robot", choices=sorted(ROBOTS) + sorted(SHAPE_ROBOTS), default="go2")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.out is None:
        args.out = os.path.join(here, "..", "go2_sim2real_locomotion_rl_amd", "model", f"{args.robot}_model.json")
    model = build_model(args.assets, base_init_pos=ROBOTS.get(args.robot, {}).get("base_init_pos", (0.0, 0.0, 0.42)), robot=args.robot)
    txt = json.dumps(model, indent=1, sort_keys=True)
    with open(args.out, "w") as f:
        f.write(txt + "\n")
    print(f"wrote {args.out}: links={len(model['links'])} dofs={len(model['dofs'])} geoms={len(model['geoms'])} "
          f"pairs={model['n_possible_pairs']} filtered={model['neutral_filtered_pairs']} "
          f"meaninertia={model['meaninertia']:.6f} sha256={hashlib.sha256(txt.encode()).hexdigest()[:16]}")


if __name__ == "__main__":
    main()
