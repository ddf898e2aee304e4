"""Packs the compiled Go2 model tables (model/go2_model.json) into the binary "GO2M" v1 blob that
`go2sim_create` consumes.  The layout is documented in include/go2sim.h."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_JSON = os.path.join(_HERE, "model", "go2_model.json")
MAGIC = 0x4D324F47
VERSION = 1


def load_model_json(path=MODEL_JSON):
    with open(path) as f:
        return json.load(f)


def pack_model(model=None):
    m = load_model_json() if model is None else model
    F, I = [], []
    sol, col = m["solver"], m["collider"]
    F += [m["substep_dt"], *m["gravity"], m["eps"], sol["tolerance"], sol["ls_tolerance"], m["meaninertia"],
          col["mc_perturbation"], col["mc_tolerance"], col["mpr_to_gjk_overlap_ratio"], col["ccd_eps"],
          col["ccd_tolerance"], 0.0, 0.0, 0.0]
    for l in m["links"]:
        F += [*l["pos"], *l["quat"], *l["inertial_pos"], *l["inertial_quat"],
              *np.asarray(l["inertial_i"]).reshape(-1).tolist(), l["inertial_mass"], *l["invweight"]]
        I += [l["parent"], l["root"], l["entity"], l["is_fixed"], l["joint_start"], l["joint_end"], l["dof_start"],
              l["dof_end"], l["q_start"], l["q_end"], l["n_dofs"], l["geom_start"], l["geom_end"]]
    for j in m["joints"]:
        F += [*j["pos"], *j["sol_params"]]
        I += [j["type"], j["link"], j["q_start"], j["dof_start"], j["dof_end"]]
    for d in m["dofs"]:
        F += [*d["motion_ang"], *d["motion_vel"], *d["limit"], d["invweight"], d["armature"], d["damping"],
              d["stiffness"], d["frictionloss"], d["kp"], d["kv"], *d["force_range"]]
    F += list(m["qpos0"])
    for g in m["geoms"]:
        rim = np.zeros((32, 2))
        if g["rim"]:
            rim[:] = np.asarray(g["rim"])
        F += [*g["pos"], *g["quat"], *g["data"], g["friction"], *g["sol_params"], *g["center"],
              *np.asarray(g["init_aabb"]).reshape(-1).tolist(), *rim.reshape(-1).tolist()]
        I += [g["type"], g["link"], g["is_convex"]]
    F += list(m["mass_parent_mask"])
    for e in m["entities"]:
        I += [e["link_start"], e["link_end"], e["dof_start"], e["dof_end"], e["geom_start"], e["geom_end"]]
    I += list(m["collision_pair_idx"])
    I += list(m["support_theta_to_ring"])
    Fa = np.asarray(F, dtype=np.float32)
    Ia = np.asarray(I, dtype=np.int32)
    H = np.zeros(32, dtype=np.int32)
    H[:20] = [MAGIC, VERSION, len(m["links"]), len(m["joints"]), len(m["dofs"]), len(m["qpos0"]), len(m["geoms"]),
              len(m["entities"]), m["n_possible_pairs"], col["max_collision_pairs"], col["max_contact_pairs"],
              col["max_broad_pairs"], col["n_contacts_per_pair"], sol["iterations"], sol["ls_iterations"],
              col["ccd_iterations"], 180, 32, Fa.size, Ia.size]
    return H.tobytes() + Fa.tobytes() + Ia.tobytes()
