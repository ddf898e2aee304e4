"""The compiled model tables (go2_sim2real_locomotion_rl_amd/model/go2_model.json) and the binary blob."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from go2_sim2real_locomotion_rl_amd.capi import REPO_ROOT
from go2_sim2real_locomotion_rl_amd.model_blob import MODEL_JSON, load_model_json, pack_model

# URDF facts (genesis/assets/urdf/go2/urdf/go2.urdf)
LINK_ORDER = ["planeLink", "base", "FL_hip", "FR_hip", "RL_hip", "RR_hip", "FL_thigh", "FR_thigh", "RL_thigh", "RR_thigh", "FL_calf",
              "FR_calf", "RL_calf", "RR_calf"]


def test_shapes_and_order():
    m = load_model_json()
    assert [l["name"] for l in m["links"]] == LINK_ORDER  # plane first, robot links breadth-first (urdf.py:52-90)
    assert len(m["dofs"]) == 18 and len(m["qpos0"]) == 19 and len(m["geoms"]) == 28 and len(m["joints"]) == 13
    assert m["n_possible_pairs"] == 319  # 27 robot-ground + 292 non-adjacent self pairs
    assert m["collider"]["max_contact_pairs"] == 150 and m["collider"]["max_broad_pairs"] == 240
    assert m["substep_dt"] == 0.01 and m["joints"][0]["sol_params"][0] == 0.02  # timeconst = max(0.01, 2*dt_sub)


def test_masses_and_limits():
    m = load_model_json()
    mass = {l["name"]: l["inertial_mass"] for l in m["links"]}
    assert mass["base"] == pytest.approx(6.921 + 0.001 + 0.001)  # Head_upper/lower, imu, radar merged
    assert mass["FL_calf"] == pytest.approx(0.154 + 0.04)       # foot merged into calf
    assert sum(v for k, v in mass.items() if k != "planeLink") == pytest.approx(15.019, abs=1e-9)
    d = m["dofs"]
    assert [x["armature"] for x in d] == [0.0] * 6 + [0.1] * 12
    assert d[6]["limit"] == [-1.0472, 1.0472] and d[6]["force_range"] == [-23.7, 23.7]
    assert d[14]["force_range"] == [-35.55, 35.55] and d[14]["limit"] == [-2.7227, -0.83776]
    assert d[12]["limit"] == [-0.5236, 4.5379]  # rear thigh


def test_principal_inertia_frames_reconstruct_the_urdf_tensor():
    """inertial_quat / inertial_i (MuJoCo-style eigen-decomposition) must reproduce the FL_hip URDF inertia."""
    m = load_model_json()
    l = next(x for x in m["links"] if x["name"] == "FL_hip")
    w, x, y, z = l["inertial_quat"]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    I = R @ np.array(l["inertial_i"]) @ R.T
    urdf = np.array([[0.00048, -3.01e-6, 1.11e-6], [-3.01e-6, 0.000884, -1.42e-6], [1.11e-6, -1.42e-6, 0.000596]])
    assert np.allclose(I, urdf, atol=1e-10)
    ev = np.diag(np.array(l["inertial_i"]))
    assert ev[0] >= ev[1] >= ev[2] > 0  # MuJoCo sorts principal moments in decreasing order


def test_invweight_sanity():
    m = load_model_json()
    base = next(x for x in m["links"] if x["name"] == "base")
    assert base["invweight"][0] == pytest.approx(1.0 / 15.019, rel=0.05)  # translational inverse weight ~ 1/total mass
    assert m["dofs"][0]["invweight"] == pytest.approx(base["invweight"][0], rel=0.02)
    assert 2.0 < m["meaninertia"] < 3.5


def test_collision_pairs():
    m = load_model_json()
    ng = 28
    P = np.array(m["collision_pair_idx"]).reshape(ng, ng)
    assert (P[np.tril_indices(ng)] == -1).all()
    assert (P[0, 1:] >= 0).all()  # ground vs every robot geom
    link = [g["link"] for g in m["geoms"]]
    parent = [l["parent"] for l in m["links"]]
    for a in range(1, ng):
        for b in range(a + 1, ng):
            la, lb = link[a], link[b]
            same = la == lb
            adjacent = parent[lb] == la or parent[la] == lb
            assert (P[a, b] == -1) == (same or adjacent)
    vals = np.sort(P[P >= 0])
    assert np.array_equal(vals, np.arange(319))


def test_blob_layout():
    b = pack_model()
    H = np.frombuffer(b[:128], np.int32)
    assert H[0] == 0x4D324F47 and H[1] == 1 and list(H[2:8]) == [14, 13, 18, 19, 28, 2]
    assert len(b) == 128 + 4 * (H[18] + H[19])


def test_committed_json_matches_generator():
    """tests/golden/go2_model.sha256 pins the committed tables; when the reference assets are present the generator is
    re-run and must reproduce the committed JSON bit for bit."""
    txt = open(MODEL_JSON).read()
    sha = hashlib.sha256(txt.encode()).hexdigest()
    pin = open(os.path.join(REPO_ROOT, "tests", "golden", "go2_model.sha256")).read().split()[0]
    assert sha == pin
    assets = "/root/reference/genesis/assets"
    if not os.path.isdir(assets):
        pytest.skip("reference assets not present (GPU box)")
    spec = importlib.util.spec_from_file_location("compile_go2_model", os.path.join(REPO_ROOT, "tools", "compile_go2_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    regenerated = json.dumps(mod.build_model(assets), indent=1, sort_keys=True) + "\n"
    assert regenerated == txt


# ---- a second robot of the reference's benchmark set through the same model compiler / kernels (ANYmal-C) ----
ANYMAL_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go2_sim2real_locomotion_rl_amd", "model", "anymal_c_model.json")


def _anymal_sim(lib, n_envs, seed=1):
    """Scene of tests/test_rigid_benchmarks.py:378-412 (`anymal`): plane + anymal_c.urdf at z = 0.8, kp = 1000 on the 12 motors, position targets 0."""
    from go2_sim2real_locomotion_rl_amd.capi import C, Go2Sim
    from go2_sim2real_locomotion_rl_amd.model_blob import load_model_json, pack_model

    m = load_model_json(ANYMAL_JSON)
    sim = Go2Sim(lib, pack_model(m), n_envs, 0, seed)
    for k in range(12):
        eff = abs(m["dofs"][6 + k]["force_range"][1])
        sim.set_dof_gains(6 + k, 1000.0, 10.0, -eff, eff)
    return sim, m, C


def test_anymal_c_model_shape_and_padding():
    from go2_sim2real_locomotion_rl_amd.model_blob import load_model_json

    m = load_model_json(ANYMAL_JSON)
    assert len(m["links"]) == 14 and len(m["dofs"]) == 18 and len(m["geoms"]) == 28 and m["n_real_geoms"] == 14
    pads = [g for g in m["geoms"] if g["link"] == 0][1:]
    assert len(pads) == 14 and all(g["type"] == 1 and g["data"][0] == 1e-3 for g in pads)       # inert spheres inside the ground link's geom range
    pidx = np.array(m["collision_pair_idx"]).reshape(28, 28)
    assert (pidx[1:15, :] == -1).all() and (pidx[:, 1:15] == -1).all(), "the padding takes part in no collision pair"
    assert m["links"][0]["geom_end"] == 15 and m["links"][1]["geom_start"] == 15
    mass = sum(l["inertial_mass"] for l in m["links"][1:])
    assert mass == pytest.approx(52.1, abs=1.0)                                                   # ANYmal C: about 52 kg
    assert all(d["limit"][0] < -1e29 and d["limit"][1] > 1e29 for d in m["dofs"][6:]), "anymal_c.urdf gives no joint bounds: unbounded (urdf.py:269-274)"


def test_anymal_c_stands_on_the_oracle(oracle_lib):
    sim, m, C = _anymal_sim(oracle_lib, 3)
    B = 3
    mode = np.zeros((18, B), np.int32); mode[6:] = 2
    sim.set_field_np(C["GO2SIM_F_CTRL_POS"], np.zeros((18, B), np.float32)); sim.set_field_np(C["GO2SIM_I_CTRL_MODE"], mode)
    sim.reset_caches(); sim.forward_kinematics()
    for _ in range(250):
        sim.scene_step(1)
    weight = 9.81 * sum(l["inertial_mass"] for l in m["links"][1:])
    cf = sim.get_field_np(C["GO2SIM_F_CONTACT_FORCE"]).reshape(14, 3, B)
    assert np.allclose(cf[1:, 2].sum(0), weight, rtol=2e-3), "the four feet carry the robot"
    assert (sim.get_field_np(C["GO2SIM_I_N_CONTACTS"])[0] == 4).all() and sim.check_errno() == 0
    z = sim.get_field_np(C["GO2SIM_F_QPOS"])[2]
    assert (z > 0.55).all() and (z < 0.70).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference/genesis/assets"), reason="the URDF assets exist in the build container only")
def test_anymal_c_model_is_what_the_compiler_produces():
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("compile_model", os.path.join(os.path.dirname(ANYMAL_JSON), "..", "..", "tools", "compile_go2_model.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    model = mod.build_model("/root/reference/genesis/assets", base_init_pos=mod.ROBOTS["anymal_c"]["base_init_pos"], robot="anymal_c")
    assert json.loads(json.dumps(model, sort_keys=True)) == json.load(open(ANYMAL_JSON))


@pytest.mark.gpu
def test_anymal_c_scene_step_hip_equals_oracle(oracle_lib, hip_lib):
    """The `anymal_random` benchmark loop (tests/test_rigid_benchmarks.py:415-466): random position targets in [-0.05, 0.05] every step."""
    import torch

    B = 64
    cpu, m, C = _anymal_sim(oracle_lib, B)
    gpu, _, _ = _anymal_sim(hip_lib, B)
    dev = torch.device("cuda:0")
    mode = np.zeros((18, B), np.int32); mode[6:] = 2
    rng = np.random.default_rng(0)
    put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cpu.set_field_np(C["GO2SIM_I_CTRL_MODE"], mode); gpu.set_field(C["GO2SIM_I_CTRL_MODE"], put(mode))
    cpu.reset_caches(); cpu.forward_kinematics(); gpu.reset_caches(); gpu.forward_kinematics()
    for s in range(120):
        ctrl = np.zeros((18, B), np.float32); ctrl[6:] = rng.uniform(-0.05, 0.05, (12, B)) * (4.0 if s > 60 else 1.0)
        cpu.set_field_np(C["GO2SIM_F_CTRL_POS"], ctrl); gpu.set_field(C["GO2SIM_F_CTRL_POS"], put(ctrl))
        cpu.scene_step(1); gpu.scene_step(1)
        for name in ("F_QPOS", "F_VEL", "F_CONTACT_FORCE"):
            k, _ = gpu.field_size(C["GO2SIM_" + name])
            t = torch.zeros(k, B, device=dev); gpu.get_field(C["GO2SIM_" + name], t); torch.cuda.synchronize()
            assert np.array_equal(t.cpu().numpy().view(np.int32), cpu.get_field_np(C["GO2SIM_" + name]).view(np.int32)), f"step {s}: {name}"
    assert gpu.check_errno() == 0
