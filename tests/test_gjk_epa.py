"""Safe GJK + EPA fallback (include/go2sim_gjk.h; reference collider/gjk.py:1200-1416, epa.py:970-1295) on analytic cases.

Parity unpinned: the reference holds no fixtures for this path; the checks are known answers (sphere / box / cylinder against the ground slab)
and agreement with the independent MPR query of the same poses."""
import ctypes

import numpy as np
import pytest

from go2_sim2real_locomotion_rl_amd.capi import Go2Sim
from go2_sim2real_locomotion_rl_amd.model_blob import load_model_json

I = [1.0, 0.0, 0.0, 0.0]


@pytest.fixture(scope="module")
def query(oracle_lib, blob):
    sim = Go2Sim(oracle_lib, blob, 1, 0, 1)

    def q(which, a, b, pa, qa, pb, qb):
        out = np.zeros(8, np.float32)
        arrs = [np.asarray(x, np.float32) for x in (pa, qa, pb, qb)]
        rc = oracle_lib.lib.go2sim_cpu_debug_narrowphase(sim.h, which, a, b, *[x.ctypes.data_as(ctypes.c_void_p) for x in arrs], out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        return dict(is_col=bool(out[0]), pen=float(out[1]), normal=out[2:5].copy(), pos=out[5:8].copy())

    q.sim = sim
    return q


@pytest.fixture(scope="module")
def geoms():
    g = load_model_json()["geoms"]
    return dict(all=g, sphere=[i for i, x in enumerate(g) if x["type"] == 1][-1], box=[i for i, x in enumerate(g) if x["type"] == 5 and i > 0][0],
                cyl=[i for i, x in enumerate(g) if x["type"] == 3][0], ground_half=g[0]["data"][2] / 2)


@pytest.mark.parametrize("depth", [0.001, 0.005, 0.02])
def test_sphere_on_ground(query, geoms, depth):
    r = geoms["all"][geoms["sphere"]]["data"][0]
    res = query(1, geoms["sphere"], 0, [0.1, 0.2, r - depth], I, [0, 0, -geoms["ground_half"]], I)
    assert res["is_col"] and res["pen"] == pytest.approx(depth, abs=2e-6)
    assert np.allclose(res["normal"], [0, 0, 1], atol=2e-3) and np.allclose(res["pos"], [0.1, 0.2, -depth / 2], atol=1e-4)


def test_separated_geoms_do_not_collide(query, geoms):
    r = geoms["all"][geoms["sphere"]]["data"][0]
    assert not query(1, geoms["sphere"], 0, [0, 0, r + 0.01], I, [0, 0, -geoms["ground_half"]], I)["is_col"]
    hz = geoms["all"][geoms["box"]]["data"][2] / 2
    assert not query(1, 0, geoms["box"], [0, 0, -geoms["ground_half"]], I, [0, 0, hz + 0.02], I)["is_col"]


@pytest.mark.parametrize("quat", [I, [0.9, 0.1, 0.3, 0.05], [0.5, 0.5, -0.5, 0.5]])
def test_box_box_matches_mpr_and_geometry(query, geoms, quat):
    quat = np.asarray(quat, np.float64); quat = quat / np.linalg.norm(quat)
    a = query(0, 0, geoms["box"], [0, 0, -geoms["ground_half"]], I, [0.3, -0.1, 0.04], quat)
    b = query(1, 0, geoms["box"], [0, 0, -geoms["ground_half"]], I, [0.3, -0.1, 0.04], quat)
    assert a["is_col"] and b["is_col"]
    assert b["pen"] == pytest.approx(a["pen"], abs=2e-5)
    assert np.allclose(b["normal"], [0, 0, -1], atol=1e-3)
    # the deepest box corner below the ground plane z = 0
    w, x, y, z = quat
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    h = np.asarray(geoms["all"][geoms["box"]]["data"][:3]) / 2
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * h
    assert b["pen"] == pytest.approx(-(corners @ R.T)[:, 2].min() - 0.04, abs=2e-5)


def test_cylinder_on_ground(query, geoms):
    r = geoms["all"][geoms["cyl"]]["data"][0]
    lying = [np.sqrt(0.5), np.sqrt(0.5), 0.0, 0.0]       # axis along y: the rim touches the ground
    res = query(1, 0, geoms["cyl"], [0, 0, -geoms["ground_half"]], I, [0, 0, r - 0.01], lying)
    ref = query(0, 0, geoms["cyl"], [0, 0, -geoms["ground_half"]], I, [0, 0, r - 0.01], lying)
    assert res["is_col"] and res["pen"] == pytest.approx(ref["pen"], abs=2e-5) and res["pen"] == pytest.approx(0.01, abs=2e-4)
