"""Boundary checks that need no GPU: header <-> libraries."""
import ctypes
import os

import pytest

from go2_sim2real_locomotion_rl_amd import capi


def test_header_enums_parsed():
    C = capi.C
    assert C["GO2SIM_NL"] == 14 and C["GO2SIM_ND"] == 18 and C["GO2SIM_NQ"] == 19 and C["GO2SIM_NG"] == 28
    assert C["GO2SIM_R_COUNT"] == 35 and C["GO2SIM_R_FEET_STANCE"] == 18  # 19 walk + 14 base-env + 2 stair-env terms
    assert C["GO2SIM_FC_DEFAULT_DOF_POS0"] == C["GO2SIM_FC_TORQUE_LIMIT0"] + 12
    assert C["GO2SIM_IC_COUNT"] > C["GO2SIM_IC_FREEZE_CURRICULUM"]
    assert len(capi.DECLARED_FUNCS) >= 25


def test_hip_library_exports_every_declared_symbol(libs_built):
    """The product library loads and exports the whole C ABI (no compute call without a GPU)."""
    lib = ctypes.CDLL(capi.HIP_LIB)
    for name in capi.DECLARED_FUNCS:
        assert hasattr(lib, name), f"libgo2sim.so does not export {name}"


def test_cpu_twin_exports_every_declared_symbol(libs_built):
    lib = ctypes.CDLL(capi.CPU_LIB)
    for name in capi.DECLARED_FUNCS:
        twin = "go2sim_cpu_" + name[len("go2sim_"):]
        assert hasattr(lib, twin), f"oracle does not export {twin}"


def test_product_fails_loudly_without_gpu(libs_built, blob):
    """No CPU fallback: go2sim_create must fail when no HIP device exists."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load_hip_lib()
    with pytest.raises(capi.Go2SimError):
        capi.Go2Sim(lib, blob, 4, 0, 1)


def test_missing_library_raises(tmp_path):
    with pytest.raises(capi.Go2SimError):
        capi.Go2SimLib(str(tmp_path / "nope.so"), "go2sim_")


def test_product_does_not_reference_oracle():
    """The product package must not import / link anything under oracle/."""
    pkg = os.path.join(capi.REPO_ROOT, "go2_sim2real_locomotion_rl_amd")
    offenders = []
    for root, _, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith((".py", ".hip", ".h", ".cpp")):
                continue
            txt = open(os.path.join(root, fn)).read()
            if fn in ("capi.py", "build.py"):
                continue  # path constants for the test-only loader / the build recipe
            if "libgo2sim_cpu" in txt or "go2sim_cpu_" in txt or "load_cpu_oracle_lib" in txt:
                offenders.append(fn)
    assert not offenders, offenders


def test_bad_arguments_return_status(oracle_lib, blob):
    h = ctypes.c_void_p()
    assert oracle_lib.fn("create")(blob, ctypes.c_size_t(len(blob)), 0, 0, ctypes.c_uint64(1), ctypes.byref(h)) == capi.C["GO2SIM_E_BADARG"]
    bad = b"\x00" * len(blob)
    assert oracle_lib.fn("create")(bad, ctypes.c_size_t(len(bad)), 4, 0, ctypes.c_uint64(1), ctypes.byref(h)) == capi.C["GO2SIM_E_BADMODEL"]
