#!/usr/bin/env python3
"""GPU-vs-oracle probe at substep granularity (development aid)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from go2_sim2real_locomotion_rl_amd.capi import C, Go2Sim, load_cpu_oracle_lib, load_hip_lib
from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg, get_walk_cfgs
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NSUB = int(sys.argv[2]) if len(sys.argv) > 2 else 40
from go2_sim2real_locomotion_rl_amd.model_blob import load_model_json
_m = load_model_json()
if len(sys.argv) > 3:
    _m["solver"]["iterations"] = int(sys.argv[3])
if len(sys.argv) > 4:
    _m["solver"]["ls_iterations"] = int(sys.argv[4])
blob = pack_model(_m)
cpu = Go2Sim(load_cpu_oracle_lib(fast=True), blob, B, 0, 7)
gpu = Go2Sim(load_hip_lib(), blob, B, 0, 7)
f, i, names = flatten_walk_cfg(B, *get_walk_cfgs())
cpu.env_configure(f, i); gpu.env_configure(f, i)
cpu.env_reset(); gpu.env_reset()
dev = torch.device("cuda:0")
fields = ["F_QPOS", "F_VEL", "F_MASS_MAT", "F_FORCE", "F_ACC_SMOOTH", "I_N_BROAD", "F_SORT_VALUE", "I_N_CONTACTS", "I_CONTACT_GEOMS", "F_CONTACT_POS",
          "F_CONTACT_NORMAL", "F_CONTACT_PEN", "F_NORMAL_CACHE", "I_N_CONSTRAINTS", "I_SOLVER_ITERS", "F_EFC_FORCE", "F_QFRC_CONSTRAINT", "F_QACC_WS",
          "F_ACC", "F_CONTACT_FORCE", "F_LINK_POS", "F_LINK_QUAT", "F_LINK_CDVEL"]

def gpu_field(name):
    k, is_int = gpu.field_size(C["GO2SIM_" + name])
    t = torch.zeros(k, B, dtype=torch.int32 if is_int else torch.float32, device=dev)
    gpu.get_field(C["GO2SIM_" + name], t)
    return t.cpu().numpy()

for s in range(NSUB):
    cpu.substep(); gpu.substep()
    torch.cuda.synchronize()
    line = f"sub {s:3d}"
    first_bad = None
    for fn in fields:
        a, b = cpu.get_field_np(C["GO2SIM_" + fn]), gpu_field(fn)
        nd = int((a.view(np.int32) != b.view(np.int32)).sum())
        if nd and first_bad is None:
            first_bad = (fn, a, b)
        line += f" {fn[2:]}:{nd}"
    print(line, flush=True)
    if first_bad is not None:
        fn, a, b = first_bad
        idx = np.argwhere(a.view(np.int32) != b.view(np.int32))
        print("first differing field", fn, "count", len(idx))
        for (j, e) in idx[:12]:
            print("   elem", j, "env", e, "cpu", repr(a[j, e]), "gpu", repr(b[j, e]))
        e0 = idx[0][1]
        print(" env", e0, "n_contacts cpu/gpu", cpu.get_field_np(C["GO2SIM_I_N_CONTACTS"])[0, e0], gpu_field("I_N_CONTACTS")[0, e0])
        cg = cpu.get_field_np(C["GO2SIM_I_CONTACT_GEOMS"])[:, e0]
        nc = cpu.get_field_np(C["GO2SIM_I_N_CONTACTS"])[0, e0]
        print(" contact geoms", cg[:nc], cg[150:150 + nc])
        print(" cpu pen", cpu.get_field_np(C["GO2SIM_F_CONTACT_PEN"])[:nc, e0], "gpu pen", gpu_field("F_CONTACT_PEN")[:nc, e0])
        import ctypes
        for nm in ["cdof_ang", "cdof_vel", "mass_L", "jac", "diag", "aref", "efc_D", "Jaref", "jv", "H", "grad", "Mgrad", "search", "qacc", "Ma", "mv"]:
            kk = ctypes.c_int(); ptr = ctypes.c_void_p(); ii = ctypes.c_int()
            assert gpu.L.lib.go2sim_debug_field(gpu.h, nm.encode(), ctypes.byref(ptr), ctypes.byref(kk), ctypes.byref(ii)) == 0
            k = kk.value
            a = np.zeros((k, B), np.float32)
            assert cpu.L.lib.go2sim_cpu_debug_get(cpu.h, nm.encode(), a.ctypes.data_as(ctypes.c_void_p), ctypes.byref(kk)) == 0
            t = torch.zeros(k, B, device=dev)
            import ctypes as ct
            torch.cuda.synchronize()
            # device-to-device copy through torch: wrap raw pointer
            buf = (ct.c_float * (k * B)).from_buffer_copy(b"\0" * (4 * k * B))
            hip = ct.CDLL("libamdhip64.so")
            hip.hipMemcpy(ct.c_void_p(t.data_ptr()), ptr, ct.c_size_t(4 * k * B), ct.c_int(3))
            g = t.cpu().numpy()
            rows = 8 * 18 if nm == "jac" else (8 if nm in ("diag", "aref", "efc_D", "Jaref", "jv") else k)
            nd = int((a[:rows, e0].view(np.int32) != g[:rows, e0].view(np.int32)).sum())
            print(f"   {nm}: differing words in env {e0} (first {rows}): {nd}")
            if (nd and nd < 40) or nm == 'sv':
                for j in (range(96) if nm == 'sv' else np.argwhere(a[:rows, e0].view(np.int32) != g[:rows, e0].view(np.int32))[:6, 0]):
                    print("      idx", j, "cpu", repr(a[j, e0]), "gpu", repr(g[j, e0]))
        break
