#!/usr/bin/env python3
"""One GJK / EPA query per launch (go2sim_debug_narrowphase: a single lane), repeated: for rocprofv3 --kernel-trace --stats (duration of
k_debug_narrowphase) and --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES (dynamic instruction count of a query)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go2_sim2real_locomotion_rl_amd.capi import Go2Sim, load_hip_lib
from go2_sim2real_locomotion_rl_amd.model_blob import load_model_json, pack_model

which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = load_hip_lib()
sim = Go2Sim(lib, pack_model(), 1, 0, 1)
g = load_model_json()["geoms"]
foot = [i for i, x in enumerate(g) if x["type"] == 1][-1]
r = g[foot]["data"][0]
half = g[0]["data"][2] / 2
I = np.array([1, 0, 0, 0], np.float32)
out = np.zeros(8, np.float32)
p = lambda a: np.asarray(a, np.float32).ctypes.data_as(ctypes.c_void_p)
pa, pb = np.array([0.1, 0.2, r - 0.005], np.float32), np.array([0, 0, -half], np.float32)
for _ in range(n):
    rc = lib.lib.go2sim_debug_narrowphase(sim.h, which, foot, 0, p(pa), p(I), p(pb), p(I), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
print("which", which, "result", out)
