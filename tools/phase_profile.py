#!/usr/bin/env python3
"""Per-phase cycle accounting of the team kernels (needs a -DGO2SIM_PHASE_PROFILE build: tools/phase_profile.py builds its own .so)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_actions
from go2_sim2real_locomotion_rl_amd import capi
from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg, get_walk_cfgs
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

so = os.path.join(ROOT, "tools", "libgo2sim_prof.so")
if not os.path.exists(so):
    raise SystemExit("build first: hipcc ... -DGO2SIM_PHASE_PROFILE -o tools/libgo2sim_prof.so")
lib = capi.Go2SimLib(so, "go2sim_")
B = 4096
dev = torch.device("cuda", 0)
sim = capi.Go2Sim(lib, pack_model(), B, 0, 1)
f, i, _ = flatten_walk_cfg(B, *get_walk_cfgs(), freeze_curriculum=True)
sim.env_configure(f, i); sim.env_reset()
N = 100
act = make_actions(N + 50, B, dev)
obs = torch.zeros(B, 49, device=dev); priv = torch.zeros(B, 104, device=dev); rew = torch.zeros(B, device=dev)
rst = torch.zeros(B, dtype=torch.uint8, device=dev); to = torch.zeros(B, device=dev)
for s in range(50):
    sim.env_step(act[s], obs, priv, rew, rst, to)
out = (ctypes.c_ulonglong * 64)()
lib.lib.go2sim_debug_phases(sim.h, out, 1)
for s in range(50, 50 + N):
    sim.env_step(act[s], obs, priv, rew, rst, to)
lib.lib.go2sim_debug_phases(sim.h, out, 1)
n_wg = (B + 3) // 4 * 2 * N   # workgroups x launches
tot = sum(out)
for k in range(64):
    if out[k]:
        print(f"phase {k:2d}: {out[k] / n_wg:10.0f} cycles/WG-launch  {100.0 * out[k] / tot:5.1f}%")
print("total cycles per WG-launch", tot / n_wg)
