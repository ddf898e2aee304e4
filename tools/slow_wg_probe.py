#!/usr/bin/env python3
"""Who are the slowest workgroups of a launch?  Per env step of the window: the solver / collide workgroups with the most cycles (prof build),
together with the contact count, row count and Newton iterations of the envs they hold.   usage: slow_wg_probe.py [W] [N]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import NOBS, NPRIV, make_actions
from go2_sim2real_locomotion_rl_amd import build, capi
from go2_sim2real_locomotion_rl_amd.capi import C
from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg, get_walk_cfgs
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

so = build.build_hip_variant("prof", ["-DGO2SIM_PHASE_PROFILE"], verbose=False)
lib = capi.Go2SimLib(so, "go2sim_")
B = 4096
dev = torch.device("cuda", 0)
sim = capi.Go2Sim(lib, pack_model(), B, 0, 1)
f, i, _ = flatten_walk_cfg(B, *get_walk_cfgs(), freeze_curriculum=True)
sim.env_configure(f, i); sim.env_reset()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
act = make_actions(N + W, B, dev, workload="walk")
obs = torch.zeros(B, NOBS["walk"], device=dev); priv = torch.zeros(B, NPRIV["walk"], device=dev); rew = torch.zeros(B, device=dev)
rst = torch.zeros(B, dtype=torch.uint8, device=dev); to = torch.zeros(B, device=dev)
PH_MAX_WG = 8192
out = (ctypes.c_ulonglong * (64 * PH_MAX_WG))()
buf = torch.zeros(B, dtype=torch.int32, device=dev)
def field(name):
    sim.get_field(C["GO2SIM_" + name], buf); return buf.cpu().numpy().copy()
def xcd_block(bid, n):
    q, r, x, i_ = n >> 3, n & 7, bid & 7, bid >> 3
    return np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + i_
for s in range(W):
    sim.env_step(act[s], obs, priv, rew, rst, to)
lib.lib.go2sim_debug_phases(sim.h, out, 1)
SOLVER = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11]; COLLIDE = [30, 31, 32, 33, 26, 27, 28, 29]
NAMES = {0: "stage", 1: "rows", 2: "init", 3: "hess", 4: "chol", 5: "grad", 6: "ls", 7: "upd", 8: "flip", 9: "prol", 11: "commit", 30: "aabb", 31: "sort", 32: "cand", 33: "narrow"}
allrows = []
for s in range(W, W + N):
    sim.env_step(act[s], obs, priv, rew, rst, to)
    lib.lib.go2sim_debug_phases(sim.h, out, 1)
    a = np.frombuffer(out, dtype=np.uint64).reshape(PH_MAX_WG, 64).astype(np.float64)
    ncon, iters, nc = field("I_N_CONSTRAINTS"), field("I_SOLVER_ITERS"), field("I_N_CONTACTS")       # of the step's second substep
    n_wg = B // 2
    tot = a[:n_wg, SOLVER].sum(1)
    lb = xcd_block(np.arange(n_wg), n_wg)
    order = np.argsort(-tot)[:3]
    print(f"step {s}: solver WG cycles (both launches) mean {tot.mean():.0f} p99 {np.percentile(tot, 99):.0f} max {tot.max():.0f}; n_con mean {ncon.mean():.1f} max {ncon.max()} frac>32 {(ncon > 32).mean():.4f}; iters mean {iters.mean():.2f} max {iters.max()}")
    for w in order:
        e0 = lb[w] * 2
        print(f"    wg {w}: {tot[w]:.0f} cyc  " + " ".join(f"{NAMES[k]}={a[w, k]:.0f}" for k in SOLVER if a[w, k] > 0) + f" [ls: setup {a[w, 50]:.0f} newton {a[w, 51]:.0f} ({a[w, 52]:.0f} points) bracket {a[w, 53]:.0f} ({a[w, 54]:.0f} rounds)]" + f" | envs {e0},{e0 + 1}: n_con {ncon[e0]},{ncon[e0 + 1]} iters {iters[e0]},{iters[e0 + 1]} contacts {nc[e0]},{nc[e0 + 1]}")
    # cost model: cycles of a WG vs the max of its two envs' iterations / rows
    it2 = np.maximum(iters[lb * 2], iters[lb * 2 + 1]); nc2 = np.maximum(ncon[lb * 2], ncon[lb * 2 + 1]); itsum = iters[lb * 2] + iters[lb * 2 + 1]
    allrows.append(np.stack([tot, it2, nc2, itsum], 1))
    tc = a[:B // 4, COLLIDE].sum(1)
    oc = np.argsort(-tc)[:2]
    lbc = xcd_block(np.arange(B // 4), B // 4)
    gj = field("I_GJK_FALLBACK") if "GO2SIM_I_GJK_FALLBACK" in C else None
    print(f"         collide WG cycles mean {tc.mean():.0f} p99 {np.percentile(tc, 99):.0f} max {tc.max():.0f}")
    for w in oc:
        e0 = lbc[w] * 4
        print(f"    cwg {w}: {tc[w]:.0f} cyc " + " ".join(f"{NAMES.get(k, k)}={a[w, k]:.0f}" for k in COLLIDE if a[w, k] > 0) + f" gjk(34)={a[w, 34]:.0f}x{a[w, 35]:.0f} mpr(36)={a[w, 36]:.0f}x{a[w, 37]:.0f} | contacts {nc[e0:e0 + 4].tolist()}")
print(f"line search, all workgroups of the last step, both launches: setup {a[:n_wg, 50].mean():.0f}  newton steps {a[:n_wg, 51].mean():.0f} cycles / {a[:n_wg, 52].mean():.2f} points  bracket {a[:n_wg, 53].mean():.0f} cycles / {a[:n_wg, 54].mean():.2f} rounds (wave level: the union of the two envs of a wavefront)")
r = np.concatenate(allrows)
print("correlation of solver WG cycles with max iters of its envs: %.3f, with max rows: %.3f, with sum of iters: %.3f" % (np.corrcoef(r[:, 0], r[:, 1])[0, 1], np.corrcoef(r[:, 0], r[:, 2])[0, 1], np.corrcoef(r[:, 0], r[:, 3])[0, 1]))
for it in range(0, int(r[:, 1].max()) + 1, 2):
    m = (r[:, 1] >= it) & (r[:, 1] < it + 2)
    if m.sum() > 5:
        print(f"  max iters {it:2d}-{it + 1:2d} (second substep): {m.sum():6d} WG-steps, mean cycles {r[m, 0].mean():8.0f}, max {r[m, 0].max():8.0f}")
