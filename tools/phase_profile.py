#!/usr/bin/env python3
"""Per-phase cycle accounting of the team kernels (needs a -DGO2SIM_PHASE_PROFILE build: tools/phase_profile.py builds its own .so)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_actions
from go2_sim2real_locomotion_rl_amd import capi
from go2_sim2real_locomotion_rl_amd.configs import build_stair_terrain, flatten_walk_cfg, get_stair_cfgs, get_walk_cfgs
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

from go2_sim2real_locomotion_rl_amd import build
so = build.build_hip_variant("prof", ["-DGO2SIM_PHASE_PROFILE"], verbose=False)     # tools/lib_prof.so, rebuilt when the sources are newer
lib = capi.Go2SimLib(so, "go2sim_")
B = 4096
dev = torch.device("cuda", 0)
sim = capi.Go2Sim(lib, pack_model(), B, 0, 1)
WORKLOAD = os.environ.get("GO2SIM_PROFILE_WORKLOAD", "walk")                    # walk | stairs (BASELINE configs[2])
cfgs = get_stair_cfgs() if WORKLOAD == "stairs" else get_walk_cfgs()
if WORKLOAD == "stairs":
    hf, info = build_stair_terrain(cfgs[0]["terrain"])
    sim.set_terrain(hf, info["horizontal_scale"], info["vertical_scale"], info["terrain_origin"])
f, i, _ = flatten_walk_cfg(B, *cfgs, freeze_curriculum=True)
sim.env_configure(f, i); sim.env_reset()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 50      # warm-up steps from the reset (5 = the landing window of the driver's bench run)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
act = make_actions(N + W, B, dev, workload=WORKLOAD)
from bench import NOBS, NPRIV
obs = torch.zeros(B, NOBS[WORKLOAD], device=dev); priv = torch.zeros(B, NPRIV[WORKLOAD], device=dev); rew = torch.zeros(B, device=dev)
rst = torch.zeros(B, dtype=torch.uint8, device=dev); to = torch.zeros(B, device=dev)
for s in range(W):
    sim.env_step(act[s], obs, priv, rew, rst, to)
PH_MAX_WG = 8192
out = (ctypes.c_ulonglong * (64 * PH_MAX_WG))()
lib.lib.go2sim_debug_phases(sim.h, out, 1)
GROUPS_ = {"post_a": [12, 13, 14, 15, 16, 17], "solver": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11], "dynamics": list(range(20, 26)), "collide": list(range(30, 34)) + [26, 27, 28, 29], "integrate_fk": [40, 41, 18, 19, 55]}
def xcd_block(bid, n):
    """logical block of physical workgroup `bid` in a grid of n (xcd_block() of csrc/go2sim.hip)"""
    q, r, x, i = n >> 3, n & 7, bid & 7, bid >> 3
    return np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + i


if len(sys.argv) > 3 and sys.argv[3] == "decouple":
    # what the grid-wide kernel boundaries cost.  Per env step: S = sum over the kernels of the slowest workgroup's cycles (what
    # the launch structure pays) against C = the longest per-env chain, max over envs of the sum over the kernels of the cycles of the workgroup
    # that held the env (what a launch structure without joins between envs would pay).  Collide runs 4 envs per workgroup (T = 16), solver / dynamics 2.
    S_all, C_all, S2_all, C2_all = [], [], [], []
    env = np.arange(B)
    for s in range(W, W + N):
        sim.env_step(act[s], obs, priv, rew, rst, to)
        lib.lib.go2sim_debug_phases(sim.h, out, 1)
        a = np.frombuffer(out, dtype=np.uint64).reshape(PH_MAX_WG, 64).astype(np.float64)
        per_env, S = np.zeros(B), 0.0
        per_env2, S2 = np.zeros(B), 0.0
        for g, ids, epw in (("collide", GROUPS_["collide"], 4), ("solver", GROUPS_["solver"], 2), ("dynamics", GROUPS_["dynamics"], 2), ("integrate_fk", GROUPS_["integrate_fk"], 2)):
            n_wg = B // epw
            t = a[:n_wg, ids].sum(1)                                       # both substep launches of the step
            lb = xcd_block(np.arange(n_wg), n_wg)
            t_env = np.zeros(B)
            for slot in range(epw):
                t_env[lb * epw + slot] = t
            per_env += t_env; S += t.max()
            if g in ("collide", "solver"):
                per_env2 += t_env; S2 += t.max()
        S_all.append(S); C_all.append(per_env.max()); S2_all.append(S2); C2_all.append(per_env2.max())
    S_all, C_all, S2_all, C2_all = map(np.array, (S_all, C_all, S2_all, C2_all))
    print(f"decoupling bound over {N} steps after {W} warm-up steps, {B} envs (cycles per env step, both substeps):")
    print(f"  all four substep kernels : sum of slowest workgroups {S_all.mean():9.0f}   longest per-env chain {C_all.mean():9.0f}   gap {100 * (1 - C_all.mean() / S_all.mean()):5.1f} %")
    print(f"  collide + solver only    : sum of slowest workgroups {S2_all.mean():9.0f}   longest per-env chain {C2_all.mean():9.0f}   gap {100 * (1 - C2_all.mean() / S2_all.mean()):5.1f} %")
    print("  (integrate_fk: the T = 16 launch of the second substep is attributed with the T = 32 map; it is constant-time, so the gap is unaffected)")
    raise SystemExit(0)
if len(sys.argv) > 3 and sys.argv[3] == "each":
    # one read-out per env step: what a single launch waits for is its slowest workgroup, which per-run sums average away
    rows = {g: [] for g in GROUPS_}
    worst = {g: None for g in GROUPS_}
    for s in range(W, W + N):
        sim.env_step(act[s], obs, priv, rew, rst, to)
        lib.lib.go2sim_debug_phases(sim.h, out, 1)
        a = np.frombuffer(out, dtype=np.uint64).reshape(PH_MAX_WG, 64).astype(np.float64)
        for g, ids in GROUPS_.items():
            tot = a[:, ids].sum(1) / (1 if g == "post_a" else 2)                                  # two launches per env step (fused kernels: their phases land in both groups)
            used = tot > 0
            if not used.any(): continue
            srt = np.sort(tot[used])
            rows[g].append((srt.mean(), srt[int(0.99 * len(srt))], srt[int(0.999 * len(srt))], srt[-1]))
            i = int(np.argmax(tot))
            if worst[g] is None or tot[i] > worst[g][0]:
                worst[g] = (tot[i], {k: a[i, k] / (1 if g == "post_a" else 2) for k in ids}, {k: a[i, k] for k in (range(50, 55) if g == "solver" else list(range(34, 50)) + list(range(56, 64)))})
    for g, r in rows.items():
        if not r: continue
        r = np.array(r)
        print(f"== {g}: cycles per WG-launch over {N} single steps: mean {r[:, 0].mean():9.0f}  p99 {r[:, 1].mean():9.0f}  p99.9 {r[:, 2].mean():9.0f}  max (mean over steps) {r[:, 3].mean():9.0f}  max (worst step) {r[:, 3].max():9.0f}")
        print("   worst workgroup: " + "  ".join(f"ph{k}={v:.0f}" for k, v in worst[g][1].items() if v > 0) + ("  | counters (both launches) " + " ".join(f"{k}:{v:.0f}" for k, v in worst[g][2].items()) if g in ("solver", "collide") else ""))
    raise SystemExit(0)
for s in range(W, W + N):
    sim.env_step(act[s], obs, priv, rew, rst, to)
lib.lib.go2sim_debug_phases(sim.h, out, 1)
a = np.frombuffer(out, dtype=np.uint64).reshape(PH_MAX_WG, 64).astype(np.float64)
launches = 2 * N                                   # substep kernels: two launches per env step
GROUPS = {"solver": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11], "dynamics": list(range(20, 26)), "collide": list(range(30, 34)) + [26, 27, 28, 29], "integrate_fk": [40, 41, 18, 19, 55]}   # (18 / 19 / 55: parts of 41; ids 34-39, 42-49: PHD sections, printed above)
NAMES = {0: "stage", 1: "rows", 2: "init Ma/Jaref/update", 3: "Hessian", 4: "Cholesky factor", 5: "gradient solve", 6: "line search", 7: "qacc/constraint update",
         8: "active-set change test (FAST ORDER) / incremental Cholesky", 9: "prologue", 11: "commit", 30: "AABB + clear", 31: "endpoint sort", 32: "candidate pairs", 33: "narrow phase", 26: "terrain pair setup/count", 27: "terrain descriptors", 28: "terrain prism MPR", 29: "terrain replay"}
# sections inside lane-divergent code (PHD): cycles at id, number of executions at id + 1
for name, i in (("GJK / EPA query, cooperative (quad)", 34), ("GJK / EPA query, one lane (perturbed detections)", 62), ("  of which GJK", 38), ("  of which EPA + witness", 42), ("    EPA nearest-face scan", 56), ("    EPA horizon walk", 58), ("    EPA face attachment", 60), ("  support pair evaluations (GJK / EPA)", 48), ("MPR query", 36)):
    cyc, cnt = a[:, i], a[:, i + 1]
    if cnt.sum() > 0:
        print(f"-- {name}: {cnt.sum() / launches:8.1f} executions per launch (wave level), {cyc.sum() / cnt.sum():9.0f} cycles each; per WG-launch mean {cyc.mean() * PH_MAX_WG / max(1, (cyc > 0).sum()) / launches:9.0f}, slowest 1% {np.sort(cyc)[-max(1, int((cyc > 0).sum()) // 100):].mean() / launches:9.0f}")
cnt = a[:, 50:55].sum(0) / launches
if cnt.sum() > 0:
    wg = max(1, int((a[:, 0] > 0).sum()))
    print(f"-- incremental Cholesky per WG-launch: serial-form calls {cnt[0] / wg:.2f} with {cnt[1] / wg:.2f} flipped-row passes; pipelined batches {cnt[2] / wg:.2f} "
          f"with {cnt[3] / wg:.2f} flipped rows (larger team) in {cnt[4] / wg:.1f} time steps")
    tot50 = a[:, 50:55]
    order = np.argsort(-a[:, 8])[:max(1, wg // 100)]
    c1 = tot50[order].mean(0) / launches
    print(f"   slowest 1% (by incremental time): serial calls {c1[0]:.2f} / passes {c1[1]:.2f}; pipelined batches {c1[2]:.2f} / rows {c1[3]:.2f} / time steps {c1[4]:.1f}")
for g, ids in GROUPS.items():
    sub = a[:, ids]
    tot = sub.sum(1)
    used = tot > 0
    if not used.any():
        continue
    n_wg = int(used.sum())
    order = np.argsort(-tot)
    top = order[:max(1, n_wg // 100)]               # the slowest 1 % of the workgroups: what a launch at full residency waits for
    print(f"== {g}: {n_wg} workgroups, {launches} launches; cycles per WG-launch: mean {tot[used].mean() / launches:9.0f}   slowest 1% {tot[top].mean() / launches:9.0f}")
    for k, i in enumerate(ids):
        if sub[:, k].sum() == 0:
            continue
        print(f"   phase {i:2d} {NAMES.get(i, ''):24s} mean {sub[used, k].mean() / launches:9.0f} ({100 * sub[used, k].sum() / tot[used].sum():5.1f}%)   slowest 1% {sub[top, k].mean() / launches:9.0f} ({100 * sub[top, k].sum() / tot[top].sum():5.1f}%)")
