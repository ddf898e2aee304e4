#!/usr/bin/env python3
"""What a launch costs: per-workgroup wall-clock stamps of one env step.

    python tools/launch_overhead.py [n_envs] [warmup] [steps]          (builds tools/lib_stamp.so = the product source with -DGO2SIM_STAMP)

Every workgroup of the step kernels stamps s_memrealtime (100 MHz, one clock for the whole device) at entry and after its last memory operation
retired.  One step's stamps give, per launch k of the 9: span_k = latest end - earliest start (the time in which at least one workgroup of the
launch may be running), ramp_k = latest start - earliest start (dispatch ramp), and gap_k = earliest start of launch k+1 - latest end of launch k:
the time in which NO workgroup runs (end-of-kernel write-back, dependency resolution of the graph, dispatch).  sum(span) + sum(gap) is the step
as the device sees it; the step graph is used as in the product (plain launches with GO2SIM_NO_GRAPH=1; GO2SIM_NO_FUSE_SOLVE=1 for the nine-launch form)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import NOBS, NPRIV, make_actions
from go2_sim2real_locomotion_rl_amd import build, capi
from go2_sim2real_locomotion_rl_amd.configs import flatten_walk_cfg, get_walk_cfgs
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
so = build.build_hip_variant(os.environ.get("GO2SIM_STAMP_VARIANT", "stamp"), ["-DGO2SIM_STAMP"] + os.environ.get("GO2SIM_STAMP_FLAGS", "").split(), verbose=False)
lib = capi.Go2SimLib(so, "go2sim_")
dev = torch.device("cuda", 0)
sim = capi.Go2Sim(lib, pack_model(), B, 0, 1)
f, i, _ = flatten_walk_cfg(B, *get_walk_cfgs(), freeze_curriculum=True)
sim.env_configure(f, i); sim.env_reset()
act = make_actions(N + W, B, dev, workload="walk")
obs = torch.zeros(B, NOBS["walk"], device=dev); priv = torch.zeros(B, NPRIV["walk"], device=dev); rew = torch.zeros(B, device=dev)
rst = torch.zeros(B, dtype=torch.uint8, device=dev); to = torch.zeros(B, device=dev)
KINDS, MAXWG, DEPTH = 8, 4096, 4
NAMES = ["pre_dynamics", "collide", "solve (+ integrate / kinematics / dynamics when fused)", "integrate_fk_dynamics", "integrate_fk", "post_a", "post_b"]
ORDER = []      # (kind, which launch of that kind in the step), filled per step from the stamps
stamps = np.zeros((KINDS, MAXWG, DEPTH, 2), np.uint64); counts = np.zeros((KINDS, MAXWG), np.uint32)
for s in range(W):
    sim.env_step(act[s], obs, priv, rew, rst, to)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rows, host_ms = [], []
for s in range(W, W + N):
    lib.lib.go2sim_debug_stamps(sim.h, stamps.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
    before = counts.copy()
    ev0.record(); sim.env_step(act[s], obs, priv, rew, rst, to); ev1.record(); torch.cuda.synchronize()
    host_ms.append(ev0.elapsed_time(ev1))
    lib.lib.go2sim_debug_stamps(sim.h, stamps.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p))
    launches = []
    for kind in range(len(NAMES)):                                       # every launch of the step, whatever the fusion state of the build: ordered by start
        n_new = int((counts[kind] - before[kind]).max())
        for which in range(n_new):
            wg = np.flatnonzero(counts[kind] - before[kind] > which)
            idx = (before[kind][wg] + which) % DEPTH
            t0 = stamps[kind, wg, idx, 0].astype(np.int64); t1 = stamps[kind, wg, idx, 1].astype(np.int64)
            launches.append((t0.min(), t0.max(), t1.max(), np.mean(t1 - t0), len(wg), kind, which))
    launches.sort(key=lambda l: l[0])
    ORDER = [(l[5], l[6]) for l in launches]
    rows.append(launches)
a = np.array([[(l[2] - l[0], l[1] - l[0], l[3], l[4]) for l in r] for r in rows], np.float64)        # [step, launch, (span, ramp, mean wg, n_wg)]
gaps = np.array([[r[k + 1][0] - r[k][2] for k in range(len(ORDER) - 1)] for r in rows], np.float64)
tick_us = 0.01
print(f"launch overhead, {B} envs, steps {W}..{W + N} after the reset (walk, action set C), s_memrealtime ticks of 10 ns converted to us; means over {N} steps")
print(f"{'launch':26s} {'wgs':>5s} {'span':>8s} {'ramp':>7s} {'mean wg':>8s} {'gap after':>10s}")
for k, (kind, which) in enumerate(ORDER):
    g = gaps[:, k].mean() * tick_us if k < len(ORDER) - 1 else float('nan')
    print(f"{(NAMES[kind] if kind != 2 else 'solve' + (' + integrate' if len(ORDER) < 9 else '')) + ('' if kind not in (1, 2) else f' #{which + 1}'):26s} {int(a[0, k, 3]):5d} {a[:, k, 0].mean() * tick_us:8.2f} {a[:, k, 1].mean() * tick_us:7.2f} {a[:, k, 2].mean() * tick_us:8.2f} {g:10.2f}")
span_sum, gap_sum = a[:, :, 0].sum(1).mean() * tick_us, gaps.sum(1).mean() * tick_us
first_to_last = np.mean([r[-1][2] - r[0][0] for r in rows]) * tick_us
print(f"sum of spans {span_sum:.1f} us + sum of the {len(ORDER) - 1} gaps {gap_sum:.1f} us = {span_sum + gap_sum:.1f} us (first workgroup start to last workgroup end {first_to_last:.1f} us); "
      f"HIP events around the step {1e3 * np.mean(host_ms):.1f} us (incl. one synchronising read-out per step)")
print(f"mean gap {gaps.mean() * tick_us:.2f} us per launch boundary, min {gaps.min() * tick_us:.2f}, max {gaps.max() * tick_us:.2f}; gaps are {100 * gap_sum / (span_sum + gap_sum):.1f} % of the step")
