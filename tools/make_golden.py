#!/usr/bin/env python3
"""Generates tests/golden/walk_b4_seed1.npz : a small fixed-seed Go2 walk trajectory (BASELINE.json configs[0]
shape: flat plane, num_envs=4).

SELF-GENERATED fixture: produced by this repo's CPU oracle (oracle/go2sim_cpu.cpp), NOT by the reference
(which cannot run in this pipeline).  It is a regression pin for the oracle and a
portable expected-output vector for the GPU path; it does not by itself establish parity with the reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from go2_sim2real_locomotion_rl_amd.capi import load_cpu_oracle_lib  # noqa: E402
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model  # noqa: E402
from util import CpuEnv  # noqa: E402

B, STEPS, SEED = 4, 60, 1


def actions():
    """N(0,1)*0.5 from a counter-based stream keyed (seed, step)."""
    out = np.zeros((STEPS, B, 16), np.float32)
    for s in range(STEPS):
        out[s] = (0.5 * np.random.default_rng([SEED, s]).standard_normal((B, 16))).astype(np.float32)
    return out


def main():
    env = CpuEnv(load_cpu_oracle_lib(fast=True), pack_model(), B, seed=SEED)
    env.reset()
    acts = actions()
    obs, priv, rew, rst, to, ncon, qpos = [], [], [], [], [], [], []
    for a in acts:
        o, p, r, d, t = env.step(a)
        obs.append(o.copy()); priv.append(p.copy()); rew.append(r.copy()); rst.append(d.copy()); to.append(t.copy())
        ncon.append(env.field("I_N_CONTACTS")[0].copy()); qpos.append(env.field("F_QPOS").copy())
    out = os.path.join(ROOT, "tests", "golden", "walk_b4_seed1.npz")
    np.savez_compressed(out, actions=acts, obs=np.array(obs), priv=np.array(priv), rew=np.array(rew), reset=np.array(rst), time_out=np.array(to),
                        n_contacts=np.array(ncon), qpos=np.array(qpos), seed=SEED)
    print("wrote", out, "resets:", int(np.array(rst).sum()), "mean contacts:", float(np.mean(ncon)))


if __name__ == "__main__":
    main()
