#!/usr/bin/env python3
"""Bracket-step trace of the exact line search on two builds of the HIP library (GPU box):

    python tools/repro_bracket/trace_bracket.py

builds  tools/lib_brdbg.so         = -DGO2SIM_FAST_ORDER=0 -DGO2SIM_BRACKET_DEBUG                      (update_bracket out of line: agrees with the oracle)
        tools/lib_brdbg_inline.so  = -DGO2SIM_FAST_ORDER=0 -DGO2SIM_BRACKET_DEBUG -DGO2SIM_BRACKET_INLINE  (update_bracket inlined)
runs tests/test_parity_gpu.py::test_env_step_bit_exact[4-80-0.5-1] (4 envs, one env per wavefront: GO2SIM_SOLVER_TEAM=64) on both and prints the
first bracket step whose logged values differ: the three candidate points, both brackets before and after the step, the next alphas and the flags."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GO2SIM_SOLVER_TEAM"] = "64"
from go2_sim2real_locomotion_rl_amd import build  # noqa: E402
from go2_sim2real_locomotion_rl_amd.capi import Go2SimLib  # noqa: E402
from go2_sim2real_locomotion_rl_amd.model_blob import pack_model  # noqa: E402
from util import GpuEnv, make_actions  # noqa: E402

FIELDS = (["ls_it", "gtol"] + [f"al{i}" for i in range(3)] + [f"cost{i}" for i in range(3)] + [f"grad{i}" for i in range(3)] + [f"hess{i}" for i in range(3)] +
          [f"p1_in.{k}" for k in ("alpha", "cost", "grad", "hess")] + [f"p2_in.{k}" for k in ("alpha", "cost", "grad", "hess")] +
          [f"p1_out.{k}" for k in ("alpha", "cost", "grad", "hess")] + [f"p2_out.{k}" for k in ("alpha", "cost", "grad", "hess")] +
          ["p1_next_alpha", "p2_next_alpha", "b1", "b2"])
ENVS, CAP, W = 4, 8192, 40


def trace(name, flags, steps=80):
    so = build.build_hip_variant(name, ["-DGO2SIM_FAST_ORDER=0", "-DGO2SIM_BRACKET_DEBUG"] + flags, verbose=False)
    lib = Go2SimLib(os.path.abspath(so), "go2sim_")
    env = GpuEnv(lib, pack_model(), 4, seed=1)
    env.reset()
    obs = []
    for a in make_actions(steps, 4, seed=1, kind="0.5"):
        obs.append(env.step(a)[0].copy())
    out = np.zeros(ENVS * CAP * W, np.float32); cnt = np.zeros(ENVS, np.int32)
    rc = lib.lib.go2sim_debug_brlog(env.sim.h, out.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out.reshape(ENVS, CAP, W), cnt, np.array(obs)


def main():
    a, ca, oa = trace("brdbg", [])
    b, cb, ob = trace("brdbg_inline", ["-DGO2SIM_BRACKET_INLINE"])
    print("bracket steps logged per env: out of line", ca.tolist(), " inlined", cb.tolist())
    d = np.abs(oa - ob).reshape(len(oa), -1).max(1)
    first_obs = int(np.argmax(d > 0)) if (d > 0).any() else -1
    print("first env step with different observations:", first_obs)
    for e in range(ENVS):
        n = min(ca[e], cb[e])
        same = (a[e, :n].view(np.int32) == b[e, :n].view(np.int32)).all(1)
        if same.all():
            print(f"env {e}: the first {n} bracket steps are bit-identical")
            continue
        k = int(np.argmin(same))
        print(f"env {e}: bracket step {k} is the first that differs")
        for j, name in enumerate(FIELDS[:len(a[e, k])]):
            x, y = a[e, k, j], b[e, k, j]
            mark = "" if np.float32(x).view(np.int32) == np.float32(y).view(np.int32) else "   <-- differs"
            print(f"    {name:14s} out-of-line {x:+.9e}   inlined {y:+.9e}{mark}")


if __name__ == "__main__":
    main()
