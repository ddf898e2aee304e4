"""Documents the one compiler workaround of the product ("update_bracket"): the bracket step of the exact line search
(`update_bracket` in csrc/go2sim.hip; update_bracket_no_eval_local, constraint/solver.py:2212-2243) is kept out of line.

The same source built with -DGO2SIM_BRACKET_INLINE (the function inlined into k_constraint_solve_team, hipcc 7.2 -O3) stops agreeing with the
CPU oracle after a few dozen steps of the walk task; every other build variant tried agrees bit for bit (tools/repro_bracket/README.md holds the
matrix).  This test runs the inlined build and EXPECTS the mismatch: it is an xfail.  If it ever passes (XPASS) the compiler no longer shows the
behaviour and the `__noinline__` can go."""
import os

import numpy as np
import pytest

from util import CpuEnv, GpuEnv, bits_equal, make_actions

pytestmark = pytest.mark.gpu


@pytest.mark.xfail(strict=False, reason="hipcc 7.2 -O3 with update_bracket inlined: solver results leave the oracle (kept __noinline__ in the product)")
def test_inlined_update_bracket_build_matches_oracle(oracle_lib, blob):
    from go2_sim2real_locomotion_rl_amd import build
    from go2_sim2real_locomotion_rl_amd.capi import Go2SimLib

    so = build.build_hip_variant("bracket_inline", ["-DGO2SIM_BRACKET_INLINE"], verbose=False)
    lib = Go2SimLib(os.path.abspath(so), "go2sim_")
    n, steps = 64, 150
    cpu, gpu = CpuEnv(oracle_lib, blob, n, seed=7), GpuEnv(lib, blob, n, seed=7)
    cpu.reset(); gpu.reset()
    for s, a in enumerate(make_actions(steps, n, seed=7, kind="mixed")):
        oc, pc, rc, dc, tc = cpu.step(a)
        og, pg, rg, dg, tg = gpu.step(a)
        assert np.array_equal(dc, dg) and bits_equal(oc, og) and bits_equal(rc, rg), f"inlined build differs from the oracle at step {s}"
