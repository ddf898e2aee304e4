#!/usr/bin/env python3
"""profiles/rNN_pmc_sq.json from the SQ pass of tools/profile_round.sh (pmc_summary.py --json), stamped with bench.source_hash() so that bench.py
refuses it for any other build of the kernels.   usage: make_pmc_sq.py SQ.json OUT.json "<command line that was profiled>"

Derived per kernel (what `roofline_issue` of the bench line reports for the dominant kernel):
  valu_issue_util  = 4 x SQ_INSTS_VALU / (1024 SIMDs x kernel cycles): the share of the chip's vector issue slots the launch used over its DURATION
                     (a wave64 instruction occupies its SIMD for 4 cycles; kernel cycles = average duration x the nominal 2.4 GHz)
  valu_busy_while_alive = 4 x SQ_INSTS_VALU / (4 x SQ_WAVE_CYCLES / waves_per_simd): the same while the mean wave is alive
  lane_occupancy   = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per vector instruction
  waves_per_simd   = SQ_WAVES / 1024
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash

CLOCK_MHZ, N_SIMD = 2400.0, 1024
sq = json.load(open(sys.argv[1]))
CLASS = [("k_constraint_solve", r"^k_constraint_solve_team"), ("k_solve_integrate_dyn", r"^k_solve_integrate_team<.*true>"), ("k_solve_integrate", r"^k_solve_integrate_team<.*false>"),
         ("k_collide", r"^k_collide_team"), ("k_pre_dynamics", r"^k_pre_dynamics_team"),
         ("k_integrate_fk", r"^k_integrate_fk_team"), ("k_integrate_fk_dynamics", r"^k_integrate_fk_dynamics_team"),
         ("k_env_post_a", r"^k_env_post_a"), ("k_env_post_b", r"^k_env_post_b")]
kernels = {}
for name, pat in CLASS:
    rec = [v for k, v in sq.items() if re.match(pat, k)]
    if not rec:
        continue
    r = rec[0]
    cyc = r["avg_us"] * CLOCK_MHZ
    waves = r.get("SQ_WAVES", 0.0)
    d = {"launches_averaged": r["n"], "avg_us_under_pmc": round(r["avg_us"], 2)}
    d.update({k: round(v, 1) for k, v in r.items() if k.startswith("SQ_") or k.startswith("GRBM_")})
    if "SQ_INSTS_VALU" in r:
        d["valu_issue_util"] = round(4.0 * r["SQ_INSTS_VALU"] / (N_SIMD * cyc), 4)
        if r.get("SQ_WAVE_CYCLES"):
            wps = max(1.0, waves / N_SIMD)
            d["valu_busy_while_alive"] = round(4.0 * r["SQ_INSTS_VALU"] / (4.0 * r["SQ_WAVE_CYCLES"] / wps), 4)
    if r.get("SQ_THREAD_CYCLES_VALU") and r.get("SQ_ACTIVE_INST_VALU"):
        d["lane_occupancy"] = round(r["SQ_THREAD_CYCLES_VALU"] / (64.0 * r["SQ_ACTIVE_INST_VALU"]), 4)
    d["waves_per_simd"] = round(waves / N_SIMD, 3)
    kernels[name] = d
if "k_constraint_solve" not in kernels and "k_solve_integrate_dyn" in kernels:     # flat ground: the solve lives in k_solve_integrate_team; the bench's solver class reads the first-substep variant
    kernels["k_constraint_solve"] = dict(kernels["k_solve_integrate_dyn"], note="k_solve_integrate_team<.., true>: solve + integrate + kinematics + next dynamics in one launch")
doc = {"_comment": __doc__.split("Derived per kernel")[1].strip(), "command": sys.argv[3], "source_sha256": source_hash(), "clock_mhz_nominal": CLOCK_MHZ,
       "simds": N_SIMD, "kernels": kernels}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: {q: v[q] for q in ("avg_us_under_pmc", "valu_issue_util", "valu_busy_while_alive", "lane_occupancy", "waves_per_simd") if q in v} for k, v in kernels.items()}, indent=1))
