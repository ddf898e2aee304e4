"""Build recipes.  `build_hip()` compiles the product (hipcc, gfx950); `build_oracle()` compiles the CPU
oracle (g++) -- building the checker is not using it."""
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HIP_SRC = os.path.join(_HERE, "csrc", "go2sim.hip")
HIP_SRC_POLICY = os.path.join(_HERE, "csrc", "go2sim_policy.hip")
HIP_HDR_GJK = os.path.join(_HERE, "csrc", "go2sim_gjk_dev.h")
HIP_LIB = os.path.join(_HERE, "csrc", "libgo2sim.so")
ORACLE_SRC = os.path.join(REPO_ROOT, "oracle", "go2sim_cpu.cpp")
ORACLE_SRC_POLICY = os.path.join(REPO_ROOT, "oracle", "policy_cpu.cpp")
ORACLE_LIB = os.path.join(REPO_ROOT, "oracle", "libgo2sim_cpu.so")             # strict: the reference's CPU (serial) summation order
ORACLE_LIB_FAST = os.path.join(REPO_ROOT, "oracle", "libgo2sim_cpu_fast.so")   # -DGO2SIM_FAST_ORDER: mirrors the HIP product's reduction order bit for bit

# -ffp-contract=off on BOTH sides is part of the numeric contract (include/go2sim_detmath.h):
# identical IEEE binary32 operation sequences => bit-identical CPU/GPU results.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
             "-Wno-unused-const-variable"]
CPU_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _newer(target, *sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _headers():
    inc = os.path.join(REPO_ROOT, "include")
    return [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]


def build_hip(force=False, verbose=True):
    if not force and _newer(HIP_LIB, HIP_SRC, HIP_SRC_POLICY, HIP_HDR_GJK, *_headers()):
        return HIP_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIP_FLAGS, HIP_SRC, HIP_SRC_POLICY, "-o", HIP_LIB]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return HIP_LIB


def build_hip_variant(name, extra_flags, force=False, verbose=True):
    """A diagnostic BUILD of the product library with extra -D flags (tools/lib_<name>.so); test infrastructure for builds the
    product does not ship -- e.g. ("bracket_inline", ["-DGO2SIM_BRACKET_INLINE"]), see DESIGN.md "update_bracket"."""
    out = os.path.join(REPO_ROOT, "tools", f"lib_{name}.so")
    if not force and _newer(out, HIP_SRC, HIP_SRC_POLICY, HIP_HDR_GJK, *_headers()):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIP_FLAGS, *extra_flags, HIP_SRC, HIP_SRC_POLICY, "-o", out]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_oracle(force=False, verbose=True):
    """Both oracle builds: the strict one (reference CPU order) and the FAST ORDER mirror of the HIP product (oracle/go2sim_cpu.cpp header)."""
    deps = [ORACLE_SRC, ORACLE_SRC_POLICY, os.path.join(REPO_ROOT, "oracle", "gjk_epa_cpu.h"), *_headers()]
    for lib, extra in ((ORACLE_LIB, []), (ORACLE_LIB_FAST, ["-DGO2SIM_FAST_ORDER"])):
        if not force and _newer(lib, *deps):
            continue
        cmd = ["g++", *CPU_FLAGS, *extra, ORACLE_SRC, ORACLE_SRC_POLICY, "-o", lib]
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return ORACLE_LIB


def build_oracle_variant(name, extra_flags, force=False, verbose=True):
    """A diagnostic BUILD of the CPU oracle (oracle/libgo2sim_cpu_<name>.so), e.g. ("rng_const_fast", ["-DGO2SIM_RNG_CONST", "-DGO2SIM_FAST_ORDER"]):
    the generator replaced by the constant schedule of include/go2sim_detmath.h, the counterpart of the reference env files run with
    torch.rand / randn_like / randint / randperm replaced by the same schedule (tools/make_ref_env_fixtures.py)."""
    out = os.path.join(REPO_ROOT, "oracle", f"libgo2sim_cpu_{name}.so")
    deps = [ORACLE_SRC, ORACLE_SRC_POLICY, os.path.join(REPO_ROOT, "oracle", "gjk_epa_cpu.h"), *_headers()]
    if not force and _newer(out, *deps):
        return out
    cmd = ["g++", *CPU_FLAGS, *extra_flags, ORACLE_SRC, ORACLE_SRC_POLICY, "-o", out]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


# diagnostic builds the tests use (name -> extra flags); `python -m go2_sim2real_locomotion_rl_amd.build --variants` builds them all so that they
# travel to the GPU box with the snapshot instead of being compiled there
HIP_VARIANTS = {
    "bracket_inline": ["-DGO2SIM_BRACKET_INLINE"],          # tests/test_bracket_inline.py (expected failure)
    "rng_const": ["-DGO2SIM_RNG_CONST"],                     # tests/test_ref_env_rng_fixtures.py
    "strict": ["-DGO2SIM_FAST_ORDER=0"],                     # tests/test_strict_order_gpu.py: the reference's summation order
}
ORACLE_VARIANTS = {
    "rng_const": ["-DGO2SIM_RNG_CONST"],
    "rng_const_fast": ["-DGO2SIM_RNG_CONST", "-DGO2SIM_FAST_ORDER"],
}


# SHAPE VARIANTS: the same sources compiled for another link / dof / qpos / geom / joint count (include/go2sim.h GO2SIM_NL ...), in the reference's
# summation order (-DGO2SIM_FAST_ORDER=0: the FAST ORDER block forms are laid out for 18 dofs).  Test infrastructure for the reference's analytic known
# answers (tests/test_analytic_shapes.py; models: tools/compile_go2_model.py --robot pendulum | double_pendulum | box).
SHAPES = {
    "pendulum": dict(NL=3, ND=1, NQ=1, NG=2, NJ=1),           # plane | fixed base, arm + point mass (1 continuous joint)
    "double_pendulum": dict(NL=4, ND=2, NQ=2, NG=3, NJ=2),
    "box": dict(NL=2, ND=6, NQ=7, NG=2, NJ=1),                 # plane | free cube
}


def _shape_flags(name):
    return [f"-DGO2SIM_{k}={v}" for k, v in SHAPES[name].items()]


def build_shape_variant(name, hip=True, force=False, verbose=True):
    """(oracle/libgo2sim_cpu_shape_<name>.so, tools/lib_shape_<name>.so or None)"""
    cpu = build_oracle_variant("shape_" + name, _shape_flags(name), force=force, verbose=verbose)
    gpu = build_hip_variant("shape_" + name, ["-DGO2SIM_FAST_ORDER=0", *_shape_flags(name)], force=force, verbose=verbose) if hip else None
    return cpu, gpu


def build_variants(force=False, verbose=True, jobs=4):
    """Every diagnostic / shape-variant library the tests load, `jobs` compilers at a time (each build is one hipcc or g++ process; a fresh tree needs nine
    HIP builds of about a minute each)."""
    from concurrent.futures import ThreadPoolExecutor

    have_hip = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")
    tasks = [(build_oracle_variant, ("shape_" + n, _shape_flags(n))) for n in SHAPES]
    tasks += [(build_oracle_variant, (n, f)) for n, f in ORACLE_VARIANTS.items()]
    if have_hip:
        tasks += [(build_hip_variant, ("shape_" + n, ["-DGO2SIM_FAST_ORDER=0", *_shape_flags(n)])) for n in SHAPES]
        tasks += [(build_hip_variant, (n, f)) for n, f in HIP_VARIANTS.items()]
    with ThreadPoolExecutor(max_workers=max(1, int(jobs))) as pool:
        futs = [pool.submit(fn, *args, force=force, verbose=verbose) for fn, args in tasks]
        for f in futs:
            f.result()


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    build_oracle(force="--force" in sys.argv)
    if "--variants" in sys.argv:
        build_variants(force="--force" in sys.argv)
