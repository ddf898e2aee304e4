"""include/go2sim_detmath.h: accuracy vs numpy (float64) and the Philox known-answer test."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from go2_sim2real_locomotion_rl_amd.capi import REPO_ROOT

SRC = r"""
#include "go2sim_detmath.h"
extern "C" {
void dm_test(int fn, int n, const float* x, const float* y, float* out) {
  for (int i = 0; i < n; ++i) {
    switch (fn) {
      case 0: out[i] = dm_sin(x[i]); break; case 1: out[i] = dm_cos(x[i]); break; case 2: out[i] = dm_atan2(y[i], x[i]); break;
      case 3: out[i] = dm_acos(x[i]); break; case 4: out[i] = dm_exp(x[i]); break; case 5: out[i] = dm_log(x[i]); break;
      case 6: out[i] = dm_pow(x[i], y[i]); break;
    }
  }
}
void dm_philox_test(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
  dm_u4 r = dm_philox(c0, c1, c2, c3, k0, k1); for (int i = 0; i < 4; ++i) out[i] = r.v[i]; }
void dm_normal_test(int n, float* out) { for (int i = 0; i < n; i += 2) { dm_u4 r = dm_philox(i, 0, 0, 0, 1, 2); dm_normal2(r.v[0], r.v[1], out + i, out + i + 1); } }
}
"""


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    d = tmp_path_factory.mktemp("dm")
    src = d / "dm.cpp"
    src.write_text(SRC)
    so = d / "dm.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(REPO_ROOT, "include"), str(src), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _call(dm, fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.zeros_like(x)
    dm.dm_test(fn, x.size, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-45)


def test_trig_accuracy(dm):
    rng = np.random.default_rng(0)
    x = rng.uniform(-20, 20, 200000).astype(np.float32)
    keep = np.abs(np.sin(x.astype(np.float64))) > 1e-3
    assert _ulp_err(_call(dm, 0, x), np.sin(x.astype(np.float64)))[keep].max() < 2.5
    keep = np.abs(np.cos(x.astype(np.float64))) > 1e-3
    assert _ulp_err(_call(dm, 1, x), np.cos(x.astype(np.float64)))[keep].max() < 2.5


def test_inverse_trig_exp_log_accuracy(dm):
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, 200000).astype(np.float32); y = rng.uniform(-1, 1, 200000).astype(np.float32)
    assert _ulp_err(_call(dm, 2, x, y), np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 4.5
    u = rng.uniform(-0.9999, 0.9999, 200000).astype(np.float32)
    assert _ulp_err(_call(dm, 3, u), np.arccos(u.astype(np.float64))).max() < 2.5
    v = rng.uniform(-30, 30, 200000).astype(np.float32)
    assert _ulp_err(_call(dm, 4, v), np.exp(v.astype(np.float64))).max() < 2.0
    w = rng.uniform(1e-6, 100, 200000).astype(np.float32)
    keep = np.abs(np.log(w.astype(np.float64))) > 1e-3
    assert _ulp_err(_call(dm, 5, w), np.log(w.astype(np.float64)))[keep].max() < 2.0


def test_edge_cases(dm):
    assert _call(dm, 3, [1.0, -1.0, 1.5, -1.5]).tolist() == [0.0, np.float32(np.pi), 0.0, np.float32(np.pi)]
    assert _call(dm, 2, [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]).tolist() == [0.0, np.float32(np.pi / 2), np.float32(np.pi)]
    x = np.float32([0.3, 1.7, 0.0])
    assert np.array_equal(_call(dm, 6, x, np.float32([2, 2, 2])), x * x)  # pow(x, 2) is the exact product
    assert _call(dm, 4, [0.0])[0] == 1.0


def test_philox_known_answers(dm):
    out = (ctypes.c_uint * 4)()
    dm.dm_philox_test(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]  # Random123 kat_vectors
    dm.dm_philox_test(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_normal_moments(dm):
    n = 200000
    out = np.zeros(n, np.float32)
    dm.dm_normal_test(n, out.ctypes.data_as(ctypes.c_void_p))
    assert abs(out.mean()) < 0.01 and abs(out.var() - 1.0) < 0.02
