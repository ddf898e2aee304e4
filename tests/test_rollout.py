"""Rollout storage (include/go2sim_policy.h): GAE(lambda) returns + advantage normalisation.  Oracle vs the rsl_rl 2.2.4 formulas written
in plain PyTorch (RolloutStorage.compute_returns + PPO.process_env_step), GPU vs oracle bit for bit, global statistics over two ranks (gloo)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from go2_sim2real_locomotion_rl_amd.rollout import RolloutBuffers

GAMMA, LAM = 0.99, 0.95


def make_data(T, B, seed):
    g = torch.Generator().manual_seed(seed)
    rew = torch.randn(T, B, generator=g) * 0.1
    val = torch.randn(T, B, generator=g)
    don = (torch.rand(T, B, generator=g) < 0.05).to(torch.uint8)
    tmo = ((torch.rand(T, B, generator=g) < 0.5) & don.bool()).float()
    last = torch.randn(B, generator=g)
    return rew, val, don, tmo, last


def torch_reference(rew, val, don, tmo, last):
    """rsl_rl 2.2.4: PPO.process_env_step (time-out bootstrap) + RolloutStorage.compute_returns."""
    T, B = rew.shape
    rewards = rew + GAMMA * (val * tmo)
    returns = torch.zeros(T, B)
    advantage = torch.zeros(B)
    for step in reversed(range(T)):
        next_values = last if step == T - 1 else val[step + 1]
        next_is_not_terminal = 1.0 - don[step].float()
        delta = rewards[step] + next_is_not_terminal * GAMMA * next_values - val[step]
        advantage = delta + next_is_not_terminal * GAMMA * LAM * advantage
        returns[step] = advantage + val[step]
    adv = returns - val
    return returns, adv, (adv - adv.mean()) / (adv.std() + 1e-8)


def run(lib, data, device_tensors=False):
    rew, val, don, tmo, last = data
    T, B = rew.shape
    rb = RolloutBuffers(lib, T, B)
    dev = "cuda" if device_tensors else "cpu"
    mom = torch.zeros(3, dtype=torch.float64, device=dev)
    for t in range(T):
        rb.add(t, rew[t].contiguous().to(dev), don[t].contiguous().to(dev), val[t].contiguous().to(dev), tmo[t].contiguous().to(dev), GAMMA)
    rb.compute_returns(last.to(dev), GAMMA, LAM, mom)
    return rb, mom


def read(rb, which, device_tensors=False):
    """[T, B] float32 copy of one buffer of the handle (host memory for the oracle, device memory for the product)."""
    import ctypes

    n = rb.T * rb.B
    if device_tensors:
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(rb.ptr(which)), ctypes.c_size_t(4 * n), ctypes.c_int(3))   # device to device
        assert rc == 0
        return out.cpu().numpy().reshape(rb.T, rb.B)
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(rb.ptr(which))).reshape(rb.T, rb.B).copy()


@pytest.mark.parametrize("T,B", [(24, 300), (24, 4096), (1, 1), (5, 257)])
def test_oracle_gae_matches_rsl_rl_formulas(oracle_lib, T, B):
    data = make_data(T, B, seed=T + B)
    rb, mom = run(oracle_lib, data)
    ret_ref, adv_ref, norm_ref = torch_reference(*data)
    assert np.array_equal(read(rb, "RETURNS"), ret_ref.numpy())               # same fp32 operation order as the torch loop
    assert np.array_equal(read(rb, "ADVANTAGES"), adv_ref.numpy())
    assert mom[2].item() == T * B and abs(mom[0].item() - adv_ref.double().sum().item()) < 1e-9 * max(1, T * B)
    if T * B > 1:
        rb.normalize(mom)
        assert np.allclose(read(rb, "ADVANTAGES"), norm_ref.numpy(), rtol=1e-5, atol=1e-6)


def _worker(rank, world, port, lib_path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from go2_sim2real_locomotion_rl_amd.capi import Go2SimLib
    from go2_sim2real_locomotion_rl_amd.distributed import allgather_moments

    lib = Go2SimLib(lib_path, "go2sim_cpu_")
    data = make_data(24, 96, seed=100 + rank)                   # every rank owns its own envs
    rb, mom = run(lib, data)
    rb.normalize(allgather_moments(mom))
    q.put((rank, read(rb, "ADVANTAGES")))
    dist.barrier()
    dist.destroy_process_group()


def test_global_normalisation_world2(oracle_lib):
    """Two ranks (gloo): the normalised advantages equal those of one process that owns all envs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, oracle_lib.path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [make_data(24, 96, seed=100 + r) for r in range(2)]
    both = tuple(torch.cat([parts[0][k], parts[1][k]], dim=-1) for k in range(5))
    rb, mom = run(oracle_lib, both)
    rb.normalize(mom)
    ref = read(rb, "ADVANTAGES")
    assert np.allclose(np.concatenate([got[0], got[1]], axis=1), ref, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(24, 4096), (24, 300), (3, 1)])
def test_gpu_gae_bit_exact_vs_oracle(oracle_lib, hip_lib, T, B):
    data = make_data(T, B, seed=B)
    rc, mc = run(oracle_lib, data)
    rg, mg = run(hip_lib, data, device_tensors=True)
    torch.cuda.synchronize()
    assert np.array_equal(read(rc, "RETURNS"), read(rg, "RETURNS", device_tensors=True))
    assert np.array_equal(read(rc, "ADVANTAGES"), read(rg, "ADVANTAGES", device_tensors=True))
    assert np.array_equal(mc.numpy(), mg.cpu().numpy())                         # fixed summation tree => identical float64 moments
    if T * B > 1:
        rc.normalize(mc); rg.normalize(mg)
        torch.cuda.synchronize()
        assert np.array_equal(read(rc, "ADVANTAGES"), read(rg, "ADVANTAGES", device_tensors=True))


@pytest.mark.gpu
def test_rollout_storage_class(hip_lib):
    from go2_sim2real_locomotion_rl_amd import RolloutStorage

    T, B = 24, 512
    rew, val, don, tmo, last = make_data(T, B, seed=5)
    st = RolloutStorage(T, B)
    for t in range(T):
        st.add_transitions(t, rew[t].cuda(), don[t].cuda(), val[t].cuda().unsqueeze(-1), tmo[t].cuda(), gamma=GAMMA)
    ret, adv = st.compute_returns(last.cuda().unsqueeze(-1), GAMMA, LAM)
    torch.cuda.synchronize()
    ret_ref, _, norm_ref = torch_reference(rew, val, don, tmo, last)
    assert torch.equal(ret.cpu(), ret_ref) and torch.allclose(adv.cpu(), norm_ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(st.dones.cpu(), don)
