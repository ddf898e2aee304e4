"""Policy inference (include/go2sim_policy.h): CPU oracle vs a plain PyTorch fp32 reference of the same modules, GPU vs the oracle.

The reference-side modules are rsl_rl's ActorCritic (rsl-rl-lib==2.2.4, not importable here): nn.Sequential(Linear, ELU, ...) + Normal.
Tolerances: oracle vs torch fp32 differ by summation order only -> |diff| <= 2e-5 + 2e-5 |ref| at unit-scale activations;
GPU vs oracle: tolerance 0 (the fp32 matrix instruction is a k-ordered fma chain and the oracle walks K in the same order)."""
import ctypes
import math

import numpy as np
import pytest
import torch

from go2_sim2real_locomotion_rl_amd.policy import Mlp, flatten_sequential, policy_act

ACTOR = [49, 512, 256, 128, 16]
CRITIC = [104, 512, 256, 128, 1]


def torch_mlp(dims, seed):
    torch.manual_seed(seed)
    layers = []
    for l in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[l], dims[l + 1]))
        if l < len(dims) - 2:
            layers.append(torch.nn.ELU())
    return torch.nn.Sequential(*layers)


def make(lib, dims, seed):
    net = torch_mlp(dims, seed)
    sd = {f"net.{k}": v for k, v in net.state_dict().items()}
    params, d = flatten_sequential(sd, "net", len(dims) - 1)
    assert d == dims
    return net, Mlp(lib, dims, params), params


@pytest.mark.parametrize("dims,rows", [(ACTOR, 70), (CRITIC, 33), ([7, 5, 3], 1), ([20, 16], 17), ([3, 512, 512, 512, 512, 2], 5)])
def test_oracle_mlp_matches_torch_fp32(oracle_lib, dims, rows):
    net, mlp, _ = make(oracle_lib, dims, seed=len(dims) + rows)
    x = torch.randn(rows, dims[0])
    y = np.zeros((rows, dims[-1]), np.float32)
    mlp.forward(np.ascontiguousarray(x.numpy()), y, rows)
    ref = net(x).detach().numpy()
    assert np.all(np.abs(y - ref) <= 2e-5 + 2e-5 * np.abs(ref)), float(np.abs(y - ref).max())


def test_oracle_mlp_argument_checks(oracle_lib):
    h = ctypes.c_void_p()
    dims = (ctypes.c_int * 3)(4, 8, 2)
    p = np.zeros(4 * 8 + 8 + 8 * 2 + 2, np.float32)
    f = oracle_lib.fn("mlp_create")
    assert f(0, dims, 2, p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.size - 1), ctypes.byref(h)) != 0      # wrong parameter count
    big = (ctypes.c_int * 2)(4, 4096)
    assert f(0, big, 1, p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.size), ctypes.byref(h)) != 0            # too wide
    assert f(0, dims, 2, p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.size), ctypes.byref(h)) == 0
    assert oracle_lib.fn("mlp_forward")(h, None, None, 3, None) != 0
    assert oracle_lib.fn("mlp_destroy")(h) == 0


def test_oracle_policy_act_matches_torch_normal(oracle_lib):
    """ActorCritic.act / evaluate / get_actions_log_prob semantics against torch.distributions.Normal."""
    B = 300
    anet, actor, _ = make(oracle_lib, ACTOR, 1)
    cnet, critic, _ = make(oracle_lib, CRITIC, 2)
    obs, cobs = torch.randn(B, 49), torch.randn(B, 104)
    std = (0.3 + torch.rand(16)).numpy().astype(np.float32)
    act = np.zeros((B, 16), np.float32); mean = np.zeros((B, 16), np.float32); val = np.zeros(B, np.float32); lp = np.zeros(B, np.float32)
    policy_act(oracle_lib, actor, critic, obs.numpy(), cobs.numpy(), std, B, 5, 7, False, act, mean, val, lp)
    mu_ref = anet(obs).detach()
    assert np.allclose(mean, mu_ref.numpy(), rtol=2e-5, atol=2e-5)
    assert np.allclose(val, cnet(cobs).detach().numpy()[:, 0], rtol=2e-5, atol=2e-5)
    dist = torch.distributions.Normal(torch.from_numpy(mean), torch.from_numpy(std).expand(B, 16))
    assert np.allclose(lp, dist.log_prob(torch.from_numpy(act)).sum(-1).numpy(), rtol=1e-5, atol=2e-4)
    z = (act - mean) / std                                           # the noise is standard normal, independent per row / action / step
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05 and abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.15
    act2 = np.zeros_like(act)
    policy_act(oracle_lib, actor, critic, obs.numpy(), cobs.numpy(), std, B, 5, 8, False, act2, None, val, None)
    assert not np.array_equal(act, act2)                             # new step -> new noise
    policy_act(oracle_lib, actor, critic, obs.numpy(), cobs.numpy(), std, B, 5, 7, False, act2, None, val, None)
    assert np.array_equal(act, act2)                                 # counter-based: same (seed, step) -> same sample
    policy_act(oracle_lib, actor, None, obs.numpy(), None, std, B, 5, 7, True, act2, None, None, lp)
    assert np.array_equal(act2, mean)                                # act_inference
    assert np.allclose(lp, -(np.log(std).sum() + 16 * 0.5 * math.log(2 * math.pi)), atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,rows", [(ACTOR, 4096), (CRITIC, 1000), ([7, 5, 3], 1), ([20, 16], 17), ([3, 512, 512, 512, 512, 2], 37), ([49, 512, 256, 128, 12], 130)])
def test_gpu_mlp_bit_exact_vs_oracle(oracle_lib, hip_lib, dims, rows):
    net, cpu, params = make(oracle_lib, dims, seed=rows)
    gpu = Mlp(hip_lib, dims, params)
    x = torch.randn(rows, dims[0])
    yc = np.zeros((rows, dims[-1]), np.float32)
    cpu.forward(np.ascontiguousarray(x.numpy()), yc, rows)
    xg = x.cuda(); yg = torch.full((rows, dims[-1]), float("nan"), device="cuda")
    gpu.forward(xg, yg, rows, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = yg.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), yc.view(np.uint32)), float(np.abs(got - yc).max())
    ref = net(x).detach().numpy()                                   # and the plain PyTorch fp32 reference of the same op
    assert np.all(np.abs(got - ref) <= 2e-5 + 2e-5 * np.abs(ref))
    new = (params * 0.5).astype(np.float32)                          # parameter update path
    gpu.set_params(new); cpu.set_params(new)
    cpu.forward(np.ascontiguousarray(x.numpy()), yc, rows)
    gpu.forward(xg, yg, rows, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(yg.cpu().numpy().view(np.uint32), yc.view(np.uint32))


@pytest.mark.gpu
def test_gpu_policy_act_bit_exact_vs_oracle(oracle_lib, hip_lib):
    B = 4096
    _, ca, pa = make(oracle_lib, ACTOR, 3); _, cc, pc = make(oracle_lib, CRITIC, 4)
    ga, gc = Mlp(hip_lib, ACTOR, pa), Mlp(hip_lib, CRITIC, pc)
    obs, cobs = torch.randn(B, 49), torch.randn(B, 104)
    std = (0.3 + torch.rand(16))
    out_c = [np.zeros((B, 16), np.float32), np.zeros((B, 16), np.float32), np.zeros(B, np.float32), np.zeros(B, np.float32)]
    policy_act(oracle_lib, ca, cc, obs.numpy(), cobs.numpy(), std.numpy(), B, 9, 3, False, *out_c)
    out_g = [torch.zeros(B, 16, device="cuda"), torch.zeros(B, 16, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")]
    policy_act(hip_lib, ga, gc, obs.cuda(), cobs.cuda(), std.cuda(), B, 9, 3, False, *out_g, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for c, g in zip(out_c, out_g):
        assert np.array_equal(c.view(np.uint32), g.cpu().numpy().view(np.uint32))


@pytest.mark.gpu
def test_actor_critic_class_with_env(hip_lib):
    """The rsl_rl-shaped class drives the env in a closed loop (what OnPolicyRunner's rollout does per step)."""
    from go2_sim2real_locomotion_rl_amd import ActorCritic, Go2Env, init
    from go2_sim2real_locomotion_rl_amd.configs import get_walk_cfgs

    init(seed=1)
    env = Go2Env(64, *get_walk_cfgs())
    pol = ActorCritic(env.num_obs, env.num_privileged_obs, env.num_actions, [512, 256, 128], [512, 256, 128], activation="elu", init_noise_std=1.0)
    ref_actor = torch_mlp(ACTOR, 0)
    sd = {f"actor.{k}": v for k, v in ref_actor.state_dict().items()}
    sd.update({f"critic.{k}": v for k, v in torch_mlp(CRITIC, 1).state_dict().items()})
    sd["std"] = 0.1 * torch.ones(16)
    pol.load_state_dict(sd)
    obs, extras = env.get_observations()
    for _ in range(30):
        actions = pol.act(obs, extras["observations"]["critic"])
        assert pol.values.shape == (64, 1) and pol.get_actions_log_prob(actions).shape == (64,)
        obs, rew, dones, extras = env.step(actions)
    torch.cuda.synchronize()
    mu = ref_actor.cuda()(obs).detach()
    assert torch.allclose(pol.act_inference(obs), mu, rtol=2e-5, atol=2e-5)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    pol.act(obs, extras["observations"]["critic"])
    assert torch.equal(pol.evaluate(extras["observations"]["critic"]), pol.values)
