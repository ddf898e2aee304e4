"""Multi-GPU sharding helpers.

Besides the advantage statistics, the reference's env has ONE curriculum level and ONE set of "global" DR scalars for all its envs
(go2_env_walk.py:458-463, 737-756, 803-848).  `sync_env_globals` keeps that true for a batch sharded over ranks: an all-reduce of the four
curriculum counters (+ the friction-throttle counter), the same state machine on every rank, and a broadcast of rank 0's draws.

Environments are fully independent, so the hot path shards as contiguous blocks of envs per GPU with NO
physics traffic.  The only collective the path needs is the rollout advantage-normalisation statistics:
an all-gather of [sum, sum of squares, count] (3 floats per rank) over RCCL/xGMI (backend "nccl" on ROCm,
"gloo" in the CPU tests).  The reference has no distributed code on this path (its only multi-GPU artefact
is examples/ddp_multi_gpu.py:36-90, one independent scene per rank with seed=local_rank)."""
import torch
import torch.distributed as dist


def shard_seed(seed: int, rank: int) -> int:
    """Per-rank RNG stream (examples/ddp_multi_gpu.py:58 uses seed=local_rank)."""
    return int(seed) + int(rank)


def shard_envs(total_envs: int, world_size: int, rank: int):
    """Contiguous block of envs owned by `rank` -> (start, count)."""
    base, rem = divmod(total_envs, world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def global_mean_std(x: torch.Tensor, group=None):
    """Mean / unbiased std (``torch.std``) of `x` over all ranks, via ONE all-gather of 3 values per rank.  The moments are accumulated in
    float64, as ``k_adv_normalize`` (csrc/go2sim_policy.hip) does, so that both advantage normalisations of the package agree."""
    xd = x.reshape(-1).to(torch.float64)
    m = torch.stack([xd.sum(), (xd * xd).sum(), torch.tensor(float(xd.numel()), device=x.device, dtype=torch.float64)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        out = torch.empty(3 * world, device=m.device, dtype=m.dtype)
        dist.all_gather_into_tensor(out, m, group=group)
        m = out.view(world, 3).sum(0)
    n = m[2]
    mean = m[0] / n
    var = torch.clamp((m[1] - n * mean * mean) / torch.clamp(n - 1.0, min=1.0), min=0.0)
    return mean.to(torch.float32), torch.sqrt(var).to(torch.float32)


def normalize_advantages(adv: torch.Tensor, group=None, eps: float = 1e-8) -> torch.Tensor:
    """rsl_rl 2.2.4 PPO: ``(adv - adv.mean()) / (adv.std() + 1e-8)`` with the statistics taken over all ranks -- the formula of
    ``go2sim_rollout_normalize`` (unbiased std, eps added to the std)."""
    mean, std = global_mean_std(adv, group)
    return (adv - mean) / (std + eps)


def allgather_moments(moments3: torch.Tensor, group=None) -> torch.Tensor:
    """Global [sum, sum of squares, count] from the per-rank moments of go2sim_rollout_compute_returns (float64[3]): one all-gather of
    3 values per rank (RCCL over xGMI with backend "nccl", gloo in the CPU tests), summed in rank order so that every rank gets the same bits."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        out = torch.empty(3 * world, device=moments3.device, dtype=moments3.dtype)
        dist.all_gather_into_tensor(out, moments3.contiguous(), group=group)
        out = out.view(world, 3)
        g = out[0].clone()
        for r in range(1, world):
            g = g + out[r]
        return g
    return moments3


def sync_env_globals(sim, group=None, stream=None, initial=False):
    """One batch sharded over ranks, one curriculum / one set of global DR scalars.  `sim` is a capi.Go2Sim configured with
    ``shared_globals=True`` (GO2SIM_IC_SHARED_GLOBALS).  Call it on every rank at the same cadence (once per rollout):
      1. this shard's increments of [episodes, time-outs, tracking sum, tracking n, throttle resets] (go2_env_walk.py:460-463, 712-715, 744)
      2. all-reduce(sum) of those 5 float64 (RCCL over xGMI with backend "nccl"; gloo in the CPU tests)
      3. every rank runs _maybe_update_curriculum_on_reset / sample_level / the global DR draws on the SAME summed counters
      4. broadcast of rank 0's draws (friction, mass shift, COM shift, leg-mass shifts and the sampled DR level t_sample: 10 float64), applied
         to all envs of every shard.
    Returns (summed counters, the 10 scalars now in force; device tensors under nccl).  Without a process group it degenerates to the single-shard case.
    Call it ONCE with ``initial=True`` between env_configure and the constructor's env_reset as well (Go2Env and bench.make_sim do): that first apply draws
    t_sample at level_init and the global scalars on rank 0 -- with the batch's env count as the friction-throttle increment, as the constructor's reset_idx
    counts it -- and every shard starts where the single-process env starts."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    src = (dist.get_global_rank(group, 0) if group is not None else 0) if multi else 0
    if multi and dist.get_backend(group) == "nccl":
        # RCCL: counters and scalars stay on the device from the shard's Glob to the collective and back (three single-thread kernels of the library
        # around two collectives, all ordered on torch's current stream); nothing is copied to the host and nothing synchronises
        dev = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        counters = torch.empty(5, dtype=torch.float64, device=dev)
        dr = torch.empty(10, dtype=torch.float64, device=dev)
        sim.env_sync_counters_dev(counters, stream)
        if initial:
            counters[4] += float(sim.n_envs)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
        sim.env_sync_apply_dev(counters, dr, stream)
        dist.broadcast(dr, src=src, group=group)
        sim.env_set_global_dr_dev(dr, stream)
        return counters, dr
    counters = torch.from_numpy(sim.env_sync_counters(stream))       # gloo (CPU tests, rehearsal) and the single-shard case: host arrays
    if initial:
        counters[4] += float(sim.n_envs)
    if multi:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    summed = counters.numpy()
    dr = torch.from_numpy(sim.env_sync_apply(summed, stream))
    if multi:
        dist.broadcast(dr, src=src, group=group)
    dr = dr.numpy()
    sim.env_set_global_dr(dr, stream)
    return summed, dr
