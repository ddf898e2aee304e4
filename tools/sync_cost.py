"""Cost of one rollout end of the sharded bench loop on one GPU (no collectives): RolloutStorage.compute_returns + distributed.sync_env_globals."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from go2_sim2real_locomotion_rl_amd.capi import load_hip_lib
from go2_sim2real_locomotion_rl_amd import RolloutStorage, distributed as d
B = 4096
dev = torch.device("cuda", 0)
sim = bench.make_sim(load_hip_lib(), B, 0, 1, "walk", shared_globals=True)
act = bench.make_actions(200, B, dev)
buf = bench.Buffers(B, "walk", dev)
stream = torch.cuda.current_stream().cuda_stream
storage = RolloutStorage(24, B, device=dev)
z = torch.zeros(B, device=dev)
for s in range(48):
    sim.env_step(act[s], buf.obs, buf.priv, storage.rewards[s % 24], storage.dones[s % 24], buf.to, stream)
torch.cuda.synchronize()
def run(n, with_sync):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(n):
        t = s % 24
        sim.env_step(act[48 + s], buf.obs, buf.priv, storage.rewards[t], storage.dones[t], buf.to, stream)
        if with_sync and t == 23:
            storage.compute_returns(z, 0.99, 0.95)
            d.sync_env_globals(sim, None, stream)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for rep in range(3):
    a = run(96, False); b = run(96, True)
    print(f"96 steps: plain {a*1e3:.2f} ms, with 4 rollout ends {b*1e3:.2f} ms -> {(b-a)/4*1e6:.0f} us per rollout end (compute_returns + sync_env_globals, no collectives)")
t0 = time.perf_counter(); storage.compute_returns(z, 0.99, 0.95); torch.cuda.synchronize(); print("compute_returns alone", (time.perf_counter()-t0)*1e6, "us")
t0 = time.perf_counter(); d.sync_env_globals(sim, None, stream); torch.cuda.synchronize(); print("sync_env_globals alone", (time.perf_counter()-t0)*1e6, "us")
