"""Per-kernel time of the reset-heavy action set B (0.5 N(0,1) actions: falls, resets, worst-case contacts), walk, 4096 envs: run under
rocprofv3 --kernel-trace --stats:   rocprofv3 --kernel-trace --stats -d out -o p -- python3 tools/set_b_kernels.py [workload] [A|B|C]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from go2_sim2real_locomotion_rl_amd.capi import load_hip_lib
B, W, N = 4096, 200, 300
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "walk"; KIND = sys.argv[2] if len(sys.argv) > 2 else "B"
dev = torch.device("cuda", 0)
sim = bench.make_sim(load_hip_lib(), B, 0, 1, WORKLOAD)
act = bench.make_actions(W + N, B, dev, workload=WORKLOAD, kind=KIND)
buf = bench.Buffers(B, WORKLOAD, dev)
if len(sys.argv) > 3 and sys.argv[3] == "stagger":        # episode lengths spread over the whole episode: a few envs time out on EVERY step (what training looks like)
    g = torch.Generator(device="cpu").manual_seed(7)
    sim.env_set_episode_length(torch.randint(0, 1000, (B,), generator=g, dtype=torch.int32).to(dev))
for s in range(W + N):
    sim.env_step(act[s], buf.obs, buf.priv, buf.rew, buf.rst, buf.to)
torch.cuda.synchronize()
print("resets in the last step", int(buf.rst.sum()))
