#!/usr/bin/env python3
"""Would two independent half-batch chains on two streams beat one full-batch chain?  (feasibility probe for a split step graph)
usage: two_chain_probe.py [warmup] [steps]   -- compares 1 x 4096 envs on one stream with 2 x 2048 (and 4 x 1024) envs on separate streams, same workload."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from go2_sim2real_locomotion_rl_amd.capi import load_hip_lib

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
lib = load_hip_lib()
for parts in (1, 2, 4):
    n = 4096 // parts
    sims = [bench.make_sim(lib, n, 0, 1 + p, "walk") for p in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    acts = [bench.make_actions(W + K, n, dev) for _ in range(parts)]
    bufs = [bench.Buffers(n, "walk", dev) for _ in range(parts)]
    torch.cuda.synchronize()
    def step(s):
        for p in range(parts):
            b = bufs[p]
            sims[p].env_step(acts[p][s], b.obs, b.priv, b.rew, b.rst, b.to, streams[p].cuda_stream)
    for s in range(W):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        step(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{parts} chain(s) x {n} envs: {4096 * K / dt / 1e6:.2f} M env-steps/s, {dt / K * 1e3:.4f} ms per step of all 4096 envs (steps {W}..{W + K})", flush=True)
    del sims
