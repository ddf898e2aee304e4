import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from go2_sim2real_locomotion_rl_amd.capi import load_hip_lib
side = torch.cuda.Stream()
for B, use_side in ((256, False), (256, True), (4096, False), (4096, True)):
  with torch.cuda.stream(side if use_side else torch.cuda.default_stream()):
      sim = bench.make_sim(load_hip_lib(), B, 0, 1, "walk")
      dev = torch.device("cuda", 0)
      act = bench.make_actions(700, B, dev)
      obs = torch.zeros(B, 49, device=dev); priv = torch.zeros(B, 104, device=dev); rew = torch.zeros(B, device=dev)
      rst = torch.zeros(B, dtype=torch.uint8, device=dev); to = torch.zeros(B, device=dev)
      s0 = torch.cuda.current_stream().cuda_stream
      for s in range(100): sim.env_step(act[s], obs, priv, rew, rst, to, s0)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for s in range(100, 700): sim.env_step(act[s], obs, priv, rew, rst, to, s0)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      print(f"B={B} side_stream={use_side}: CPU enqueue {1e3*(t1-t0)/600:.4f} ms/step, wall {1e3*(t2-t0)/600:.4f} ms/step", flush=True)

# where the host time goes: ctypes call overhead vs the launches inside the library
import ctypes
lib = load_hip_lib()
k, isint = ctypes.c_int(), ctypes.c_int()
t0 = time.perf_counter()
for _ in range(20000): lib.fn("field_size")(ctypes.c_int(0), ctypes.byref(k), ctypes.byref(isint))
t1 = time.perf_counter()
print(f"trivial ctypes call: {1e6*(t1-t0)/20000:.2f} us")
x = torch.zeros(4096, 16, device="cuda")
t0 = time.perf_counter()
for s in range(20000): y = act[s % 600]
t1 = time.perf_counter()
print(f"act[s] slicing: {1e6*(t1-t0)/20000:.2f} us")
t0 = time.perf_counter()
for s in range(20000): p = (ctypes.c_void_p(obs.data_ptr()), ctypes.c_void_p(priv.data_ptr()), ctypes.c_void_p(rew.data_ptr()), ctypes.c_void_p(rst.data_ptr()), ctypes.c_void_p(to.data_ptr()), ctypes.c_void_p(x.data_ptr()))
t1 = time.perf_counter()
print(f"6 x data_ptr -> c_void_p: {1e6*(t1-t0)/20000:.2f} us")
