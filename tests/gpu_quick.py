"""Quick GPU bring-up script (not a test): python tests/gpu_quick.py"""
import sys, time, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
import __graft_entry__ as g
g.smoke()
for n in (256, 4096):
    env = FetchVecEnv("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
    env.reset(seed=0)
    a = torch.rand(n, 4, device="cuda:0") * 2 - 1
    for _ in range(3): env.step(a)
    torch.cuda.synchronize(); t = time.time()
    K = 20
    for _ in range(K): env.step(a)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    print(f"N={n}: {dt*1e3:.2f} ms/step, {n/dt:.0f} env-steps/s, status max {int(env.status.max())}")
