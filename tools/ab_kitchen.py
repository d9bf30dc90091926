import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from gymnasium_robotics_amd import make_vec
n = 16384
gen = torch.Generator(device="cuda:0")
env = make_vec("FrankaKitchen-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env.set_elapsed(np.arange(n) % 280)
gen.manual_seed(5)
for _ in range(3): env.step(torch.rand(n, 9, device="cuda:0", generator=gen) * 2 - 1)
env.kernel_events = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12): env.step(torch.rand(n, 9, device="cuda:0", generator=gen) * 2 - 1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
print(os.environ.get("GRX_HIP_LIB", "default"), f"{dt*1e3:.2f} ms/step, kernel {np.mean([a.elapsed_time(b) for a, b in env.kernel_events]):.2f} ms, {n/dt:,.0f} env-steps/s", env.status_counts())
