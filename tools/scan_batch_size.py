import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import gymnasium_robotics_amd as grx
for n in (2048, 2304, 4096, 4608, 6912, 8192, 9216):
    env = grx.make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    for _ in range(5): env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize(); t = time.time(); K = 30
    for _ in range(K): env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    print(f"N={n}  {dt*1e3:.3f} ms/step  {n/dt:,.0f} env-steps/s")
    env.close()
