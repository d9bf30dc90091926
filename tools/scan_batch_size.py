"""env-steps/s of the Fetch bench configuration (staggered episodes, same-step autoreset, HER excluded) against worlds per GPU, one process:
    python tools/scan_batch_size.py > profiles/scan_batch_size_r02.txt
2 048 worlds = one world per wave slot (8 worlds per CU x 256 CUs)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gymnasium_robotics_amd as grx  # noqa: E402

for n in (1024, 2048, 4096, 6144, 8192, 12288, 16384, 32768, 65536):
    env = grx.make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
    env.reset(seed=0)
    env._elapsed[:] = np.arange(n) % 50
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    for _ in range(25):
        env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    env.kernel_events = []
    torch.cuda.synchronize(); t = time.perf_counter(); K = 40
    for _ in range(K):
        env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    k = float(np.mean([a.elapsed_time(b) for a, b in env.kernel_events]))
    print(f"N={n:6d}  {dt * 1e3:8.3f} ms per vector step (step kernel {k:8.3f} ms)  {n / dt:12,.0f} env-steps/s  = {n / 2048:5.1f} worlds per wave slot", flush=True)
    env.close()
    del env
    torch.cuda.empty_cache()
