"""A/B of the broad-phase skin list of the kitchen scene in ONE process on one GPU (box-to-box variance is +-4 %): env-steps/s of FrankaKitchen-v1 for several
skin radii (0 = the flat sweep of all 3 736 candidate pairs in every substep).    python tools/skin_probe.py [worlds] > profiles/skin_probe_r02.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd import make_vec  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
gen = torch.Generator(device="cuda:0")
for radius in (0.0, 0.05, 0.1, 0.15, 0.2, 0.0, 0.1):
    env = make_vec("FrankaKitchen-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", skin_radius=radius)
    env.reset(seed=0)
    env._elapsed[:] = np.arange(n) % 280
    gen.manual_seed(5)
    for _ in range(3):
        env.step(torch.rand(n, 9, device="cuda:0", generator=gen) * 2 - 1)
    env.kernel_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        env.step(torch.rand(n, 9, device="cuda:0", generator=gen) * 2 - 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12
    k = float(np.mean([a.elapsed_time(b) for a, b in env.kernel_events]))
    extra = ""
    if radius:
        h = env._skin[:, 0].cpu().numpy()
        extra = f"  list length p50 {np.median(h):.0f} p99 {np.quantile(h, 0.99):.0f} max {h.max()}"
    print(f"skin radius {radius:4.2f} m: {dt * 1e3:7.2f} ms per env.step of {n} worlds, step kernel {k:7.2f} ms, {n / dt:,.0f} env-steps/s{extra}", flush=True)
    del env
    torch.cuda.empty_cache()
