"""The PCIe-inclusive rate of the reference-shaped call: FetchVecEnv(output="numpy") takes host actions and returns host float64 arrays every step (actions H2D, packed
rows D2H, one synchronisation per step), against output="torch" (everything stays in HBM: what bench.py times).  Run on the GPU box:  python tools/pcie_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

import gymnasium_robotics_amd as grx  # noqa: E402

n, steps = 4096, 100
for output in ("torch", "numpy"):
    env = grx.make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output=output, autoreset_mode="same_step")
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (steps + 10, n, 4)).astype(np.float32)
    dev = torch.from_numpy(acts).cuda() if output == "torch" else None
    for t in range(10):
        env.step(dev[t] if output == "torch" else acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(10, steps + 10):
        env.step(dev[t] if output == "torch" else acts[t])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"FetchPickAndPlace-v4, {n} worlds, output={output!r}: {n * steps / dt:,.0f} env-steps/s ({1e3 * dt / steps:.2f} ms per step)")
    env.close()
