"""Do out-of-phase sub-batches fill the tail of a Fetch step launch?  K FetchPickAndPlace environments of 4096 / K worlds each, every one on its own stream, stepped round-robin by
ONE host thread (each env.step() is enqueue-only), against one environment of 4096 worlds.  Same total worlds, same work per world-step (no HER here: env.step only).
    python tools/async_probe.py [total_worlds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd import make_vec
total = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
for K in (1, 2, 4, 1, 2, 4):
    n = total // K
    envs, streams, gens = [], [], []
    for k in range(K):
        e = make_vec("FetchPickAndPlace-v4", num_envs=n, device=dev, output="torch", autoreset_mode="same_step")
        e.reset(seed=100 * k)
        e._elapsed[:] = np.arange(n) % 50
        g = torch.Generator(device=dev); g.manual_seed(k)
        envs.append(e); streams.append(torch.cuda.Stream(device=dev)); gens.append(g)
    torch.cuda.synchronize()
    def sweep():
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                envs[k].step(torch.rand(n, 4, device=dev, generator=gens[k]) * 2 - 1)
    for _ in range(60):
        sweep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = 200
    for _ in range(S):
        sweep()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flagged = sum(int(((e.status >> 16) & 6).ne(0).sum()) for e in envs)
    print(f"{K} x {n} worlds on {K} stream(s): {total * S / dt:,.0f} env-steps/s, {dt / S * 1e3:.3f} ms per sweep of {total} world-steps; flagged worlds {flagged}")
    del envs
