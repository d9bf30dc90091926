"""Is a workload's step HOST-bound?  Wall time the host spends inside env.step() (enqueue only, never synchronised) against the device time of the step's launches.
    python tools/host_time_probe.py hand_touch [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "hand_touch"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = bench.WORKLOADS[w]
n = W["worlds"]
env = bench.make_env(w, n, "cuda:0", 0)
env.reset(seed=0)
bench._set_elapsed(env, np.arange(n) % (env.max_episode_steps or W["horizon"]))
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
na = env.single_action_space.shape[0]
for _ in range(30):
    env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1
    h0 = time.perf_counter()
    env.step(a)
    host.append(time.perf_counter() - h0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host = np.array(host) * 1e3
print(f"{w}: {steps} steps, host time inside step() mean {host.mean():.2f} ms (p50 {np.median(host):.2f}, p90 {np.quantile(host, .9):.2f}, max {host.max():.2f}); enqueue loop {1e3 * (t1 - t0) / steps:.2f} ms per step, "
      f"with the final drain {1e3 * (t2 - t0) / steps:.2f} ms per step (queue behind the host at the end: {1e3 * (t2 - t1):.1f} ms)")
