"""Per-world durations of one FetchPickAndPlace step launch (the cost words the step kernel writes for the cost-ordered dispatch) against the launch's
makespan and the perfectly balanced bound; with GRX_HIP_LIB=<-DGRX_PROFILE_ITER build> the cost words are (Newton iterations | contacts << 16) instead.
    python tools/cost_probe.py [out.npy]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd import make_vec
n = 4096
env = make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
env.kernel_events = []
for k in range(40):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
raw = env.cost.cpu().numpy()
if len(sys.argv) > 1:
    np.save(sys.argv[1], np.stack([raw, env._elapsed.astype(np.int32)]))
if "iter" not in os.environ.get("GRX_HIP_LIB", ""):
    c = raw.astype(np.float64) * 0.08   # us
    k = np.mean([a.elapsed_time(b) for a, b in env.kernel_events[10:]])
    top = np.sort(c)[::-1]
    print("slowest worlds (us):", " ".join(f"{x:.0f}" for x in top[:12]), "| worlds above 1.5 / 2.0 ms:", int((c > 1500).sum()), int((c > 2000).sum()))
    print(f"world durations (us): min {c.min():.0f} p10 {np.quantile(c,.1):.0f} p50 {np.median(c):.0f} p90 {np.quantile(c,.9):.0f} p99 {np.quantile(c,.99):.0f} max {c.max():.0f}; sum/2048 slots = {c.sum()/2048:.0f} us; kernel {k*1e3:.0f} us")
