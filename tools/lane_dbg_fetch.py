import os, sys, time
os.environ.setdefault("GRX_FETCH_HANDOFF", "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
n = int(os.environ.get("N", 4096))
env = FetchVecEnv("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
for k in range(60):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
env.kernel_events, env.step_events = [], []
env.lane.trace = []
SYNC = os.environ.get('SYNC', '1') == '1'
for k in range(20):
    t0 = time.perf_counter()
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    if SYNC:
        torch.cuda.synchronize()
        ent = env.lane.entered_last_step()
        print(f"step {k}: lane {env.lane.count():4d} cap {env.lane.cap_cur:4d} entrants {len(ent):3d} (large {int((ent >> 30 & 1).sum())})")
torch.cuda.synchronize()
for k in range(20):
    a, b = env.kernel_events[k]; l0, l1 = env.step_events[k]; ev0, t0, t1, x0, x1 = env.lane.trace[k]
    print(f"step {k}: relative to the step's start: fast kernel {ev0.elapsed_time(a):.2f} -> {ev0.elapsed_time(b):.2f} | lane kernel {ev0.elapsed_time(t0):.2f} -> {ev0.elapsed_time(t1):.2f} | entry launch {ev0.elapsed_time(x0):.2f} -> {ev0.elapsed_time(x1):.2f} | group {l0.elapsed_time(l1):.2f} ms")
