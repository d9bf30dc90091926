"""Per-step HIP-event timeline of the kitchen's launch group (fast kernel, standing lane, entry launch), at the bench's stationary regime.
    python tools/lane_dbg_kitchen.py [worlds] [preroll steps]        (env: GRX_LANE_POLL, GRX_LANE_TTL, GRX_LANE_MARGIN, GRX_LANE_FIRST as in core.OverflowLane)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import gymnasium_robotics_amd as grx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 60
env = grx.make_vec("FrankaKitchen-v1", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0)
env._elapsed[:] = np.arange(n) % (env.max_episode_steps or 280)
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
A = env.single_action_space.shape[0]
for k in range(pre):
    env.step(torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
env.kernel_events, env.step_events = [], []
env.lane.trace = []
K = 12
info = []
for k in range(K):
    env.step(torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize()
    ent = env.lane.entered_last_step()
    info.append((env.lane.count(), env.lane.cap_cur, len(ent)))
for k in range(K):
    a, b = env.kernel_events[k]; l0, l1 = env.step_events[k]; ev0, t0, t1, x0, x1 = env.lane.trace[k]
    print(f"step {k}: lane {info[k][0]:4d} cap {info[k][1]:4d} entrants {info[k][2]:3d} | fast kernel {ev0.elapsed_time(a):.2f} -> {ev0.elapsed_time(b):.2f} | lane kernel {ev0.elapsed_time(t0):.2f} -> {ev0.elapsed_time(t1):.2f} | "
          f"entry launch {ev0.elapsed_time(x0):.2f} -> {ev0.elapsed_time(x1):.2f} | group {l0.elapsed_time(l1):.2f} ms")
