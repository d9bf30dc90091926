"""Occupancy of the overflow lane (core.OverflowLane) along a random-action rollout: worlds in the lane, worlds that entered through a serialised re-run (synchronises every
step: diagnostics only).   python tools/lane_probe.py [env id] [worlds] [steps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.abspath(__import__("os").path.join(__import__("os").path.dirname(__file__), "..")))
import gymnasium_robotics_amd as grx
env_id = sys.argv[1] if len(sys.argv) > 1 else "FetchPickAndPlace-v4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0)
env._elapsed[:] = np.arange(n) % (env.max_episode_steps or 50)
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
A = env.single_action_space.shape[0]
lane, entered, ms = [], [], []
for t in range(steps):
    a = torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(a)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    L = env.lane
    lane.append(L.count()); entered.append(len(L.entered_last_step()))
print(env_id, n, "worlds: lane size per step", lane)
print("entered by re-run per step", entered)
print("ms per step (synchronised)", ["%.1f" % x for x in ms])
print("sticky", env.status_counts())
