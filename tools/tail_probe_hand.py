"""tools/tail_probe.py for the Shadow-hand families (their cost arrays exist only with balance=True): launch time against work per wave slot.   python tools/tail_probe_hand.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gymnasium_robotics_amd as grx
for env_id, n, horizon in (("HandReach-v3", 16384, 50), ("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", 16384, 100), ("HandManipulateEggRotate-v1", 16384, 100)):
    env = grx.make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step", balance=True); env.reset(seed=0)
    env._elapsed[:] = np.arange(n) % horizon
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    for _ in range(30): env.step(torch.rand(n, 20, device="cuda:0", generator=g) * 2 - 1)
    env.kernel_events = []
    res = []
    for _ in range(5):
        env.step(torch.rand(n, 20, device="cuda:0", generator=g) * 2 - 1); torch.cuda.synchronize()
        c = env.cost.cpu().numpy().astype(np.float64) * 0.08
        a, b = env.kernel_events[-1]; res.append((a.elapsed_time(b), c.sum() / 2048 / 1e3, np.percentile(c, 50) / 1e3, c.max() / 1e3))
    r = np.mean(res, axis=0)
    print(f"{env_id}: kernel {r[0]:.2f} ms, sum(world time) / 2048 slots = {r[1]:.2f} ms ({100 * (r[0] / r[1] - 1):.0f} % above), world p50 {r[2]:.2f} ms, max {r[3]:.2f} ms")
    del env
