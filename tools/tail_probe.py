"""Launch time against work / slots for the families with a cost array (80 ns units per world, written by every step launch): how much of a step launch is tail.
   python tools/tail_probe.py   (on the GPU box)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
for w, slots_per_cu in (("adroit", 7), ("adroit_pen", 8), ("adroit_relocate", 5), ("adroit_door", 6), ("kitchen", 6)):
    if w == "adroit_pen":
        os.environ["GRX_ADROIT_BALANCE"] = "1"
    W = bench.WORKLOADS[w]; n = W["worlds"]
    env = bench.make_env(w, n, "cuda:0", 0); env.reset(seed=0)
    bench._set_elapsed(env, np.arange(n) % (env.max_episode_steps or W["horizon"]))
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    A = env.single_action_space.shape[0]
    for _ in range(40 if w != "kitchen" else 60): env.step(torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1)
    env.kernel_events = []
    res = []
    for _ in range(5):
        env.step(torch.rand(n, A, device="cuda:0", generator=g) * 2 - 1); torch.cuda.synchronize()
        c = env.cost.cpu().numpy().astype(np.float64) * 0.08   # us
        a, b = env.kernel_events[-1]; res.append((a.elapsed_time(b), c.sum() / (256 * slots_per_cu) / 1e3, np.percentile(c, 50) / 1e3, c.max() / 1e3))
    r = np.mean(res, axis=0)
    print(f"{w}: kernel {r[0]:.2f} ms, sum(world time) / {256 * slots_per_cu} slots = {r[1]:.2f} ms ({100 * (r[0] / r[1] - 1):.0f} % above), world p50 {r[2]:.2f} ms, max {r[3]:.2f} ms")
    os.environ.pop("GRX_ADROIT_BALANCE", None)
    del env
