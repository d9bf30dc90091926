"""Host time of every native call of an env.step() (a proxy around the ctypes library times each entry point; the loop never synchronises).
    python tools/host_calls_probe.py [env_id] [n_worlds] [steps]"""
import os, sys, time, collections
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd import make_vec
env_id = sys.argv[1] if len(sys.argv) > 1 else "FetchPickAndPlace-v4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
env = make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
acc = collections.defaultdict(lambda: [0.0, 0])
class Proxy:
    def __init__(self, L): self.__dict__["_L"] = L
    def __getattr__(self, name):
        f = getattr(self.__dict__["_L"], name)
        if not callable(f): return f
        def timed(*a):
            t = time.perf_counter(); r = f(*a); d = time.perf_counter() - t
            acc[name][0] += d; acc[name][1] += 1
            return r
        return timed
env.reset(seed=0); env._elapsed[:] = np.arange(n) % (env.max_episode_steps or 50)
na = env.single_action_space.shape[0]
a = torch.rand(n, na, device="cuda:0") * 2 - 1
for _ in range(60): env.step(a)
torch.cuda.synchronize()
real = env._L
env._L = Proxy(real)
if getattr(env, "lane", None) is not None and hasattr(env.lane, "_L"): env.lane._L = Proxy(env.lane._L)
t0 = time.perf_counter()
for _ in range(steps): env.step(a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{env_id} {n} worlds: host {1e3 * (t1 - t0) / steps:.3f} ms per step (enqueue only), {1e3 * (t2 - t0) / steps:.3f} ms with the drain")
for k, (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:36s} {c / steps:5.2f} calls/step  {1e6 * s / max(c, 1):8.1f} us each  {1e3 * s / steps:7.3f} ms/step")
