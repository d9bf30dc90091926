"""Where the HOST time of a Fetch env.step() goes (cProfile over a loop that never synchronises; small batch so that the host, not the device, is the bottleneck).
    python tools/host_profile_fetch.py [n_worlds] [steps]"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gymnasium_robotics_amd import make_vec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = make_vec(os.environ.get("GRX_PROBE_ID", "FetchPickAndPlace-v4"), num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
na = env.single_action_space.shape[0]
a = torch.rand(n, na, device="cuda:0") * 2 - 1
for _ in range(100):
    env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{n} worlds: enqueue loop {1e3 * (t1 - t0) / steps:.3f} ms per step, with the drain {1e3 * (t2 - t0) / steps:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps):
    env.step(a)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
