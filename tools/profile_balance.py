"""Load balance across worlds of one Fetch step launch (profiling build, run on the GPU box):
    bash tools/build_prof.sh; python tools/profile_balance.py [n_worlds]
Per-world start / end timestamps (wall_clock64, 100 MHz) -> distribution of world durations, launch makespan, and what an ideal
(perfectly balanced) schedule of the same durations on the same number of wave slots would take."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gymnasium_robotics_amd as grx  # noqa: E402
from gymnasium_robotics_amd import _native  # noqa: E402

_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", "libgrx_hip_prof.so")   # tools/build_prof.sh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = grx.make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
env.reset(seed=0)
L = _native.lib()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
for k in range(30):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * n))()
L.grx_profile_world_spans.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.grx_profile_world_spans(buf, n)
sp = np.array(list(buf), dtype=np.int64).reshape(n, 2)
t0 = sp[:, 0].min()
start, end = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0      # microseconds
dur = end - start
print(f"{n} worlds: makespan {end.max():.0f} us; world duration min {dur.min():.0f} p10 {np.quantile(dur, .1):.0f} p50 {np.median(dur):.0f} p90 {np.quantile(dur, .9):.0f} "
      f"p99 {np.quantile(dur, .99):.0f} max {dur.max():.0f} us; sum {dur.sum()/1e3:.1f} ms")
slots = 2304
print(f"sum / {slots} slots = {dur.sum() / slots:.0f} us (perfect balance at the same per-world speed)")
late = np.sort(start)
print("start times: first wave of worlds <", f"{late[min(slots, n) - 1]:.0f} us;", "last start", f"{late[-1]:.0f} us")
for f in (0.1, 0.3, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    t = f * end.max()
    print(f"  t = {t:6.0f} us ({int(100*f):2d} % of the launch): {int(((start <= t) & (end > t)).sum()):5d} worlds running, {int((start > t).sum()):5d} not started")
first = start < 5.0
print(f"worlds started at t=0: {int(first.sum())}: duration p50 {np.median(dur[first]):.0f} max {dur[first].max():.0f}; later starters: {int((~first).sum())}: duration p50 "
      f"{np.median(dur[~first]) if (~first).any() else 0:.0f} max {dur[~first].max() if (~first).any() else 0:.0f}")
# who is slow?  correlate with contact / row counts of the final substep
st = env.status.cpu().numpy()
q = env.qpos.cpu().numpy()
z = q[:, -5]
print("corr(duration, object height)", np.corrcoef(dur, z)[0, 1], " slowest 1% mean z", z[np.argsort(dur)[-n // 100:]].mean(), " all mean z", z.mean())
np.save(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "world_spans.npy"), sp)
