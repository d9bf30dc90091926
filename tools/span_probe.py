"""Start / end time of every world of one FetchPickAndPlace step launch (profiling build: sh tools/build_prof.sh fetch): who ends the launch, and when did it start.
    python tools/span_probe.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd import _native, make_vec
_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", os.environ.get("GRX_SPAN_LIB", "libgrx_span.so"))      # product build + the two time stamps (-DGRX_WORLD_SPAN): same occupancy as the shipped kernel
n = 4096
env = make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
for k in range(40):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
order = env.order.cpu().numpy().copy()      # the order the NEXT launch will use; the last launch used the one before: take one more step and keep both
env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
L = _native.lib()
buf = (ctypes.c_longlong * (2 * n))()
L.grx_profile_world_spans.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.grx_profile_world_spans(buf, n)
S = np.frombuffer(buf, dtype=np.int64).reshape(n, 2).astype(np.float64) * 0.01      # us (100 MHz)
t0 = S[:, 0].min()
start, end = S[:, 0] - t0, S[:, 1] - t0
dur = end - start
pos = np.empty(n, int); pos[order] = np.arange(n)          # dispatch position of every world
print(f"launch span {end.max():.0f} us; worlds started after 100 us: {(start > 100).sum()}; median duration {np.median(dur):.0f} us")
last = np.argsort(-end)[:24]
print("the worlds that end the launch: end, start, duration, dispatch position (block), slice")
for w in last:
    print(f"  world {w:5d}: end {end[w]:7.0f} start {start[w]:7.0f} dur {dur[w]:6.0f} block {pos[w]:5d} (position {pos[w] >> 3} of slice {pos[w] & 7})")
second = start > 100
print("second-round starts (us): p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(start[second], [1, 10, 50, 90, 99, 100])))
for s in range(8):
    m = (pos & 7) == s
    print(f"  slice {s}: first-round worlds {(m & ~second).sum()}, last start {start[m].max():.0f}, last end {end[m].max():.0f}, stragglers (> 1.8 ms) {(m & (dur > 1800)).sum()}")
