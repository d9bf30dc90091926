"""Where the overflow lane's per-step cost goes on FetchPickAndPlace (4096 worlds): host time of the calls and GPU time (HIP events) of the two large-table launches."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import gymnasium_robotics_amd as grx
from gymnasium_robotics_amd import core
n = 4096
env = grx.make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0)
env._elapsed[:] = np.arange(n) % 50
L = env.lane
host, gpu = {"large": [], "fast": []}, {"large": [], "fast": []}
orig_step = L.step
def timed(kind, fn):
    def wrap(b):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t0 = time.perf_counter(); fn(b); host[kind].append(time.perf_counter() - t0); e1.record(); gpu[kind].append((e0, e1))
    return wrap
def step(mask, lf, ll, fast_bufs):
    return orig_step(mask, timed("fast", lf), timed("large", ll), fast_bufs)
L.step = step
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
tt, marks = [], []
order = []
_orig_timed = timed
def timed(kind, fn):      # remember the order of the launches inside a step
    w = _orig_timed(kind, fn)
    def wrap(b):
        w(b); order.append((kind, gpu[kind][-1]))
    return wrap
for t in range(60):
    a = torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1
    order.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    t0 = time.perf_counter(); env.step(a); tt.append(time.perf_counter() - t0)
    s1.record()
    marks.append((s0, s1, list(order)))
torch.cuda.synchronize()
rows = []
for s0, s1, od in marks[10:]:
    if len(od) != 3:
        continue
    (k0, (a0, b0)), (k1, (a1, b1)), (k2, (a2, b2)) = od
    rows.append([s0.elapsed_time(a0), a0.elapsed_time(b0), b0.elapsed_time(a1), a1.elapsed_time(b1), b1.elapsed_time(a2), a2.elapsed_time(b2), b2.elapsed_time(s1), s0.elapsed_time(s1)])
rows = np.array(rows) * 1e3
print("order of launches:", [k for k, _ in marks[-1][2]])
fs = np.array([[s0.elapsed_time(od[1][1][0]), od[1][1][1].elapsed_time(s1), s0.elapsed_time(od[0][1][0]), s0.elapsed_time(od[0][1][1])] for s0, s1, od in marks[10:] if len(od) == 3]) * 1e3
print("median us: step begin -> fast kernel start %.0f | fast kernel end -> step end %.0f | step begin -> first large launch start %.0f, end %.0f" % tuple(np.median(fs, axis=0)))
print("median us: begin->L0 %.0f | L0 %.0f | L0->fast %.0f | fast %.0f | fast->L1 %.0f | L1 %.0f | L1->end %.0f | whole step %.0f" % tuple(np.median(rows, axis=0)))
gaps = [marks[i][1].elapsed_time(marks[i + 1][0]) * 1e3 for i in range(10, len(marks) - 1)]
print("median us between the end of a step and the start of the next (rand + host): %.0f" % np.median(gaps))
for k in host:
    print(k, "host call us: median %.1f max %.1f | gpu event us: median %.1f max %.1f  (n=%d)" % (np.median(host[k]) * 1e6, np.max(host[k]) * 1e6, np.median([a.elapsed_time(b) * 1e3 for a, b in gpu[k]]), np.max([a.elapsed_time(b) * 1e3 for a, b in gpu[k]]), len(host[k])))
print("host time of env.step us: median %.1f" % (np.median(tt[10:]) * 1e6))
