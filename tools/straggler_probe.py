"""Which stages make a slow world slow: per-world stage cycles of one step launch (profiling build: sh tools/build_prof.sh fetch|kitchen|...).
    python tools/straggler_probe.py [env id] [worlds] [pre-roll steps] > profiles/stragglers_r02_fetch.txt"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd import _native, make_vec
_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", os.environ.get("GRX_PROF_LIB", "libgrx_hip_prof.so"))
NP = 56
env_id = sys.argv[1] if len(sys.argv) > 1 else "FetchPickAndPlace-v4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 40
env = make_vec(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % (env.max_episode_steps or 50)
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
NA = env.single_action_space.shape[0]
for k in range(pre):
    env.step(torch.rand(n, NA, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
L = _native.lib()
buf = (ctypes.c_int * (NP * n))()
L.grx_profile_world_stages.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.grx_profile_world_stages(buf, n)
P = np.array(list(buf), dtype=np.float64).reshape(n, NP)
tot = P[:, :16 + 24].sum(axis=1)      # the time slots only: every cycle is booked on exactly one of them; the slots behind are counters
names = ["kinematics", "inertia", "collision", "constraint", "velocity", "M solve", "newton eval", "newton grad", "newton hessian", "newton factor", "newton linesearch", "newton final", "euler", "other"]
SUB = {0: "constraint: count", 1: "constraint: scan", 2: "constraint: equality rows", 3: "constraint: friction/limit rows", 4: "constraint: row params", 5: "constraint: contact J",
       6: "velocity: rne a", 7: "velocity: rne b", 8: "velocity: passive/actuation", 9: "kinematics: bodies", 10: "kinematics: sites/frames", 11: "collision: box-box queue",
       12: "collision: geom frames", 13: "collision: narrow phase rounds", 14: "inertia: cinert/cdof", 15: "inertia: crb/M", 16: "collision: hull pairs", 19: "collision: candidate sweep", 20: "collision: regroup", 21: "hull pairs: set-up", 22: "hull pairs: cached direction check", 23: "hull pairs: portal search",
       24: "COUNT hull pairs queued", 25: "COUNT portal searches", 26: "COUNT support evaluations in portal searches", 27: "COUNT hull vertices scanned by them", 28: "(inside portal search) cycles in the two support scans"}
label = {k: v for k, v in enumerate(names)}
label.update({16 + k: v for k, v in SUB.items()})
slow = np.argsort(-tot)[:40]
typ = np.argsort(tot)[n // 2 - 200: n // 2 + 200]
q = np.quantile(tot, [0.5, 0.9, 0.99, 0.999, 1.0])
print(f"{env_id} @{n}: cycles per env.step p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; sum over worlds {tot.sum():.3e}; worlds above 2 x median: {int((tot > 2 * q[0]).sum())} holding {100 * tot[tot > 2 * q[0]].sum() / tot.sum():.1f} % of all cycles")
lane = getattr(env, "lane", None)
if lane is not None and getattr(lane, "mode", "") == "lane":
    torch.cuda.synchronize()
    fl = lane.nxt.flags.cpu().numpy().astype(bool)[:n]      # (after a step: nxt = the list the LAST step's lane launch walked)
    if fl is not None and fl.any():
        print(f"worlds in the standing lane: {int(fl.sum())}; their cycles: mean {tot[fl].mean():.0f} (x{tot[fl].mean() / q[0]:.2f} the median world) p90 {np.quantile(tot[fl], .9):.0f} max {tot[fl].max():.0f}; share of all cycles {100 * tot[fl].sum() / tot.sum():.1f} %")
print(f"cycles per env.step: median world {np.median(tot):.0f}, slowest 40 worlds mean {tot[slow].mean():.0f} (x{tot[slow].mean() / np.median(tot):.2f})")
print(f"{'stage':34s} {'typical world':>14s} {'slowest 40':>12s} {'difference':>12s}")
rows = sorted(label, key=lambda k: -(P[slow, k].mean() - P[typ, k].mean()))
for k in rows:
    a, b = P[typ, k].mean(), P[slow, k].mean()
    if a > 0 or b > 0:
        print(f"{label[k]:34s} {a:14.0f} {b:12.0f} {b - a:12.0f}")
