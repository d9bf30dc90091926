"""Table demand and hull-search rate of FetchPickAndPlace worlds under the bench rollout (profiling build: sh tools/build_prof.sh fetch): per world and env.step
the largest row count / Jacobian-pool demand / contact count of its substeps and the number of portal searches -- the data the capacities of the fast kernel and the
population of the overflow lane are chosen from.
    python tools/demand_probe.py [steps] > profiles/demand_r03_fetch.txt"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd import _native, make_vec
_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", os.environ.get("GRX_PROF_LIB", "libgrx_hip_prof.so"))
NP, n = 56, 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
env = make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
L = _native.lib()
L.grx_profile_world_stages.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_int * (NP * n))()
rows, pool, ncon, srch, queued = [], [], [], [], []
for k in range(steps):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize()
    L.grx_profile_world_stages(buf, n)
    P = np.frombuffer(buf, dtype=np.int32).reshape(n, NP)
    rows.append(P[:, 16 + 32].copy()); pool.append(P[:, 16 + 33].copy()); ncon.append(P[:, 16 + 34].copy()); srch.append(P[:, 16 + 25].copy()); queued.append(P[:, 16 + 24].copy())
rows, pool, ncon, srch, queued = (np.array(x) for x in (rows, pool, ncon, srch, queued))      # [steps, n]
if os.environ.get("GRX_DEMAND_NPZ"):      # the raw per-world-step maxima, for offline evaluation of capacity triples
    np.savez_compressed(os.environ["GRX_DEMAND_NPZ"], rows=rows.astype(np.int16), pool=pool.astype(np.int16), ncon=ncon.astype(np.int16))
q = [50, 90, 95, 98, 99, 99.5, 99.9, 100]
print(f"FetchPickAndPlace-v4, {n} worlds, {steps} steps, uniform random actions, staggered same-step resets; per world and env.step")
for name, a in (("rows (max over the substeps)", rows), ("Jacobian-pool words", pool), ("contacts", ncon), ("hull pairs queued per step (20 substeps)", queued), ("portal searches per step", srch)):
    print(f"{name:44s} " + " ".join(f"p{p}={np.percentile(a, p):.0f}" for p in q))
print("worlds x steps with at least one portal search: %.3f %%" % (100.0 * (srch > 0).mean()))
s = srch > 0
new = s[1:] & ~s[:-1]
print("... of which the world had none in the previous step (entrants): %.3f %% of world-steps = %.1f worlds per step of %d" % (100.0 * new.mean(), new.sum(axis=1).mean(), n))
for lag in (2, 4, 8):
    seen = np.zeros_like(s[lag:])
    for j in range(1, lag + 1):
        seen |= s[lag - j: len(s) - j]
    e = s[lag:] & ~seen
    print(f"... none in the previous {lag} steps: {e.sum(axis=1).mean():.2f} worlds per step")
qd = queued > 20
print("worlds x steps with more hull pairs queued than the one persistent pair (> 20 per step): %.3f %%" % (100.0 * qd.mean()))
e = s[1:] & ~qd[:-1] & ~s[:-1]
print("search in this step although the previous step queued only the persistent pair and searched nothing: %.2f worlds per step" % e.sum(axis=1).mean())
for cap_r, cap_p, cap_c in ((144, 1984, 32), (112, 1520, 32), (96, 1024, 24), (80, 768, 24), (64, 512, 16), (64, 640, 20)):
    over = (rows > cap_r) | (pool > cap_p) | (ncon > cap_c)
    soft = (rows > 0.8 * cap_r) | (pool > 0.8 * cap_p) | (ncon > 0.8 * cap_c)
    ent = over[1:] & ~soft[:-1]
    print(f"capacity rows {cap_r} pool {cap_p} contacts {cap_c}: over {100.0 * over.mean():.3f} % of world-steps, within 80 % {100.0 * soft.mean():.3f} %, over without having been within 80 % the step before: {ent.sum(axis=1).mean():.2f} worlds per step")
