"""What the portal searches of the slow FetchPickAndPlace worlds end in (profiling build with -DGRX_PROBE_HULL: sh tools/build_prof.sh fetch -DGRX_PROBE_HULL):
per world of one step launch the number of searches, how many found a contact, how many ended on a separating direction, and the pair searched last."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd import _native, make_vec
_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", "libgrx_hip_prof.so")
NP, n = 56, 4096
env = make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
L = _native.lib()
L.grx_profile_world_stages.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_int * (NP * n))()
T = env.model.tables
gname = {v: k.replace("robot0:", "") for k, v in env.model.names["geom"].items()}
for k in range(40):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
L.grx_profile_world_stages(buf, n)
P = np.frombuffer(buf, dtype=np.int32).reshape(n, NP)
srch, con, sep, last, queued, evals = P[:, 16 + 25], P[:, 16 + 35], P[:, 16 + 36], P[:, 16 + 37], P[:, 16 + 24], P[:, 16 + 26]
print("worlds with searches:", int((srch > 0).sum()), "searches", int(srch.sum()), "contacts", int(con.sum()), "separating direction", int(sep.sum()), "neither", int((srch - con - sep).sum()))
order = np.argsort(-srch)[:30]
print("row of a world without searches:", P[np.argmin(srch)].tolist())
print("row of the world with most searches:", P[order[0]].tolist())
qpos = env.qpos.cpu().numpy()
for w in order:
    p = int(last[w]); g1, g2 = int(T["pair_geom1"][p]), int(T["pair_geom2"][p])
    print(f"world {w}: queued {queued[w]} searches {srch[w]} evals {evals[w]} contact {con[w]} sep {sep[w]} last pair {p} ({gname[g1]} / {gname[g2]}) pan {qpos[w, 6]:.2f} lift {qpos[w, 7]:.2f} roll {qpos[w, 8]:.2f} elbow {qpos[w, 9]:.2f}")
