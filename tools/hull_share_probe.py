"""How many FetchPickAndPlace worlds have ANY hull-pair activity in a step, and how many of them are new (profiling build: sh tools/build_prof.sh fetch).  The sizing question of a
fast kernel WITHOUT the hull routine that hands such worlds off: the hand-off rate is the number of NEW worlds per step, the hull kernel's load the total.
    python tools/hull_share_probe.py [worlds] > profiles/hull_share_r06_fetch.txt"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gymnasium_robotics_amd import _native, make_vec
_native.LIB_PATH = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", os.environ.get("GRX_PROF_LIB", "libgrx_hip_prof.so"))
NP, n = 56, 4096
env = make_vec("FetchPickAndPlace-v4", num_envs=n, device="cuda:0", output="torch", autoreset_mode="same_step")
env.reset(seed=0); env._elapsed[:] = np.arange(n) % 50
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
L = _native.lib()
L.grx_profile_world_stages.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_int * (NP * n))()
for k in range(60):      # pre-roll: one episode horizon and a bit
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
prev_q = prev_s = None
print("FetchPickAndPlace-v4, 4096 worlds, staggered episodes, after a 60-step pre-roll; per step: worlds with >= 1 hull pair QUEUED (passed the OBB filter and the gates), of them new since the previous step; worlds with >= 1 portal SEARCH (the cached direction did not separate), new; worlds with a hull contact")
tq = tn = ts = tsn = 0
for k in range(30):
    env.step(torch.rand(n, 4, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize()
    L.grx_profile_world_stages(buf, n)
    P = np.frombuffer(buf, dtype=np.int32).reshape(n, NP).copy()
    queued, srch = P[:, 16 + 24] > 0, P[:, 16 + 25] > 0
    rows, pool, con = P[:, 16 + 32], P[:, 16 + 33], P[:, 16 + 34]
    newq = int((queued & ~prev_q).sum()) if prev_q is not None else -1
    news = int((srch & ~prev_s).sum()) if prev_s is not None else -1
    print(f"step {k:2d}: queued {int(queued.sum()):4d} (new {newq:3d})  searching {int(srch.sum()):4d} (new {news:3d})  | table demand over 96 rows: {int((rows > 96).sum()):3d}  over 1024 pool words: {int((pool > 1024).sum()):3d}  over 24 contacts: {int((con > 24).sum()):3d}"
          f"  | over 80 rows {int((rows > 80).sum()):3d} over 768 words {int((pool > 768).sum()):3d} over 20 contacts {int((con > 20).sum()):3d}")
    if prev_q is not None:
        tq += int(queued.sum()); tn += newq; ts += int(srch.sum()); tsn += news
    prev_q, prev_s = queued, srch
print(f"mean per step: queued {tq / 29:.1f} worlds ({100 * tq / 29 / n:.2f} %), new {tn / 29:.1f}; searching {ts / 29:.1f} ({100 * ts / 29 / n:.2f} %), new {tsn / 29:.1f}")
