"""Per-stage cycle breakdown of grx_fetch_step_kernel (needs a GPU).  Builds/loads the -DGRX_PROFILE variant of the
HIP library, runs a few steps and prints shader-clock cycles per stage for one representative world.
    python tools/profile_stages.py [n_worlds]
"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gymnasium_robotics_amd import _native
prof_so = os.path.join(ROOT, "gymnasium_robotics_amd", "_lib", "libgrx_hip_prof.so")
if not os.path.exists(prof_so):
    from __graft_entry__ import HIPCC_FLAGS
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-DGRX_PROFILE", "-o", prof_so,
                           os.path.join(ROOT, "gymnasium_robotics_amd", "csrc", "grx_kernels.hip")])
_native.LIB_PATH = prof_so
from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv, HandReachVecEnv
from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv
from gymnasium_robotics_amd.envs.adroit import AdroitVecEnv as AdroitHammerVecEnv
from gymnasium_robotics_amd.envs.kitchen import KitchenVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env_id = sys.argv[2] if len(sys.argv) > 2 else "FetchPickAndPlace-v4"
Env = KitchenVecEnv if env_id.startswith("FrankaKitchen") else AdroitHammerVecEnv if env_id.startswith("Adroit") else FetchVecEnv if env_id.startswith("Fetch") else AntMazeVecEnv if env_id.startswith("AntMaze") else (HandReachVecEnv if env_id.startswith("HandReach") else HandBlockVecEnv)
env = Env(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
env.reset(seed=0)
NA = env.single_action_space.shape[0]
L = _native.lib()
names = ["kinematics", "inertia_cdof_crb_M", "collision", "make_constraint", "velocity_rne", "M_factor_solve", "newton_eval", "newton_grad",
         "newton_hessian", "newton_factor_solve", "newton_linesearch", "newton_final", "euler", "other"]
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
NP = 56
tot = np.zeros(NP)
K = 10
for k in range(K):
    a = torch.rand(n, NA, device="cuda:0", generator=g) * 2 - 1
    if NA == 4:
        a[:, 2] = -a[:, 2].abs()  # push towards the table so that contacts are active
    else:
        a *= 0.3                   # gentle hand motion keeps the object in the hand
    L.grx_profile_reset()
    env.step(a)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * NP)()
    L.grx_profile_read(out)
    if k >= 2:
        tot += np.array(list(out), dtype=np.float64) / n  # kernel sums over worlds
tot /= (K - 2)
s = tot.sum()
print(f"{env_id}: cycles per env.step (all substeps of the step), mean over {n} worlds: {s:.0f}  ")
for nm, v in zip(names, tot):
    print(f"  {nm:22s} {v:12.0f}  {100*v/s:5.1f}%")
SUB = {0: "constraint: count", 1: "constraint: scan", 2: "constraint: equality rows", 3: "constraint: friction/limit rows", 4: "constraint: row params", 5: "constraint: contact J",
       6: "velocity: rne a", 7: "velocity: rne b", 8: "velocity: passive/actuation", 9: "kinematics: bodies", 10: "kinematics: sites/frames", 11: "collision: box-box queue",
       12: "collision: geom frames", 13: "collision: narrow phase rounds", 14: "inertia: cinert/cdof", 15: "inertia: crb/M", 16: "collision: hull pairs", 17: "solver aux a", 18: "solver aux b",
       19: "collision: candidate sweep", 20: "collision: survivor regroup", 21: "hull pairs: set-up", 22: "hull pairs: cached direction check", 23: "hull pairs: portal search",
       24: "COUNT hull pairs queued", 25: "COUNT portal searches", 26: "COUNT support evaluations", 27: "COUNT hull vertices scanned", 28: "(inside the portal searches) cycles in the support scans", 29: "COUNT Newton iterations", 30: "COUNT constrained solves", 31: "COUNT constraint rows (sum over solves)",
       32: "MAX rows demanded (per world; summed over worlds here)", 33: "MAX Jacobian-pool words demanded", 34: "MAX contacts", 36: "collision: skin-list check / rebuild", 37: "COUNT candidates swept (sum over substeps)", 38: "COUNT sweep survivors (sum over substeps)"}
for k in range(16, NP):
    if tot[k] > 0:
        print(f"  sub[{k-16:2d}] {SUB.get(k - 16, ''):32s} {tot[k]:12.0f}  {100*tot[k]/s:5.1f}%")
