"""Diagnostic (not a test): per-snapshot teacher-forced error distribution on the GPU."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
for task in ("FetchReach", "FetchPush", "FetchPickAndPlace"):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"fetch_{task}_teacher.npz"))
    n = g["obs"].shape[0]
    env = FetchVecEnv(task + "-v4", num_envs=n, device="cuda:0", autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    env.load_world_rows({k: g[k] for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal")})
    obs, r, _, _, info = env.step(g["action"])
    err = np.abs(obs["observation"] - g["obs"]).max(axis=1)
    order = np.argsort(-err)[:8]
    print(task, "quantiles 50/90/99/max:", np.quantile(err, [0.5, 0.9, 0.99, 1.0]))
    for i in order:
        print(f"   snap {i:3d} err {err[i]:.2e} gap {g['activation_gap'][i]:.2e} nefc {g['nefc'][i]} ncon {g['ncon'][i]} argmax {np.abs(obs['observation'][i]-g['obs'][i]).argmax()}")
