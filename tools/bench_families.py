"""Throughput of the non-headline families on one MI355X (the headline metric is bench.py):  python tools/bench_families.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gymnasium_robotics_amd.envs.fetch import FetchVecEnv
from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv, HandReachVecEnv
from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv, PointMazeVecEnv

CASES = ((FetchVecEnv, "FetchReach-v4", 4096, 4), (FetchVecEnv, "FetchPush-v4", 4096, 4), (FetchVecEnv, "FetchSlide-v4", 4096, 4), (FetchVecEnv, "FetchPickAndPlace-v4", 4096, 4),
         (HandReachVecEnv, "HandReach-v3", 4096, 20), (HandReachVecEnv, "HandReach-v3", 16384, 20),
         (HandBlockVecEnv, "HandManipulateBlockRotateXYZ-v1", 4096, 20), (HandBlockVecEnv, "HandManipulateBlockRotateXYZ-v1", 16384, 20),
         (HandBlockVecEnv, "HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", 16384, 20),
         (HandBlockVecEnv, "HandManipulateEggRotate-v1", 16384, 20), (HandBlockVecEnv, "HandManipulatePenRotate-v1", 16384, 20),
         (AntMazeVecEnv, "AntMaze_Large_Diverse_GR-v5", 8192, 8), (PointMazeVecEnv, "PointMaze_Large_Diverse_GR-v3", 65536, 2))
for cls, env_id, n, na in CASES:
    env = cls(env_id, num_envs=n, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
    env.reset(seed=0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    for _ in range(5):
        env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize(); t = time.time(); K = 30
    for _ in range(K):
        env.step(torch.rand(n, na, device="cuda:0", generator=g) * 2 - 1)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    lds = env._L.grx_model_lds_bytes(env._h)
    st = env.status
    print(f"{env_id}: N={n}  {dt*1e3:.2f} ms/step  {n/dt:,.0f} env-steps/s  LDS/world {lds} B  status max {int(st.max())} (worlds flagged {int((st != 0).sum())})")
    env.close()
