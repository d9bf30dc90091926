import sys, numpy as np, torch, os
sys.path.insert(0, '.')
import gymnasium_robotics_amd as grx
env = grx.make_vec("FetchSlide-v4", num_envs=4, device="cuda:0", output="torch", autoreset_mode="disabled", max_episode_steps=None)
obs, _ = env.reset(seed=0)
rec = {"qpos": [env.qpos.cpu().numpy().copy()], "qvel": [env.qvel.cpu().numpy().copy()], "qacc_ws": [env.qacc_ws.cpu().numpy().copy()], "mocap": [env.mocap.cpu().numpy().copy()], "aux": [env.aux.cpu().numpy().copy()], "act": []}
rng = np.random.default_rng(3)
for t in range(12):
    a = rng.uniform(-1, 1, (4, 4)).astype(np.float32); a[:, 2] = -1.0 if t < 6 else 0.0
    env.step(torch.from_numpy(a).cuda())
    rec["act"].append(a)
    for k in ("qpos", "qvel", "qacc_ws", "mocap", "aux"):
        rec[k].append(getattr(env, k).cpu().numpy().copy())
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/slide_probe_%s.npz" % os.environ.get("TAG", "default"), **{k: np.asarray(v) for k, v in rec.items()})
print("saved")
