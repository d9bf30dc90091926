"""Record TRUE-MuJoCo teacher-forcing fixtures with the reference itself.

NOT runnable in the build image (neither `mujoco` nor `gymnasium` is installed there, and there is no network): run it on a
machine with `pip install gymnasium-robotics mujoco`, commit the produced `tests/golden/mujoco_<id>.npz`, and
`tests/test_gpu_mujoco_golden.py` will compare the HIP path against them (it skips while the files are absent).  That turns the
oracle's "parity unpinned" status into parity pinned against the reference.

    python tools/record_golden.py [env ids ...]

Each fixture holds, per snapshot: the full pre-step MjData state the device kernels take as input (qpos, qvel, qacc_warmstart,
mocap pose, the stale gripper-body pose the Fetch _set_action reads), the goal, the action, and the reference's outputs
(observation, achieved goal, reward, success), plus `mujoco.__version__`.
"""
import sys

import numpy as np

DEFAULT_IDS = ["FetchReach-v4", "FetchPush-v4", "FetchSlide-v4", "FetchPickAndPlace-v4", "HandReach-v3", "HandManipulateBlockRotateXYZ-v1",
               "HandManipulateEggRotate-v1", "HandManipulatePenRotate-v1",   # Slide / Egg exercise MuJoCo's convex collider: record mujoco.__version__ (libccd MPR <= 3.1, native GJK/EPA later)
               "AdroitHandHammer-v2", "AdroitHandDoor-v2", "AdroitHandPen-v2", "AdroitHandRelocate-v2", "FrankaKitchen-v1", "AntMaze_UMaze-v5", "PointMaze_UMaze-v3"]


def record_plain(env_id, episodes=6, steps=50, seed0=0):
    """Adroit (plain Env; the per-episode MODEL edits of reset_model are recorded: body_pos / body_quat / site_pos rows), FrankaKitchen (noise ratios 0: the
    device path is fed no noise; _last_robot_qpos recorded) and the mazes (goal = the target site)."""
    import gymnasium as gym
    import gymnasium_robotics
    import mujoco

    gym.register_envs(gymnasium_robotics)
    kw = dict(robot_noise_ratio=0.0, object_noise_ratio=0.0) if env_id.startswith("FrankaKitchen") else {}
    top = gym.make(env_id, **kw).unwrapped
    sim = top.robot_env if env_id.startswith("FrankaKitchen") else (getattr(top, "ant_env", None) or getattr(top, "point_env", None) or top)
    model, data = sim.model, sim.data
    rng = np.random.default_rng(1234)
    rec = {k: [] for k in ("qpos", "qvel", "qacc_ws", "action", "obs", "reward", "success", "seed", "t", "edit", "target", "last_qpos", "goal")}
    bid = lambda name: mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_BODY, name)
    for ep in range(episodes):
        top.reset(seed=seed0 + ep)
        for t in range(steps):
            a = rng.uniform(-1, 1, top.action_space.shape[0]).astype(np.float32)
            rec["qpos"].append(data.qpos.copy()); rec["qvel"].append(data.qvel.copy()); rec["qacc_ws"].append(data.qacc_warmstart.copy()); rec["action"].append(a)
            edit, target, last, goal = np.zeros(4), np.zeros(3), np.zeros(9), np.zeros(2)
            if env_id.startswith("AdroitHandHammer"):
                edit[:3] = model.body_pos[bid("nail_board")]
            elif env_id.startswith("AdroitHandDoor"):
                edit[:3] = model.body_pos[bid("frame")]
            elif env_id.startswith("AdroitHandPen"):
                edit[:] = model.body_quat[bid("target")]
            elif env_id.startswith("AdroitHandRelocate"):
                edit[:3] = model.body_pos[bid("Object")]
                target[:] = model.site_pos[mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_SITE, "target")]
            elif env_id.startswith("FrankaKitchen"):
                last[:] = sim._last_robot_qpos
            else:
                goal[:] = top.goal
            rec["edit"].append(edit); rec["target"].append(target); rec["last_qpos"].append(last); rec["goal"].append(goal)
            obs, r, term, trunc, info = top.step(a)
            rec["obs"].append(obs["observation"] if isinstance(obs, dict) else obs); rec["reward"].append(r)
            rec["success"].append(float(info.get("success", info.get("is_success", 0.0)))); rec["seed"].append(seed0 + ep); rec["t"].append(t)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["mujoco_version"] = np.frombuffer(mujoco.__version__.encode(), dtype=np.uint8)
    out["nq"], out["nv"] = np.int64(model.nq), np.int64(model.nv)
    return out


def record(env_id, episodes=6, steps=50, seed0=0):
    import gymnasium as gym
    import gymnasium_robotics
    import mujoco

    gym.register_envs(gymnasium_robotics)
    env = gym.make(env_id).unwrapped
    model, data = env.model, env.data
    rng = np.random.default_rng(1234)
    keys = ("qpos", "qvel", "qacc_ws", "mocap", "aux", "goal", "action", "obs", "achieved", "reward", "success", "seed", "t")
    rec = {k: [] for k in keys}
    fetch = env_id.startswith("Fetch")
    grip = mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_BODY, "robot0:gripper_link") if fetch else -1
    for ep in range(episodes):
        env.reset(seed=seed0 + ep)
        for t in range(steps):
            a = rng.uniform(-1, 1, env.action_space.shape[0]).astype(np.float32)
            rec["qpos"].append(data.qpos.copy()); rec["qvel"].append(data.qvel.copy()); rec["qacc_ws"].append(data.qacc_warmstart.copy())
            rec["mocap"].append(np.concatenate([data.mocap_pos.ravel(), data.mocap_quat.ravel()]) if model.nmocap else np.zeros(0))
            rec["aux"].append(np.concatenate([data.xpos[grip], data.xquat[grip], [0.0]]) if fetch else np.zeros(8))
            rec["goal"].append(env.goal.copy()); rec["action"].append(a)
            obs, r, term, trunc, info = env.step(a)
            rec["obs"].append(obs["observation"]); rec["achieved"].append(obs["achieved_goal"]); rec["reward"].append(r)
            rec["success"].append(info["is_success"]); rec["seed"].append(seed0 + ep); rec["t"].append(t)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["mujoco_version"] = np.frombuffer(mujoco.__version__.encode(), dtype=np.uint8)
    out["nq"], out["nv"] = np.int64(model.nq), np.int64(model.nv)
    return out


if __name__ == "__main__":
    import os

    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    for env_id in (sys.argv[1:] or DEFAULT_IDS):
        d = record(env_id) if env_id.startswith(("Fetch", "HandReach", "HandManipulate")) else record_plain(env_id)
        path = os.path.join(out_dir, f"mujoco_{env_id}.npz")
        np.savez_compressed(path, **d)
        print(env_id, d["obs"].shape, "->", path)
