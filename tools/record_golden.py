"""Record TRUE-MuJoCo teacher-forcing fixtures with the reference itself -- ONE command on any machine that has the reference:

    pip install gymnasium-robotics mujoco
    python record_golden.py [--out DIR] [env ids ...]          # default DIR: <this file>/../tests/golden if it exists, else ./golden

The script is SELF-CONTAINED (imports only numpy, gymnasium, gymnasium_robotics, mujoco: copy this one file anywhere) and prints the MuJoCo / gymnasium-robotics
versions and a SHA-256 per fixture.  It cannot run in the build image (no mujoco / gymnasium wheel, no network).  Commit the produced
`tests/golden/mujoco_<id>.npz`: `tests/test_gpu_mujoco_golden.py` (HIP path) and `tests/test_cpu_mujoco_golden.py` (oracle) then compare against them, which turns
the oracle's "parity unpinned" status into parity pinned against the reference.  `GRX_REQUIRE_MUJOCO_GOLDEN=1` makes both test files FAIL instead of skip
while no fixture is committed (CI of a machine that is supposed to have them).

Each fixture holds, per snapshot: the full pre-step MjData state the device kernels take as input (qpos, qvel, qacc_warmstart,
mocap pose, the stale gripper-body pose the Fetch _set_action reads), the goal, the action, and the reference's outputs
(observation, achieved goal, reward, success), plus `mujoco.__version__` (libccd MPR up to 3.1, the native GJK / EPA collider later: the convex-pair
families -- FetchSlide, the egg, Adroit pen / hammer, FrankaKitchen -- depend on it).
"""
import sys

import numpy as np

DEFAULT_IDS = ["FetchReach-v4", "FetchPush-v4", "FetchSlide-v4", "FetchPickAndPlace-v4", "HandReach-v3", "HandManipulateBlockRotateXYZ-v1",
               "HandManipulateEggRotate-v1", "HandManipulatePenRotate-v1",   # Slide / Egg exercise MuJoCo's convex collider: record mujoco.__version__ (libccd MPR <= 3.1, native GJK/EPA later)
               "HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1",      # BASELINE configs[2]: the 153-word observation
               "AdroitHandHammer-v2", "AdroitHandDoor-v2", "AdroitHandPen-v2", "AdroitHandRelocate-v2", "FrankaKitchen-v1", "AntMaze_UMaze-v5",
               "AntMaze_Large_Diverse_GR-v5",                                  # BASELINE configs[3] itself
               "PointMaze_UMaze-v3"]


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
    import hashlib
    import os

    args = sys.argv[1:]
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "..", "tests", "golden") if os.path.isdir(os.path.join(here, "..", "tests", "golden")) else os.path.join(os.getcwd(), "golden")
    if "--out" in args:
        k = args.index("--out")
        out_dir = args[k + 1]
        del args[k: k + 2]
    os.makedirs(out_dir, exist_ok=True)
    import gymnasium
    import gymnasium_robotics
    import mujoco

    print(f"mujoco {mujoco.__version__}, gymnasium {gymnasium.__version__}, gymnasium-robotics {getattr(gymnasium_robotics, '__version__', '?')}, numpy {np.__version__}")
    for env_id in (args or DEFAULT_IDS):
        d = record(env_id) if env_id.startswith(("Fetch", "HandReach", "HandManipulate")) else record_plain(env_id)
        path = os.path.join(out_dir, f"mujoco_{env_id}.npz")
        np.savez_compressed(path, **d)
        with open(path, "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()
        print(f"{env_id}: {d['obs'].shape[0]} snapshots x obs {d['obs'].shape[1]} (nq {int(d['nq'])}, nv {int(d['nv'])}) -> {path}  sha256 {digest}")
