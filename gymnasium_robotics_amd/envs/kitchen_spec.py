"""FrankaKitchen-v1 task description shared by the device env and the test oracle (host logic only).

Mirrors /root/reference/gymnasium_robotics/envs/franka_kitchen/franka_env.py (FrankaRobot: velocity command -> position target on the PREVIOUS noisy
joint reading, position / velocity bounds and noise amplitudes of franka_config.xml, frame_skip 40, :53-171) and kitchen_env.py (KitchenEnv: the seven
tasks' qpos indices and goals :19-37, BONUS_THRESH :38, init_qpos :246-279, the 59-vector observation with object noise :356-384, reward = number of
tasks completed in the step :340-354, task bookkeeping and termination :386-423, registration gymnasium_robotics/__init__.py:1117-1122).
"""
import os
from typing import Optional

import numpy as np

FRAME_SKIP = 40                 # franka_env.py:54
MAX_EPISODE_STEPS = 280         # __init__.py:1120
BONUS_THRESH = 0.3              # kitchen_env.py:38
N_ROBOT = 9
OBS_DIM = 59                    # 9 + 9 robot, 21 + 20 objects (kitchen_env.py:109)
KITCHEN_XML = os.path.join("kitchen_franka", "kitchen_assets", "kitchen_env_model.xml")
FRANKA_CONFIG = os.path.join("kitchen_franka", "franka_assets", "franka_config.xml")
_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")

# kitchen_env.py:19-37 (dict order = the order of the tasks everywhere below)
OBS_ELEMENT_INDICES = {
    "bottom burner": np.array([11, 12]), "top burner": np.array([15, 16]), "light switch": np.array([17, 18]), "slide cabinet": np.array([19]),
    "hinge cabinet": np.array([20, 21]), "microwave": np.array([22]), "kettle": np.array([23, 24, 25, 26, 27, 28, 29]),
}
OBS_ELEMENT_GOALS = {
    "bottom burner": np.array([-0.88, -0.01]), "top burner": np.array([-0.92, -0.01]), "light switch": np.array([-0.69, -0.05]),
    "slide cabinet": np.array([0.37]), "hinge cabinet": np.array([0.0, 1.45]), "microwave": np.array([-0.75]),
    "kettle": np.array([-0.23, 0.75, 1.62, 0.99, 0.0, 0.0, -0.06]),
}
TASKS = list(OBS_ELEMENT_GOALS)
# kitchen_env.py:246-279
INIT_QPOS = np.array([
    1.48388023e-01, -1.76848573e00, 1.84390296e00, -2.47685760e00, 2.60252026e-01, 7.12533105e-01, 1.59515394e00, 4.79267505e-02, 3.71350919e-02,
    -2.66279850e-04, -5.18043486e-05, 3.12877220e-05, -4.51199853e-05, -3.90842156e-06, -4.22629655e-05, 6.28065475e-05, 4.04984708e-05, 4.62730939e-04,
    -2.26906415e-04, -4.65501369e-04, -6.44129196e-03, -1.77048263e-03, 1.08009684e-03, -2.69397440e-01, 3.50383255e-01, 1.61944683e00, 1.00618764e00,
    4.06395120e-03, -6.62095997e-03, -2.68278933e-04])
# no engine sites (the task reads qpos only).  Capacities of the FAST kernel: a finger pad pressed flat on a box is a box-box pair with up to 8 contacts of 10 pyramid rows
# (condim 6) each; rounds 2 - 4 gave the fast kernel 192 rows / 2 240 pool words / 32 contacts = 31.9 KB of LDS = 5 worlds per CU.  The step kernel's throughput is nearly
# proportional to the worlds a CU holds (profiles/ab_r05_two_worlds_occupancy.txt) and the overflow lane (400 / 8 160 / 64 tables) steps whatever exceeds the fast tables, so the
# fast tables are a throughput choice: 128 / 1 280 / 24 = 26.1 KB = 6 worlds per CU measures +5.7 % (fast launch 47.6 -> 41.7 ms, the lane's launches then end the step at 45.2 ms:
# profiles/ab_r05_kitchen_capacity.txt).  csrc/grx_kernels.hip GRX_KITCHEN_CAP must agree.
KITCHEN_CAPACITY = {"maxcon": 24, "maxefc": 128, "jpool": 1280}
KITCHEN_COMPILE = dict(keep_sites=[])


def read_franka_config(path: str, nv: int):
    """franka_env.py:177-202 (_read_specs_from_config): pos_bound / vel_bound [nv, 2], pos_noise_amp / vel_noise_amp [nv] of the nodes qpos0 .. qpos<nv-1>"""
    import xml.etree.ElementTree as ET

    root = ET.parse(path).getroot()
    get = lambda i, key: np.array(root.find(f"qpos{i}").get(key).split(), dtype=float)
    return dict(pos_bound=np.stack([get(i, "pos_bound") for i in range(nv)]), vel_bound=np.stack([get(i, "vel_bound") for i in range(nv)]),
                pos_noise_amp=np.array([get(i, "pos_noise_amp")[0] for i in range(nv)]), vel_noise_amp=np.array([get(i, "vel_noise_amp")[0] for i in range(nv)]))


def load_kitchen_model(assets_root: Optional[str] = None, capacity=None):
    """the compiled kitchen scene; model.info["franka_config"] carries the numbers of franka_config.xml (the asset is not shipped)"""
    from ..mjcf import compile_mjcf, load_model

    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        m = compile_mjcf(os.path.join(assets_root, KITCHEN_XML), capacity=capacity or KITCHEN_CAPACITY, **KITCHEN_COMPILE)
        cfg = read_franka_config(os.path.join(assets_root, FRANKA_CONFIG), m.dim("nv"))
        m.info["franka_config"] = {k: v.tolist() for k, v in cfg.items()}
        return m
    path = os.path.join(_MODELS_DIR, "kitchen.npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist (no packaged model and no assets_root given)")
    cap = dict(capacity or KITCHEN_CAPACITY)
    if os.environ.get("GRX_KITCHEN_CAP"):      # "rows,pool,contacts": A/B of the fast kernel's tables (needs a library built with the matching -DGRX_KITCHEN_CAP to stay on the specialised kernel)
        cap = dict(zip(("maxefc", "jpool", "maxcon"), (int(x) for x in os.environ["GRX_KITCHEN_CAP"].split(","))))
    return load_model(path).with_capacity(**cap)


def franka_config(model):
    return {k: np.asarray(v, dtype=np.float64) for k, v in model.info["franka_config"].items()}


def noise_scales(model, robot_noise_ratio: float, object_noise_ratio: float) -> np.ndarray:
    """per-element factor of the 59 uniform(-1, 1) draws of one observation, in draw order = observation order: robot qpos (9), robot qvel (9)
    (franka_env.py:118-127), object qpos (21, amplitudes pos_noise_amp[8:]), object qvel (20, vel_noise_amp[9:]) (kitchen_env.py:361-369)"""
    c = franka_config(model)
    return np.concatenate([robot_noise_ratio * c["pos_noise_amp"][:9], robot_noise_ratio * c["vel_noise_amp"][:9],
                           object_noise_ratio * c["pos_noise_amp"][8:], object_noise_ratio * c["vel_noise_amp"][9:]])


def control_targets(model, action, last_robot_qpos, dt: float):
    """FrankaRobot.step -> ctrl (franka_env.py:92-103,136-171): clip, denormalise (act_mid 0, act_rng 2), velocity limits, position target on the previous
    noisy reading, position limits.  action [..., 9], last_robot_qpos [..., 9]"""
    c = franka_config(model)
    a = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
    a = 0.0 + a * 2.0
    vel = np.clip(a, c["vel_bound"][:9, 0], c["vel_bound"][:9, 1])
    pos = np.asarray(last_robot_qpos, dtype=np.float64) + vel * dt
    return np.clip(pos, c["pos_bound"][:9, 0], c["pos_bound"][:9, 1])


def task_mask(tasks) -> int:
    for t in tasks:
        if t not in OBS_ELEMENT_GOALS:
            raise ValueError(f"The task {t} cannot be found the the list of possible goals: {OBS_ELEMENT_GOALS.keys()}")   # kitchen_env.py:291-294
    return sum(1 << TASKS.index(t) for t in set(tasks))


def completed_mask(qpos) -> np.ndarray:
    """bit k set <=> |qpos[idx_k] - goal_k| < BONUS_THRESH (kitchen_env.py:346-351), for a batch [..., 30] of TRUE (noise-free) qpos"""
    qpos = np.asarray(qpos, dtype=np.float64)
    out = np.zeros(qpos.shape[:-1], np.int64)
    for k, t in enumerate(TASKS):
        d = np.linalg.norm(qpos[..., OBS_ELEMENT_INDICES[t]] - OBS_ELEMENT_GOALS[t], axis=-1)
        out |= (d < BONUS_THRESH).astype(np.int64) << k
    return out


def make_kitchen_task(model, robot_noise_ratio: float, object_noise_ratio: float):
    from .. import _native

    t = _native.KitchenTaskStruct()
    t.n_substeps, t.obs_dim = FRAME_SKIP, OBS_DIM
    t.dt = float(model.opt("timestep")) * FRAME_SKIP
    c = franka_config(model)
    for i in range(9):
        t.vel_lo[i], t.vel_hi[i], t.pos_lo[i], t.pos_hi[i] = c["vel_bound"][i, 0], c["vel_bound"][i, 1], c["pos_bound"][i, 0], c["pos_bound"][i, 1]
    ns = noise_scales(model, robot_noise_ratio, object_noise_ratio)
    for i in range(OBS_DIM):
        t.noise_scale[i] = ns[i]
    k = 0
    for j, name in enumerate(TASKS):
        idx, goal = OBS_ELEMENT_INDICES[name], OBS_ELEMENT_GOALS[name]
        t.task_adr[j], t.task_num[j] = int(idx[0]), len(idx)
        for g in goal:
            t.task_goal[k] = g
            k += 1
    t.bonus_thresh = BONUS_THRESH
    return t
