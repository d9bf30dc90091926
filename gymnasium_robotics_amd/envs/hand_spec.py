"""Shadow Dexterous Hand task description shared by the device env and the test oracle (host logic only).

Mirrors /root/reference/gymnasium_robotics/envs/shadow_dexterous_hand/reach.py (ids, initial pose, goal sampling) and
hand_env.py:36-58 (absolute position control: ctrl = centre + action * half-range, clipped to the actuator ctrlrange).
"""
import numpy as np

FINGERTIP_SITE_NAMES = ["robot0:S_fftip", "robot0:S_mftip", "robot0:S_rftip", "robot0:S_lftip", "robot0:S_thtip"]  # reach.py:8-14

# reach.py:16-41 (values are data, not code)
DEFAULT_INITIAL_QPOS = {
    "robot0:WRJ1": -0.16514339750464327, "robot0:WRJ0": -0.31973286565062153,
    "robot0:FFJ3": 0.14340512546557435, "robot0:FFJ2": 0.32028208333591573, "robot0:FFJ1": 0.7126053607727917, "robot0:FFJ0": 0.6705281001412586,
    "robot0:MFJ3": 0.000246444303701037, "robot0:MFJ2": 0.3152655251085491, "robot0:MFJ1": 0.7659800313729842, "robot0:MFJ0": 0.7323156897425923,
    "robot0:RFJ3": 0.00038520700007378114, "robot0:RFJ2": 0.36743546201985233, "robot0:RFJ1": 0.7119514095008576, "robot0:RFJ0": 0.6699446327514138,
    "robot0:LFJ4": 0.0525442258033891, "robot0:LFJ3": -0.13615534724474673, "robot0:LFJ2": 0.39872030433433003, "robot0:LFJ1": 0.7415570009679252,
    "robot0:LFJ0": 0.704096378652974,
    "robot0:THJ4": 0.003673823825070126, "robot0:THJ3": 0.5506291436028695, "robot0:THJ2": -0.014515151997119306,
    "robot0:THJ1": -0.0015229223564485414, "robot0:THJ0": -0.7894883021600622,
}
DISTANCE_THRESHOLD = 0.01   # reach.py:60
N_SUBSTEPS = 20             # reach.py:61
MAX_EPISODE_STEPS = 50      # __init__.py:90-95
N_ACTIONS = 20              # hand_env.py:27


def parse_hand_reach_id(env_id: str):
    """'HandReach-v3' / 'HandReachDense-v3' (gymnasium_robotics/__init__.py:90-95) -> reward_type."""
    name, _, version = env_id.rpartition("-")
    if version not in ("v3", "v2", "v1") or name not in ("HandReach", "HandReachDense"):
        raise ValueError(f"unknown HandReach id {env_id!r}")
    return "dense" if name.endswith("Dense") else "sparse"


def initial_qpos_vector(model) -> np.ndarray:
    """qpos after _env_setup (reach.py:408-411): the named joints set to DEFAULT_INITIAL_QPOS."""
    q = np.array(model.tables["qpos0"], dtype=np.float64).copy()
    jq = model.tables["jnt_qposadr"]
    for name, value in DEFAULT_INITIAL_QPOS.items():
        q[int(np.asarray(jq).reshape(-1)[model.names["joint"][name]])] = value
    return q


def fingertip_site_ids(model):
    return [int(model.names["site"][n]) for n in FINGERTIP_SITE_NAMES]


_FINGERS = [name for name in FINGERTIP_SITE_NAMES if name != "robot0:S_thtip"]
_THUMB_IDX = FINGERTIP_SITE_NAMES.index("robot0:S_thtip")
_MEET_OFFSET = np.array([0.0, -0.09, 0.05])


def sample_hand_reach_goal(np_random: np.random.Generator, initial_goal: np.ndarray, palm_xpos: np.ndarray) -> np.ndarray:
    """reach.py:99-126, draw for draw: choice of the finger, normal meeting-point noise, the 10 % 'stay' branch.
    `Generator.choice(list)` without p draws `integers(0, len)` (checked stream-for-stream in tests/test_cpu_hand.py); using the
    latter skips the list -> array conversion, which dominated the host cost of resetting thousands of worlds."""
    finger_idx = FINGERTIP_SITE_NAMES.index(_FINGERS[np_random.integers(0, len(_FINGERS))])
    meeting_pos = np.asarray(palm_xpos, dtype=np.float64) + _MEET_OFFSET
    meeting_pos = meeting_pos + np_random.normal(scale=0.005, size=3)
    goal = np.array(initial_goal, dtype=np.float64).reshape(-1, 3)
    for idx in (_THUMB_IDX, finger_idx):
        offset_direction = meeting_pos - goal[idx]
        offset_direction /= np.sqrt(offset_direction @ offset_direction)
        goal[idx] = meeting_pos - 0.005 * offset_direction
    if np_random.uniform() < 0.1:
        # With some probability all fingers are asked to move back to the origin (reach.py:122-125)
        goal = np.array(initial_goal, dtype=np.float64)
    return goal.reshape(-1)


def sample_hand_reach_goal_batch(np_randoms, initial_goal: np.ndarray, palm_xpos: np.ndarray) -> np.ndarray:
    """sample_hand_reach_goal for many worlds: three draws per world from its own generator (finger index, 3 normals, 1 uniform -- the
    reference's order), the geometry once on arrays."""
    n = len(np_randoms)
    fi, noise, stay = np.zeros(n, np.int64), np.zeros((n, 3)), np.zeros(n)
    finger_of = np.array([FINGERTIP_SITE_NAMES.index(f) for f in _FINGERS])
    for k, r in enumerate(np_randoms):
        fi[k] = r.integers(0, len(_FINGERS))
        noise[k] = r.normal(scale=0.005, size=3)
        stay[k] = r.uniform()
    meeting = np.asarray(palm_xpos, dtype=np.float64) + _MEET_OFFSET
    meeting = meeting + noise                                            # [n, 3]
    goal = np.tile(np.array(initial_goal, dtype=np.float64).reshape(1, -1, 3), (n, 1, 1))
    rows = np.arange(n)
    for idx in (np.full(n, _THUMB_IDX), finger_of[fi]):
        d = meeting - goal[rows, idx]
        d = d / np.sqrt(d[:, 0:1] * d[:, 0:1] + d[:, 1:2] * d[:, 1:2] + d[:, 2:3] * d[:, 2:3])
        goal[rows, idx] = meeting - 0.005 * d
    goal = goal.reshape(n, -1)
    goal[stay < 0.1] = np.array(initial_goal, dtype=np.float64).reshape(-1)
    return goal


def hand_reach_reward(achieved, desired, reward_type="sparse"):
    """reach.py:92-97."""
    d = np.linalg.norm(np.asarray(achieved) - np.asarray(desired), axis=-1)
    return -(d > DISTANCE_THRESHOLD).astype(np.float32) if reward_type == "sparse" else -d


# ---- C struct mirror of GrxHandTask (csrc/grx_hand_task.h) / grx_hand_task (include/grx_capi.h)
import ctypes  # noqa: E402


class HandTaskStruct(ctypes.Structure):
    _fields_ = [("n_substeps", ctypes.c_int), ("sparse_reward", ctypes.c_int), ("site", ctypes.c_int * 5), ("palm_body", ctypes.c_int),
                ("distance_threshold", ctypes.c_double), ("kind", ctypes.c_int), ("nq_robot", ctypes.c_int), ("obj_qadr", ctypes.c_int),
                ("obj_dadr", ctypes.c_int), ("ignore_position", ctypes.c_int), ("ignore_rotation", ctypes.c_int), ("rotation_threshold", ctypes.c_float), ("ignore_z", ctypes.c_int), ("touch_mode", ctypes.c_int)]


def make_hand_task(model, reward_type="sparse") -> HandTaskStruct:
    t = HandTaskStruct()
    t.n_substeps, t.sparse_reward = N_SUBSTEPS, int(reward_type == "sparse")
    for k, sid in enumerate(fingertip_site_ids(model)):
        t.site[k] = sid
    t.palm_body = int(model.names["body"]["robot0:palm"])
    t.distance_threshold = DISTANCE_THRESHOLD
    return t
