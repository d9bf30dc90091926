"""Adroit hand task descriptions shared by the device env and the test oracle (host logic only).

Mirrors /root/reference/gymnasium_robotics/envs/adroit_hand/{adroit_hammer,adroit_door,adroit_pen,adroit_relocate}.py: the actuator gain / bias
rewrite of the constructors (adroit_hammer.py:234-262, identical in the other three), the action scaling (:291-293), frame_skip = 5 (:211), the
observations, the dense / sparse rewards and success flags, each reset_model's draws, and the registered ids (gymnasium_robotics/__init__.py:990-1101).

Per-world MODEL edits of the reference (reset_model writes model.body_pos / body_quat / site_pos) are per-world STATE here:
  hammer    body_pos[nail_board, 2] = U(0.1, 0.25)                                  -> shift offset (0, 0, z - z0)
  door      body_pos[frame] = (U(-0.3,-0.2), U(0.25,0.35), U(0.252,0.35))           -> shift offset pos - pos0
  pen       body_quat[target] = euler2quat((U(-1,1), U(-1,1), 0))                    -> shift rotation about the target body's origin
  relocate  body_pos[Object, :2] = (U(-0.15,0.15), U(-0.15,0.3)); site_pos[target] = (U(-0.2,0.2), U(-0.2,0.2), U(0.15,0.35))
                                                                                     -> shift offset of the ball's body + the world's target vector
"""
import os
from typing import Optional

import numpy as np

from .manipulate_spec import quat_conj, quat_mul

FRAME_SKIP = 5                 # adroit_hammer.py:211 (all four)
MAX_EPISODE_STEPS = 200        # __init__.py:1092,1099,1106,1113 (all four tasks)
_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")
_no_touch = lambda name: False

# engine compile options per task: the shift group, the touch sensors the observation reads, the sites the task reads
SPECS = {
    "hammer": dict(kind=0, xml="adroit_hammer.xml", npz="adroit_hammer.npz", obs_dim=46, max_episode_steps=200, state_keys=("board_pos", "target_pos"),
                   compile=dict(shift_body="nail_board", touch_filter=lambda name: name == "S_nail", keep_sites=["S_grasp", "S_target", "nail_goal", "tool"]),
                   sites=("S_grasp", "S_target", "nail_goal", "tool")),
    "door": dict(kind=1, xml="adroit_door.xml", npz="adroit_door.npz", obs_dim=39, max_episode_steps=200, state_keys=("door_body_pos",),
                 compile=dict(shift_body="frame", touch_filter=_no_touch, keep_sites=["S_grasp", "S_handle"]), sites=("S_grasp", "S_handle")),
    "pen": dict(kind=2, xml="adroit_pen.xml", npz="adroit_pen.npz", obs_dim=45, max_episode_steps=200, state_keys=("desired_orien",),
                compile=dict(shift_body="target", shift_rotates=True, touch_filter=_no_touch,
                             keep_sites=["eps_ball", "object_top", "object_bottom", "target_top", "target_bottom"]),
                sites=("eps_ball", "object_top", "object_bottom", "target_top", "target_bottom")),
    "relocate": dict(kind=3, xml="adroit_relocate.xml", npz="adroit_relocate.npz", obs_dim=39, max_episode_steps=200, state_keys=("obj_pos", "target_pos"),
                     compile=dict(shift_body="Object", touch_filter=_no_touch, keep_sites=["S_grasp"]), sites=("S_grasp",)),
}
_ID_TASK = {"AdroitHandHammer": "hammer", "AdroitHandDoor": "door", "AdroitHandPen": "pen", "AdroitHandRelocate": "relocate"}
OBS_DIM = SPECS["hammer"]["obs_dim"]
HAMMER_XML = os.path.join("adroit_hand", "adroit_hammer.xml")
HAMMER_COMPILE = SPECS["hammer"]["compile"]


def parse_adroit_id(env_id: str):
    """'AdroitHandDoor-v1' / 'AdroitHandPenSparse-v1' / ... (also -v2: hammer, -v1 everywhere else is the registered one; both accepted) -> (task, reward_type)"""
    name, _, version = env_id.rpartition("-")
    sparse = name.endswith("Sparse")
    base = name[:-6] if sparse else name
    if version not in ("v1", "v2") or base not in _ID_TASK:
        raise KeyError(f"unknown / unsupported Adroit env id {env_id}")
    return _ID_TASK[base], ("sparse" if sparse else "dense")


def apply_actuator_overrides(model):
    """adroit_hammer.py:234-262 (door :222-250, pen :232-260, relocate :225-253): the wrist actuators A_WRJ1..A_WRJ0 get gain 10 / bias (0, -10, 0), the finger
    actuators A_FFJ3..A_THJ0 gain 1 / bias (0, -1, 0); the arm actuators keep the values of the XML.  Edits the compiled tables in place."""
    T, A = model.tables, model.names["actuator"]
    gain, bias = T["act_gainprm"].reshape(-1, 3), T["act_biasprm"].reshape(-1, 3)
    gain[A["A_WRJ1"]: A["A_WRJ0"] + 1] = [10.0, 0.0, 0.0]
    gain[A["A_FFJ3"]: A["A_THJ0"] + 1] = [1.0, 0.0, 0.0]
    bias[A["A_WRJ1"]: A["A_WRJ0"] + 1] = [0.0, -10.0, 0.0]
    bias[A["A_FFJ3"]: A["A_THJ0"] + 1] = [0.0, -1.0, 0.0]
    return model


# capacities of the fast kernels' tables per task (None: the engine defaults, 144 rows / 2 032 pool words / 32 contacts): they set the LDS footprint = the worlds a CU holds; the
# overflow lane steps the worlds that exceed them, so they are a throughput choice (measured: profiles/ab_r05_adroit_capacity.txt, ab_r05_adroit_capacity2.txt); csrc/grx_kernels.hip GRX_ADROIT_*_CAP must agree
ADROIT_CAPACITY = {"hammer": dict(maxefc=96, jpool=1024, maxcon=24), "pen": dict(maxefc=80, jpool=896, maxcon=24), "door": None, "relocate": None}


def load_adroit_model(task: str, assets_root: Optional[str] = None, capacity=None):
    from ..mjcf import compile_mjcf, load_model

    spec = SPECS[task]
    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        return apply_actuator_overrides(compile_mjcf(os.path.join(assets_root, "adroit_hand", spec["xml"]), capacity=capacity or ADROIT_CAPACITY[task], **spec["compile"]))
    path = os.path.join(_MODELS_DIR, spec["npz"])
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist (no packaged model and no assets_root given)")
    model = load_model(path)   # the packaged blob already carries the constructor's actuator rewrite
    cap = capacity or ADROIT_CAPACITY[task]
    if os.environ.get("GRX_ADROIT_CAP"):      # "rows,pool,contacts" (A/B of the fast kernels' tables; needs a library built with the matching -DGRX_ADROIT_ME / _JP / _MC to stay on the specialised kernels)
        cap = dict(zip(("maxefc", "jpool", "maxcon"), (int(x) for x in os.environ["GRX_ADROIT_CAP"].split(","))))
    return model.with_capacity(**cap) if cap else model


def load_adroit_hammer_model(assets_root: Optional[str] = None, capacity=None):
    return load_adroit_model("hammer", assets_root, capacity)


def action_scaling(model):
    """act_mean, act_rng (adroit_hammer.py:269-272)"""
    cr = np.array(model.tables["act_ctrlrange"], dtype=np.float64).reshape(-1, 2)
    return cr.mean(axis=1), 0.5 * (cr[:, 1] - cr[:, 0])


# ---------------------------------------------------------------------------------------------------- per-world model edits -> engine state
IDENTITY_SHIFT = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])


def euler2quat(euler):
    """utils/rotations.py:140-159, operation for operation (batched over leading axes)"""
    euler = np.asarray(euler, dtype=np.float64)
    ai, aj, ak = euler[..., 2] / 2, -euler[..., 1] / 2, euler[..., 0] / 2
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    quat = np.empty(euler.shape[:-1] + (4,), dtype=np.float64)
    quat[..., 0] = cj * cc + sj * ss
    quat[..., 3] = cj * sc - sj * cs
    quat[..., 2] = -(cj * ss + sj * cc)
    quat[..., 1] = cj * cs - sj * sc
    return quat


def _quat2mat(q):
    """rotation matrix of unit quaternion(s) (..., 4) -> (..., 3, 3)"""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = w * w + x * x - y * y - z * z; R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = w * w - x * x + y * y - z * z; R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = w * w - x * x - y * y + z * z
    return R


def group_shift(model, pos=None, quat=None) -> np.ndarray:
    """7-vector(s) (t, q) the engine applies to the model's shift group for model.body_pos[body] = pos and / or model.body_quat[body] = quat (one
    world, or a batch along the leading axis): members move as x -> R (x - p0) + pos with R = R(quat) R(quat0)^T, i.e. x -> R x + t with
    t = pos - R p0 (translation-only groups: R = 1, t = pos - p0)."""
    p0, q0 = np.asarray(model.info["shift_pos0"], dtype=np.float64), np.asarray(model.info.get("shift_quat0", [1.0, 0, 0, 0]), dtype=np.float64)
    lead = np.shape(pos)[:-1] if pos is not None else np.shape(quat)[:-1]
    pos = np.broadcast_to(p0, lead + (3,)) if pos is None else np.asarray(pos, dtype=np.float64)
    if quat is None:
        return np.concatenate([pos - p0, np.broadcast_to(IDENTITY_SHIFT[3:], lead + (4,))], axis=-1)
    q = np.asarray(quat, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)                 # mj_kinematics normalises body_quat before use [3P]
    rel = quat_mul(q, quat_conj(q0))
    return np.concatenate([pos - _quat2mat(rel) @ p0, rel], axis=-1)


def board_shift(model, z: float) -> np.ndarray:
    """hammer: model.body_pos[nail_board, 2] = z"""
    p = np.array(model.info["shift_pos0"], dtype=np.float64)
    p[2] = z
    return group_shift(model, pos=p)


def sample_reset_batch(task: str, rngs, model, current=None):
    """reset_model's draws for a list of per-world generators, in the reference's order -> dict(edit [k, 3 or 4] = what get_env_state reports,
    shift [k, 7], target [k, 3] or None).  Only the draws run per world; the arithmetic on them is batched.
    current ([k, 3]): the worlds' present model.body_pos rows.  reset_model rewrites only SOME components (hammer: board z, adroit_hammer.py:374-376; relocate:
    ball x / y, adroit_relocate.py:353-356); what a set_env_state / options["initial_state_dict"] wrote into the others persists across resets in the
    reference, so the rewrite starts from the current row, not from the XML pose."""
    k = len(rngs)
    p0 = np.tile(np.asarray(model.info["shift_pos0"], dtype=np.float64), (k, 1))
    if current is not None and task in ("hammer", "relocate"):
        p0 = np.array(current, dtype=np.float64).reshape(k, 3).copy()
    if task == "hammer":       # adroit_hammer.py:374-376
        p0[:, 2] = [r.uniform(low=0.1, high=0.25) for r in rngs]
        return dict(edit=p0, shift=group_shift(model, pos=p0), target=None)
    if task == "door":         # adroit_door.py:362-370
        pos = np.array([[r.uniform(low=-0.3, high=-0.2), r.uniform(low=0.25, high=0.35), r.uniform(low=0.252, high=0.35)] for r in rngs]).reshape(k, 3)
        return dict(edit=pos, shift=group_shift(model, pos=pos), target=None)
    if task == "pen":          # adroit_pen.py:380-383
        e = np.zeros((k, 3))
        e[:, :2] = np.array([[r.uniform(low=-1, high=1), r.uniform(low=-1, high=1)] for r in rngs]).reshape(k, 2)
        quat = euler2quat(e)
        return dict(edit=quat, shift=group_shift(model, quat=quat), target=None)
    if task == "relocate":     # adroit_relocate.py:353-372
        d = np.array([[r.uniform(low=-0.15, high=0.15), r.uniform(low=-0.15, high=0.3), r.uniform(low=-0.2, high=0.2), r.uniform(low=-0.2, high=0.2),
                       r.uniform(low=0.15, high=0.35)] for r in rngs]).reshape(k, 5)
        p0[:, :2] = d[:, :2]
        return dict(edit=p0, shift=group_shift(model, pos=p0), target=d[:, 2:].copy())
    raise KeyError(task)


def sample_reset(task: str, rng, model, current=None):
    """one world: dict(edit, shift, target)"""
    d = sample_reset_batch(task, [rng], model, current=None if current is None else np.asarray(current, dtype=np.float64).reshape(1, 3))
    return {k: (None if v is None else v[0]) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------------- rewards (float64, the reference's order of operations)
def hammer_reward(palm, hamm, head, nail, goal, qvel, sparse: bool):
    """adroit_hammer.py:303-324 -> (reward, goal_achieved)"""
    goal_distance = np.linalg.norm(nail - goal)
    goal_achieved = bool(goal_distance < 0.01)
    reward = 10.0 if goal_achieved else -0.1
    if not sparse:
        reward = -0.1 * np.linalg.norm(palm - hamm)
        reward -= np.linalg.norm(head - nail)
        reward -= 10 * np.linalg.norm(nail - goal)
        reward -= 1e-2 * np.linalg.norm(qvel)
        if hamm[2] > 0.04 and head[2] > 0.04:
            reward += 2
        if goal_distance < 0.020:
            reward += 25
        if goal_distance < 0.010:
            reward += 75
    return reward, goal_achieved


def door_reward(goal_distance, palm, handle, qvel, sparse: bool):
    """adroit_door.py:287-316 (goal_distance = the door hinge angle)"""
    goal_achieved = bool(goal_distance >= 1.35)
    reward = 10.0 if goal_achieved else -0.1
    if not sparse:
        reward = -0.1 * np.linalg.norm(palm - handle)
        reward += -0.1 * (goal_distance - 1.57) * (goal_distance - 1.57)
        reward += -1e-5 * np.sum(qvel ** 2)
        if goal_distance > 0.2:
            reward += 2
        if goal_distance > 1.0:
            reward += 8
        if goal_distance > 1.35:
            reward += 10
    return reward, goal_achieved


def pen_reward(obj_pos, desired_loc, obj_orien, desired_orien, sparse: bool):
    """adroit_pen.py:306-327"""
    goal_distance = np.linalg.norm(obj_pos - desired_loc)
    orien_similarity = np.dot(obj_orien, desired_orien)
    goal_achieved = bool(goal_distance < 0.075 and orien_similarity > 0.95)
    reward = 10.0 if goal_achieved else -0.1
    if not sparse:
        reward = -goal_distance + orien_similarity
        if goal_distance < 0.075 and orien_similarity > 0.9:
            reward += 10
        if goal_distance < 0.075 and orien_similarity > 0.95:
            reward += 50
        if obj_pos[2] < 0.075:
            reward -= 5
    return reward, goal_achieved


def relocate_reward(obj_pos, palm_pos, target_pos, sparse: bool):
    """adroit_relocate.py:298-321"""
    goal_distance = float(np.linalg.norm(obj_pos - target_pos))
    goal_achieved = goal_distance < 0.1
    reward = 10.0 if goal_achieved else -0.1
    if not sparse:
        reward = -0.1 * np.linalg.norm(palm_pos - obj_pos)
        if obj_pos[2] > 0.04:
            reward += 1.0
            reward += -0.5 * np.linalg.norm(palm_pos - target_pos)
            reward += -0.5 * np.linalg.norm(obj_pos - target_pos)
        if goal_distance < 0.1:
            reward += 10.0
        if goal_distance < 0.05:
            reward += 20.0
    return reward, goal_achieved


def quat2euler(quat):
    """utils/rotations.py:227-232 + :245-274 + :162-184 for one quaternion, operation for operation (quat2mat scales by s = 2 / |q|^2 instead of
    normalising; mat2euler is the fixed-axis x-y-z extraction with the gimbal branch at cy <= 4 eps)."""
    q = np.asarray(quat, dtype=np.float64)
    w, x, y, z = q
    Nq = np.sum(q * q)
    if not Nq > np.finfo(np.float64).eps:
        R = np.eye(3)
    else:
        s = 2.0 / Nq
        X, Y, Z = x * s, y * s, z * s
        wX, wY, wZ = w * X, w * Y, w * Z
        xX, xY, xZ = x * X, x * Y, x * Z
        yY, yZ, zZ = y * Y, y * Z, z * Z
        R = np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])
    cy = np.sqrt(R[2, 2] * R[2, 2] + R[1, 2] * R[1, 2])
    if cy > np.finfo(np.float64).eps * 4.0:
        return np.array([-np.arctan2(R[1, 2], R[2, 2]), -np.arctan2(-R[0, 2], cy), -np.arctan2(R[0, 1], R[0, 0])])
    return np.array([0.0, -np.arctan2(-R[0, 2], cy), -np.arctan2(-R[1, 0], R[1, 1])])


def pen_lengths(model):
    """pen_length, tar_length (adroit_pen.py:385-392): |top - bottom| of the two site pairs (rigid: the local distance)"""
    S, P = model.names["site"], np.asarray(model.tables["site_pos"], dtype=np.float64).reshape(-1, 3)
    return float(np.linalg.norm(P[S["object_top"]] - P[S["object_bottom"]])), float(np.linalg.norm(P[S["target_top"]] - P[S["target_bottom"]]))


def door_qpos_indices(model):
    """(hinge, latch): the reference reads qpos[jnt_dofadr[door_hinge]] (adroit_door.py:264-266,287) -- equal to the qpos address in this all-1-dof model -- and qpos[-1]"""
    j = model.names["joint"]["door_hinge"]
    return int(np.asarray(model.tables["jnt_dofadr"]).ravel()[j]), int(model.dim("nq")) - 1


def make_adroit_task(model, reward_type: str, task: str = "hammer"):
    from .. import _native

    spec, n = SPECS[task], model.names
    t = _native.AdroitTaskStruct()
    t.n_substeps, t.sparse_reward, t.kind = FRAME_SKIP, int(reward_type == "sparse"), spec["kind"]
    for k, name in enumerate(spec["sites"]):
        t.site[k] = int(n["site"][name])
    t.obj_body = int(n["body"]["Object"]) if task != "door" else 0
    nq = int(model.dim("nq"))
    t.nq_obs = nq - 3 if task == "door" else nq - 6
    t.obs_dim = spec["obs_dim"]
    if task == "door":
        t.qadr[0], t.qadr[1] = door_qpos_indices(model)
    if task == "pen":
        t.len[0], t.len[1] = pen_lengths(model)
    return t
