"""Adroit hand task description shared by the device env and the test oracle (host logic only).

Mirrors /root/reference/gymnasium_robotics/envs/adroit_hand/adroit_hammer.py: the actuator gain / bias rewrite of the constructor
(:234-262), the action scaling (:291-293), frame_skip = 5 (:211), the 46-vector observation (:331-357), the dense / sparse reward and the
success flag (:296-324), reset_model's board-height draw (:372-378) and the registered ids (gymnasium_robotics/__init__.py:1082-1101).
"""
import os
from typing import Optional

import numpy as np

FRAME_SKIP = 5                 # adroit_hammer.py:211
MAX_EPISODE_STEPS = 200        # __init__.py:1099
OBS_DIM = 46                   # adroit_hammer.py:205-207
HAMMER_XML = os.path.join("adroit_hand", "adroit_hammer.xml")
# engine compile options: the board is the per-world shift group (model.body_pos[nail_board] is redrawn at every reset, :374-376), the nail's
# touch sensor feeds the observation (:344-346), and only the four sites the task reads are tracked
HAMMER_COMPILE = dict(shift_body="nail_board", touch_filter=lambda name: name == "S_nail", keep_sites=["S_grasp", "S_target", "nail_goal", "tool"])
_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")


def parse_adroit_id(env_id: str):
    """'AdroitHandHammer-v2' / 'AdroitHandHammerSparse-v2' (also -v1: same environment class) -> ('hammer', reward_type)"""
    name, _, version = env_id.rpartition("-")
    if version not in ("v1", "v2") or name not in ("AdroitHandHammer", "AdroitHandHammerSparse"):
        raise KeyError(f"unknown / unsupported Adroit env id {env_id}")
    return "hammer", ("sparse" if name.endswith("Sparse") else "dense")


def apply_actuator_overrides(model):
    """adroit_hammer.py:234-262: the wrist actuators A_WRJ1..A_WRJ0 get gain 10 / bias (0, -10, 0), the finger actuators A_FFJ3..A_THJ0 gain 1 /
    bias (0, -1, 0); the two arm actuators keep the values of the XML.  Edits the compiled tables in place (MjModel is edited in place there)."""
    T, A = model.tables, model.names["actuator"]
    gain, bias = T["act_gainprm"].reshape(-1, 3), T["act_biasprm"].reshape(-1, 3)
    gain[A["A_WRJ1"]: A["A_WRJ0"] + 1] = [10.0, 0.0, 0.0]
    gain[A["A_FFJ3"]: A["A_THJ0"] + 1] = [1.0, 0.0, 0.0]
    bias[A["A_WRJ1"]: A["A_WRJ0"] + 1] = [0.0, -10.0, 0.0]
    bias[A["A_FFJ3"]: A["A_THJ0"] + 1] = [0.0, -1.0, 0.0]
    return model


def load_adroit_hammer_model(assets_root: Optional[str] = None, capacity=None):
    from ..mjcf import compile_mjcf, load_model

    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        return apply_actuator_overrides(compile_mjcf(os.path.join(assets_root, HAMMER_XML), capacity=capacity, **HAMMER_COMPILE))
    path = os.path.join(_MODELS_DIR, "adroit_hammer.npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist (no packaged model and no assets_root given)")
    return load_model(path)   # the packaged blob already carries the constructor's actuator rewrite


def action_scaling(model):
    """act_mean, act_rng (adroit_hammer.py:269-272)"""
    cr = np.array(model.tables["act_ctrlrange"], dtype=np.float64).reshape(-1, 2)
    return cr.mean(axis=1), 0.5 * (cr[:, 1] - cr[:, 0])


def board_shift(model, z: float) -> np.ndarray:
    """shift vector of the board group for model.body_pos[nail_board, 2] = z (the XML value is the zero of the shift)"""
    return np.array([0.0, 0.0, z - model.info["shift_pos0"][2]])


def hammer_reward(palm, hamm, head, nail, goal, qvel, sparse: bool):
    """adroit_hammer.py:303-324 (float64 arithmetic, the reference's order of operations) -> (reward, goal_achieved)"""
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


def make_adroit_task(model, reward_type: str):
    from .. import _native

    n = model.names
    return _native.AdroitTaskStruct(FRAME_SKIP, int(reward_type == "sparse"), int(n["site"]["S_grasp"]), int(n["site"]["S_target"]), int(n["site"]["nail_goal"]),
                                    int(n["site"]["tool"]), int(n["body"]["Object"]), int(model.dim("nq")) - 6, OBS_DIM)
