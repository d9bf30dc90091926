"""Fetch family task specifications (host side).

The constants are the constructor arguments of the reference env classes
(/root/reference/gymnasium_robotics/envs/fetch/reach.py:125-147, push.py:167-190,
slide.py:166-189, pick_and_place.py:139-162) and the registry entries
(/root/reference/gymnasium_robotics/__init__.py:26-80: ids, reward_type, max_episode_steps=50).
``FetchTaskStruct`` mirrors ``struct GrxFetchTask`` in csrc/grx_fetch_task.h.
"""
import ctypes

import numpy as np

N_SUBSTEPS = 20
DISTANCE_THRESHOLD = 0.05
MAX_EPISODE_STEPS = 50

FETCH_TASKS = {
    "FetchReach": dict(xml="fetch/reach.xml", has_object=False, block_gripper=True, gripper_extra_height=0.2,
                       target_in_the_air=True, target_offset=0.0, obj_range=0.15, target_range=0.15,
                       initial_qpos={"robot0:slide0": 0.4049, "robot0:slide1": 0.48, "robot0:slide2": 0.0}),
    "FetchPush": dict(xml="fetch/push.xml", has_object=True, block_gripper=True, gripper_extra_height=0.0,
                      target_in_the_air=False, target_offset=0.0, obj_range=0.15, target_range=0.15,
                      initial_qpos={"robot0:slide0": 0.405, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                    "object0:joint": [1.25, 0.53, 0.4, 1.0, 0.0, 0.0, 0.0]}),
    "FetchSlide": dict(xml="fetch/slide.xml", has_object=True, block_gripper=True, gripper_extra_height=-0.02,
                       target_in_the_air=False, target_offset=np.array([0.4, 0.0, 0.0]), obj_range=0.1, target_range=0.3,
                       initial_qpos={"robot0:slide0": 0.05, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                     "object0:joint": [1.7, 1.1, 0.41, 1.0, 0.0, 0.0, 0.0]}),
    "FetchPickAndPlace": dict(xml="fetch/pick_and_place.xml", has_object=True, block_gripper=False,
                              gripper_extra_height=0.2, target_in_the_air=True, target_offset=0.0, obj_range=0.15,
                              target_range=0.15,
                              initial_qpos={"robot0:slide0": 0.405, "robot0:slide1": 0.48, "robot0:slide2": 0.0,
                                            "object0:joint": [1.25, 0.53, 0.4, 1.0, 0.0, 0.0, 0.0]}),
}


def parse_env_id(env_id: str):
    """'FetchPickAndPlaceDense-v4' -> ('FetchPickAndPlace', 'dense')   (registry: __init__.py:26-80)"""
    base = env_id.split("-v")[0]
    reward_type = "sparse"
    if base.endswith("Dense"):
        base, reward_type = base[: -len("Dense")], "dense"
    if base not in FETCH_TASKS:
        raise KeyError(f"unknown Fetch env id {env_id}")
    return base, reward_type


class FetchTaskStruct(ctypes.Structure):
    _fields_ = [
        ("has_object", ctypes.c_int), ("block_gripper", ctypes.c_int), ("n_substeps", ctypes.c_int),
        ("sparse_reward", ctypes.c_int), ("grip_body", ctypes.c_int), ("grip_relpos", ctypes.c_float * 3),
        ("grip_relquat", ctypes.c_float * 4), ("site_grip", ctypes.c_int), ("site_obj", ctypes.c_int),
        ("jq_rf", ctypes.c_int), ("jq_lf", ctypes.c_int), ("jd_rf", ctypes.c_int), ("jd_lf", ctypes.c_int),
        ("obs_dim", ctypes.c_int), ("goal_dim", ctypes.c_int), ("dt", ctypes.c_float),
        ("distance_threshold", ctypes.c_double),
    ]


def make_fetch_task(model, task: str, reward_type: str = "sparse") -> FetchTaskStruct:
    cfg = FETCH_TASKS[task]
    n, T = model.names, model.tables
    gb, gp, gq = model.info["body_orig"]["robot0:gripper_link"]
    jq, jd = T["jnt_qposadr"].ravel(), T["jnt_dofadr"].ravel()
    rf, lf = n["joint"]["robot0:r_gripper_finger_joint"], n["joint"]["robot0:l_gripper_finger_joint"]
    t = FetchTaskStruct()
    t.has_object, t.block_gripper = int(cfg["has_object"]), int(cfg["block_gripper"])
    t.n_substeps, t.sparse_reward = N_SUBSTEPS, int(reward_type == "sparse")
    t.grip_body = int(gb)
    t.grip_relpos[:] = [float(x) for x in gp]
    t.grip_relquat[:] = [float(x) for x in gq]
    t.site_grip = n["site"]["robot0:grip"]
    t.site_obj = n["site"].get("object0", 0)
    t.jq_rf, t.jq_lf, t.jd_rf, t.jd_lf = int(jq[rf]), int(jq[lf]), int(jd[rf]), int(jd[lf])
    t.obs_dim = 25 if cfg["has_object"] else 10
    t.goal_dim = 3
    t.distance_threshold = DISTANCE_THRESHOLD
    t.dt = float(N_SUBSTEPS * model.opt("timestep"))
    return t
