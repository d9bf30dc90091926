"""HandManipulateBlock* task description shared by the device env and the test oracle (host logic only).

Mirrors /root/reference/gymnasium_robotics/envs/shadow_dexterous_hand/manipulate.py (reset pose randomisation, goal sampling,
goal distance / reward / success) and manipulate_block.py:214-230 (target position range), plus the quaternion helpers of
gymnasium_robotics/utils/rotations.py that those use.  The helpers are pinned against the reference's own code through
tests/golden/ref_rotations.npz (tools/make_reference_vectors.py).
"""
import itertools

import numpy as np

TARGET_POSITION_RANGE = np.array([(-0.04, 0.04), (-0.06, 0.02), (0.0, 0.06)])   # manipulate_block.py:226
DISTANCE_THRESHOLD, ROTATION_THRESHOLD = 0.01, 0.1                               # manipulate.py:32-33
N_SUBSTEPS, MAX_EPISODE_STEPS, SETTLE_STEPS, PALM_HEIGHT = 20, 100, 10, 0.04     # manipulate.py:34,205-216; __init__.py:284
# registered ids (gymnasium_robotics/__init__.py:124-341, 644-800): name -> (object, target_position, target_rotation)
BLOCK_VARIANTS = {
    "HandManipulateBlockRotateZ": ("block", "ignore", "z"), "HandManipulateBlockRotateParallel": ("block", "ignore", "parallel"),
    "HandManipulateBlockRotateXYZ": ("block", "ignore", "xyz"), "HandManipulateBlockFull": ("block", "random", "xyz"),
    "HandManipulateBlock": ("block", "random", "xyz"),
    "HandManipulateEggRotate": ("egg", "ignore", "xyz"), "HandManipulateEggFull": ("egg", "random", "xyz"), "HandManipulateEgg": ("egg", "random", "xyz"),
    "HandManipulatePenRotate": ("pen", "ignore", "xyz"), "HandManipulatePenFull": ("pen", "random", "xyz"), "HandManipulatePen": ("pen", "random", "xyz"),
}
NO_TOUCH_IDS = ("HandManipulateBlockFull", "HandManipulateEggFull", "HandManipulatePenFull")   # the *Full ids have no touch-sensor twins
# per-object constructor constants: manipulate_block.py:214-230, manipulate_pen.py:216-235
OBJECTS = {
    "block": dict(xml="manipulate_block", randomize_initial_rotation=True, ignore_z_target_rotation=False, distance_threshold=0.01),
    # manipulate_egg.py:216-230: the Manipulate defaults; the ellipsoid goes through the general convex (MPR) narrow phase
    "egg": dict(xml="manipulate_egg", randomize_initial_rotation=True, ignore_z_target_rotation=False, distance_threshold=0.01),
    "pen": dict(xml="manipulate_pen", randomize_initial_rotation=False, ignore_z_target_rotation=True, distance_threshold=0.05),
}


TOUCH_MODES = {"off": 0, "sensordata": 1, "boolean": 2, "log": 3}   # manipulate_touch_sensors.py:36-41, 124-131
N_TOUCH = 92


def touch_filter(sensor_name: str) -> bool:
    """The sensors the touch envs read: names containing 'robot0:TS_' (manipulate_touch_sensors.py:66-81)."""
    return "robot0:TS_" in sensor_name


def parse_block_id(env_id: str):
    """'HandManipulateBlockRotateXYZ-v1', '..._ContinuousTouchSensors-v1', '..._BooleanTouchSensorsDense-v1', ... ->
    (target_position, target_rotation, reward_type, touch_get_obs)   (gymnasium_robotics/__init__.py:124-341)."""
    name, _, version = env_id.rpartition("-")
    dense = name.endswith("Dense")
    base = name[:-5] if dense else name
    touch = "off"
    for suffix, mode in (("_ContinuousTouchSensors", "sensordata"), ("_BooleanTouchSensors", "boolean")):
        if base.endswith(suffix):
            base, touch = base[: -len(suffix)], mode
    if version != "v1" or base not in BLOCK_VARIANTS or (touch != "off" and base in NO_TOUCH_IDS):
        raise ValueError(f"unknown HandManipulate id {env_id!r}")
    _obj, tp, tr = BLOCK_VARIANTS[base]
    return (tp, tr, "dense" if dense else "sparse", touch)


def object_of(env_id: str) -> str:
    """'block', 'egg' or 'pen'."""
    name = env_id.rpartition("-")[0]
    name = name[:-5] if name.endswith("Dense") else name
    for suffix in ("_ContinuousTouchSensors", "_BooleanTouchSensors"):
        name = name[: -len(suffix)] if name.endswith(suffix) else name
    if name not in BLOCK_VARIANTS:
        raise ValueError(f"unknown HandManipulate id {env_id!r}")
    return BLOCK_VARIANTS[name][0]


def drop_target_body(root):
    """MJCF mutation: remove the 'target' body.  It is a free, non-colliding (contype = conaffinity = 0) visual marker whose state
    never enters observation, reward or goal (manipulate.py:298-316); without it the device model has nv = 30 instead of 36."""
    for wb in root.iter("worldbody"):
        for b in list(wb):
            if b.tag == "body" and b.attrib.get("name") == "target":
                wb.remove(b)


# ---- quaternion helpers (w, x, y, z), restating utils/rotations.py
def quat_conj(q):
    q = np.asarray(q, dtype=np.float64)
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def quat_mul(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    aw, av, bw, bv = a[..., :1], a[..., 1:], b[..., :1], b[..., 1:]
    return np.concatenate([aw * bw - np.sum(av * bv, axis=-1, keepdims=True), aw * bv + bw * av + np.cross(av, bv)], axis=-1)


def quat_from_angle_and_axis(angle, axis):
    """manipulate.py:10-15."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    q = np.concatenate([[np.cos(angle / 2.0)], np.sin(angle / 2.0) * axis])
    return q / np.linalg.norm(q)


def euler2quat(euler):
    """rotations.py:140-159: intrinsic x-y-z Euler angles, i.e. the quaternion product qx(e0) qy(e1) qz(e2)."""
    e = np.asarray(euler, dtype=np.float64)
    half = e / 2.0
    zeros = np.zeros_like(half[..., 0])
    qx = np.stack([np.cos(half[..., 0]), np.sin(half[..., 0]), zeros, zeros], axis=-1)
    qy = np.stack([np.cos(half[..., 1]), zeros, np.sin(half[..., 1]), zeros], axis=-1)
    qz = np.stack([np.cos(half[..., 2]), zeros, zeros, np.sin(half[..., 2])], axis=-1)
    return quat_mul(qx, quat_mul(qy, qz))


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat2euler(q):
    """rotations.py:227-232 for one quaternion."""
    q = np.asarray(q, dtype=np.float64)
    return _mat2euler(_quat2mat(q / np.sqrt(np.sum(q * q)))) if np.sum(q * q) > np.finfo(np.float64).eps else np.zeros(3)


def _mat2euler(R):
    """rotations.py:162-184 for one matrix (fixed-axis x-y-z)."""
    cy = np.sqrt(R[2, 2] * R[2, 2] + R[1, 2] * R[1, 2])
    if cy > np.finfo(np.float64).eps * 4.0:
        return np.array([-np.arctan2(R[1, 2], R[2, 2]), -np.arctan2(-R[0, 2], cy), -np.arctan2(R[0, 1], R[0, 0])])
    return np.array([0.0, -np.arctan2(-R[0, 2], cy), -np.arctan2(-R[1, 0], R[1, 1])])


def canonical_parallel_quats():
    """manipulate.py:66-68: euler2quat of the canonical Euler triples rotations.get_parallel_rotations() (rotations.py:394-408)
    returns -- first occurrence, over itertools.product of (0, 90, -90, 180) degrees per axis, of each distinct triple."""
    mult90 = [0, np.pi / 2, -np.pi / 2, np.pi]
    rots = []
    for e in itertools.product(mult90, repeat=3):
        canonical = np.round(_mat2euler(_quat2mat(euler2quat(np.array(e)))) / (np.pi / 2))
        if canonical[0] == -2:
            canonical[0] = 2
        if canonical[2] == -2:
            canonical[2] = 2
        canonical = canonical * (np.pi / 2)
        if all((canonical != r).any() for r in rots):
            rots.append(canonical)
    assert len(rots) == 24
    return [euler2quat(r) for r in rots]


# ---- reset / goal sampling (draw-for-draw) ------------------------------------------------------------------------------
def sample_reset_object_pose(np_random, initial_pos, initial_quat, target_position, target_rotation, pquats=None,
                             randomize_initial_rotation=True, randomize_initial_position=True):
    """manipulate.py:170-203 (MujocoManipulateEnv._reset_sim): object pose before the settle steps."""
    pos, quat = np.array(initial_pos, dtype=np.float64), np.array(initial_quat, dtype=np.float64)
    if randomize_initial_rotation:
        if target_rotation == "z":
            angle = np_random.uniform(-np.pi, np.pi)
            quat = quat_mul(quat, quat_from_angle_and_axis(angle, np.array([0.0, 0.0, 1.0])))
        elif target_rotation == "parallel":
            angle = np_random.uniform(-np.pi, np.pi)
            z_quat = quat_from_angle_and_axis(angle, np.array([0.0, 0.0, 1.0]))
            parallel_quat = pquats[np_random.integers(len(pquats))]
            quat = quat_mul(quat, quat_mul(z_quat, parallel_quat))
        elif target_rotation in ("xyz", "ignore"):
            angle = np_random.uniform(-np.pi, np.pi)
            axis = np_random.uniform(-1.0, 1.0, size=3)
            quat = quat_mul(quat, quat_from_angle_and_axis(angle, axis))
        elif target_rotation != "fixed":
            raise ValueError(f'Unknown target_rotation option "{target_rotation}".')
    if randomize_initial_position and target_position != "fixed":
        pos = pos + np_random.normal(size=3, scale=0.005)
    quat = quat / np.linalg.norm(quat)
    return np.concatenate([pos, quat])


def sample_block_goal(np_random, object_qpos, target_position, target_rotation, pquats=None):
    """manipulate.py:226-279 (_sample_goal) from the object's CURRENT pose (after the settle steps)."""
    object_qpos = np.asarray(object_qpos, dtype=np.float64)
    if target_position == "random":
        offset = np_random.uniform(TARGET_POSITION_RANGE[:, 0], TARGET_POSITION_RANGE[:, 1])
        target_pos = object_qpos[:3] + offset
    elif target_position in ("ignore", "fixed"):
        target_pos = object_qpos[:3].copy()
    else:
        raise ValueError(f'Unknown target_position option "{target_position}".')
    if target_rotation == "z":
        target_quat = quat_from_angle_and_axis(np_random.uniform(-np.pi, np.pi), np.array([0.0, 0.0, 1.0]))
    elif target_rotation == "parallel":
        target_quat = quat_from_angle_and_axis(np_random.uniform(-np.pi, np.pi), np.array([0.0, 0.0, 1.0]))
        target_quat = quat_mul(target_quat, pquats[np_random.integers(len(pquats))])
    elif target_rotation == "xyz":
        angle = np_random.uniform(-np.pi, np.pi)
        axis = np_random.uniform(-1.0, 1.0, size=3)
        target_quat = quat_from_angle_and_axis(angle, axis)
    else:
        # 'ignore' / 'fixed' call data.get_joint_qpos, which the mujoco bindings do not have (manipulate.py:270): no registered id
        raise ValueError(f'Unknown target_rotation option "{target_rotation}".')
    target_quat = target_quat / np.linalg.norm(target_quat)
    return np.concatenate([target_pos, target_quat])


def _quat_from_angle_and_axis_batch(angle, axis):
    axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    q = np.concatenate([np.cos(angle / 2.0)[:, None], np.sin(angle / 2.0)[:, None] * axis], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def _draw_rotation(np_randoms, target_rotation, npar):
    """per-world draws of one random rotation, in the reference's order: angle [, parallel index | axis]"""
    n = len(np_randoms)
    ang, axis, pidx = np.zeros(n), np.tile(np.array([0.0, 0.0, 1.0]), (n, 1)), np.zeros(n, np.int64)
    for k, r in enumerate(np_randoms):
        ang[k] = r.uniform(-np.pi, np.pi)
        if target_rotation == "parallel":
            pidx[k] = r.integers(npar)
        elif target_rotation != "z":
            axis[k] = r.uniform(-1.0, 1.0, size=3)
    return ang, axis, pidx


def sample_reset_object_pose_batch(np_randoms, initial_pos, initial_quat, target_position, target_rotation, pquats=None,
                                   randomize_initial_rotation=True, randomize_initial_position=True):
    """sample_reset_object_pose for many worlds: the draws come from each world's own generator in the reference's order, the quaternion
    algebra runs once on arrays (a reset of 4 096 worlds spent 170 ms in per-world numpy calls)."""
    n = len(np_randoms)
    pos, quat = np.tile(np.asarray(initial_pos, dtype=np.float64), (n, 1)), np.tile(np.asarray(initial_quat, dtype=np.float64), (n, 1))
    if target_rotation not in ("z", "parallel", "xyz", "ignore", "fixed"):
        raise ValueError(f'Unknown target_rotation option "{target_rotation}".')
    rotate = randomize_initial_rotation and target_rotation != "fixed"
    move = randomize_initial_position and target_position != "fixed"
    ang = axis = pidx = None
    noise = np.zeros((n, 3))
    if rotate and move:          # both kinds of draws interleave per world: angle [..], then the position noise
        ang, axis, pidx = np.zeros(n), np.tile(np.array([0.0, 0.0, 1.0]), (n, 1)), np.zeros(n, np.int64)
        for k, r in enumerate(np_randoms):
            a_, x_, p_ = _draw_rotation([r], target_rotation, len(pquats) if pquats is not None else 0)
            ang[k], axis[k], pidx[k] = a_[0], x_[0], p_[0]
            noise[k] = r.normal(size=3, scale=0.005)
    elif rotate:
        ang, axis, pidx = _draw_rotation(np_randoms, target_rotation, len(pquats) if pquats is not None else 0)
    elif move:
        for k, r in enumerate(np_randoms):
            noise[k] = r.normal(size=3, scale=0.005)
    if rotate:
        off = _quat_from_angle_and_axis_batch(ang, axis)
        if target_rotation == "parallel":
            off = quat_mul(off, np.asarray(pquats)[pidx])
        quat = quat_mul(quat, off)
    pos = pos + noise
    quat = quat / np.linalg.norm(quat, axis=-1, keepdims=True)
    return np.concatenate([pos, quat], axis=-1)


def sample_block_goal_batch(np_randoms, object_qpos, target_position, target_rotation, pquats=None):
    """sample_block_goal for many worlds (object_qpos [n, 7]): per-world draws, array algebra."""
    object_qpos = np.asarray(object_qpos, dtype=np.float64)
    n = len(np_randoms)
    if target_position not in ("random", "ignore", "fixed"):
        raise ValueError(f'Unknown target_position option "{target_position}".')
    if target_rotation not in ("z", "parallel", "xyz"):
        raise ValueError(f'Unknown target_rotation option "{target_rotation}".')
    offset = np.zeros((n, 3))
    ang, axis, pidx = np.zeros(n), np.tile(np.array([0.0, 0.0, 1.0]), (n, 1)), np.zeros(n, np.int64)
    for k, r in enumerate(np_randoms):
        if target_position == "random":
            offset[k] = r.uniform(TARGET_POSITION_RANGE[:, 0], TARGET_POSITION_RANGE[:, 1])
        a_, x_, p_ = _draw_rotation([r], target_rotation, len(pquats) if pquats is not None else 0)
        ang[k], axis[k], pidx[k] = a_[0], x_[0], p_[0]
    target_pos = object_qpos[:, :3] + offset
    target_quat = _quat_from_angle_and_axis_batch(ang, axis)
    if target_rotation == "parallel":
        target_quat = quat_mul(target_quat, np.asarray(pquats)[pidx])
    target_quat = target_quat / np.linalg.norm(target_quat, axis=-1, keepdims=True)
    return np.concatenate([target_pos, target_quat], axis=-1)


def block_goal_distance(goal_a, goal_b, target_position, target_rotation, ignore_z=False):
    """manipulate.py:87-118; ignore_z (pen variants): quat_a takes the z Euler angle of quat_b before the comparison."""
    goal_a, goal_b = np.asarray(goal_a, dtype=np.float64), np.asarray(goal_b, dtype=np.float64)
    d_pos = np.zeros_like(goal_a[..., 0])
    d_rot = np.zeros_like(goal_b[..., 0])
    if target_position != "ignore":
        d_pos = np.linalg.norm(goal_a[..., :3] - goal_b[..., :3], axis=-1)
    if target_rotation != "ignore":
        quat_a = goal_a[..., 3:]
        if ignore_z:
            flat_a, flat_b = quat_a.reshape(-1, 4), goal_b[..., 3:].reshape(-1, 4)
            fixed = np.empty_like(flat_a)
            for k in range(len(flat_a)):
                ea, eb = quat2euler(flat_a[k]), quat2euler(flat_b[k])
                ea[2] = eb[2]
                fixed[k] = euler2quat(ea)
            quat_a = fixed.reshape(quat_a.shape)
        w = quat_mul(quat_a, quat_conj(goal_b[..., 3:]))[..., 0]
        d_rot = 2 * np.arccos(np.clip(w, -1.0, 1.0))
    return d_pos, d_rot


def block_is_success(achieved, desired, target_position, target_rotation, ignore_z=False, distance_threshold=DISTANCE_THRESHOLD):
    d_pos, d_rot = block_goal_distance(achieved, desired, target_position, target_rotation, ignore_z)
    return ((d_pos < distance_threshold) & (d_rot < ROTATION_THRESHOLD)).astype(np.float32)


def block_reward(achieved, desired, target_position, target_rotation, reward_type, ignore_z=False, distance_threshold=DISTANCE_THRESHOLD):
    """manipulate.py:120-128."""
    if reward_type == "sparse":
        return block_is_success(achieved, desired, target_position, target_rotation, ignore_z, distance_threshold) - 1.0
    d_pos, d_rot = block_goal_distance(achieved, desired, target_position, target_rotation, ignore_z)
    return -(10.0 * d_pos + d_rot)


# ---- C struct (GrxHandTask with kind = 1, csrc/grx_hand_task.h) ------------------------------------------------------------
def make_block_task(model, target_position, target_rotation, reward_type, touch_get_obs="off", obj="block"):
    from .hand_spec import HandTaskStruct

    t = HandTaskStruct()
    t.n_substeps, t.sparse_reward = N_SUBSTEPS, int(reward_type == "sparse")
    t.palm_body = int(model.names["body"]["robot0:palm"])
    t.distance_threshold, t.rotation_threshold = OBJECTS[obj]["distance_threshold"], ROTATION_THRESHOLD   # the env may override the former
    t.ignore_z = int(OBJECTS[obj]["ignore_z_target_rotation"])
    j = int(model.names["joint"]["object:joint"])
    t.kind, t.nq_robot = 1, 24
    t.obj_qadr = int(np.asarray(model.tables["jnt_qposadr"]).reshape(-1)[j])
    t.obj_dadr = int(np.asarray(model.tables["jnt_dofadr"]).reshape(-1)[j])
    t.ignore_position, t.ignore_rotation = int(target_position == "ignore"), int(target_rotation == "ignore")
    t.touch_mode = TOUCH_MODES[touch_get_obs]
    if t.touch_mode and len(model.tables["touch_body"]) != N_TOUCH:
        raise ValueError("touch observations need the model compiled with its 92 touch zones")
    robot = [n for n in model.names["joint"] if n.startswith("robot")]
    assert len(robot) == t.nq_robot and max(model.names["joint"][n] for n in robot) == t.nq_robot - 1  # robot joints come first
    return t
