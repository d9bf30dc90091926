"""Vectors produced by EXECUTING the reference's own host-side task code (goal / reset sampling, rewards, success flags) in the
build container, where neither gymnasium nor mujoco is installed:

    python tools/make_reference_host_vectors.py   ->  tests/golden/ref_host_logic.npz

The reference modules are imported from /root/reference under two stand-ins: a minimal `gymnasium` module (only the names the
imports touch: error, logger, spaces, Env, utils.ezpickle.EzPickle) and a `gymnasium_robotics` package object whose __path__
points at the reference tree, so that its __init__ (which registers environments and imports pettingzoo) is not executed.
`import mujoco` fails inside the reference's own try/except (robot_env.py:19-26).  No environment is constructed (that needs
MuJoCo); the reference METHODS are called unbound on a plain namespace that carries the attributes they read, with recording
doubles for `self._utils` / `self._mujoco`.  What is pinned is therefore exactly the reference's arithmetic and its order of
np_random draws -- the things this repo restates on the host (envs/fetch.py, envs/hand_spec.py, envs/manipulate_spec.py).
"""
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_host_logic.npz")


def install_stand_ins():
    gym = types.ModuleType("gymnasium")

    class _Error(Exception):
        pass

    err = types.ModuleType("gymnasium.error")
    err.Error = _Error
    err.DependencyNotInstalled = type("DependencyNotInstalled", (_Error,), {})
    logger = types.ModuleType("gymnasium.logger")
    logger.warn = logger.info = logger.error = lambda *a, **k: None
    spaces = types.ModuleType("gymnasium.spaces")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    spaces.Box = spaces.Dict = spaces.Space = _Space
    utils = types.ModuleType("gymnasium.utils")
    ez = types.ModuleType("gymnasium.utils.ezpickle")

    class EzPickle:
        def __init__(self, *a, **k):
            pass

    ez.EzPickle = EzPickle
    utils.ezpickle, utils.EzPickle = ez, EzPickle
    gym.Env = type("Env", (), {})
    gym.error, gym.logger, gym.spaces, gym.utils = err, logger, spaces, utils
    for name, mod in (("gymnasium", gym), ("gymnasium.error", err), ("gymnasium.logger", logger), ("gymnasium.spaces", spaces),
                      ("gymnasium.utils", utils), ("gymnasium.utils.ezpickle", ez)):
        sys.modules[name] = mod
    pkg = types.ModuleType("gymnasium_robotics")
    pkg.__path__ = [os.path.join(REF_ROOT, "gymnasium_robotics")]
    sys.modules["gymnasium_robotics"] = pkg


class Recorder:
    """Stands in for self._utils / self._mujoco: answers the queries from a table, records what is written."""

    def __init__(self, answers=None):
        self.answers, self.calls = dict(answers or {}), []

    def __getattr__(self, name):
        def call(*args, **kw):
            self.calls.append((name, args))
            if name in self.answers:
                a = self.answers[name]
                return a(*args) if callable(a) else np.array(a, dtype=np.float64).copy()
            return None
        return call


def rng(seed):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))     # gymnasium.utils.seeding.np_random(seed)


def main():
    install_stand_ins()
    from gymnasium_robotics.envs.fetch import fetch_env
    from gymnasium_robotics.envs.shadow_dexterous_hand import manipulate, reach

    out = {}
    # ------------------------------------------------------------------ Fetch (fetch_env.py:74-80,153-170,375-402)
    F = fetch_env.MujocoFetchEnv
    g0 = np.array([1.3419, 0.7491, 0.555])                                      # the documented settled gripper position
    cases = {"reach": dict(has_object=False, target_in_the_air=True, target_offset=0.0, obj_range=0.15, target_range=0.15, height_offset=0.0),
             "push": dict(has_object=True, target_in_the_air=False, target_offset=0.0, obj_range=0.15, target_range=0.15, height_offset=0.42473),
             "slide": dict(has_object=True, target_in_the_air=False, target_offset=np.array([0.4, 0.0, 0.0]), obj_range=0.1, target_range=0.3, height_offset=0.41401),
             "pick": dict(has_object=True, target_in_the_air=True, target_offset=0.0, obj_range=0.15, target_range=0.15, height_offset=0.42473)}
    for name, cfg in cases.items():
        obj_xy, goals = [], []
        for seed in range(64):
            ns = types.SimpleNamespace(np_random=rng(seed), initial_gripper_xpos=g0.copy(), initial_time=0.0, initial_qpos=np.zeros(3), initial_qvel=np.zeros(3),
                                       data=types.SimpleNamespace(time=0.0, qpos=np.zeros(3), qvel=np.zeros(3), act=np.zeros(0)), model=types.SimpleNamespace(na=0),
                                       _mujoco=Recorder(), _utils=Recorder({"get_joint_qpos": [1.25, 0.53, 0.4, 1, 0, 0, 0]}), **cfg)
            assert F._reset_sim(ns) is True                                      # the reference's reset: draws the object position
            sets = [a for n_, a in ns._utils.calls if n_ == "set_joint_qpos"]
            obj_xy.append(sets[0][3][:2] if cfg["has_object"] else [np.nan, np.nan])
            goals.append(F._sample_goal(ns))                                     # ... then the goal, from the same stream (robot_env.py:176-183)
        out[f"fetch_{name}_object_xy"], out[f"fetch_{name}_goal"] = np.array(obj_xy, dtype=np.float64), np.array(goals)
    r = np.random.default_rng(7)
    ag, dg = r.uniform(-0.2, 0.2, (5, 33, 3)), r.uniform(-0.2, 0.2, (5, 33, 3))
    dg[0, :5] = ag[0, :5] + np.array([0.05, 0, 0])                               # distance == threshold up to rounding: strict comparisons
    for rt in ("sparse", "dense"):
        ns = types.SimpleNamespace(reward_type=rt, distance_threshold=0.05)
        out[f"fetch_reward_{rt}"] = F.compute_reward(ns, ag, dg, {})
    out["fetch_success"], out["fetch_ag"], out["fetch_dg"] = F._is_success(types.SimpleNamespace(distance_threshold=0.05), ag, dg), ag, dg
    out["fetch_gripper_xpos"] = g0

    # ------------------------------------------------------------------ HandReach (reach.py:92-134)
    R = reach.MujocoHandReachEnv
    tips = r.uniform(-0.05, 0.05, (5, 3)) + np.array([1.0, 0.87, 0.2])
    names = ["robot0:S_fftip", "robot0:S_mftip", "robot0:S_rftip", "robot0:S_lftip", "robot0:S_thtip"]
    init_goal, palm = tips.ravel().copy(), np.array([1.0, 0.9, 0.15])
    goals = []
    for seed in range(64):
        ns = types.SimpleNamespace(np_random=rng(seed), initial_goal=init_goal.copy(), palm_xpos=palm.copy())
        goals.append(R._sample_goal(ns))
    out["hand_reach_goal"], out["hand_reach_initial_goal"], out["hand_reach_palm"] = np.array(goals), init_goal, palm
    a15, b15 = r.uniform(-0.02, 0.02, (4, 17, 15)), r.uniform(-0.02, 0.02, (4, 17, 15))
    for rt in ("sparse", "dense"):
        out[f"hand_reach_reward_{rt}"] = R.compute_reward(types.SimpleNamespace(reward_type=rt, distance_threshold=0.01), a15, b15, {})
    out["hand_reach_success"], out["hand_reach_ag"], out["hand_reach_dg"] = R._is_success(types.SimpleNamespace(distance_threshold=0.01), a15, b15), a15, b15

    # ------------------------------------------------------------------ HandManipulate (manipulate.py:87-142,154-279)
    M = manipulate.MujocoManipulateEnv
    obj0 = np.array([1.0, 0.87, 0.2, 1.0, 0.0, 0.0, 0.0])
    tpr = np.array([(-0.04, 0.04), (-0.06, 0.02), (0.0, 0.06)])
    # the registered variants (gymnasium_robotics/__init__.py:124-800); target_rotation "ignore" / "fixed" reads self.data.get_joint_qpos, a
    # mujoco_py-only call: the reference itself cannot run those options on the mujoco bindings
    variants = [("ignore", "z"), ("ignore", "parallel"), ("ignore", "xyz"), ("random", "xyz")]
    from gymnasium_robotics.utils import rotations
    for tp, tr in variants:
        resets, goals = [], []
        for seed in range(48):
            utils = Recorder({"get_joint_qpos": obj0, "get_site_xpos": [1.0, 0.87, 0.2]})
            ns = types.SimpleNamespace(np_random=rng(seed), target_position=tp, target_rotation=tr, target_position_range=tpr, parallel_quats=[rotations.euler2quat(q) for q in rotations.get_parallel_rotations()],
                                       randomize_initial_rotation=True, randomize_initial_position=True, initial_time=0.0, initial_qpos=np.zeros(3), initial_qvel=np.zeros(3),
                                       data=types.SimpleNamespace(time=0.0, qpos=np.zeros(3), qvel=np.zeros(3), act=np.zeros(0), site_xpos=np.array([[1.0, 0.87, 0.2]])),
                                       model=types.SimpleNamespace(na=0), _mujoco=Recorder(), _utils=utils,
                                       _model_names=types.SimpleNamespace(_site_name2id={"object:center": 0}), _set_action=lambda a: None, n_substeps=20,
                                       _is_on_palm=lambda: True)
            ns._set_action = lambda a: None
            ok = M._reset_sim(ns)                                               # pose randomisation (+ the settle loop, which only steps the doubles)
            assert ok
            sets = [a for n_, a in utils.calls if n_ == "set_joint_qpos"]
            resets.append(np.asarray(sets[0][3], dtype=np.float64))
            goals.append(M._sample_goal(ns))
        out[f"manip_{tp}_{tr}_reset_pose"], out[f"manip_{tp}_{tr}_goal"] = np.array(resets), np.array(goals)
    # one batch dimension: the reference's rotations.quat_mul asserts on inputs with more (rotations.py:283-303)
    qa = r.normal(size=(63, 4)); qa /= np.linalg.norm(qa, axis=-1, keepdims=True)
    qb = r.normal(size=(63, 4)); qb /= np.linalg.norm(qb, axis=-1, keepdims=True)
    qb[:6] = qa[:6]; qb[6:12] = -qa[6:12]                                      # identical and antipodal orientations
    pa, pb = r.uniform(-0.02, 0.02, (63, 3)), r.uniform(-0.02, 0.02, (63, 3))
    pb[:20] = pa[:20] + r.uniform(-0.004, 0.004, (20, 3))                       # some pairs inside the 1 cm threshold
    ga, gb = np.concatenate([pa, qa], -1), np.concatenate([pb, qb], -1)
    out["manip_ga"], out["manip_gb"] = ga, gb
    for tp, tr in (("ignore", "xyz"), ("random", "xyz"), ("random", "ignore")):   # the distance itself supports every combination
        for rt in ("sparse", "dense"):
            ns = types.SimpleNamespace(target_position=tp, target_rotation=tr, ignore_z_target_rotation=False, reward_type=rt, distance_threshold=0.01, rotation_threshold=0.1)
            ns._goal_distance = lambda a, b, ns=ns: M._goal_distance(ns, a, b)
            ns._is_success = lambda a, b, ns=ns: M._is_success(ns, a, b)
            out[f"manip_{tp}_{tr}_reward_{rt}"] = M.compute_reward(ns, ga, gb, {})
        out[f"manip_{tp}_{tr}_success"] = M._is_success(ns, ga, gb)
        dp, dr = M._goal_distance(ns, ga, gb)
        out[f"manip_{tp}_{tr}_dpos"], out[f"manip_{tp}_{tr}_drot"] = dp, dr
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
