"""gymnasium façade (gymnasium_robotics_amd/gym_compat.py) driven with a recording stand-in for gymnasium (not installed in the build image):
every served id is registered with the reference's TimeLimit and entry points that resolve; the space conversion keeps bounds, shapes and dtypes."""
import importlib
import sys
import types

import numpy as np
import pytest


@pytest.fixture
def fake_gymnasium(monkeypatch):
    calls = []
    gym = types.ModuleType("gymnasium")
    gym.register = lambda **kw: calls.append(kw)
    gym.Env = type("Env", (), {"reset": lambda self, seed=None, options=None: None})
    sp = types.ModuleType("gymnasium.spaces")

    class Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Dict(dict):
        pass
    sp.Box, sp.Dict = Box, Dict
    vec = types.ModuleType("gymnasium.vector")
    vec.VectorEnv = type("VectorEnv", (), {})
    vec.AutoresetMode = types.SimpleNamespace(NEXT_STEP="NextStep", SAME_STEP="SameStep", DISABLED="Disabled")
    vec.utils = types.SimpleNamespace(batch_space=lambda space, n: ("batched", space, n))
    gym.spaces, gym.vector = sp, vec
    for name, mod in (("gymnasium", gym), ("gymnasium.spaces", sp), ("gymnasium.vector", vec)):
        monkeypatch.setitem(sys.modules, name, mod)
    return gym, calls


def test_every_served_id_is_registered_with_the_reference_time_limit(fake_gymnasium):
    gym, calls = fake_gymnasium
    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd import gym_compat

    ids = gym_compat.register_envs()
    assert len(ids) == len(calls) == len(grx.registered_env_ids()) and all(i.startswith("grx/") for i in ids)
    by_id = {c["id"]: c for c in calls}
    # the reference's register() calls (gymnasium_robotics/__init__.py): max_episode_steps per family
    want = {"grx/FetchPickAndPlace-v4": 50, "grx/HandReachDense-v3": 50, "grx/HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1": 100,
            "grx/AdroitHandRelocateSparse-v2": 200, "grx/PointMaze_UMaze-v3": 300, "grx/PointMaze_Large_Diverse_GR-v3": 800, "grx/AntMaze_UMaze-v5": 700,
            "grx/AntMaze_Large_Diverse_GR-v5": 1000, "grx/AntMaze_Medium-v5": 1000, "grx/PointMaze_MediumDense-v3": 600, "grx/FrankaKitchen-v1": 280}
    for gid, limit in want.items():
        assert by_id[gid]["max_episode_steps"] == limit, gid
        assert by_id[gid]["kwargs"] == {"env_id": gid[4:]}
    for c in calls[:3]:
        for key in ("entry_point", "vector_entry_point"):
            mod, fn = c[key].split(":")
            assert callable(getattr(importlib.import_module(mod), fn))
    assert gym_compat.register_envs(namespace=None)[0] == grx.registered_env_ids()[0]


def test_space_conversion_and_adapter_classes(fake_gymnasium):
    gym, _ = fake_gymnasium
    from gymnasium_robotics_amd import gym_compat
    from gymnasium_robotics_amd.spaces import Box, Dict

    d = Dict(dict(observation=Box(-np.inf, np.inf, (25,), np.float64), achieved_goal=Box(-np.inf, np.inf, (3,), np.float64)))
    g = gym_compat._to_gym_space(d, gym)
    assert isinstance(g, gym.spaces.Dict) and g["observation"].shape == (25,) and g["observation"].dtype == np.float64 and np.isneginf(g["achieved_goal"].low).all()
    a = gym_compat._to_gym_space(Box(-1.0, 1.0, (4,), np.float32), gym)
    assert a.dtype == np.float32 and a.low.min() == -1.0 and a.high.max() == 1.0
    assert issubclass(gym_compat.make_vector_env_class(), gym.vector.VectorEnv) and issubclass(gym_compat.make_single_env_class(), gym.Env)
    assert gym_compat._autoreset_mode(gym, "same_step") == "SameStep"
    with pytest.raises(Exception):   # no GPU here: constructing an env must fail loudly (no CPU fallback), after the render_mode check
        gym_compat.vector_entry_point("FetchReach-v4", num_envs=2)
    with pytest.raises(ValueError, match="do not render"):
        gym_compat.vector_entry_point("FetchReach-v4", num_envs=2, render_mode="human")


def test_missing_gymnasium_feature_is_reported(fake_gymnasium):
    gym, _ = fake_gymnasium
    from gymnasium_robotics_amd import gym_compat

    del gym.vector
    with pytest.raises(ImportError, match="vector"):
        gym_compat.register_envs()
