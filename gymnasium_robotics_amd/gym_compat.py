"""Gymnasium façade: register the accelerated ids with a real ``gymnasium`` when one is importable.

The reference registers every id through ``gymnasium.register`` (gymnasium_robotics/__init__.py:12-1201, ``register_robotics_envs``) and the user
builds batches with ``gymnasium.make_vec(id, num_envs=N)``.  This module does the same for the device environments:

    import gymnasium, gymnasium_robotics_amd.gym_compat as grx_gym
    grx_gym.register_envs()                                    # ids get the "grx/" namespace unless namespace=None (then they shadow the reference's)
    envs = gymnasium.make_vec("grx/FetchPickAndPlace-v4", num_envs=4096)          # -> vector_entry_point below
    obs, info = envs.reset(seed=0); obs, r, term, trunc, info = envs.step(actions)

gymnasium is NOT installed in the build image: nothing here is imported by the package, the tests drive it with a recording stand-in
(tests/test_cpu_gym_compat.py), and every gymnasium attribute used is listed in ``_need`` so that a version without it fails at registration.
``GymVectorEnv`` adapts a device environment to ``gymnasium.vector.VectorEnv``: gymnasium space objects, ``metadata["autoreset_mode"]``,
``render_mode`` / ``spec`` attributes, numpy in / numpy out (``output="numpy"``), ``close``; ``single_entry_point`` gives ``gymnasium.make(id)`` a
one-world ``gymnasium.Env`` view with the reference's un-batched shapes so that the reference's own tests (tests/test_envs.py: ``check_env``,
determinism, pickling) can run against this back-end.
"""
import numpy as np

_need = ("register", "Env", "spaces", "vector")


def _gym():
    import gymnasium

    missing = [n for n in _need if not hasattr(gymnasium, n)]
    if missing:
        raise ImportError(f"gymnasium lacks {missing}: version >= 1.0 is required (the reference pins gymnasium>=1.2.0, pyproject.toml:29)")
    return gymnasium


def _to_gym_space(space, gym):
    """gymnasium_robotics_amd.spaces.Box / Dict -> gymnasium.spaces.Box / Dict"""
    from .spaces import Dict

    if isinstance(space, Dict):
        return gym.spaces.Dict({k: _to_gym_space(v, gym) for k, v in space.items()})
    return gym.spaces.Box(low=space.low, high=space.high, shape=space.shape, dtype=space.dtype)


def _autoreset_mode(gym, mode):
    enum = getattr(gym.vector, "AutoresetMode", None)       # gymnasium >= 1.1
    return mode if enum is None else {"next_step": enum.NEXT_STEP, "same_step": enum.SAME_STEP, "disabled": enum.DISABLED}[mode]


def make_vector_env_class():
    """Built lazily so that importing this module does not import gymnasium."""
    gym = _gym()

    class GymVectorEnv(gym.vector.VectorEnv):
        """gymnasium.vector.VectorEnv over one batched device environment (N worlds = one kernel launch per step)"""

        def __init__(self, env_id, num_envs=1, render_mode=None, **kwargs):
            from . import make_vec

            if render_mode is not None:
                raise ValueError("the device environments do not render (the reference renders through MuJoCo's OpenGL context, robot_env.py:320-338)")
            kwargs.setdefault("autoreset_mode", "next_step")
            self.env = make_vec(env_id, num_envs=num_envs, output="numpy", **kwargs)
            self.num_envs, self.render_mode = self.env.num_envs, None
            self.single_observation_space = _to_gym_space(self.env.single_observation_space, gym)
            self.single_action_space = _to_gym_space(self.env.single_action_space, gym)
            self.observation_space = gym.vector.utils.batch_space(self.single_observation_space, self.num_envs)
            self.action_space = gym.vector.utils.batch_space(self.single_action_space, self.num_envs)
            self.metadata = {"render_modes": [], "autoreset_mode": _autoreset_mode(gym, self.env.autoreset_mode)}

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, actions):
            obs, reward, terminated, truncated, info = self.env.step(np.asarray(actions, dtype=np.float32))
            return obs, np.asarray(reward, dtype=np.float64), terminated, truncated, info

        # the multi-goal API of the reference's GoalEnv (core.py:45-114), batched
        def compute_reward(self, achieved_goal, desired_goal, info):
            return self.env.compute_reward(achieved_goal, desired_goal, info)

        def compute_terminated(self, achieved_goal, desired_goal, info):
            return self.env.compute_terminated(achieved_goal, desired_goal, info)

        def compute_truncated(self, achieved_goal, desired_goal, info):
            return self.env.compute_truncated(achieved_goal, desired_goal, info)

        def close_extras(self, **kwargs):
            self.env.close()

    return GymVectorEnv


def make_single_env_class():
    gym = _gym()

    class GymSingleEnv(gym.Env):
        """gymnasium.Env view of ONE world (num_envs = 1, autoreset disabled: gymnasium.make wraps it in TimeLimit itself), with the reference's
        un-batched shapes and scalar reward / flags -- what gymnasium.utils.env_checker.check_env and the reference's tests/test_envs.py expect."""

        metadata = {"render_modes": []}

        def __init__(self, env_id, render_mode=None, **kwargs):
            from . import make_vec

            if render_mode is not None:
                raise ValueError("the device environments do not render")
            self.env = make_vec(env_id, num_envs=1, output="numpy", autoreset_mode="disabled", max_episode_steps=None, **kwargs)
            self.render_mode = None
            self.observation_space = _to_gym_space(self.env.single_observation_space, gym)
            self.action_space = _to_gym_space(self.env.single_action_space, gym)

        @staticmethod
        def _first(x):
            return {k: v[0] for k, v in x.items()} if isinstance(x, dict) else x[0]

        def reset(self, *, seed=None, options=None):
            super().reset(seed=seed)
            obs, info = self.env.reset(seed=seed, options=options)
            return self._first(obs), {k: self._first(v) if hasattr(v, "__len__") and len(v) == 1 else v for k, v in info.items()}

        def step(self, action):
            obs, reward, terminated, truncated, info = self.env.step(np.asarray(action, dtype=np.float32)[None])
            info = {k: (v[0] if hasattr(v, "__len__") and len(v) == 1 else v) for k, v in info.items()}
            return self._first(obs), float(reward[0]), bool(terminated[0]), bool(truncated[0]), info

        def compute_reward(self, achieved_goal, desired_goal, info):
            return self.env.compute_reward(achieved_goal, desired_goal, info)

        def compute_terminated(self, achieved_goal, desired_goal, info):
            return self.env.compute_terminated(achieved_goal, desired_goal, info)

        def compute_truncated(self, achieved_goal, desired_goal, info):
            return self.env.compute_truncated(achieved_goal, desired_goal, info)

        def close(self):
            self.env.close()

    return GymSingleEnv


def vector_entry_point(env_id=None, num_envs=1, **kwargs):
    """``vector_entry_point`` of the registered specs: gymnasium.make_vec(id, num_envs=N, **kw) lands here (kwargs of the spec carry the bare id)."""
    return make_vector_env_class()(env_id, num_envs=num_envs, **kwargs)


def single_entry_point(env_id=None, **kwargs):
    return make_single_env_class()(env_id, **kwargs)


def max_episode_steps_of(env_id: str) -> int:
    """the reference's TimeLimit of every served id (gymnasium_robotics/__init__.py: max_episode_steps=... of each register call)"""
    from . import env_family
    from .envs import maze_spec

    family = env_family(env_id)
    if family in ("fetch", "hand_reach"):
        return 50                      # __init__.py:51,119
    if family == "hand_manipulate":
        return 100                     # __init__.py:141 ...
    if family == "adroit":
        return 200                     # __init__.py:1092-1113
    if family == "kitchen":
        return 280                     # __init__.py:1120
    return (maze_spec.parse_point_maze_id if family == "point_maze" else maze_spec.parse_ant_maze_id)(env_id)[2]


def register_envs(namespace="grx"):
    """gymnasium.register every served id.  namespace=None registers the bare ids (shadowing gymnasium_robotics' own registration)."""
    gym = _gym()
    from . import registered_env_ids

    done = []
    for env_id in registered_env_ids():
        gid = env_id if namespace is None else f"{namespace}/{env_id}"
        gym.register(id=gid, entry_point="gymnasium_robotics_amd.gym_compat:single_entry_point", vector_entry_point="gymnasium_robotics_amd.gym_compat:vector_entry_point",
                     max_episode_steps=max_episode_steps_of(env_id), kwargs={"env_id": env_id})
        done.append(gid)
    return done
