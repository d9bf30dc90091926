"""gymnasium_robotics_amd: the env.step() hot path of Gymnasium-Robotics on MI355X.

The reference registers its ids with gymnasium (gymnasium_robotics/__init__.py:12-1201, ``register_robotics_envs``) and a user
builds N environments with ``gymnasium.make_vec(id, num_envs=N)``.  This package has no gymnasium dependency; the same ids map to
the batched device environments here:

    import gymnasium_robotics_amd as grx
    envs = grx.make_vec("FetchPickAndPlace-v4", num_envs=4096)          # one world per 64-lane wavefront on cuda:0
    obs, info = envs.reset(seed=0)
    obs, reward, terminated, truncated, info = envs.step(actions)        # VectorEnv contract

``registered_env_ids()`` lists every id this build serves: all 121 ids the reference registers (Fetch, Shadow hand, Adroit hand, Maze, FrankaKitchen).
``UnsupportedEnvError`` is what an id listed in ``_NOT_SERVED`` would raise (an extension hook: the table has been empty since round 2).
Nothing here imports torch or loads the HIP library until an environment is constructed.
"""
from typing import List

__all__ = ["make_vec", "registered_env_ids", "env_family", "UnsupportedEnvError", "PipelinedVecEnv"]


class UnsupportedEnvError(KeyError):
    """The reference registers this id, the device engine does not serve it (DESIGN.md section 8 lists what is out of scope)."""


_NOT_SERVED = {}   # extension hook: reference id prefix -> why the engine does not serve it.  Empty since round 2 (FrankaKitchen was the last entry); env_family() consults it first


def _fetch_ids() -> List[str]:
    # gymnasium_robotics/__init__.py:26-80: sparse + Dense twins, versions v1..v4 share the constructor arguments; v4 = mujoco bindings
    from .envs.fetch_spec import FETCH_TASKS

    return [f"{t}{sfx}-v4" for t in FETCH_TASKS for sfx in ("", "Dense")]


def _hand_reach_ids() -> List[str]:
    return ["HandReach-v3", "HandReachDense-v3"]   # __init__.py:82-121


def _hand_manipulate_ids() -> List[str]:
    # __init__.py:124-341 (block), 644-800 (pen): base ids, Dense twins, and the two touch-sensor twins of every non-*Full id
    from .envs.manipulate_spec import BLOCK_VARIANTS, NO_TOUCH_IDS

    ids = []
    for base in BLOCK_VARIANTS:
        touches = [""] if base in NO_TOUCH_IDS else ["", "_BooleanTouchSensors", "_ContinuousTouchSensors"]
        ids += [f"{base}{t}{sfx}-v1" for t in touches for sfx in ("", "Dense")]
    return ids


def _maze_ids() -> List[str]:
    # __init__.py:839-1078: PointMaze_<map>[Dense]-v3, AntMaze_<map>[Dense]-v5
    from .envs.maze_spec import MAPS

    return [f"{agent}_{m}{sfx}-{ver}" for agent, ver in (("PointMaze", "v3"), ("AntMaze", "v5")) for m in MAPS for sfx in ("", "Dense")]


def _adroit_ids() -> List[str]:
    return [f"AdroitHand{task}{sfx}-{ver}" for task in ("Door", "Hammer", "Pen", "Relocate") for sfx in ("", "Sparse") for ver in ("v1", "v2")]   # __init__.py:1078-1115 (v1: same class, deprecated alias)


def registered_env_ids() -> List[str]:
    """Every id ``make_vec`` serves, in the reference's registration order of families."""
    return _fetch_ids() + _hand_reach_ids() + _hand_manipulate_ids() + _maze_ids() + _adroit_ids() + ["FrankaKitchen-v1"]   # __init__.py:1117-1122


def env_family(env_id: str) -> str:
    """'fetch' | 'hand_reach' | 'hand_manipulate' | 'adroit' | 'kitchen' | 'point_maze' | 'ant_maze' for a served id; raises for the rest."""
    for prefix, why in _NOT_SERVED.items():
        if env_id.startswith(prefix):
            raise UnsupportedEnvError(f"{env_id}: not served by this build -- {why}")
    if env_id not in registered_env_ids():
        raise KeyError(f"unknown env id {env_id!r}; see gymnasium_robotics_amd.registered_env_ids()")
    if env_id.startswith("Fetch"):
        return "fetch"
    if env_id.startswith("HandReach"):
        return "hand_reach"
    if env_id.startswith("HandManipulate"):
        return "hand_manipulate"
    if env_id.startswith("AdroitHand"):
        return "adroit"
    if env_id.startswith("FrankaKitchen"):
        return "kitchen"
    return "point_maze" if env_id.startswith("PointMaze") else "ant_maze"


def make_vec(env_id: str, num_envs: int = 1, **kwargs):
    """The batched device environment for a reference id (stands in for ``gymnasium.make_vec(env_id, num_envs=...)``).
    Keyword arguments go to the environment class (device, autoreset_mode, max_episode_steps, seed_offset, and the
    constructor arguments the reference's classes take: reward_type is implied by the id, distance_threshold,
    continuing_task / reset_target for the mazes, ...)."""
    family = env_family(env_id)
    if family == "fetch":
        from .envs.fetch import FetchVecEnv as cls
    elif family == "hand_reach":
        from .envs.hand import HandReachVecEnv as cls
    elif family == "hand_manipulate":
        from .envs.hand import HandBlockVecEnv as cls
    elif family == "adroit":
        from .envs.adroit import AdroitVecEnv as cls
    elif family == "kitchen":
        from .envs.kitchen import KitchenVecEnv as cls
    elif family == "point_maze":
        from .envs.point_maze import PointMazeVecEnv as cls
    else:
        from .envs.point_maze import AntMazeVecEnv as cls
    return cls(env_id, num_envs=num_envs, **kwargs)


def __getattr__(name):      # (lazy: the pipeline module is only needed by callers that step sub-batches out of phase)
    if name == "PipelinedVecEnv":
        from .pipeline import PipelinedVecEnv

        return PipelinedVecEnv
    raise AttributeError(name)
