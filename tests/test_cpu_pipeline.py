"""PipelinedVecEnv (gymnasium_robotics_amd/pipeline.py) host logic with stand-in stage environments: world identity of the stages (seed offsets, slices), argument checks,
the lazy export.  The device behaviour (stages == the plain environment, world by world) is tests/test_gpu_pipeline.py."""
import numpy as np
import pytest


class _Stage:
    def __init__(self, env_id, num_envs, device=None, seed_offset=0, **kw):
        self.env_id, self.num_envs, self.device, self.seed_offset, self.kw = env_id, num_envs, device or "cpu", seed_offset, kw
        self.max_episode_steps, self.steps, self.closed = 50, 0, False
        self.single_action_space = self.single_observation_space = None

    def reset(self, *, seed=None, options=None):
        # (what every family does: a scalar seed fans out over the stage's worlds with the stage's offset, a sequence is one seed per world of THIS environment)
        self.seeds = None if seed is None else ([seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed))
        self.options = options
        assert self.seeds is None or len(self.seeds) == self.num_envs
        return np.asarray(self.seeds, dtype=np.float64), {}

    def step(self, a):
        self.steps += 1
        return np.asarray(a) + self.seed_offset, None, None, None, {"stage_steps": self.steps}

    def close(self):
        self.closed = True

    def get_state(self):
        return {"steps": self.steps, "off": self.seed_offset}

    def set_state(self, st):
        assert st["off"] == self.seed_offset
        self.steps = st["steps"]


def test_stages_are_the_plain_environments_worlds():
    import gymnasium_robotics_amd as grx

    pe = grx.PipelinedVecEnv("FetchPickAndPlace-v4", 12, stages=3, device="cpu", make_stage=_Stage, seed_offset=100, output="torch")
    assert pe.num_stages == 3 and pe.stage_size == 4 and [e.seed_offset for e in pe.stage_envs] == [100, 104, 108]
    assert all(e.kw == {"output": "torch"} and e.num_envs == 4 for e in pe.stage_envs)
    outs = pe.reset(seed=7)
    seeds = np.concatenate([o for o, _ in outs])
    assert np.array_equal(seeds, 7 + 100 + np.arange(12))      # world i of the whole batch gets seed + offset + i, as one plain environment of 12 worlds would give it
    assert [pe.world_slice(k) for k in range(3)] == [slice(0, 4), slice(4, 8), slice(8, 12)]
    assert pe.stream(1) is None                                  # CPU stand-ins: no streams, `on` is a no-op context
    with pe.on(2):
        o, *_ , info = pe.step_stage(2, np.zeros(4))
    assert np.array_equal(o, np.full(4, 108.0)) and info["stage_steps"] == 1 and [e.steps for e in pe.stage_envs] == [0, 0, 1]
    ck = pe.get_state()
    pe.step_stage(0, np.zeros(4)); pe.step_stage(2, np.zeros(4))
    pe.set_state(ck)
    assert [e.steps for e in pe.stage_envs] == [0, 0, 1]
    other = grx.PipelinedVecEnv("FetchPickAndPlace-v4", 12, stages=2, device="cpu", make_stage=_Stage)
    with pytest.raises(ValueError, match="does not fit"):
        other.set_state(ck)
    pe.synchronize(); pe.close()
    assert all(e.closed for e in pe.stage_envs)


def test_per_world_seeds_and_options_are_sliced_per_stage():
    """a sequence of seeds (one per world of the whole batch) and per-world option rows reach the stage that holds the world; shared option values reach every stage"""
    import gymnasium_robotics_amd as grx

    pe = grx.PipelinedVecEnv("AdroitHandHammer-v2", 12, stages=3, device="cpu", make_stage=_Stage)
    seeds = [1000 + 7 * i for i in range(12)]
    rows = np.arange(12 * 5, dtype=np.float64).reshape(12, 5)
    outs = pe.reset(seed=seeds, options={"goal_cell": np.array([3, 4]), "initial_state_dict": {"qpos": rows, "board_pos": rows[:, :3].tolist()}, "flag": True})
    assert np.array_equal(np.concatenate([o for o, _ in outs]), np.asarray(seeds, dtype=np.float64))      # no stage saw another stage's seeds
    for k, e in enumerate(pe.stage_envs):
        sl = pe.world_slice(k)
        assert np.array_equal(e.options["goal_cell"], [3, 4]) and e.options["flag"] is True
        assert np.array_equal(e.options["initial_state_dict"]["qpos"], rows[sl]) and np.array_equal(e.options["initial_state_dict"]["board_pos"], rows[sl, :3])
    with pytest.raises(ValueError, match="one seed per world"):
        pe.reset(seed=[1, 2, 3])
    a12 = grx.PipelinedVecEnv("PointMaze_UMaze-v3", 2, stages=2, device="cpu", make_stage=_Stage)      # a 2-vector option with num_envs == 2 is still a shared cell, not per-world rows
    a12.reset(seed=None, options={"goal_cell": np.array([1, 1])})
    assert all(np.array_equal(e.options["goal_cell"], [1, 1]) for e in a12.stage_envs)


def test_argument_checks():
    from gymnasium_robotics_amd.pipeline import PipelinedVecEnv

    with pytest.raises(ValueError, match="multiple of stages"):
        PipelinedVecEnv("FetchReach-v4", 10, stages=4, make_stage=_Stage)
    with pytest.raises(ValueError):
        PipelinedVecEnv("FetchReach-v4", 8, stages=0, make_stage=_Stage)
    import gymnasium_robotics_amd as grx

    with pytest.raises(AttributeError):
        grx.no_such_name
