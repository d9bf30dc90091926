"""Out-of-phase sub-batches of one vector environment (EnvPool-style asynchronous stepping, on one GPU).

A step launch ends when its slowest world does: one wavefront steps one world, a world in contact-rich poses takes up to 2.6x the median, and
while those worlds finish most of the chip's wave slots are empty.  Nothing inside ONE `env.step()` can fill them -- the next step needs this one's observations --
but a second, independent sub-batch can: `PipelinedVecEnv` owns K vector environments of `num_envs / K` worlds each (world j of stage k is world
`k * num_envs / K + j` of the plain environment: same seeds, same episodes), every stage on its own HIP stream.  `step_stage(k, actions)` only ENQUEUES (no
`env.step()` of this package waits for the device), so a caller that walks the stages round-robin -- policy on stage k's observations, step stage k, next stage --
keeps K launches in flight that drift out of phase: the tail of one is filled by the body of the next.  Measured on MI355X, FetchPickAndPlace-v4, `env.step()`
only (tools/async_probe.py): 8 192 worlds 1.56 M -> 1.84 M env-steps/s with K = 2 (+18 %), 4 096 worlds 1.31 -> 1.39 M (+7 %).  K = 2 is the sweet spot: more
stages need more HIP hardware queues than the runtime's default four (`GPU_MAX_HW_QUEUES=16` in the environment BEFORE the first HIP call; unrelated streams that
share a hardware queue serialise) and gain nothing further.

The reference has no counterpart (its users reach for `gymnasium.vector.AsyncVectorEnv` [3P], one process per environment); results are the plain environment's,
world by world (tests/test_gpu_pipeline.py), because worlds never interact.

Stream rule (the usual one): whatever produces stage k's actions must run on, or be waited for by, stage k's stream -- `with env.on(k): a = policy(obs_k);
out = env.step_stage(k, a)`.  Tensors returned by `step_stage(k, ...)` are valid on stage k's stream until that stage's next step.
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional


class PipelinedVecEnv:
    def __init__(self, env_id: str, num_envs: int, stages: int = 2, device: Optional[str] = None, make_stage: Optional[Callable] = None, seed_offset: int = 0, **kwargs):
        if stages < 1 or num_envs % stages:
            raise ValueError(f"num_envs ({num_envs}) must be a positive multiple of stages ({stages})")
        if make_stage is None:
            from . import make_vec as make_stage
        self.env_id, self.num_envs, self.num_stages, self.stage_size = env_id, int(num_envs), int(stages), num_envs // stages
        self.stage_envs = [make_stage(env_id, num_envs=self.stage_size, device=device, seed_offset=seed_offset + k * self.stage_size, **kwargs) for k in range(stages)]
        self.device = getattr(self.stage_envs[0], "device", device)
        self._streams = None
        if str(self.device).startswith("cuda"):
            import torch

            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(stages)]
            for s in self._streams:      # the constructors uploaded models and zeroed buffers on the caller's stream
                s.wait_stream(torch.cuda.current_stream(self.device))
        e0 = self.stage_envs[0]
        self.single_action_space, self.single_observation_space = getattr(e0, "single_action_space", None), getattr(e0, "single_observation_space", None)
        self.max_episode_steps = getattr(e0, "max_episode_steps", None)

    # ------------------------------------------------------------------ stages
    def world_slice(self, k: int) -> slice:
        """the worlds of the plain environment that stage k holds"""
        return slice(k * self.stage_size, (k + 1) * self.stage_size)

    def stream(self, k: int):
        return None if self._streams is None else self._streams[k]

    def on(self, k: int):
        """context manager: stage k's stream is the current stream inside"""
        if self._streams is None:
            return contextlib.nullcontext()
        import torch

        return torch.cuda.stream(self._streams[k])

    def reset(self, *, seed=None, options=None):
        """every stage reset on its own stream; returns [(obs, info)] per stage.  A scalar seed gives world i of the WHOLE batch the seed `seed + i`, as the plain environment
        does (the stages carry their `seed_offset`); a SEQUENCE of seeds -- one per world of the whole batch -- and per-world entries of `options` (_slice_options) are sliced per stage, so that world i gets the i-th seed / option row whatever stage holds it."""
        import numpy as np

        if seed is not None and not np.isscalar(seed):
            seed = list(seed)
            if len(seed) != self.num_envs:
                raise ValueError(f"reset(seed=<sequence>) needs one seed per world: got {len(seed)}, expected {self.num_envs}")
        out = []
        for k, e in enumerate(self.stage_envs):
            sl = self.world_slice(k)
            sd = seed if (seed is None or np.isscalar(seed)) else seed[sl]
            opt = self._slice_options(options, sl)
            with self.on(k):
                out.append(e.reset(seed=sd, options=opt))
        return out

    def _slice_options(self, options, sl):
        """per-world entries of `options` -- arrays of two or more dimensions whose leading one is num_envs, e.g. the [N, nq] rows of Adroit's options["initial_state_dict"]
        (adroit_hammer.py:343-345) -- cut to the stage's worlds, nested dicts followed; scalars and vectors (the mazes' 'goal_cell' / 'reset_cell') are shared by every world"""
        import numpy as np

        if isinstance(options, dict):
            return {name: self._slice_options(v, sl) for name, v in options.items()}
        if isinstance(options, (list, tuple, np.ndarray)):
            a = np.asarray(options)
            if a.ndim >= 2 and a.shape[0] == self.num_envs:
                return a[sl]
        return options

    def step_stage(self, k: int, actions):
        """`env.step(actions)` of stage k, enqueued on its stream: (obs, reward, terminated, truncated, info) of its `stage_size` worlds"""
        with self.on(k):
            return self.stage_envs[k].step(actions)

    def get_state(self):
        """checkpoint of every stage (GoalVecEnv.get_state: everything that determines the future at a step boundary); synchronises"""
        out = []
        for k, e in enumerate(self.stage_envs):
            with self.on(k):
                out.append(e.get_state())
        return {"stages": out, "num_envs": self.num_envs, "env_id": self.env_id}

    def set_state(self, state):
        if state.get("num_envs") != self.num_envs or state.get("env_id") != self.env_id or len(state.get("stages", ())) != self.num_stages:
            raise ValueError(f"checkpoint of {state.get('env_id')!r} with {state.get('num_envs')} worlds in {len(state.get('stages', ()))} stages does not fit "
                             f"{self.env_id!r} with {self.num_envs} worlds in {self.num_stages} stages")
        for k, (e, st) in enumerate(zip(self.stage_envs, state["stages"])):
            with self.on(k):
                e.set_state(st)

    def synchronize(self):
        for s in self._streams or ():
            s.synchronize()

    def close(self):
        for e in self.stage_envs:
            if hasattr(e, "close"):
                e.close()
