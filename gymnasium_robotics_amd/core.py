"""Vectorised restatement of the multi-goal API (/root/reference/gymnasium_robotics/core.py:8-114).

``GoalVecEnv`` keeps the GoalEnv contract -- dict observations with the keys ``observation`` /
``achieved_goal`` / ``desired_goal`` (core.py:31-43) and the three ``compute_*`` methods that accept
arbitrary leading batch dimensions (core.py:45-114) -- with a leading ``num_envs`` axis on everything,
following the gymnasium.vector.VectorEnv conventions [3P].
"""
import numpy as np

from .spaces import Dict


class GoalVecEnv:
    """Constructor arguments are recorded so that the env pickles the way gymnasium's EzPickle [3P] envs do: by value of its
    constructor call (the reference's envs derive from EzPickle, e.g. shadow_dexterous_hand/reach.py:56,78-87; its tests
    tests/test_envs.py:164-178 and tests/envs/hand/test_reach.py pickle every env).  Device state is not serialised."""

    metadata = {"render_modes": []}

    def __new__(cls, *args, **kwargs):
        obj = super().__new__(cls)
        obj._ezpickle_args, obj._ezpickle_kwargs = args, kwargs
        return obj

    def __getstate__(self):
        return {"_ezpickle_args": self._ezpickle_args, "_ezpickle_kwargs": self._ezpickle_kwargs}

    def __setstate__(self, d):
        out = type(self)(*d["_ezpickle_args"], **d["_ezpickle_kwargs"])
        self.__dict__.update(out.__dict__)
        out.__dict__.pop("_h", None)   # the native handle now belongs to self: do not let the temporary destroy it

    num_envs: int
    single_observation_space: Dict
    observation_space: Dict
    REQUIRED_KEYS = ("observation", "achieved_goal", "desired_goal")

    def _check_goal_space(self):
        if not isinstance(self.single_observation_space, Dict):
            raise TypeError("GoalEnv requires an observation space of type Dict")
        for key in self.REQUIRED_KEYS:
            if key not in self.single_observation_space:
                raise KeyError(f'GoalEnv requires the "{key}" key to be part of the observation dictionary.')

    def compute_reward(self, achieved_goal, desired_goal, info):
        raise NotImplementedError

    def compute_terminated(self, achieved_goal, desired_goal, info):
        raise NotImplementedError

    def compute_truncated(self, achieved_goal, desired_goal, info):
        raise NotImplementedError

    # The step kernel is launched on the caller's stream and `step` returns without waiting for it when output="torch"; the split below is the
    # AsyncVectorEnv-style [3P] spelling of that: launch, do something else on the host (policy inference of the previous batch), collect.
    def step_async(self, actions):
        self._pending_step = self.step(actions)

    def step_wait(self):
        import torch

        if getattr(self, "_pending_step", None) is None:
            raise RuntimeError("step_wait() called without a pending step_async()")
        out, self._pending_step = self._pending_step, None
        torch.cuda.current_stream(self.device).synchronize()
        return out

    # `status` (int32 per world): low half = GRX_STATUS_* flags of the last launch, high half = the same flags OR-accumulated by the kernels
    # since clear_status() (include/grx_capi.h).  A capacity overflow drops contacts for one substep: training code should poll this.
    def status_counts(self):
        st = (self.status >> 16).cpu().numpy()
        return {"badnum": int((st & 1 != 0).sum()), "con_overflow": int((st & 2 != 0).sum()), "efc_overflow": int((st & 4 != 0).sum()),
                "factor": int((st & 8 != 0).sum()), "worlds": int(self.num_envs)}

    def clear_status(self):
        self.status.zero_()

    def _status_info(self, info):
        """info["status"] (this step's flags) and info["status_sticky"] for both output modes."""
        if self.output == "torch":      # two elementwise device ops, no sync
            info["status"], info["status_sticky"] = self.status & 0xFFFF, self.status >> 16
        else:
            st = self.status.cpu().numpy()
            info["status"], info["status_sticky"] = st & 0xFFFF, st >> 16
        return info

    def close(self):
        pass


class PinnedStager:
    """Index lists / small float rows for the device WITHOUT waiting for the stream.  A copy from pageable memory is stream-ordered and synchronous for the
    host: issued behind a step kernel it makes the host wait for that kernel, and the reset draws that follow run with the GPU idle.  Here the data goes
    through a ring of pinned buffers and is only enqueued; a slot is reused sixteen calls later (its copy event is checked first)."""

    def __init__(self, n: int, width: int, device, slots: int = 16):
        import torch

        self.device = device
        self._slots = [dict(idx=torch.empty(n, dtype=torch.int64, pin_memory=True), rows=torch.empty(n, max(width, 1), dtype=torch.float32, pin_memory=True), event=None)
                       for _ in range(slots)]
        self._next = 0

    def __call__(self, idx, rows=None):
        """idx (numpy ints [k]) -> int64 device tensor; with rows (numpy [k, w]): (indices, float32 device tensor [k, w])"""
        import torch

        slot = self._slots[self._next]
        self._next = (self._next + 1) % len(self._slots)
        if slot["event"] is not None:
            slot["event"].synchronize()
        k = len(idx)
        slot["idx"].numpy()[:k] = idx
        ti = torch.empty(k, dtype=torch.int64, device=self.device)
        ti.copy_(slot["idx"][:k], non_blocking=True)
        out = ti
        if rows is not None:
            rows = np.asarray(rows).reshape(k, -1)
            w = rows.shape[1]
            slot["rows"].numpy()[:k, :w] = rows
            tr = torch.empty(k, w, dtype=torch.float32, device=self.device)
            tr.copy_(slot["rows"][:k, :w], non_blocking=True)
            out = (ti, tr)
        slot["event"] = torch.cuda.Event()
        slot["event"].record(torch.cuda.current_stream(self.device))
        return out


def np_random(seed=None):
    """gymnasium.utils.seeding.np_random [3P]: PCG64 seeded through a SeedSequence."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer or None, got {seed!r}")
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy
