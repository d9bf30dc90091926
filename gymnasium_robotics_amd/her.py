"""Hindsight-experience-replay storage and relabelling on the device -- the caller of ``GoalEnv.compute_reward``.

The reference documents the use (README.md:72-76, gymnasium_robotics/core.py:45-67): "substitute a goal and recompute the reward":

    env.unwrapped.compute_reward(obs["achieved_goal"], substituted_goal, info)

With thousands of worlds stepped per launch, doing that through the host would move every trajectory over PCIe.  Here the episode rows the step
kernels already write (``env.packed`` = ``[obs | achieved | desired | reward | success]`` per world) are appended to a device buffer, and ONE kernel
(``grx_her_relabel``) gathers a batch of transitions, substitutes the goals ("future" strategy: a goal achieved later in the same episode, with
probability k / (k + 1)), recomputes reward and success with the device functions behind ``compute_reward`` and writes packed replay rows

    [obs_t | achieved_t | goal | action_t | reward | obs_t+1 | achieved_t+1 | success]

into a device ring buffer.  Nothing leaves HBM; the learner reads ``replay.rows``.
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _native

FETCH, HAND_REACH, MAZE, MANIPULATE = 0, 1, 2, 3


def reward_spec(env) -> dict:
    """kind / thresholds / flags of grx_her_relabel for a goal-conditioned device environment (the same parameters its compute_reward uses)"""
    name = type(env).__name__
    sparse = int(getattr(env, "reward_type", "sparse") == "sparse")
    if name == "FetchVecEnv":
        return dict(kind=FETCH, p0=float(env.task.distance_threshold), p1=0.0, sparse=sparse)
    if name == "HandReachVecEnv":
        return dict(kind=HAND_REACH, p0=float(env.distance_threshold), p1=0.0, sparse=sparse)
    if name in ("PointMazeVecEnv", "AntMazeVecEnv"):
        return dict(kind=MAZE, p0=0.45, p1=0.0, sparse=sparse)
    if name == "HandBlockVecEnv":   # the arguments of its grx_manip_compute_reward call (envs/hand.py:_launch_reward)
        from .envs.manipulate_spec import ROTATION_THRESHOLD

        return dict(kind=MANIPULATE, p0=float(env.distance_threshold), p1=float(ROTATION_THRESHOLD), sparse=sparse, ignore_pos=int(env.target_position == "ignore"),
                    ignore_rot=int(env.target_rotation == "ignore"), ignore_z=int(env._objcfg["ignore_z_target_rotation"]))
    raise TypeError(f"{name} is not a goal-conditioned environment")


class HerReplay:
    """A ring of the last `horizon` + 1 output rows of every world + the actions that led to them + a replay ring [capacity, OW], all on the
    environment's device.  Episodes may start at different steps in different worlds (same-step autoreset): `episode_start[w]` is the row at which
    world w's current episode began, and only rows of the current episode that are still in the ring are sampled.

        buf = HerReplay(env, horizon=50, capacity=1 << 20)
        obs, _ = env.reset(seed=0); buf.begin_episode(env.packed)
        for t in range(50):
            obs, r, term, trunc, info = env.step(a); buf.append(a, env.packed)        # append(..., reset_mask) when worlds were autoreset in this step
        buf.relabel(batch=4 * env.num_envs, k_future=4)        # one kernel: gather + goal substitution + reward recompute + replay write
    """

    def __init__(self, env, horizon: int, capacity: int, obs_dim: Optional[int] = None, goal_dim: Optional[int] = None, seed: int = 0, continuous: bool = False):
        self.env, self.T, self.N = env, int(horizon), int(env.num_envs)
        self.device = env.device
        self.W = int(env.packed.shape[1])
        self.goal_dim = int(goal_dim if goal_dim is not None else env.single_observation_space["desired_goal"].shape[0])
        self.obs_dim = int(obs_dim if obs_dim is not None else self.W - 2 * self.goal_dim - 2)
        self.act_dim = int(env.single_action_space.shape[0])
        self.OW = 2 * self.obs_dim + 3 * self.goal_dim + self.act_dim + 2
        self.spec = reward_spec(env)
        self.continuous = bool(continuous)        # False: append() past `horizon` steps is an error (one episode per buffer); True: the ring wraps
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=self.device)
        self.R = self.T + 1
        self.episode, self.actions = z(self.R, self.N, self.W), z(self.R, self.N, self.act_dim)   # actions[r] = the action that led to row r
        self.episode_start = z(self.N, dtype=torch.int32)
        # same-step autoreset with final_rows (append): where world w's previous episode began, and the absolute row index its terminal row belongs to (-1: none)
        self.prev_start, self.term_t = z(self.N, dtype=torch.int32), torch.full((self.N,), -1, dtype=torch.int32, device=self.device)
        self._final_rows = None                    # the env's [N, W] buffer of terminal rows (e.g. FetchVecEnv.final_packed); read by the relabel kernel, never copied
        self._just_ended = np.zeros(self.N, bool)  # host mirror of term_t == t
        self._start_host = np.zeros(self.N, np.int64)          # host mirror of episode_start: decides without a device sync whether anything can be sampled
        self.rows, self.capacity, self.head, self.size = z(int(capacity), self.OW), int(capacity), 0, 0
        self.t = 0                                 # absolute index of the newest row
        self._seed, self._calls = int(seed), 0     # counter-based index stream (grx_her_sample): see reseed()
        self._L = _native.lib()
        # reset masks reach the device through pinned buffers: a copy from pageable memory would make the host wait for the step kernel (core.PinnedStager)
        self._mask_pin = [dict(buf=torch.empty(self.N, dtype=torch.bool, pin_memory=True), event=None) for _ in range(4)]
        self._mask_next, self._mask_dev = 0, torch.zeros(self.N, dtype=torch.bool, device=self.device)

    # ---------------------------------------------------------------- episode storage (device copies of what the step kernel wrote)
    def begin_episode(self, packed_rows: torch.Tensor):
        self.episode[0].copy_(packed_rows)
        self.episode_start.zero_()
        self.prev_start.zero_(); self.term_t.fill_(-1)
        self._start_host[:] = 0
        self._just_ended[:] = False
        self.t = 0

    def append(self, actions: torch.Tensor, packed_rows: torch.Tensor, reset_mask: Optional[torch.Tensor] = None, final_rows: Optional[torch.Tensor] = None):
        """row t + 1 <- the rows of this step.  reset_mask (bool [N]): worlds that were autoreset inside this step -- their row is the first one of a new
        episode.  final_rows ([N, W] device buffer whose row w holds the TERMINAL packed row of a world reset in this step, e.g. FetchVecEnv.final_packed):
        the finished episode stays sampleable for this one step WITH its last transition (next observation = the terminal row), which is how a replay that
        stores whole episodes sees it (/root/reference/README.md:66-76).  Without final_rows that last transition is not stored."""
        if self.t >= self.T and not self.continuous:
            raise RuntimeError("episode buffer is full: call begin_episode()")
        self.t += 1
        r = self.t % self.R
        self.actions[r].copy_(actions)
        self.episode[r].copy_(packed_rows)
        if reset_mask is not None:     # bool [N]: numpy / CPU tensor (what the envs return) or a device tensor
            host = reset_mask.cpu().numpy() if isinstance(reset_mask, torch.Tensor) else np.asarray(reset_mask)
            self._start_host[host.astype(bool)] = self.t
            if isinstance(reset_mask, torch.Tensor) and reset_mask.device == self.device:
                dev = reset_mask
            else:
                slot = self._mask_pin[self._mask_next]
                self._mask_next = (self._mask_next + 1) % len(self._mask_pin)
                if slot["event"] is not None:
                    slot["event"].synchronize()
                slot["buf"].numpy()[:] = host.astype(bool)
                dev = self._mask_dev
                dev.copy_(slot["buf"], non_blocking=True)
                slot["event"] = torch.cuda.Event()
                slot["event"].record(torch.cuda.current_stream(self.device))
            if final_rows is not None:
                assert tuple(final_rows.shape) == (self.N, self.W) and final_rows.is_contiguous()
                self._final_rows = final_rows
            track = self._final_rows is not None
            self._just_ended = host.astype(bool) if track else self._just_ended
            _native.check(self._L.grx_her_mark_resets(dev.data_ptr(), self.N, self.t, self.episode_start.data_ptr(), self.prev_start.data_ptr() if track else None,
                                                      self.term_t.data_ptr() if track else None, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        elif self._final_rows is not None:
            self._just_ended = np.zeros(self.N, bool)

    def set_episode_start(self, starts):
        """absolute row at which every world's current episode began (e.g. negative values for episodes that were already under way at row 0)"""
        self._start_host[:] = np.asarray(starts, dtype=np.int64)
        self.episode_start.copy_(torch.from_numpy(self._start_host.astype(np.int32)).to(self.device))

    # ---------------------------------------------------------------- sampling + the fused relabel kernel
    def reseed(self, seed: int):
        """restart the index stream: the same (seed, number of sample_indices calls since) reproduces the same draws"""
        self._seed, self._calls = int(seed), 0

    def sample_indices(self, batch: int, k_future: int = 4):
        """(t, world, t_goal): a uniform world, a uniform transition of that world's current episode among the rows still in the ring; with
        probability k / (k + 1) the goal achieved at a uniformly drawn LATER row of the same episode (the "future" strategy of Andrychowicz et al.
        2017), else -1 = keep the episode's goal.  Worlds whose episode has no transition yet (just reset) are not drawn.  One kernel
        (grx_her_sample) with a counter-based generator; None when nothing can be sampled."""
        track = self._final_rows is not None
        if not ((np.maximum(self._start_host, max(self.t - self.T, 0)) < self.t) | (self._just_ended if track else False)).any():
            return None                                                                   # every world has just been reset: nothing to sample
        t, w, tg = (torch.empty(batch, dtype=torch.int32, device=self.device) for _ in range(3))
        _native.check(self._L.grx_her_sample_final(self.episode_start.data_ptr(), self.prev_start.data_ptr() if track else None, self.term_t.data_ptr() if track else None,
                                                   self.N, self.t, self.T, int(k_future), self._seed, self._calls, batch,
                                                   t.data_ptr(), w.data_ptr(), tg.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._calls += 1
        return t, w, tg

    def relabel_into(self, out: torch.Tensor, t: torch.Tensor, w: torch.Tensor, t_goal: torch.Tensor):
        """the kernel alone: out[b] <- relabelled transition (t[b], w[b], t_goal[b]); int32 index tensors on the device"""
        a = _native.HerArgsStruct()
        a.rows, a.acts = self.episode.data_ptr(), self.actions.data_ptr()
        a.T, a.N, a.W, a.obs_dim, a.goal_dim, a.act_dim = self.T, self.N, self.W, self.obs_dim, self.goal_dim, self.act_dim   # ring of T + 1 rows
        a.t_idx, a.w_idx, a.t_goal, a.out = t.data_ptr(), w.data_ptr(), t_goal.data_ptr(), out.data_ptr()
        if self._final_rows is not None:
            a.term_rows, a.term_t = self._final_rows.data_ptr(), self.term_t.data_ptr()
        for k, v in self.spec.items():
            setattr(a, k, v)
        assert out.is_contiguous() and tuple(out.shape) == (len(t), self.OW) and t.dtype == w.dtype == t_goal.dtype == torch.int32
        _native.check(self._L.grx_her_relabel(ctypes.byref(a), len(t), ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def relabel(self, batch: int, k_future: int = 4):
        """sample + relabel + write `batch` rows at the head of the replay ring; returns the view of the rows just written"""
        if batch > self.capacity:
            raise ValueError("batch larger than the replay capacity")
        if self.head + batch > self.capacity:
            self.head = 0                                    # keep every batch contiguous (a ring of whole batches)
        idx = self.sample_indices(batch, k_future)
        if idx is None:
            return self.rows[self.head: self.head]
        t, w, tg = idx
        view = self.rows[self.head: self.head + batch]
        self.relabel_into(view, t, w, tg)
        self.head += batch
        self.size = min(self.capacity, max(self.size, self.head))
        return view

    # ---------------------------------------------------------------- views of a replay row
    def split(self, rows: torch.Tensor) -> dict:
        o, g, a = self.obs_dim, self.goal_dim, self.act_dim
        c = np.cumsum([0, o, g, g, a, 1, o, g, 1])
        names = ("observation", "achieved_goal", "desired_goal", "action", "reward", "next_observation", "next_achieved_goal", "success")
        return {n: rows[:, c[i]: c[i + 1]] for i, n in enumerate(names)}
