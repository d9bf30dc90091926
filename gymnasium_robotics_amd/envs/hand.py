"""Batched Shadow Dexterous Hand reach environment on the MI355X engine (ids HandReach-v3 / HandReachDense-v3).

Host-side mirror of /root/reference/gymnasium_robotics/envs/shadow_dexterous_hand/reach.py (MujocoHandReachEnv) and
hand_env.py (MujocoHandEnv) on top of robot_env.py (BaseRobotEnv / MujocoRobotEnv): same constructor meaning, dict
observations (observation 63 = 24 joint positions | 24 joint velocities | 5 fingertip positions, goals 15), action 20
(absolute joint targets scaled to the actuator ranges), `compute_reward` on arbitrary leading batch dims, truncation at 50
steps, never terminated.  One call of `step` = one launch of `grx_hand_step_kernel` (set_action + 20 substeps + obs + reward).
"""
import ctypes
import weakref
import os
from typing import Optional

import numpy as np
import torch

from .. import _native
from ..core import create_rerun_model, GoalVecEnv, OverflowLane, PinnedStager, np_random
from ..mjcf import CompiledModel, compile_mjcf, load_model
from ..spaces import Box, Dict, batch_space
from .hand_spec import (DISTANCE_THRESHOLD, MAX_EPISODE_STEPS, N_ACTIONS, initial_qpos_vector, make_hand_task, parse_hand_reach_id,
                        sample_hand_reach_goal, sample_hand_reach_goal_batch)

_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")
GOAL_DIM = 15


# the task reads the five fingertip sites only (reach.py:398-405); without an object the hand produces a handful of contacts
# (explicit finger pairs), 24 friction-loss rows and a few limit / tendon rows (peaks in random rollouts: 4 contacts, 33 rows, < 100 pool
# words): 16 contact slots, 96 rows, 512 pool words.  Together they trim the per-world LDS footprint to 9 worlds per CU.
HAND_REACH_COMPILE = dict(keep_sites=["robot0:S_fftip", "robot0:S_mftip", "robot0:S_rftip", "robot0:S_lftip", "robot0:S_thtip"], capacity={"maxcon": 16, "maxefc": 96, "jpool": 512})


# World origin of the compiled hand models (MJCF coordinates): the hand mount sits at (1, 1.25, 0.15) (assets/hand/robot.xml:3), the object starts at (1, 0.87, 0.2)
# (manipulate_block.xml:27), so every position of the palm's workspace is within 0.3 m of this point: an fp32 ulp of 1.5e-8 .. 3e-8 m instead of 1.2e-7
# (compile_mjcf(origin=...); measured on the fixtures: touch forces 79 % -> 100 % within 1e-4 ABSOLUTE, velocity medians 1.0e-5 -> 2.5e-6: profiles/origin_r06_emu.txt)
HAND_ORIGIN = (1.0, 1.0, 0.2)


def load_hand_reach_model(assets_root: Optional[str] = None) -> CompiledModel:
    """hand/reach.xml compiled from MJCF when an asset tree is given (assets_root / $GRX_ASSETS_ROOT), else the packaged blob."""
    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        return compile_mjcf(os.path.join(assets_root, "hand", "reach.xml"), origin=HAND_ORIGIN, **HAND_REACH_COMPILE)
    path = os.path.join(_MODELS_DIR, "hand_reach.npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist")
    return load_model(path)


HAND_SPLIT_PARTS = {"HandReachVecEnv": 4, "HandBlockVecEnv": 2}      # default of GRX_HAND_SPLIT for batches of more than 2 048 worlds (see HandReachVecEnv.__init__; profiles/ab_r06_hand_split.txt, 16 384 worlds, 1 / 2 / 4 / 5 parts:
                                                                    # HandReach 1.51 / 1.553 / 1.558 / 1.549 M; hand + touch (throughput-bound with its settle chains) 0.983 / 0.993 / 0.975 / 0.971 M)


class HandReachVecEnv(GoalVecEnv):
    def __init__(self, env_id: str = "HandReach-v3", num_envs: int = 1, device: Optional[str] = None, reward_type: Optional[str] = None,
                 relative_control: bool = False, max_episode_steps: Optional[int] = MAX_EPISODE_STEPS, autoreset_mode: str = "next_step",
                 output: str = "numpy", assets_root: Optional[str] = None, model: Optional[CompiledModel] = None, seed_offset: int = 0,
                 distance_threshold: Optional[float] = None, balance: bool = False):
        if relative_control:
            # hand_env.py:43-52 reads data.get_joint_qpos / model.actuator_names, which the mujoco bindings do not have: the
            # reference itself cannot run this branch on the mujoco (non mujoco_py) backend
            raise NotImplementedError("relative_control=True is not available on the mujoco backend of the reference either")
        self.env_id = env_id
        self.distance_threshold = DISTANCE_THRESHOLD   # reach.py:60; the manipulate envs set their own in _parse_id
        self._parse_id(env_id, reward_type)
        if distance_threshold is not None:
            self.distance_threshold = float(distance_threshold)
        self.max_episode_steps, self.autoreset_mode, self.output = max_episode_steps, autoreset_mode, output
        self.num_envs, self.seed_offset = int(num_envs), int(seed_offset)
        if not torch.cuda.is_available():
            raise RuntimeError("HandReachVecEnv needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device or "cuda:0")
        self.model = model or self._load_model(assets_root)
        self.nq, self.nv, self.nu = self.model.dim("nq"), self.model.dim("nv"), self.model.dim("nu")
        if self.nu != N_ACTIONS:
            raise ValueError("Action dimension mismatch")
        self._L = _native.lib()
        H, I, F = self.model.pack()
        self._h = _native.acquire_model(H, I, F, self.device.index or 0)   # shared with every other environment of the same compiled tables (reference-counted)
        self._h_big = create_rerun_model(self._L, self.model, self.device.index or 0)    # larger tables for the worlds that overflow a capacity (core.RERUN_CAPACITY)
        self.task = self._make_task()
        GOAL_DIM = self.GOAL_DIM
        self.obs_dim = self._obs_dim()
        n, d = self.num_envs, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=d)
        self.qpos, self.qvel, self.qacc_ws = z(n, self.nq), z(n, self.nv), z(n, self.nv)
        self.goal, self.action, self.obs, self.achieved = z(n, GOAL_DIM), z(n, self.nu), z(n, self.obs_dim), z(n, GOAL_DIM)
        self.palm, self.reward = z(n, 3), z(n)
        self.packed = z(n, self.obs_dim + 2 * GOAL_DIM + 2)   # [obs | achieved | desired | reward | success] rows written by the step kernel (cross-rank gather)
        self.success, self.status, self.mask = z(n, dtype=torch.uint8), z(n, dtype=torch.int32), torch.ones(n, dtype=torch.uint8, device=d)
        # cost-ordered dispatch (see FetchVecEnv._alloc / include/grx_capi.h): the worlds that took longest in the last launch start first.
        # Off by default for the hand: measured neutral to -1.5 % (16 384 worlds are 8 per wave slot, the tail is short; at 4 096 the cost of a
        # world under random finger motion is not persistent enough to pay for the argsort)
        self.balance = bool(balance) and n % 8 == 0 and 1024 <= n <= 65536 * 8
        self.cost = torch.zeros(n, dtype=torch.int32, device=d) if self.balance else None
        self.cost_ema = torch.zeros(n, dtype=torch.float32, device=d) if self.balance else None
        self.balance_alpha = 0.1   # weight of the newest sample in the moving average the order is sorted by (A/B on FetchPickAndPlace: 1.0 -> 2.69 ms, 0.15 -> 2.64 ms per step)
        self.order = None
        if self.balance:
            per = n // 8
            self._slice_base = (torch.arange(8, device=d, dtype=torch.int32) * per).unsqueeze(1)
            self.order = (self._slice_base + torch.arange(per, device=d, dtype=torch.int32).unsqueeze(0)).t().contiguous().view(-1)
        self._bufs, self._bufs_masked = self._make_bufs(None), self._make_bufs(self.mask)
        # no dropped contacts: the worlds that exceed a table capacity of the fast kernel are stepped on larger tables (core.OverflowLane)
        # (with the entrants picked up by polling workgroups a smaller standing lane is cheaper: margin 0.9 / ttl 4 against 0.8 / 8, +1 % on hand + touch, profiles/ab_r03_lane_size_with_polling.txt)
        self.lane = OverflowLane(n, d, self.model, self._lane_make_bufs(), mode=self.LANE_MODE, margin=0.9, ttl=4) if self._h_big is not None else None
        # SPLIT STEP (include/grx_capi.h grx_hand_buffers.split_parts; see AdroitVecEnv): a hand launch ended 15 - 17 % after the mean of its wave slots (profiles/tail_probe_r06.txt).
        # GRX_HAND_SPLIT=P (1: off).  Bit-identical to the plain launch (tests/test_gpu_hand.py::test_split_step_is_the_plain_step).
        self._split = max(1, min(int(self.task.n_substeps), 8, int(os.environ.get("GRX_HAND_SPLIT", HAND_SPLIT_PARTS.get(type(self).__name__, 1) if n > 2048 else 1)))) if n >= 64 else 1
        if self._split > 1:
            stride = -(-(self.nq + 2 * self.nv) // 16) * 16
            self._split_rows, self._split_state = z(n, stride), z(n, 4, dtype=torch.int32)
            for b in (self._bufs, self._bufs_masked):
                b.split_rows, b.split_state, b.split_stride, b.split_parts = self._split_rows.data_ptr(), self._split_state.data_ptr(), stride, self._split
            if self.lane is not None:
                self.lane._fast_grid *= self._split      # (polling workgroups of the standing lane wait for EVERY workgroup of the fast launch)
        self.single_action_space = Box(-1.0, 1.0, (self.nu,), np.float32)
        self.single_observation_space = Dict(dict(
            observation=Box(-np.inf, np.inf, (self.obs_dim,), np.float64), achieved_goal=Box(-np.inf, np.inf, (GOAL_DIM,), np.float64),
            desired_goal=Box(-np.inf, np.inf, (GOAL_DIM,), np.float64)))
        self.action_space = batch_space(self.single_action_space, n)
        self.observation_space = batch_space(self.single_observation_space, n)
        self._check_goal_space()
        self.np_randoms = [np_random(None)[0] for _ in range(n)]
        self._elapsed = np.zeros(n, np.int64)
        self._needs_reset = np.zeros(n, bool)
        self._has_reset = False
        self.kernel_events = None  # when a list: (start, end) HIP events around every step-kernel launch (benchmarks)
        self.step_events = []      # ... and around the whole launch group of a step (fast kernel + the overflow lane's launches)
        self._chain_stage = PinnedStager(n, max(self.nq, GOAL_DIM), d)      # the settle chains' start poses (side streams)
        self._dev_index = PinnedStager(n, max(self.nq, GOAL_DIM), d)   # index lists / goal rows of the autoreset path: enqueued, never waited for (core.PinnedStager)
        self._env_setup()

    # ---- hooks specialised by the manipulation envs
    GOAL_DIM = GOAL_DIM
    LANE_MODE = "entry"      # HandReach: no overflow in 4 M world-steps (profiles/soak); the hand + object models keep a standing lane (core.OverflowLane)

    def _parse_id(self, env_id, reward_type):
        self.reward_type = reward_type or parse_hand_reach_id(env_id)

    def _load_model(self, assets_root):
        return load_hand_reach_model(assets_root)

    def _make_task(self):
        t = make_hand_task(self.model, self.reward_type)
        t.distance_threshold = self.distance_threshold
        return t

    def _obs_dim(self):
        return self.nq + self.nv + self.GOAL_DIM

    def _lane_make_bufs(self):
        me = weakref.ref(self)      # (the lane must not keep the environment alive: its native model slots are released by __del__)
        return lambda m: me()._make_bufs(m, large=True)

    def _make_bufs(self, mask, large=False):
        """large: buffers of a launch of the large-table kernel of the overflow lane (no cost ordering: it runs a handful of worlds)"""
        b = _native.HandBuffersStruct()
        for name in ("qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "palm", "reward", "success", "status", "packed"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.mask = None if mask is None else mask.data_ptr()
        b.order = None if (self.order is None or large) else self.order.data_ptr()
        b.cost = None if (self.cost is None or large) else self.cost.data_ptr()
        return b

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _launch(self, bufs, forward_only, settle=False):
        """settle: a reset-time settle launch of the step kernel (manipulate.py:205-224) -- not a timed env.step(), no cost re-ordering."""
        own = bufs is self._bufs or bufs is self._bufs_masked      # the side arenas of the overlapped settle chains carry their own structs (no lane: a handful of freshly reset worlds)
        large = lambda b: _native.check(self._L.grx_hand_step(self._h_big, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, 0, self._stream()))
        if settle:
            fast = lambda b: _native.check(self._L.grx_hand_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, 0, self._stream()))
            if self.lane is not None and own:
                self.lane.rerun_only(self.mask if bufs is self._bufs_masked else None, bufs, fast, large)     # every masked world is stepped by the fast kernel, whatever lane it is in; an overflow is re-run at once
            else:
                fast(bufs)
            return

        def fast(b):
            timed = self.kernel_events is not None and not forward_only
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _native.check(self._L.grx_hand_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, int(forward_only), self._stream()))
            if timed:
                e1.record()
                self.kernel_events.append((e0, e1))

        if self.lane is not None and own and not forward_only:
            timed_all = self.kernel_events is not None and not forward_only      # the fast launch AND the lane's launches (side stream, joined before the entry launch): the step's device time
            if timed_all:
                l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                l0.record()
            self.lane.step(self.mask if bufs is self._bufs_masked else None, fast, large, fast_bufs=bufs)
            if timed_all:
                l1.record()
                self.step_events.append((l0, l1))
        else:
            fast(bufs)
        if self.balance and not forward_only:
            _native.check(self._L.grx_order_by_cost(self.cost.data_ptr(), self.cost_ema.data_ptr(), self.balance_alpha, self.num_envs, self.order.data_ptr(), self._stream()))

    # ------------------------------------------------------------------ _env_setup (reach.py:408-416) on the device
    def _env_setup(self):
        self._initial_qpos = torch.from_numpy(initial_qpos_vector(self.model).astype(np.float32)).to(self.device)
        with torch.cuda.device(self.device):
            self.qpos[:] = self._initial_qpos
            self.qvel.zero_()
            self.qacc_ws.zero_()
            self._launch(self._bufs, True)
            torch.cuda.synchronize(self.device)
        if int(self.status[0]) != 0:
            raise RuntimeError(f"engine reported status {int(self.status[0])} during env setup")
        self.initial_goal = self.achieved[0].double().cpu().numpy().copy()
        self.palm_xpos = self.palm[0].double().cpu().numpy().copy()

    def _begin_overlapped_reset(self):
        return None

    def _after_step_launch(self, spec):
        pass

    def _cancel_chains(self):
        pass

    def _before_step_launch(self, spec):
        pass

    # ------------------------------------------------------------------ reset (robot_env.py:154-182, 300-313; reach.py:99-126)
    def _reset_worlds(self, idx):
        if len(idx) == 0:
            return
        goals = sample_hand_reach_goal_batch([self.np_randoms[w] for w in idx], self.initial_goal, self.palm_xpos)   # host draws: the step kernel may still be running
        ti, tg = self._dev_index(np.asarray(idx, dtype=np.int64), goals)
        self.qpos[ti] = self._initial_qpos
        self.qvel.index_fill_(0, ti, 0.0)      # (x[ti] = 0.0 would upload a host scalar: a synchronising copy)
        self.qacc_ws.index_fill_(0, ti, 0.0)
        self.goal[ti] = tg
        self.mask.zero_()
        self.mask.index_fill_(0, ti, 1)
        self._launch(self._bufs_masked, True)
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    def reset(self, *, seed=None, options=None):
        with torch.cuda.device(self.device):
            self._cancel_chains()      # BEFORE re-seeding: a cancelled chain hands its draws back to the generator they came from, never to a freshly seeded one
            if seed is not None:
                seeds = [seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed)
                self.np_randoms = [np_random(s)[0] for s in seeds]
            self._reset_worlds(np.arange(self.num_envs))
        self._has_reset = True
        return self._obs_dict(), {}

    # ------------------------------------------------------------------ step (robot_env.py:114-152)
    def step(self, actions):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        a = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions, dtype=np.float32))
        if tuple(a.shape) != (self.num_envs, self.nu):
            raise ValueError("Action dimension mismatch")
        self.action.copy_(a.to(torch.float32), non_blocking=True)
        with torch.cuda.device(self.device):
            pending = np.nonzero(self._needs_reset)[0] if self.autoreset_mode == "next_step" else np.zeros(0, np.int64)
            spec = self._begin_overlapped_reset() if self.autoreset_mode == "same_step" else None
            if spec is not None:
                self._before_step_launch(spec)
            if len(pending):
                self.mask.fill_(1)
                self.mask.index_fill_(0, self._stage_idx(pending), 0)      # (pinned staging + index_fill_: nothing here waits for the running kernel)
                self._launch(self._bufs_masked, False)
            else:
                self._launch(self._bufs, False)
            if spec is not None:
                self._after_step_launch(spec)   # host-side draws of the overlapped resets, while the step kernel runs
            stepped = ~self._needs_reset
            self._elapsed[stepped] += 1
            truncated = np.zeros(self.num_envs, bool)
            if self.max_episode_steps is not None:
                truncated = stepped & (self._elapsed >= self.max_episode_steps)
            terminated = np.zeros(self.num_envs, bool)
            info = {}
            if len(pending):
                self._reset_worlds(pending)
                tp = self._stage_idx(pending)
                self.reward.index_fill_(0, tp, 0.0)
                self.packed[:, -2].index_fill_(0, tp, 0.0)      # the packed row (cross-rank gather, HER) reports the same reward as reward[]
            if self.autoreset_mode == "same_step" and truncated.any():
                done = np.nonzero(truncated)[0]
                td = self._dev_index(done)
                info["final_obs"] = self._obs_dict(rows=done, ti=td)   # the terminal observation (bootstrapping), as FetchVecEnv reports it
                keep_r, keep_s, keep_st = self.reward.clone(), self.success.clone(), self.status.clone()
                if spec is not None:
                    restore = self._finish_overlapped_reset(spec, done)     # False: every world came from a settle chain, committed by one kernel that leaves these words alone
                else:
                    self._reset_worlds(done)
                    restore = True
                if restore:
                    self.reward.copy_(keep_r)
                    self.success.copy_(keep_s)
                    self.status.copy_((keep_st & 0xFFFF) | (self.status & -65536))   # this step's flags are the step launch's, not the reset launches'; sticky bits keep accumulating
                    self.packed[td, -2] = keep_r[td]
                    self.packed[td, -1] = keep_s[td].float()
            elif self.autoreset_mode == "next_step":
                self._needs_reset |= truncated
        obs = self._obs_dict()
        if self.output == "torch":
            info["is_success"] = self.success
            return obs, self.reward, torch.from_numpy(terminated), torch.from_numpy(truncated), self._status_info(info)
        info["is_success"] = self.success.cpu().numpy().astype(np.float32)
        self._status_info(info)
        r = self.reward.cpu().numpy()
        return obs, (r if self.reward_type == "sparse" else r.astype(np.float64)), terminated, truncated, info

    def _obs_dict(self, rows=None, ti=None):
        if self.output == "torch":
            if rows is not None and ti is None:
                ti = self._dev_index(rows)
            sel = (lambda t: t) if rows is None else (lambda t: t[ti])
            return {"observation": sel(self.obs), "achieved_goal": sel(self.achieved), "desired_goal": sel(self.goal)}
        sel = (lambda a: a) if rows is None else (lambda a: a[rows])
        return {"observation": sel(self.obs.double().cpu().numpy()), "achieved_goal": sel(self.achieved.double().cpu().numpy()),
                "desired_goal": sel(self.goal.double().cpu().numpy())}

    # ------------------------------------------------------------------ GoalEnv API (reach.py:92-97; core.py:45-114)
    def compute_reward(self, achieved_goal, desired_goal, info=None):
        as_numpy = not isinstance(achieved_goal, torch.Tensor)
        ag = torch.as_tensor(np.asarray(achieved_goal, dtype=np.float32) if as_numpy else achieved_goal, dtype=torch.float32, device=self.device).contiguous()
        dg = torch.as_tensor(np.asarray(desired_goal, dtype=np.float32) if as_numpy else desired_goal, dtype=torch.float32, device=self.device).contiguous()
        if ag.shape != dg.shape or ag.shape[-1] != self.GOAL_DIM:
            raise ValueError(f"achieved_goal and desired_goal must have the same (..., {self.GOAL_DIM}) shape")
        out = torch.empty(ag.shape[:-1], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._launch_reward(ag, dg, out)
        if not as_numpy:
            return out
        r = out.cpu().numpy()
        return r if self.reward_type == "sparse" else r.astype(np.float64)

    def _launch_reward(self, ag, dg, out):
        _native.check(self._L.grx_goal_compute_reward(ag.data_ptr(), dg.data_ptr(), out.numel(), self.GOAL_DIM, self.distance_threshold,
                                                      int(self.reward_type == "sparse"), out.data_ptr(), self._stream()))

    def compute_terminated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(achieved_goal).shape[:-1], bool)  # robot_env.py:106-108

    def compute_truncated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(achieved_goal).shape[:-1], bool)  # robot_env.py:110-112

    # ---- checkpoint hooks (core.GoalVecEnv.get_state / set_state): the settle chains of the overlapped same-step autoreset are state -- arena rows, the worlds they
    # belong to, the step they are due at, the generator positions they would hand back -- and they live on side streams
    def _ckpt_quiesce(self):
        for c in getattr(self, "_chains", None) or ():
            if c["event"] is not None:
                c["event"].synchronize()

    def _ckpt_extra_get(self):
        import copy

        chains = [dict(worlds=np.asarray(c["worlds"]).copy(), lo=int(c["lo"]), k=int(c["k"]), due_at=int(c["due_at"]), ok=None if c["ok"] is None else np.asarray(c["ok"]).copy(),
                       rng_states=copy.deepcopy(c.get("rng_states", [])), obj=c["obj_host"][: c["k"]].numpy().copy()) for c in (getattr(self, "_chains", None) or ())]
        return {"chains": chains, "arena": getattr(self, "_ar", None) is not None}

    def _ckpt_extra_set(self, extra):
        if getattr(self, "_ar", None) is not None:      # chains of the run that is being replaced: wait for them, drop them (their draws are overwritten with the generators' states)
            self._ckpt_quiesce()
            self._chains, self._chain_started[:] = [], False
        if not extra.get("arena"):
            return
        self._arena()
        chains = []
        for c in extra["chains"]:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))      # (set_state synchronises before it returns: the arena rows restored behind this are in place when the event is consulted)
            buf = self._chain_obj[c["due_at"] % len(self._chain_obj)]
            buf[: c["k"]] = torch.from_numpy(c["obj"])
            chains.append(dict(worlds=c["worlds"].copy(), ti=self._dev_index(c["worlds"]), lo=c["lo"], k=c["k"], due_at=c["due_at"], ready=ev, event=ev, ok=None if c["ok"] is None else c["ok"].copy(),
                               rng_states=c["rng_states"], obj_host=buf))
        self._chains = chains

    def close(self):
        if getattr(self, "_h", None):
            _native.release_model(self._h)
            self._h = None
        if getattr(self, "_h_big", None):
            _native.release_model(self._h_big)
            self._h_big = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# Engine capacities of the hand + object models.  With split dof spans a finger-object row takes 12-20 pool words (6 rows per condim-4
# contact); random rollouts peak at 10 contacts / 69 rows.  24 contacts / 112 rows / 1 024 pool words give 8 worlds per CU; worlds that
# exceed them in a substep drop the excess contacts for that substep and raise GRX_ST_EFC_OVERFLOW in `status`.
HAND_MANIP_CAPACITY = {"maxcon": 24, "maxefc": 112, "jpool": 1024}


def load_hand_block_model(assets_root: Optional[str] = None, touch: bool = False, obj: str = "block") -> CompiledModel:
    """hand/manipulate_{block,pen}[_touch_sensors].xml without its visual-only target body (manipulate_spec.drop_target_body); with
    touch=True the 92 'robot0:TS_*' touch zones are compiled into the touch_* tables."""
    from .manipulate_spec import OBJECTS, drop_target_body, touch_filter

    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        xml = OBJECTS[obj]["xml"] + ("_touch_sensors.xml" if touch else ".xml")
        # the task reads no site (manipulate.py:298-316 use qpos / qvel only): none is tracked by the engine
        return compile_mjcf(os.path.join(assets_root, "hand", xml), mutate=drop_target_body, touch_filter=touch_filter if touch else None, keep_sites=[], origin=HAND_ORIGIN,
                            capacity=dict(HAND_MANIP_CAPACITY, jpool=928) if touch else HAND_MANIP_CAPACITY)   # touch keeps contact data out of the overlay: same 16 LDS granules with a slightly smaller pool
    path = os.path.join(_MODELS_DIR, f"hand_{obj}_touch.npz" if touch else f"hand_{obj}.npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist")
    return load_model(path)


class HandBlockVecEnv(HandReachVecEnv):
    """Batched HandManipulateBlock{, RotateZ, RotateParallel, RotateXYZ, Full} and HandManipulatePen{, Rotate, Full}
    [_ContinuousTouchSensors | _BooleanTouchSensors][Dense]-v1 (pen: manipulate_pen.py:216-235 -- no initial rotation noise, z rotation
    ignored in the goal distance, position threshold 0.05)
    (/root/reference/gymnasium_robotics/envs/shadow_dexterous_hand/manipulate.py: MujocoManipulateEnv; manipulate_block.py:214-230).
    Observation 61 = 24 robot joint positions | 24 velocities | object velocity 6 | object pose 7; goals are 7-vector poses.
    The visual-only, non-colliding `target` body of the MJCF is not simulated (its state is not observable through the env API)."""
    LANE_MODE = "lane"
    chain_events = None      # set to [] to collect (step started, step due, worlds, start event, end event) of every settle chain
    CHAIN_LOOKAHEAD = int(os.environ.get("GRX_CHAIN_LOOKAHEAD", 2))      # steps before the time limit at which a world's settle chain is started (same-step autoreset); 3 measured no faster (below)

    _chain_first = os.environ.get("GRX_CHAIN_FIRST", "0") != "0"      # settle chains launched ahead of the step launch (see _before_step_launch)
    _fused_settle = os.environ.get("GRX_HAND_FUSED_SETTLE", "1") != "0"      # a chain's ten settle steps as ONE repeat launch (include/grx_capi.h grx_hand_step_repeat); 0: ten launches (A/B)

    GOAL_DIM = 7

    def __init__(self, env_id: str = "HandManipulateBlockRotateXYZ-v1", num_envs: int = 1, max_episode_steps: Optional[int] = 100,
                 target_position: Optional[str] = None, target_rotation: Optional[str] = None, **kw):
        # constructor overrides of the registered values, as gym.make(id, target_position="fixed") allows (tests/envs/hand/test_manipulate.py:21)
        self._tp_override, self._tr_override = target_position, target_rotation
        super().__init__(env_id, num_envs, max_episode_steps=max_episode_steps, **kw)

    def _parse_id(self, env_id, reward_type):
        from .manipulate_spec import canonical_parallel_quats, parse_block_id

        from .manipulate_spec import OBJECTS, object_of

        self.target_position, self.target_rotation, rt, self.touch_get_obs = parse_block_id(env_id)
        if self._tp_override is not None:
            if self._tp_override not in ("ignore", "random", "fixed"):
                raise ValueError(f'Unknown target_position option "{self._tp_override}".')
            self.target_position = self._tp_override
        if self._tr_override is not None:
            if self._tr_override not in ("z", "parallel", "xyz"):      # "ignore" / "fixed" call a mujoco_py-only API in the reference (manipulate.py:270)
                raise ValueError(f'Unknown target_rotation option "{self._tr_override}".')
            self.target_rotation = self._tr_override
        self.object = object_of(env_id)
        self._objcfg = OBJECTS[self.object]
        self.distance_threshold = self._objcfg["distance_threshold"]
        self.reward_type = reward_type or rt
        self._pquats = canonical_parallel_quats()

    def _load_model(self, assets_root):
        return load_hand_block_model(assets_root, touch=self.touch_get_obs != "off", obj=self.object)

    def _make_task(self):
        from .manipulate_spec import make_block_task

        return make_block_task(self.model, self.target_position, self.target_rotation, self.reward_type, self.touch_get_obs, self.object)

    def _obs_dim(self):
        from .manipulate_spec import N_TOUCH

        return 2 * 24 + 6 + 7 + (N_TOUCH if self.touch_get_obs != "off" else 0)   # 61, or 153 with touch (manipulate_touch_sensors.py:113-137)

    def _env_setup(self):
        # manipulate.py:149-152 with initial_qpos = {}: the model's qpos0
        self._initial_qpos = torch.from_numpy(self.model.tables["qpos0"].astype(np.float32)).to(self.device)
        self._initial_qpos_host = torch.from_numpy(self.model.tables["qpos0"].astype(np.float32))
        self._qa = int(self.task.obj_qadr)
        # The model's world frame is palm-centred (load_hand_block_model: HAND_ORIGIN): the qpos ROWS hold the object's position relative to model.origin, in fp32;
        # the reference's arithmetic on object positions (reset pose, on-palm test, goal) is done here in fp64 in the MJCF's frame, _obj_rows / _obj_world convert.
        self._origin = self.model.origin
        self._obj0 = self._obj_world(self.model.tables["qpos0"][self._qa: self._qa + 7].astype(np.float64)[None])[0]
        self.reset_attempts = np.zeros(self.num_envs, np.int64)

    def _obj_world(self, rows):
        """object pose rows [k, 7] of the device state (model frame) -> fp64 poses in the MJCF's world frame"""
        out = np.array(rows, dtype=np.float64)
        out[:, :3] += self._origin
        return out

    def _obj_rows(self, poses):
        """fp64 object poses [k, 7] in the MJCF's world frame -> float32 rows of the device state (model frame); the subtraction is done in fp64"""
        out = np.array(poses, dtype=np.float64)
        out[:, :3] -= self._origin
        return out.astype(np.float32)

    # manipulate.py:154-224 (_reset_sim: pose randomisation, ten settle steps with a zero action, on-palm test, retried until it
    # holds -- robot_env.py:163-171) and :226-279 (_sample_goal from the settled pose).  The reference does not call mj_resetData
    # here, so the warm start of the previous episode survives the reset.
    def _reset_worlds(self, idx):
        if len(idx) == 0:
            return
        pending = np.asarray(idx, dtype=np.int64)
        self.reset_attempts[pending] = 0
        self._settle_until_on_palm(pending)
        self._sample_goals(idx)

    def _settle_until_on_palm(self, pending):
        """the retry loop of robot_env.py:163-171 around _reset_sim (manipulate.py:154-224) for the listed worlds, in the main buffers"""
        from .manipulate_spec import PALM_HEIGHT, SETTLE_STEPS, sample_reset_object_pose_batch

        saved_action = self.action.clone()
        while len(pending):
            self.reset_attempts[pending] += 1
            poses = sample_reset_object_pose_batch([self.np_randoms[w] for w in pending], self._obj0[:3], self._obj0[3:], self.target_position, self.target_rotation,
                                                   self._pquats, randomize_initial_rotation=self._objcfg["randomize_initial_rotation"])
            ti = torch.from_numpy(pending).to(self.device)
            q = self._initial_qpos.unsqueeze(0).repeat(len(pending), 1)
            q[:, self._qa: self._qa + 7] = torch.from_numpy(self._obj_rows(poses)).to(self.device)
            self.qpos[ti] = q
            self.qvel.index_fill_(0, ti, 0.0)
            self.action.zero_()
            self.mask.zero_()
            self.mask.index_fill_(0, ti, 1)
            for _ in range(SETTLE_STEPS):
                self._launch(self._bufs_masked, False, settle=True)
            z = self.qpos[ti, self._qa + 2].double().cpu().numpy() + self._origin[2]
            pending = pending[~(z > PALM_HEIGHT)]
        self.action.copy_(saved_action)

    def _sample_goals(self, idx):
        from .manipulate_spec import sample_block_goal_batch

        ti = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(self.device)
        obj = self._obj_world(self.qpos[ti, self._qa: self._qa + 7].double().cpu().numpy())
        goals = sample_block_goal_batch([self.np_randoms[w] for w in idx], obj, self.target_position, self.target_rotation, self._pquats)
        self.goal[ti] = torch.from_numpy(goals.astype(np.float32)).to(self.device)
        gd = self.goal.shape[1]   # the settle launches wrote the packed rows against the PREVIOUS goal: the reset row carries the new one
        self.packed[ti, self.obs_dim + gd: self.obs_dim + 2 * gd] = self.goal[ti]
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    # ---- same-step autoreset at the time limit, overlapped with the step launches.  Which worlds hit the limit in a step() is known in advance
    # (manipulate envs never terminate early: compute_terminated is always False, core.py:100-101), and their reset needs nothing from the last
    # steps of the old episode except the solver's warm start.  The ten settle steps are ten DEPENDENT launches of ~1.8 ms each however few worlds they
    # carry (one wavefront's latency through 20 substeps) -- 18 ms against the 12 ms step of 16 384 worlds.  So the settle chain of the worlds that
    # will hit the limit at the NEXT step is started now, on a high-priority side stream over a compacted side arena, and runs underneath two step
    # launches; when its worlds are done the settled state is scattered into the main buffers.  Same draws, same kernels, same number of settle steps
    # as the sequential path; the one difference: the settle's first solve starts from the warm start the world had when the chain was started (one or
    # two steps before the end of the episode) instead of the one left by the final step.
    def _arena(self):
        if getattr(self, "_ar", None) is None:
            n, d = 2 * self.num_envs, self.device
            z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=d)
            ar = {k: z(n, getattr(self, k).shape[1]) for k in ("qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "palm", "packed")}
            ar.update(reward=z(n), success=z(n, dtype=torch.uint8), status=z(n, dtype=torch.int32))
            self._ar, self._ar_head, self._chains, self._step_no = ar, 0, [], 0
            self._chain_pin, self._chain_pin_ref, self._chain_ms = None, 0.0, {}
            self._chain_pin_enabled = os.environ.get("GRX_CHAIN_PIN", "0") != "0"      # opt-in: see _pin_chain_stream
            self._chain_started = np.zeros(self.num_envs, bool)
            self._chain_obj = [torch.empty(self.num_envs, 7, dtype=torch.float32, pin_memory=True) for _ in range(self.CHAIN_LOOKAHEAD + 2)]      # settled object poses of the chains in flight (one buffer per due step)
            self._side = [torch.cuda.Stream(device=d, priority=int(os.environ.get("GRX_CHAIN_PRIO", "-1"))) for _ in range(int(os.environ.get("GRX_CHAIN_STREAMS", "3")))]   # one per chain generation in flight (round 6: until one of them is found to be the fast one, _pin_chain_stream)
            self._goal_side = torch.cuda.Stream(device=d)      # the goal rows of finished chains (_early_goals)
        return self._ar

    def _arena_bufs(self, lo):
        b = _native.HandBuffersStruct()
        for name, t in self._ar.items():
            setattr(b, name, t[lo:].data_ptr())
        b.mask = b.order = b.cost = None
        return b

    def _reserve_chain(self, worlds, due_at):
        """BEFORE the step launch (main stream): arena rows for the chain, snapshot of the worlds' warm start (the step kernel is about to overwrite it)"""
        ar, k, cap = self._arena(), len(worlds), 2 * self.num_envs
        lo = self._ar_head if self._ar_head + k <= cap else 0
        if any(lo < c["lo"] + c["k"] and c["lo"] < lo + k for c in self._chains):
            return None                                         # no room next to the chains in flight: these worlds take the sequential path
        self._ar_head = lo + k
        ti = self._dev_index(worlds)
        ar["qacc_ws"][lo: lo + k] = self.qacc_ws[ti]
        ar["goal"][lo: lo + k] = self.goal[ti]
        ar["status"][lo: lo + k] = 0
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        c = dict(worlds=worlds, ti=ti, lo=lo, k=k, due_at=due_at, ready=ready, event=None, ok=None)
        self._chains.append(c)
        self._chain_started[worlds] = True
        return c

    def _start_chain(self, c):
        """AFTER the step launch: the pose draws (host) and the ten settle launches on the side stream; nothing here touches the main stream"""
        from .manipulate_spec import SETTLE_STEPS, sample_reset_object_pose_batch

        worlds, lo, k, ar = c["worlds"], c["lo"], c["k"], self._ar
        self.reset_attempts[worlds] = 1
        # the chain draws from the worlds' own generators BEFORE their episodes have ended: if it is abandoned (an explicit reset() / set-state call) the draws are
        # given back, so that a world's stream only ever advances by draws the reference's sequential reset would have made (_cancel_chains)
        c["rng_states"] = [self.np_randoms[w].bit_generator.state for w in worlds]
        poses = sample_reset_object_pose_batch([self.np_randoms[w] for w in worlds], self._obj0[:3], self._obj0[3:], self.target_position, self.target_rotation,
                                               self._pquats, randomize_initial_rotation=self._objcfg["randomize_initial_rotation"])
        q = self._initial_qpos_host.unsqueeze(0).repeat(k, 1)
        q[:, self._qa: self._qa + 7] = torch.from_numpy(self._obj_rows(poses))
        # Which side stream: round-robin -- until the measured chain times say that ONE of the streams gets wave slots beside the step kernel and the others do not
        # (profiles/chain_probe_r06.txt: of K high-priority streams exactly one runs a chain in 17 ms, the others in 30 ms = behind the step launch's dispatch, whatever the
        # lane's priority or the number of hardware queues); then every chain goes to that stream (_pin_chain_stream).  A chain is a single launch of ~17 ms and one is started
        # per step, so they follow each other on one stream; the goal rows go through another stream (_early_goals), never behind the next chain's kernel.
        c["stream_idx"] = self._chain_pin if self._chain_pin is not None else c["due_at"] % len(self._side)
        side = self._side[c["stream_idx"]]
        with torch.cuda.stream(side):
            side.wait_event(c["ready"])
            e0 = c["t0"] = torch.cuda.Event(enable_timing=True); e0.record(side)
            _, tq = self._chain_stage(np.arange(k), q.numpy())      # pinned staging: enqueued on the side stream, never waited for
            ar["qpos"][lo: lo + k] = tq[:, : self.nq]
            ar["qvel"][lo: lo + k].zero_()
            bufs, sp = self._arena_bufs(lo), ctypes.c_void_p(side.cuda_stream)
            c["pre"] = torch.cuda.Event()
            c["pre"].record(side)      # (everything of this chain but its settle launch is done when this fires: see _before_step_launch)
            # the arena's action rows stay zero: _set_action(np.zeros(20)) (manipulate.py:206-216).  ONE repeat launch (include/grx_capi.h grx_hand_step_repeat: bit-identical to the ten
            # launches, which beside the step kernel each wait for wave slots and end with their slowest world); GRX_HAND_FUSED_SETTLE=0: the ten launches (A/B)
            if self._fused_settle:
                _native.check(self._L.grx_hand_step_repeat(self._h, ctypes.byref(self.task), ctypes.byref(bufs), k, SETTLE_STEPS, sp))
            else:
                for _ in range(SETTLE_STEPS):
                    _native.check(self._L.grx_hand_step(self._h, ctypes.byref(self.task), ctypes.byref(bufs), k, 0, sp))
            # the settled object poses travel to a pinned host buffer behind the last settle launch: when the goals are drawn they are already there (round 4 read them with a
            # blocking .cpu() at that point: a one-block kernel and a copy that had to find a slot on a GPU saturated by the step kernel)
            c["obj_host"] = self._chain_obj[c["due_at"] % len(self._chain_obj)]
            c["obj_host"][:k].copy_(ar["qpos"][lo: lo + k, self._qa: self._qa + 7], non_blocking=True)
            c["event"] = torch.cuda.Event(enable_timing=True)
            c["event"].record(side)
            if self.chain_events is not None:
                self.chain_events.append((self._step_no, c["due_at"], k, e0, c["event"]))

    def _early_goals(self, c):
        """a chain that is committed at the end of THIS step: wait for it (it has had a whole step), read the settled object poses through the side stream,
        draw the goals of the worlds whose object stayed on the palm and park them in the arena -- all while the step kernel runs"""
        from .manipulate_spec import PALM_HEIGHT, sample_block_goal_batch

        lo, k, ar, side = c["lo"], c["k"], self._ar, self._goal_side
        with torch.cuda.stream(side):
            c["event"].synchronize()      # (started CHAIN_LOOKAHEAD steps ago: normally long done)
            self._pin_chain_stream(c)
            obj = self._obj_world(c["obj_host"][:k].numpy())      # the fp32 rows the kernel wrote, widened on the host (what .double() did on the device), in the MJCF's frame
            ok = obj[:, 2] > PALM_HEIGHT
            if ok.any():
                goals = sample_block_goal_batch([self.np_randoms[w] for w in c["worlds"][ok]], obj[ok], self.target_position, self.target_rotation, self._pquats)
                rows, tg = self._chain_stage(np.nonzero(ok)[0] + lo, goals.astype(np.float32))      # pinned staging: enqueued, never waited for
                ar["goal"][rows] = tg[:, : ar["goal"].shape[1]]
            c["ok"] = ok
            c["event"] = torch.cuda.Event()
            c["event"].record(side)

    def _pin_chain_stream(self, c):
        """measured device time of a finished chain -> which side stream the next chains use (see _start_chain).  Pinned when one stream's chains take less than 3/4 of every
        other stream's (medians of the last samples); released when the pinned stream stops being that fast.  OPT-IN (GRX_CHAIN_PIN=1): with the goal rows on their own stream
        (_goal_side, which alone measured +1.5 %) no stream was the fast one any more in the A/B (profiles/ab_r06_hand_chain_pin.txt: every chain 29 - 30 ms, pinned = round-robin),
        so the pinned state never occurred there."""
        if not self._chain_pin_enabled or c.get("t0") is None or len(self._side) < 2:
            return
        ms = c["t0"].elapsed_time(c["event"])
        hist = self._chain_ms.setdefault(c["stream_idx"], [])
        hist.append(ms)
        del hist[:-4]
        med = {i: sorted(h)[len(h) // 2] for i, h in self._chain_ms.items() if len(h) >= 2}
        if self._chain_pin is None:
            if len(med) == len(self._side):
                best = min(med, key=med.get)
                if all(med[best] < 0.75 * v for i, v in med.items() if i != best):
                    self._chain_pin, self._chain_pin_ref = best, min(v for i, v in med.items() if i != best)
        elif c["stream_idx"] == self._chain_pin and len(hist) >= 2 and med[self._chain_pin] > 0.9 * self._chain_pin_ref:
            self._chain_pin, self._chain_ms = None, {}      # no longer the fast one: measure again

    def _begin_overlapped_reset(self):
        if self.max_episode_steps is None:
            return None
        self._arena()
        self._step_no += 1
        rem = self.max_episode_steps - self._elapsed            # steps left before this one
        fresh = ~self._needs_reset & ~self._chain_started
        new = []
        # a chain is started CHAIN_LOOKAHEAD steps before its worlds hit the limit (start-up: as late as it must).  Round 5 measured the chains (tools/host_profile_hand.py): ten
        # dependent settle launches beside a 16.7 ms step kernel take 19 - 33 ms of device time, i.e. a chain started in step s is done before the kernel of step s + 2 starts.
        # The host's wait for it (13 ms per step inside Event.synchronize) is BACK-PRESSURE -- the host runs one step ahead of the device and needs the settled poses to draw
        # the goals -- not device idle time: a lookahead of 3 (one more step of slack, a staler warm start) measured 0.907 M against 0.918 M env-steps/s for 2.
        for k in range(1, self.CHAIN_LOOKAHEAD + 1):
            worlds = np.nonzero(fresh & ((rem <= 1) if k == 1 else (rem == k)))[0]
            if len(worlds):
                c = self._reserve_chain(worlds, self._step_no + k - 1)
                if c is not None:
                    new.append(c)
        return self._step_no, new

    def _before_step_launch(self, spec):
        """GRX_CHAIN_FIRST=1: the new chains' launches go to the device AHEAD of the step launch, which waits (on the device) until the chain's kernel is the next packet of its
        queue.  Beside a step launch that is still dispatching, a side stream's workgroups get no wave slot (profiles/chain_probe_r06.txt: 30 ms chains = the step launch's
        dispatch + their own 16 ms); started first, a chain's 164 workgroups take their slots at once and the step launch fills the rest of the chip."""
        if not self._chain_first:
            return
        main = torch.cuda.current_stream(self.device)
        for c in spec[1]:
            self._start_chain(c)
            main.wait_event(c["pre"])

    def _after_step_launch(self, spec):
        step_no, new = spec
        for c in new:                      # first: the new chains need every millisecond of the two steps they have
            if c.get("event") is None:
                self._start_chain(c)
        for c in self._chains:
            if c["due_at"] <= step_no and c["ok"] is None and not any(c is x for x in new):
                self._early_goals(c)

    def _finish_overlapped_reset(self, spec, done_idx):
        from .manipulate_spec import PALM_HEIGHT

        step_no = spec[0]
        mine = [c for c in self._chains if c["due_at"] <= step_no]
        self._chains = [c for c in self._chains if c["due_at"] > step_no]
        covered = np.zeros(self.num_envs, bool)
        ar, failed, gd = self._ar, [], self.goal.shape[1]
        for c in mine:
            if c["ok"] is None:                     # started in this very step (start-up): its goals have not been drawn yet
                self._early_goals(c)
            torch.cuda.current_stream(self.device).wait_event(c["event"])
            lo, k, ti = c["lo"], c["k"], c["ti"]
            a = _native.HandCommitArgsStruct()     # one kernel: state, outputs, goal, the packed row (its reward / success words stay the finished episode's), sticky status bits
            a.idx, a.k, a.nq, a.nv, a.obs_dim, a.goal_dim = ti.data_ptr(), k, self.nq, self.nv, self.obs_dim, gd
            for name in ("qpos", "qvel", "qacc_ws", "obs", "achieved", "palm", "goal", "packed", "status"):
                setattr(a, "s_" + name, ar[name][lo:].data_ptr())
                setattr(a, name, getattr(self, name).data_ptr())
            _native.check(self._L.grx_hand_commit_rows(ctypes.byref(a), self._stream()))
            failed.append(c["worlds"][~c["ok"]])
            covered[c["worlds"]] = True
            self._chain_started[c["worlds"]] = False
        if (covered & ~np.isin(np.arange(self.num_envs), done_idx)).any():
            raise RuntimeError("a settle chain finished for a world that is not at its time limit")
        failed = np.concatenate(failed) if failed else np.zeros(0, np.int64)
        if len(failed):                                   # object fell off the palm: the sequential retry loop takes over from the state just copied
            self._settle_until_on_palm(failed)
        rest = np.asarray(done_idx)[~covered[done_idx]]   # worlds without a chain (no room in the arena): the sequential path
        if len(rest):
            self.reset_attempts[rest] = 0
            self._settle_until_on_palm(rest)
        late = np.concatenate([failed, rest]).astype(np.int64)
        if len(late):
            self._sample_goals(late)
        self._elapsed[done_idx] = 0
        self._needs_reset[done_idx] = False
        return len(late) > 0        # settle launches ran on the main buffers: their reward / success / status words belong to the reset, not to this step

    def _cancel_chains(self):
        """reset() / set-state calls: settle chains in flight belong to episodes that no longer exist"""
        if getattr(self, "_ar", None) is not None:
            for c in self._chains:
                if c["event"] is not None:
                    c["event"].synchronize()
                for w, st in zip(c["worlds"], c.get("rng_states", ())):      # hand the chain's draws back to the worlds' generators
                    self.np_randoms[w].bit_generator.state = st
            self._chains, self._chain_started[:] = [], False

    def _launch_reward(self, ag, dg, out):
        from .manipulate_spec import ROTATION_THRESHOLD as RT

        _native.check(self._L.grx_manip_compute_reward(ag.data_ptr(), dg.data_ptr(), out.numel(), int(self.target_position == "ignore"),
                                                       int(self.target_rotation == "ignore"), int(self._objcfg["ignore_z_target_rotation"]),
                                                       self.distance_threshold, RT, int(self.reward_type == "sparse"),
                                                       out.data_ptr(), self._stream()))


HandManipulateVecEnv = HandBlockVecEnv   # the class covers the block and the pen objects
