"""Batched FrankaKitchen-v1 environments on the MI355X engine (host side, Python).

Vectorised drop-in for KitchenEnv (/root/reference/gymnasium_robotics/envs/franka_kitchen/kitchen_env.py, id FrankaKitchen-v1,
gymnasium_robotics/__init__.py:1117-1122) and the FrankaRobot it wraps (franka_env.py).  Per step: ONE launch of grx_kitchen_step_kernel
(velocity command -> position targets on the previous noisy joint reading, 40 physics substeps on the 124-geom scene with its five joint equalities,
the 59-vector observation with noise, the seven tasks' completion tests).  The observation noise -- 59 uniform(-1, 1) draws per world and step from each
world's numpy PCG64 stream -- and the task bookkeeping of KitchenEnv.step (tasks_to_complete / episode completions as per-world bit masks) run

* output="torch": ON THE DEVICE.  The streams' 128-bit states live in HBM and are advanced by grx_uniform_rows_device (bit-equal to Generator.uniform), the
  bookkeeping, the TimeLimit and the autoreset decision are one kernel (grx_kitchen_bookkeeping), the reset of the finished worlds is a masked forward launch:
  step() enqueues eight launches and returns device tensors -- no host synchronisation, no PCIe traffic (round 2: one 64 KB read-back and 3.9 MB of noise per step);
* output="numpy": on the host (C loop over the streams, one upload; the completion bits are read back), the path the fixtures pin.

Observations: {"observation": (N, 59), "achieved_goal": {task: (N, k)}, "desired_goal": {task: (N, k)}} -- the reference's dict-of-dicts with a
leading world axis.  reward = number of still-open tasks completed in this step; terminated when every task of the episode has been completed
(terminate_on_tasks_completed).
"""
import ctypes
import weakref
from typing import Optional

import numpy as np
import torch

from .. import _native
from ..core import KITCHEN_RERUN_CAPACITY, cost_order_alloc, cost_order_update, create_rerun_model, GoalVecEnv, OverflowLane, np_random
from ..mjcf import CompiledModel
from ..spaces import Box, Dict, batch_space
from .kitchen_spec import (INIT_QPOS, MAX_EPISODE_STEPS, OBS_DIM, OBS_ELEMENT_GOALS, OBS_ELEMENT_INDICES, TASKS, load_kitchen_model, make_kitchen_task, task_mask)


KITCHEN_SPLIT_PARTS = 1      # default of GRX_KITCHEN_SPLIT (see KitchenVecEnv.__init__): OFF -- measured (profiles/ab_r06_kitchen_split.txt, 16 384 worlds): 2 parts take the step kernel from 36.7 to 35.5 ms but the step still ends
                             # with the lane's launch (38.9 ms: its last entrant's re-run): 0.418 -> 0.419 M; 4 / 5 / 8 parts lose 2 - 5 % (every part rebuilds the skin list)


class KitchenVecEnv(GoalVecEnv):
    """autoreset_mode: "next_step" (Gymnasium >= 1.0 default), "same_step" or "disabled"; output: "numpy" (float64 arrays like the reference) or
    "torch" (the fp32 device tensors the kernel wrote; achieved goals are views of the qpos buffer)."""

    def __init__(self, env_id: str = "FrankaKitchen-v1", num_envs: int = 1, device: Optional[str] = None, tasks_to_complete=None,
                 terminate_on_tasks_completed: bool = True, remove_task_when_completed: bool = True, object_noise_ratio: float = 0.0005,
                 robot_noise_ratio: float = 0.01, max_episode_steps: Optional[int] = MAX_EPISODE_STEPS, autoreset_mode: str = "next_step", output: str = "numpy",
                 assets_root: Optional[str] = None, model: Optional[CompiledModel] = None, seed_offset: int = 0, skin_radius: float = 0.1):
        if env_id != "FrankaKitchen-v1":
            raise KeyError(f"unknown / unsupported kitchen env id {env_id}")
        if autoreset_mode not in ("next_step", "same_step", "disabled"):
            raise ValueError(f"unknown autoreset_mode {autoreset_mode}")
        self.env_id = env_id
        self.tasks = list(OBS_ELEMENT_GOALS) if tasks_to_complete is None else list(tasks_to_complete)
        self._all_mask = task_mask(self.tasks)                 # raises the reference's ValueError for an unknown task (kitchen_env.py:291-294)
        self.terminate_on_tasks_completed, self.remove_task_when_completed = bool(terminate_on_tasks_completed), bool(remove_task_when_completed)
        self.robot_noise_ratio, self.object_noise_ratio = float(robot_noise_ratio), float(object_noise_ratio)
        self.num_envs, self.max_episode_steps, self.autoreset_mode, self.output, self.seed_offset = int(num_envs), max_episode_steps, autoreset_mode, output, int(seed_offset)
        if not torch.cuda.is_available():
            raise RuntimeError("KitchenVecEnv needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device or "cuda:0")
        self.model = model or load_kitchen_model(assets_root)
        self.nq, self.nv, self.nu = self.model.dim("nq"), self.model.dim("nv"), self.model.dim("nu")
        self.obs_dim = OBS_DIM
        self._L = _native.lib()
        H, I, F = self.model.pack()
        self._h = _native.acquire_model(H, I, F, self.device.index or 0)   # shared with every other environment of the same compiled tables (reference-counted)
        self.lds_bytes = self._L.grx_model_lds_bytes(self._h)
        self._h_big = create_rerun_model(self._L, self.model, self.device.index or 0, capacity=KITCHEN_RERUN_CAPACITY)    # larger tables for the worlds that overflow a capacity
        self.task = make_kitchen_task(self.model, self.robot_noise_ratio, self.object_noise_ratio)
        self._noisy = self.robot_noise_ratio != 0.0 or self.object_noise_ratio != 0.0
        n, d = self.num_envs, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=d)
        self.qpos, self.qvel, self.qacc_ws, self.last_qpos = z(n, self.nq), z(n, self.nv), z(n, self.nv), z(n, 9)
        self.action, self.obs, self.noise = z(n, self.nu), z(n, self.obs_dim), z(n, self.obs_dim)
        self.completed, self.status, self.mask = z(n, dtype=torch.int32), z(n, dtype=torch.int32), torch.ones(n, dtype=torch.uint8, device=d)
        self._noise_host = torch.empty(n, self.obs_dim, dtype=torch.float32, pin_memory=True)
        # broad-phase skin lists (csrc/grx_engine.h, grx_collision): one zeroed HBM row per world that the step kernel owns from then on -- the candidate
        # pairs within `skin_radius` of passing the bounding test and the geom positions they were collected at (16 KB per world for the 3 736 pairs)
        self.skin_radius = float(skin_radius)
        self._skin = z(n, 4 + 3 * self.model.dim("ngeom") + len(self.model.tables["devpair"]), dtype=torch.int32) if self.skin_radius > 0.0 else None
        self._bufs, self._bufs_masked = self._make_bufs(None), self._make_bufs(self.mask)
        # Cost-ordered dispatch (include/grx_capi.h grx_kitchen_buffers.order / .cost): a kitchen world takes 2.9 ms at the median and 11.5 ms at the worst, and what it costs persists
        # from step to step (which fixtures the arm touches); in index order the launch's slowest worlds start anywhere.  Measured at 16 384 worlds (profiles/ab_r06_cost_order.txt): step kernel 39.5 -> 38.3 ms, 0.4035 -> 0.410 M (+1.7 %: the lane's launch, 39.5 ms, is what the step waits for then); the kitchen + hammer batch of cfg 5: 0.399 -> 0.436 M (+9 %).  GRX_KITCHEN_BALANCE=0: off (A/B).
        import os
        self.cost = self.cost_ema = self.order = None
        self.balance = os.environ.get("GRX_KITCHEN_BALANCE", "1") != "0" and cost_order_alloc(self, n, d, self._bufs, self._bufs_masked)
        # no dropped contacts: the worlds that exceed a table capacity of the fast kernel are stepped on larger tables (core.OverflowLane)
        # (32 polling workgroups: with the fast tables cut to 128 rows / 1 280 words / 24 contacts more worlds ENTER the lane per step; measured 16 -> 0.357 M, 32 -> 0.374 M, 48 -> 0.365 M, 96 -> 0.341 M env-steps/s)
        # (round 6, with the cost-ordered dispatch: a world leaves the lane one step after it was last near a capacity and enters it at 0.9 of one -- fewer worlds on the slower large-table kernel: ttl 4 -> 1: 0.409 -> 0.418 M, margin 0.8 -> 0.9: 0.421 M; ttl 0 and margin 0.7 lose: profiles/ab_r06_kitchen_ttl*.txt)
        self.lane = OverflowLane(n, d, self.model, self._lane_make_bufs(), poll_grid=32, ttl=1, margin=0.9) if self._h_big is not None else None
        # SPLIT STEP (include/grx_capi.h grx_kitchen_buffers.split_parts; see AdroitVecEnv): a kitchen launch ended 15 % after the mean of its wave slots (profiles/tail_probe_r06.txt).
        # GRX_KITCHEN_SPLIT=P (1: off).  Bit-identical to the plain launch (tests/test_gpu_kitchen.py::test_split_step_is_the_plain_step).
        self._split = max(1, min(8, int(os.environ.get("GRX_KITCHEN_SPLIT", KITCHEN_SPLIT_PARTS if n > 2048 else 1)))) if n >= 64 else 1
        if self._split > 1:
            stride = -(-(self.nq + 2 * self.nv) // 16) * 16
            self._split_rows, self._split_state = z(n, stride), z(n, 4, dtype=torch.int32)
            for b in (self._bufs, self._bufs_masked):
                b.split_rows, b.split_state, b.split_stride, b.split_parts = self._split_rows.data_ptr(), self._split_state.data_ptr(), stride, self._split
            if self.lane is not None:
                self.lane._fast_grid *= self._split      # (polling workgroups of the standing lane wait for EVERY workgroup of the fast launch)
        self.single_action_space = Box(-1.0, 1.0, (self.nu,), np.float64)                     # franka_env.py:88
        goal_space = Dict({t: Box(-np.inf, np.inf, OBS_ELEMENT_GOALS[t].shape, np.float64) for t in self.tasks})
        self.single_observation_space = Dict(dict(desired_goal=goal_space, achieved_goal=Dict(dict(goal_space)),
                                                  observation=Box(-np.inf, np.inf, (self.obs_dim,), np.float64)))   # kitchen_env.py:315-338
        self.action_space = batch_space(self.single_action_space, n)
        self.observation_space = batch_space(self.single_observation_space, n)
        self._init_qpos = torch.from_numpy(INIT_QPOS.astype(np.float32)).to(d)
        self._goal_np = {t: np.tile(OBS_ELEMENT_GOALS[t], (n, 1)) for t in self.tasks}
        self._goal_t = {t: torch.from_numpy(self._goal_np[t].astype(np.float32)).to(d) for t in self.tasks}
        self.tasks_to_complete = np.full(n, self._all_mask, np.int64)          # per-world bit masks (bit k = TASKS[k])
        self.episode_task_completions = np.zeros(n, np.int64)
        # output="torch": bookkeeping, TimeLimit, autoreset and the noise streams on the device (GRX_KITCHEN_HOST_PATH=1: the round-2 host path, for A/B)
        import os
        self._device_path = self.output == "torch" and os.environ.get("GRX_KITCHEN_HOST_PATH") is None
        if self._device_path:
            i32 = lambda: torch.zeros(n, dtype=torch.int32, device=d)
            u8 = lambda: torch.zeros(n, dtype=torch.uint8, device=d)
            self.d_ttc, self.d_epi, self.d_elapsed, self.d_stepdone = i32(), i32(), i32(), i32()
            self.d_reward, self.d_term, self.d_trunc, self.d_needs, self.d_resetnow = z(n), u8(), u8(), u8(), u8()
            self.d_rng = torch.zeros(n, 4, dtype=torch.int64, device=d)          # the worlds' PCG64 states (state_hi, state_lo, inc_hi, inc_lo), uint64 bit patterns
            self.final_obs, self.final_qpos, self.d_final_info = z(n, self.obs_dim), z(n, self.nq), torch.zeros(n, 3, dtype=torch.int32, device=d)
            self._book = _native.KitchenBookStruct()
            for name, t in (("completed", self.completed), ("stepped", self.mask), ("tasks_to_complete", self.d_ttc), ("episode_completions", self.d_epi), ("elapsed", self.d_elapsed),
                            ("step_completions", self.d_stepdone), ("reward", self.d_reward), ("terminated", self.d_term), ("truncated", self.d_trunc), ("needs_reset", self.d_needs),
                            ("reset_now", self.d_resetnow), ("qpos", self.qpos), ("qvel", self.qvel), ("qacc_ws", self.qacc_ws), ("init_qpos", self._init_qpos)):
                setattr(self._book, name, t.data_ptr())
            self._book.nq, self._book.nv, self._book.all_mask = self.nq, self.nv, int(self._all_mask)
            self._book.max_steps = int(self.max_episode_steps or 0)
            self._book.remove_when_completed, self._book.terminate_when_completed = int(self.remove_task_when_completed), int(self.terminate_on_tasks_completed)
            self._book.mode = {"disabled": 0, "next_step": 1, "same_step": 2}[self.autoreset_mode]
            self._book.final_info = self.d_final_info.data_ptr()
        self._seed_worlds([None] * n)
        self._elapsed = np.zeros(n, np.int64)
        self._needs_reset = np.zeros(n, bool)
        self._has_reset = False
        self.step_events = []      # (start, end) HIP events around the whole launch group of a step: fast kernel + the overflow lane's launches
        self.kernel_events = None

    def _lane_make_bufs(self):
        me = weakref.ref(self)      # (the lane must not keep the environment alive: its native model slots are released by __del__)
        return lambda m: me()._make_bufs(m)

    def _make_bufs(self, mask):
        b = _native.KitchenBuffersStruct()
        for name in ("qpos", "qvel", "qacc_ws", "last_qpos", "action", "obs", "completed", "status"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.noise = self.noise.data_ptr() if self._noisy else None
        b.mask = None if mask is None else mask.data_ptr()
        if self._skin is not None:
            b.skin, b.skin_stride, b.skin_radius = self._skin.data_ptr(), self._skin.shape[1], self.skin_radius
        return b

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ per-world numpy PCG64 streams, advanced in C
    def _seed_worlds(self, seeds):
        st, mask = np.zeros((self.num_envs, 4), np.uint64), (1 << 64) - 1
        for i, sd in enumerate(seeds):
            s = np_random(sd)[0].bit_generator.state["state"]
            st[i] = [s["state"] >> 64, s["state"] & mask, s["inc"] >> 64, s["inc"] & mask]
        self._rng_state = st
        self._noise_uploaded = None
        if self._device_path:
            self.d_rng.copy_(torch.from_numpy(st.view(np.int64)))      # the streams live on the device from here on (grx_uniform_rows_device)
            return
        self._refill_noise(None)

    # Every world's NEXT 59 draws are always sitting in the pinned `_noise_host` rows: an observation (step or reset) uploads its worlds' rows, and
    # the rows are refilled from the streams while the kernel that consumes them runs -- the streams advance in exactly the order of the observations.
    def _draw_noise(self, idx=None):
        """upload the pending draws of the listed worlds (all: None) into self.noise (async); _refill_noise(idx) must follow the launch"""
        if not self._noisy:
            return
        if idx is None:
            self.noise.copy_(self._noise_host, non_blocking=True)
        else:
            ti = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64))
            self.noise[ti.to(self.device)] = self._noise_host[ti].to(self.device)
        self._noise_uploaded = torch.cuda.Event()
        self._noise_uploaded.record(torch.cuda.current_stream(self.device))

    def _refill_noise(self, idx=None):
        """draw the next 59 uniform(-1, 1) values of the listed worlds' streams (C, bit-exact with Generator.uniform) into their pinned rows"""
        if not self._noisy:
            return
        if getattr(self, "_noise_uploaded", None) is not None:
            self._noise_uploaded.synchronize()          # the copy that reads the rows must be done before they are overwritten (it precedes the kernel)
            self._noise_uploaded = None
        if idx is None:
            _native.check(self._L.grx_sample_uniform_rows(self._rng_state.ctypes.data, None, self.num_envs, self.obs_dim, self._noise_host.data_ptr()))
        else:
            idx64 = np.ascontiguousarray(idx, dtype=np.int64)
            rows = np.empty((len(idx64), self.obs_dim), np.float32)
            _native.check(self._L.grx_sample_uniform_rows(self._rng_state.ctypes.data, idx64.ctypes.data, len(idx64), self.obs_dim, rows.ctypes.data))
            self._noise_host[torch.from_numpy(idx64)] = torch.from_numpy(rows)

    def _launch(self, bufs, forward_only):
        def fast(b):
            timed = self.kernel_events is not None and not forward_only
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _native.check(self._L.grx_kitchen_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, int(forward_only), self._stream()))
            if timed:
                e1.record()
                self.kernel_events.append((e0, e1))

        if self.lane is not None and not forward_only:
            large = lambda b: _native.check(self._L.grx_kitchen_step(self._h_big, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, 0, self._stream()))
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
            cost_order_update(self)

    # ------------------------------------------------------------------ reset (kitchen_env.py:425-437 -> FrankaRobot.reset -> reset_model, franka_env.py:133-139)
    def _reset_worlds(self, idx):
        if len(idx) == 0:
            return None
        ti = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(self.device)
        self.qpos[ti] = self._init_qpos
        self.qvel.index_fill_(0, ti, 0.0)
        self.qacc_ws.index_fill_(0, ti, 0.0)
        which = idx if len(idx) < self.num_envs else None
        self._draw_noise(which)
        self.mask.zero_()
        self.mask.index_fill_(0, ti, 1)
        self._launch(self._bufs_masked, True)
        self._refill_noise(which)
        self.tasks_to_complete[idx] = self._all_mask
        self.episode_task_completions[idx] = 0
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False
        return ti

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            seeds = [seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed)
            self._seed_worlds(seeds)
        with torch.cuda.device(self.device):
            if self._device_path:
                self.qpos.copy_(self._init_qpos.expand_as(self.qpos))
                self.qvel.zero_(); self.qacc_ws.zero_()
                self.mask.fill_(1)
                self._device_noise(self.mask)
                self._launch(self._bufs_masked, True)
                self.d_ttc.fill_(int(self._all_mask))
                for t in (self.d_epi, self.d_elapsed, self.d_stepdone, self.d_needs, self.d_resetnow, self.d_term, self.d_trunc):
                    t.zero_()
                self._has_reset = True
                return self._obs_dict(), self._device_info()
            self._reset_worlds(np.arange(self.num_envs))
        self._has_reset = True
        return self._obs_dict(), self._info(np.zeros(self.num_envs, np.int64))

    # ------------------------------------------------------------------ the device path (output="torch")
    def _device_noise(self, mask):
        """the next 59 draws of the masked worlds' streams into their rows of self.noise (grx_uniform_rows_device: bit-equal to Generator.uniform(-1, 1))"""
        if self._noisy:
            _native.check(self._L.grx_uniform_rows_device(self.d_rng.data_ptr(), mask.data_ptr(), self.num_envs, self.obs_dim, self.noise.data_ptr(), self._stream()))

    def _device_info(self):
        return dict(tasks_to_complete=self.d_ttc, step_task_completions=self.d_stepdone, episode_task_completions=self.d_epi)

    def set_elapsed(self, elapsed):
        """episode positions of the worlds (benchmarks stagger the episodes so that every step carries its share of resets)"""
        self._elapsed[:] = np.asarray(elapsed, dtype=np.int64)
        if self._device_path:
            self.d_elapsed.copy_(torch.from_numpy(np.asarray(elapsed, dtype=np.int32)))

    def _step_launch_device(self):
        if self.autoreset_mode == "next_step":
            torch.bitwise_xor(self.d_needs, 1, out=self.mask)      # the worlds that finished in the previous step wait for their reset at the end of this one
        else:
            self.mask.fill_(1)
        self._device_noise(self.mask)
        self._launch(self._bufs_masked, False)

    def _step_finish_device(self):
        info = {}
        same = self.autoreset_mode == "same_step"
        if same:      # the rows of the finished episodes, before the bookkeeping kernel rewinds their state (which worlds: _final_obs, known to the device only)
            self.final_obs.copy_(self.obs); self.final_qpos.copy_(self.qpos)
        _native.check(self._L.grx_kitchen_bookkeeping(ctypes.byref(self._book), self.num_envs, self._stream()))
        if self.autoreset_mode != "disabled":
            if same:
                done = (self.d_term | self.d_trunc).view(torch.bool)
                fq = self.final_qpos
                info["final_obs"] = {"observation": self.final_obs, "achieved_goal": {t: fq[:, OBS_ELEMENT_INDICES[t][0]: OBS_ELEMENT_INDICES[t][-1] + 1] for t in self.tasks},
                                     "desired_goal": {t: self._goal_t[t] for t in self.tasks}}
                info["_final_obs"] = done          # gymnasium's vector convention: the mask of the worlds whose final_obs rows are valid
                fi = self.d_final_info
                info["final_info"] = dict(tasks_to_complete=fi[:, 0], step_task_completions=fi[:, 1], episode_task_completions=fi[:, 2])
                info["_final_info"] = done
            self.mask.copy_(self.d_resetnow)
            self._device_noise(self.mask)
            keep = self.status.clone()
            self._launch(self._bufs_masked, True)          # mj_forward + _get_obs of the worlds that were reset (every other workgroup returns at once)
            self.status.copy_((keep & 0xFFFF) | (self.status & -65536))
        info.update(self._device_info())
        return self._obs_dict(), self.d_reward, self.d_term.view(torch.bool), self.d_trunc.view(torch.bool), self._status_info(info)

    # ------------------------------------------------------------------ step (kitchen_env.py:386-423)
    def step(self, actions):
        """= step_launch(actions) + step_finish(): the first half only ENQUEUES (action copy, noise upload, the step kernel, the host draws of the next
        observation's noise while the kernel runs), the second half reads the completion bits back (the one host sync of this family: the task bookkeeping
        of KitchenEnv.step decides termination) and does the autoresets.  A caller that drives several environments from one thread launches them all
        before it finishes any (bench.py --workload mixed: kitchen and Adroit kernels share the GPU, no host threads)."""
        self.step_launch(actions)
        return self.step_finish()

    def step_launch(self, actions):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        a = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions, dtype=np.float32))
        if tuple(a.shape) != (self.num_envs, self.nu):
            raise ValueError(f"Action dimension mismatch. Expected {(self.num_envs, self.nu)}, found {tuple(a.shape)}")
        self.action.copy_(a.to(torch.float32), non_blocking=True)
        if self._device_path:
            with torch.cuda.device(self.device):
                self._step_launch_device()
            self._pending_finish = None
            return
        with torch.cuda.device(self.device):
            pending = np.nonzero(self._needs_reset)[0] if self.autoreset_mode == "next_step" else np.zeros(0, np.int64)
            stepped = ~self._needs_reset
            which = None if not len(pending) else np.nonzero(stepped)[0]
            self._draw_noise(which)
            if len(pending):
                self.mask.fill_(1)
                self.mask.index_fill_(0, self._stage_idx(pending), 0)      # (pinned staging + index_fill_: nothing here waits for the running kernel)
                self._launch(self._bufs_masked, False)
            else:
                self._launch(self._bufs, False)
            self._refill_noise(which)          # host draws for the NEXT observation of these worlds while the kernel runs
        self._pending_finish = (pending, stepped)

    def step_finish(self):
        if self._device_path:
            with torch.cuda.device(self.device):
                return self._step_finish_device()
        pending, stepped = self._pending_finish
        self._pending_finish = None
        info = {}
        with torch.cuda.device(self.device):
            done_bits = self.completed.cpu().numpy().astype(np.int64)
            # compute_reward over the tasks still open, bookkeeping of KitchenEnv.step
            step_done = np.where(stepped, done_bits & self.tasks_to_complete, 0)
            reward = np.array([bin(int(x)).count("1") for x in step_done], dtype=np.float64) if self.num_envs <= 64 else _popcount(step_done).astype(np.float64)
            if self.remove_task_when_completed:
                self.tasks_to_complete &= ~step_done
            self.episode_task_completions |= step_done
            terminated = np.zeros(self.num_envs, bool)
            if self.terminate_on_tasks_completed:
                terminated = stepped & (self.episode_task_completions == self._all_mask)
            self._elapsed[stepped] += 1
            truncated = np.zeros(self.num_envs, bool)
            if self.max_episode_steps is not None:
                truncated = stepped & (self._elapsed >= self.max_episode_steps)
            step_info = self._info(step_done)
            if len(pending):
                self._reset_worlds(pending)
                reward[pending] = 0.0
                step_info = self._info(step_done)
            done = terminated | truncated
            if self.autoreset_mode == "same_step" and done.any():
                idx = np.nonzero(done)[0]
                info["final_obs"] = self._obs_dict(rows=idx)
                info["final_info"] = {k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in step_info.items()}
                keep = self.status.clone()
                self._reset_worlds(idx)
                self.status.copy_((keep & 0xFFFF) | (self.status & -65536))
                fresh = np.asarray(step_done).copy()
                fresh[idx] = 0
                step_info = self._info(fresh)          # the reset worlds report their new episode (gymnasium's same-step convention); the last step's info is in final_info
            elif self.autoreset_mode == "next_step":
                self._needs_reset |= done
        info.update(step_info)
        r = torch.from_numpy(reward).to(self.device) if self.output == "torch" else reward
        if self.output == "torch":
            return self._obs_dict(), r, torch.from_numpy(terminated), torch.from_numpy(truncated), self._status_info(info)
        return self._obs_dict(), r, terminated, truncated, self._status_info(info)

    def _info(self, step_done):
        """the reference's info lists as per-world bit masks over kitchen_spec.TASKS (bit k = TASKS[k]); task_names(mask) turns one into the list of names"""
        return dict(tasks_to_complete=self.tasks_to_complete.copy(), step_task_completions=np.asarray(step_done, dtype=np.int64).copy(),
                    episode_task_completions=self.episode_task_completions.copy())

    @staticmethod
    def task_names(mask: int):
        return [t for k, t in enumerate(TASKS) if (int(mask) >> k) & 1]

    def _obs_dict(self, rows=None):
        if self.output == "torch":
            sel = (lambda t: t) if rows is None else (lambda t: t[torch.from_numpy(np.asarray(rows)).to(self.device)])
            qp = sel(self.qpos)
            return {"observation": sel(self.obs), "achieved_goal": {t: qp[:, OBS_ELEMENT_INDICES[t][0]: OBS_ELEMENT_INDICES[t][-1] + 1] for t in self.tasks},
                    "desired_goal": {t: sel(self._goal_t[t]) for t in self.tasks}}
        sel = (lambda a: a) if rows is None else (lambda a: a[rows])
        qp = sel(self.qpos.double().cpu().numpy())
        return {"observation": sel(self.obs.double().cpu().numpy()), "achieved_goal": {t: qp[:, OBS_ELEMENT_INDICES[t]] for t in self.tasks},
                "desired_goal": {t: sel(self._goal_np[t]) for t in self.tasks}}

    # ------------------------------------------------------------------ GoalEnv API (kitchen_env.py:340-354), batched
    def compute_reward(self, achieved_goal, desired_goal, info=None):
        """number of the env's tasks whose achieved goal is within BONUS_THRESH of the desired one, per leading index (the reference counts over its
        CURRENT tasks_to_complete, a per-env set; pass info={"tasks_to_complete": masks} to restrict the count the same way)"""
        from .kitchen_spec import BONUS_THRESH

        total = None
        masks = None if not info or "tasks_to_complete" not in info else np.asarray(info["tasks_to_complete"])
        for t in self.tasks:
            d = np.linalg.norm(np.asarray(achieved_goal[t], dtype=np.float64) - np.asarray(desired_goal[t], dtype=np.float64), axis=-1)
            c = (d < BONUS_THRESH).astype(np.float64)
            if masks is not None:
                c = c * ((masks >> TASKS.index(t)) & 1)
            total = c if total is None else total + c
        return total

    def compute_terminated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(next(iter(achieved_goal.values()))).shape[:-1], bool)

    def compute_truncated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(next(iter(achieved_goal.values()))).shape[:-1], bool)

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


def _popcount(x):
    x = np.asarray(x, dtype=np.int64)
    return sum((x >> k) & 1 for k in range(len(TASKS)))
