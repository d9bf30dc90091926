"""Batched Adroit hand environments on the MI355X engine (host side, Python).

Vectorised drop-in for AdroitHandHammerEnv / AdroitHandDoorEnv / AdroitHandPenEnv / AdroitHandRelocateEnv
(/root/reference/gymnasium_robotics/envs/adroit_hand/adroit_{hammer,door,pen,relocate}.py, ids AdroitHand<Task>[Sparse]-v2 and their -v1 aliases,
gymnasium_robotics/__init__.py:1078-1115).  Unlike the goal-conditioned families these are plain Envs: observation = one vector per world (46 / 39 /
45 / 39, not a dict), reward float, terminated always False, info["success"].
Per-step work = ONE launch of grx_adroit_step_kernel (action scaling, 5 physics substeps incl. the noslip post-solver, observation, reward,
success).  Host side of a reset: the PCG64 draws of reset_model per world; the model edits the reference makes there (model.body_pos /
body_quat / site_pos) are per-world state here (`shift`, `target`: adroit_spec.sample_reset).
"""
import ctypes
import os
import weakref
from typing import Optional

import numpy as np
import torch

from .. import _native
from ..core import cost_order_alloc, cost_order_update, create_rerun_model, GoalVecEnv, OverflowLane, PinnedStager, np_random
from ..mjcf import CompiledModel
from ..spaces import Box, batch_space
from .adroit_spec import IDENTITY_SHIFT, MAX_EPISODE_STEPS, SPECS, action_scaling, group_shift, load_adroit_model, make_adroit_task, parse_adroit_id, sample_reset_batch


ADROIT_SPLIT_PARTS = 5      # default of GRX_ADROIT_SPLIT for batches of more than 2 048 worlds = one part per substep (see AdroitVecEnv.__init__; profiles/ab_r06_adroit_split*.txt, 16 384 worlds, 1 / 2 / 3 / 5 parts:
                            # hammer 1.477 / 1.556 / 1.589 / 1.606 M, pen 2.330 / 2.463 / 2.530 / 2.576 M, relocate 1.228 / 1.200 / 1.229 / 1.278 M, door 1.395 / 1.431 / 1.458 / 1.468 M -- the last two end with their lane's launch)


class AdroitVecEnv(GoalVecEnv):
    """autoreset_mode: "next_step" (Gymnasium >= 1.0 default), "same_step" or "disabled"; output: "numpy" (float64 arrays like the reference)
    or "torch" (the fp32 device tensors the kernel wrote)."""

    CKPT_SKIP = GoalVecEnv.CKPT_SKIP + ("_ahead",)      # staging rows of the overlapped reset: written and consumed inside one step() call, never state

    def __init__(self, env_id: str = "AdroitHandHammer-v2", num_envs: int = 1, device: Optional[str] = None, reward_type: Optional[str] = None,
                 max_episode_steps: Optional[int] = MAX_EPISODE_STEPS, autoreset_mode: str = "next_step", output: str = "numpy",
                 assets_root: Optional[str] = None, model: Optional[CompiledModel] = None, seed_offset: int = 0):
        self.task_name, rt = parse_adroit_id(env_id)
        self.spec = SPECS[self.task_name]
        self.env_id, self.reward_type = env_id, reward_type or rt
        if self.reward_type not in ("dense", "sparse"):
            raise ValueError(f"Unknown reward type, expected `dense` or `sparse` but got {self.reward_type}")   # adroit_hammer.py:226-229
        if autoreset_mode not in ("next_step", "same_step", "disabled"):
            raise ValueError(f"unknown autoreset_mode {autoreset_mode}")
        self.num_envs, self.max_episode_steps, self.autoreset_mode, self.output, self.seed_offset = int(num_envs), max_episode_steps, autoreset_mode, output, int(seed_offset)
        if not torch.cuda.is_available():
            raise RuntimeError("AdroitVecEnv needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device or "cuda:0")
        self.model = model or load_adroit_model(self.task_name, assets_root)
        self.nq, self.nv, self.nu = self.model.dim("nq"), self.model.dim("nv"), self.model.dim("nu")
        self.obs_dim = self.spec["obs_dim"]
        self._L = _native.lib()
        H, I, F = self.model.pack()
        self._h = _native.acquire_model(H, I, F, self.device.index or 0)   # shared with every other environment of the same compiled tables (reference-counted)
        self.lds_bytes = self._L.grx_model_lds_bytes(self._h)
        self._h_big = create_rerun_model(self._L, self.model, self.device.index or 0)    # larger tables for the worlds that overflow a capacity (core.RERUN_CAPACITY)
        self.task = make_adroit_task(self.model, self.reward_type, self.task_name)
        n, d = self.num_envs, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=d)
        self.qpos, self.qvel, self.qacc_ws = z(n, self.nq), z(n, self.nv), z(n, self.nv)
        self.shift = torch.from_numpy(np.tile(IDENTITY_SHIFT.astype(np.float32), (n, 1))).to(d)
        self.target = z(n, 3) if self.task_name == "relocate" else None
        self.action, self.obs, self.reward = z(n, self.nu), z(n, self.obs_dim), z(n)
        self.success, self.status, self.mask = z(n, dtype=torch.uint8), z(n, dtype=torch.int32), torch.ones(n, dtype=torch.uint8, device=d)
        am, ar = action_scaling(self.model)
        self._act_mean, self._act_rng = torch.from_numpy(am.astype(np.float32)).to(d), torch.from_numpy(ar.astype(np.float32)).to(d)
        self._bufs, self._bufs_masked = self._make_bufs(None), self._make_bufs(self.mask)
        # Cost-ordered dispatch (include/grx_capi.h grx_adroit_buffers.order / .cost; see KitchenVecEnv): the step launch starts the worlds that took longest in the previous launches
        # first.  Measured at 16 384 worlds (profiles/ab_r06_cost_order.txt): hammer 1.348 -> 1.441 M (+6.9 %), door 1.230 -> 1.326 M (+7.8 %), relocate 1.088 -> 1.203 M (+10.6 %);
        # the pen's worlds all cost the same and the launch only loses the locality of the index order (2.30 -> 2.18 M): off there.  GRX_ADROIT_BALANCE=0 / 1 overrides (A/B, tests).
        self.cost = self.cost_ema = self.order = None
        self.balance_alpha = 0.3      # weight of the newest sample in the moving average (profiles/ab_r06_cost_order_lane.txt: 0.3 against 0.1: hammer +2 %, relocate +3.5 %, door -1.5 %)
        self.balance = os.environ.get("GRX_ADROIT_BALANCE", "0" if self.task_name == "pen" else "1") != "0" and cost_order_alloc(self, n, d, self._bufs, self._bufs_masked)
        self._compact_resets = os.environ.get("GRX_ADROIT_COMPACT_RESET", "1") != "0"      # (0: the masked whole-grid forward launch of rounds 3 - 4; A/B, tests)
        # no dropped contacts: the worlds that exceed a table capacity of the fast kernel are stepped on larger tables (core.OverflowLane)
        self.lane = OverflowLane(n, d, self.model, self._lane_make_bufs(), mode="lane" if self.task_name in ("door", "relocate") else "entry", lane_first=True, ttl=1) if self._h_big is not None else None      # (ttl 4 -> 1 with the cost-ordered dispatch: door +1.8 %, relocate +1 %: profiles/ab_r06_lane_ttl.txt)     # hammer / pen: no overflow in 4 M world-steps
        # SPLIT STEP (include/grx_capi.h grx_adroit_buffers.split_parts): P workgroups per world, each running its share of the 5 substeps, the world handed on through carrier rows.  An Adroit
        # world's cost changes from step to step, the cost order predicts it poorly, and a launch ended 15 - 40 % after the mean of its wave slots (profiles/tail_probe_r06.txt).
        # Bit-identical to the plain launch (tests/test_gpu_adroit.py::test_split_step_is_the_plain_step).  GRX_ADROIT_SPLIT=P (1: off).
        self._split = max(1, min(int(self.task.n_substeps), 8, int(os.environ.get("GRX_ADROIT_SPLIT", ADROIT_SPLIT_PARTS if n > 2048 else 1)))) if n >= 64 else 1
        if self._split > 1:
            stride = -(-(self.nq + 2 * self.nv) // 16) * 16
            self._split_rows, self._split_state = z(n, stride), z(n, 4, dtype=torch.int32)
            for b in (self._bufs, self._bufs_masked):
                b.split_rows, b.split_state, b.split_stride, b.split_parts = self._split_rows.data_ptr(), self._split_state.data_ptr(), stride, self._split
            if self.lane is not None:
                self.lane._fast_grid *= self._split      # (polling workgroups of the standing lane wait for EVERY workgroup of the fast launch: grx_overflow_lane.progress_total)
        self.single_action_space = Box(-1.0, 1.0, (self.nu,), np.float32)                      # adroit_hammer.py:231-233
        self.single_observation_space = Box(-np.inf, np.inf, (self.obs_dim,), np.float64)      # :205-207
        self.action_space = batch_space(self.single_action_space, n)
        self.observation_space = batch_space(self.single_observation_space, n)
        self._init_qpos = torch.from_numpy(self.model.tables["qpos0"].astype(np.float32)).to(d)   # MujocoEnv.init_qpos [3P]: data.qpos after mj_resetData
        # the model edit of each world as the reference's get_env_state reports it: body_pos (hammer board, door frame, relocate ball) or body_quat (pen target)
        edit0 = self.model.info["shift_quat0"] if self.task_name == "pen" else self.model.info["shift_pos0"]
        # hammer / door / relocate: reset_model's draws are uniforms plus additions -- they are made ON THE DEVICE from device-resident PCG64 streams
        # (grx_adroit_sample_resets_device, bit-equal to numpy), and the fp64 rows get_env_state reports live in HBM next to them.  The pen's draws go through
        # euler2quat (sin / cos: no bit-exact device twin of numpy's libm) and stay per-world numpy generators on the host, staged through pinned memory.
        self._device_draws = self.task_name in ("hammer", "door", "relocate")
        # Overlapped same-step reset (include/grx_capi.h, grx_adroit_commit_rows): the Adroit tasks never terminate, so the worlds a step truncates are known before its launch;
        # their draws and the reset-time forward pass (0.4 ms: one world's dependent chain, in line behind a 12 ms step kernel) run on a side stream beside the step kernel into
        # staged rows, one small kernel commits them behind it (AdroitHandPen: the host draws, made at the same point of every world's stream, travel through pinned memory).  GRX_ADROIT_AHEAD_RESET=0: the in-line path (A/B, tests).
        self._ahead = None
        if self.num_envs > 1 and os.environ.get("GRX_ADROIT_AHEAD_RESET", "1") != "0":
            A = dict(qpos=self._init_qpos.expand(n, -1).contiguous(), qvel=z(n, self.nv), qacc_ws=z(n, self.nv), shift=self.shift.clone(), obs=z(n, self.obs_dim), reward=z(n),
                     success=z(n, dtype=torch.uint8), status=z(n, dtype=torch.int32))
            if self.target is not None:
                A["target"] = z(n, 3)
            self._ahead = A
            b = _native.AdroitBuffersStruct()
            for name in ("qpos", "qvel", "qacc_ws", "shift", "obs", "reward", "success", "status"):
                setattr(b, name, A[name].data_ptr())
            b.target = A["target"].data_ptr() if "target" in A else None
            b.action, b.act_mean, b.act_rng, b.mask = self.action.data_ptr(), self._act_mean.data_ptr(), self._act_rng.data_ptr(), None
            self._ahead_bufs = b
            self._ahead_stream = torch.cuda.Stream(device=d)
        if self._device_draws:
            self._edit_dev = torch.from_numpy(np.tile(np.asarray(edit0, dtype=np.float64), (n, 1))).to(d)
            self._target_dev = torch.zeros(n, 3, dtype=torch.float64, device=d) if self.task_name == "relocate" else None
            self._shift_pos0 = np.ascontiguousarray(self.model.info["shift_pos0"], dtype=np.float64)
            self._seed_worlds([None] * n)
        else:
            self._model_edit = np.tile(np.asarray(edit0, dtype=np.float64), (n, 1))
            self.np_randoms = [np_random(None)[0] for _ in range(n)]
        self._elapsed = np.zeros(n, np.int64)
        self._needs_reset = np.zeros(n, bool)
        self._has_reset = False
        self.step_events = []      # (start, end) HIP events around the whole launch group of a step: fast kernel + the overflow lane's launches
        self.kernel_events = None
        self._stage = PinnedStager(n, 10, self.device)

    def _seed_worlds(self, seeds):
        """one numpy PCG64 per world, seeded like gymnasium.utils.seeding.np_random [3P]; only the raw 128-bit (state, inc) pairs are kept, on the device"""
        st = np.zeros((self.num_envs, 4), np.uint64)
        mask = (1 << 64) - 1
        for i, sd in enumerate(seeds):
            s = np_random(sd)[0].bit_generator.state["state"]
            st[i] = [s["state"] >> 64, s["state"] & mask, s["inc"] >> 64, s["inc"] & mask]
        self._rng_dev = torch.from_numpy(st.view(np.int64)).to(self.device)

    # model.body_pos / body_quat of every world as the reference's get_env_state reports it (float64 [N, 3 or 4]); reading it from a device-draw task synchronises
    @property
    def model_edit(self):
        return self._edit_dev.cpu().numpy() if self._device_draws else self._model_edit

    @model_edit.setter
    def model_edit(self, value):
        value = np.asarray(value, dtype=np.float64).reshape(self.num_envs, -1)
        if self._device_draws:
            self._edit_dev.copy_(torch.from_numpy(np.ascontiguousarray(value)).to(self.device))
        else:
            self._model_edit = value.copy()

    @property
    def target_pos(self):
        """relocate: model.site_pos[target] of every world (float64 [N, 3]); None for the other tasks"""
        return None if self.task_name != "relocate" else self._target_dev.cpu().numpy()

    @property
    def board_z(self):
        """hammer: model.body_pos[nail_board, 2] of every world"""
        return self.model_edit[:, 2]

    def _lane_make_bufs(self):
        me = weakref.ref(self)      # (the lane must not keep the environment alive: its native model slots are released by __del__)
        return lambda m: me()._make_bufs(m)

    def _make_bufs(self, mask):
        b = _native.AdroitBuffersStruct()
        for name in ("qpos", "qvel", "qacc_ws", "shift", "action", "obs", "reward", "success", "status"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.target = None if self.target is None else self.target.data_ptr()
        b.act_mean, b.act_rng = self._act_mean.data_ptr(), self._act_rng.data_ptr()
        b.mask = None if mask is None else mask.data_ptr()
        return b

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _launch(self, bufs, forward_only):
        def fast(b):
            timed = self.kernel_events is not None and not forward_only
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _native.check(self._L.grx_adroit_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, int(forward_only), self._stream()))
            if timed:
                e1.record()
                self.kernel_events.append((e0, e1))

        if self.lane is not None and not forward_only:
            large = lambda b: _native.check(self._L.grx_adroit_step(self._h_big, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, 0, self._stream()))
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
            cost_order_update(self, self.balance_alpha)

    # ------------------------------------------------------------------ reset (MujocoEnv.reset [3P] -> reset_model of the task)
    def _write_edits(self, idx, shifts, targets=None):
        """per-world model edits -> device, through pinned staging (core.PinnedStager): nothing here waits for a step kernel that is still running"""
        k = len(idx)
        rows = np.asarray(shifts, dtype=np.float32).reshape(k, 7)
        if targets is not None:
            rows = np.concatenate([rows, np.asarray(targets, dtype=np.float32).reshape(k, 3)], axis=1)
        ti, tr = self._stage(np.asarray(idx, dtype=np.int64), rows)
        self.shift[ti] = tr[:, :7]
        if targets is not None:
            self.target[ti] = tr[:, 7:10]
        return ti

    def _launch_reset_ahead(self, idx, after):
        """draws + init rows + forward pass of the listed worlds on the side stream, into the staged rows; `after`: an event of the caller's stream recorded BEFORE the step
        launch (the previous commit read the staged rows; queued behind the step launch the forward's workgroups take wave slots its first finished worlds free)"""
        A, k = self._ahead, len(idx)
        self._ahead_stream.wait_event(after)
        with torch.cuda.stream(self._ahead_stream):
            if self._device_draws:
                ti = self._stage(np.asarray(idx, dtype=np.int64))
                _native.check(self._L.grx_adroit_sample_resets_device(
                    self._rng_dev.data_ptr(), ti.data_ptr(), k, int(self.task.kind), self._shift_pos0.ctypes.data, self._edit_dev.data_ptr(),
                    None if self._target_dev is None else self._target_dev.data_ptr(), A["shift"].data_ptr(), A["target"].data_ptr() if "target" in A else None, self._stream()))
            else:      # (AdroitHandPen: the target orientation, adroit_pen.py:379-383 -- libm calls with no bit-exact device twin)
                d = sample_reset_batch(self.task_name, [self.np_randoms[w] for w in idx], self.model)
                self._model_edit[idx] = d["edit"]
                ti, tr = self._stage(np.asarray(idx, dtype=np.int64), np.asarray(d["shift"], dtype=np.float32).reshape(k, 7))
                A["shift"][ti] = tr
            A["qpos"][ti] = self._init_qpos
            A["qvel"].index_fill_(0, ti, 0.0)
            A["qacc_ws"].index_fill_(0, ti, 0.0)
            b = self._ahead_bufs
            b.compact, b.n_compact = ti.data_ptr(), k
            _native.check(self._L.grx_adroit_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, 1, self._stream()))
            done = torch.cuda.Event()
            done.record(self._ahead_stream)
        return ti, done

    def _commit_ahead(self, ahead, idx):
        """behind the step kernel and the lane's launches: the staged rows replace the live ones (grx_adroit_commit_rows)"""
        ti, done = ahead
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        A = self._ahead
        p = lambda t: None if t is None else t.data_ptr()
        a = _native.AdroitCommitArgsStruct(ti.data_ptr(), len(idx), self.nq, self.nv, self.obs_dim, A["qpos"].data_ptr(), A["qvel"].data_ptr(), A["qacc_ws"].data_ptr(),
                                           A["shift"].data_ptr(), p(A.get("target")), A["obs"].data_ptr(), A["status"].data_ptr(),
                                           self.qpos.data_ptr(), self.qvel.data_ptr(), self.qacc_ws.data_ptr(), self.shift.data_ptr(), p(self.target), self.obs.data_ptr(), self.status.data_ptr())
        _native.check(self._L.grx_adroit_commit_rows(ctypes.byref(a), self._stream()))
        ti.record_stream(main)      # allocated under the side stream, read by the commit on this one
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    def _reset_worlds(self, idx):
        if len(idx) == 0:
            return None
        if self._device_draws:      # the index list goes up through pinned memory; the draws, the fp64 edit rows and the fp32 shift / target rows are written by one kernel
            ti = self._stage(np.asarray(idx, dtype=np.int64))
            _native.check(self._L.grx_adroit_sample_resets_device(
                self._rng_dev.data_ptr(), ti.data_ptr(), len(idx), int(self.task.kind), self._shift_pos0.ctypes.data, self._edit_dev.data_ptr(),
                None if self._target_dev is None else self._target_dev.data_ptr(), self.shift.data_ptr(), None if self.target is None else self.target.data_ptr(), self._stream()))
        else:
            d = sample_reset_batch(self.task_name, [self.np_randoms[w] for w in idx], self.model)
            self._model_edit[idx] = d["edit"]
            ti = self._write_edits(idx, d["shift"], None)
        self.qpos[ti] = self._init_qpos
        self.qvel.index_fill_(0, ti, 0.0)      # (x[ti] = 0.0 would upload a host scalar: a synchronising copy)
        self.qacc_ws.index_fill_(0, ti, 0.0)
        if self._compact_resets:     # one workgroup per listed world (grx_adroit_buffers.compact) instead of a masked launch over all N
            b = self._make_bufs(None)
            b.compact, b.n_compact = ti.data_ptr(), len(idx)
            self._launch(b, True)
        else:
            self.mask.zero_()
            self.mask.index_fill_(0, ti, 1)
            self._launch(self._bufs_masked, True)
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False
        return ti

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            seeds = [seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed)
            if self._device_draws:
                self._seed_worlds(seeds)
            else:
                self.np_randoms = [np_random(s)[0] for s in seeds]
        with torch.cuda.device(self.device):
            self._reset_worlds(np.arange(self.num_envs))
            if options is not None and "initial_state_dict" in options:
                self.set_env_state(options["initial_state_dict"])
        self._has_reset = True
        return self._obs(), {}

    # ------------------------------------------------------------------ step (adroit_hammer.py:291-329)
    def step(self, actions):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        a = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions, dtype=np.float32))
        if tuple(a.shape) != (self.num_envs, self.nu):
            raise ValueError(f"Action dimension mismatch. Expected {(self.num_envs, self.nu)}, found {tuple(a.shape)}")
        self.action.copy_(a.to(torch.float32), non_blocking=True)
        info = {}
        with torch.cuda.device(self.device):
            pending = np.nonzero(self._needs_reset)[0] if self.autoreset_mode == "next_step" else np.zeros(0, np.int64)
            will, ahead = np.zeros(0, np.int64), None
            if self._ahead is not None and self.autoreset_mode == "same_step" and self.max_episode_steps is not None:
                will = np.nonzero(self._elapsed + 1 >= self.max_episode_steps)[0]      # the worlds this step truncates: the Adroit tasks have no other episode end
                if len(will):
                    before = torch.cuda.Event()
                    before.record(torch.cuda.current_stream(self.device))
            if len(pending):
                self.mask.fill_(1)
                self.mask.index_fill_(0, self._stage_idx(pending), 0)      # (pinned staging + index_fill_: nothing here waits for the running kernel)
                self._launch(self._bufs_masked, False)
            else:
                self._launch(self._bufs, False)
            if len(will):
                ahead = self._launch_reset_ahead(will, before)
            stepped = ~self._needs_reset
            self._elapsed[stepped] += 1
            truncated = np.zeros(self.num_envs, bool)
            if self.max_episode_steps is not None:
                truncated = stepped & (self._elapsed >= self.max_episode_steps)
            terminated = np.zeros(self.num_envs, bool)
            if len(pending):
                tp = self._reset_worlds(pending)
                self.reward.index_fill_(0, tp, 0.0)
            if self.autoreset_mode == "same_step" and truncated.any():
                done = np.nonzero(truncated)[0]
                if ahead is not None:
                    if not np.array_equal(will, done):
                        raise RuntimeError("overlapped reset: the worlds reset ahead of the step are not the ones it truncated")
                    torch.cuda.current_stream(self.device).wait_event(ahead[1])      # (the index list was uploaded on the side stream)
                    info["final_obs"] = self.obs[ahead[0]] if self.output == "torch" else self.obs[ahead[0]].double().cpu().numpy()
                    self._commit_ahead(ahead, done)
            if self.autoreset_mode == "same_step" and truncated.any() and ahead is None:
                done = np.nonzero(truncated)[0]
                td = self._stage(done)
                info["final_obs"] = self.obs[td].clone() if self.output == "torch" else self.obs[td].double().cpu().numpy()
                keep_r, keep_s, keep_st = self.reward.clone(), self.success.clone(), self.status.clone()
                self._reset_worlds(done)
                self.reward.copy_(keep_r)
                self.success.copy_(keep_s)
                self.status.copy_((keep_st & 0xFFFF) | (self.status & -65536))
            elif self.autoreset_mode == "next_step":
                self._needs_reset |= truncated
        if self.output == "torch":
            info["success"] = self.success.bool()
            return self.obs, self.reward, torch.from_numpy(terminated), torch.from_numpy(truncated), self._status_info(info)
        info["success"] = self.success.cpu().numpy().astype(bool)
        return self._obs(), self.reward.double().cpu().numpy(), terminated, truncated, self._status_info(info)

    def _obs(self):
        return self.obs if self.output == "torch" else self.obs.double().cpu().numpy()

    # ------------------------------------------------------------------ get_env_state / set_env_state, batched
    # (adroit_hammer.py:380-402, adroit_door.py:375-392, adroit_pen.py:399-419, adroit_relocate.py:375-410)
    def get_env_state(self):
        st = dict(qpos=self.qpos.double().cpu().numpy(), qvel=self.qvel.double().cpu().numpy())
        obs = self.obs.double().cpu().numpy()
        if self.task_name == "hammer":
            st.update(board_pos=self.model_edit.copy(), target_pos=obs[:, 42:45])
        elif self.task_name == "door":
            st.update(door_body_pos=self.model_edit.copy())
        elif self.task_name == "pen":
            st.update(desired_orien=self.model_edit.copy())
        else:   # data.xpos[Object], site S_grasp and the target site as of the last forward pass: recovered from the observation's differences
            tgt = self.target_pos.copy()
            st.update(hand_qpos=st["qpos"][:, :30].copy(), obj_pos=tgt + obs[:, 36:39], target_pos=tgt, palm_pos=tgt + obs[:, 33:36])
        return st

    def set_env_state(self, state_dict):
        n = self.num_envs
        need = dict(hammer=(("board_pos", 3),), door=(("door_body_pos", 3),), pen=(("desired_orien", 4),), relocate=(("obj_pos", 3), ("target_pos", 3)))[self.task_name]
        for key, width in (("qpos", self.nq), ("qvel", self.nv)) + need:
            if key not in state_dict or np.asarray(state_dict[key]).shape != (n, width):
                raise AssertionError(f"The state dictionary must hold `{key}` of shape {(n, width)}")
        qp = np.asarray(state_dict["qpos"], dtype=np.float64)
        if self.task_name == "pen":
            edit = np.asarray(state_dict["desired_orien"], dtype=np.float64)
            shifts = group_shift(self.model, quat=edit)
        else:
            if self.task_name == "relocate":   # model.body_pos[Object] = obj_pos - qpos[OBJTx..OBJTz] (adroit_relocate.py:405-407)
                edit = np.asarray(state_dict["obj_pos"], dtype=np.float64) - qp[:, 30:33]
                self._target_dev.copy_(torch.from_numpy(np.ascontiguousarray(state_dict["target_pos"], dtype=np.float64)).to(self.device))
            else:
                edit = np.asarray(state_dict[need[0][0]], dtype=np.float64)
            shifts = group_shift(self.model, pos=edit)
        self.model_edit = edit
        with torch.cuda.device(self.device):
            self._write_edits(np.arange(n), shifts, None if self.task_name != "relocate" else np.asarray(state_dict["target_pos"], dtype=np.float64))
            self.qpos.copy_(torch.from_numpy(qp.astype(np.float32)).to(self.device))
            self.qvel.copy_(torch.from_numpy(np.asarray(state_dict["qvel"], dtype=np.float32)).to(self.device))
            self.qacc_ws.zero_()
            self._launch(self._bufs, True)   # set_state -> mj_forward

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


AdroitHammerVecEnv = AdroitVecEnv   # round-1/2 name of the class
