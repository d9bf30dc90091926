"""Batched Fetch environments on the MI355X engine (host side, Python).

Vectorised drop-in for the reference classes
    MujocoFetchReachEnv / MujocoFetchPushEnv / MujocoFetchPickAndPlaceEnv
(/root/reference/gymnasium_robotics/envs/fetch/*.py) behind the unchanged Gymnasium GoalEnv contract:

    reset(seed=, options=) -> (obs_dict, info)            envs/robot_env.py:154-186
    step(actions)          -> (obs_dict, reward, terminated, truncated, info)   envs/robot_env.py:114-152
    compute_reward(achieved_goal, desired_goal, info)      envs/fetch/fetch_env.py:74-80 (batched, HER)

Everything per-step runs in ONE HIP kernel launch (csrc/grx_kernels.hip) over all ``num_envs`` worlds.
The host keeps what the reference keeps on the host at episode boundaries: the PCG64 draws of
``_reset_sim`` / ``_sample_goal`` (fetch_env.py:153-166,375-402) so that ``reset(seed=s)`` gives world
``i`` exactly the start state and goal the reference produces for ``seed = s + i``.
"""
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from .. import _native
from ..core import create_rerun_model, GoalVecEnv, OverflowLane, np_random
from ..mjcf import CompiledModel, compile_mjcf, load_model
from ..spaces import Box, Dict, batch_space
from .fetch_spec import DISTANCE_THRESHOLD, FETCH_TASKS, MAX_EPISODE_STEPS, N_SUBSTEPS, make_fetch_task, parse_env_id

_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")


# Engine capacities of the Fetch models: 28 contacts / 144 rows / 2 032 Jacobian-pool words (rounds 2 - 4: 32 / 144 / 1 984) = 20.4 KB of LDS per world = 16 allocation
# granules = 8 worlds per CU, which is also what the register budget of the step kernels allows (2 waves per SIMD: the hull-vs-convex
# routine needs more than the 168 VGPRs of a 3-wave build).  Round 1 ran 112 rows / 1 520 words for 9 worlds per CU and dropped contacts
# in 0.035 % of the world-steps; at these capacities the measured rate is 0.004 % (GRX_STATUS_EFC_OVERFLOW, sticky in `status`).
# Round 5 (profiles/demand_r05_fetch.txt: per world and step maxima over 400 steps of the bench rollout): the worlds that exceed these tables -- and are re-run behind the launch on
# the large ones, 1 - 3 ms once each -- exceed the POOL by a few dozen words (2 005 .. 2 054 of 1 984) while the contact list peaks at 26 of 32: 28 contacts / 2 032 words fit the
# same 16 granules (20 424 B) and halve the re-runs.
FETCH_CAPACITY = {"maxcon": 28, "maxefc": 144, "jpool": 2032, "split_spans": False}   # the arm chain is (nearly) contiguous: single-span rows
# Round 6 (opt-in, GRX_FETCH_HANDOFF=1; see __init__): tables of the FAST FetchPickAndPlace step kernel (csrc/grx_kernels.hip GrxShapeFetchPickFast: no hull routine, 168 VGPRs = three waves per SIMD, 15.2 KB of LDS = ten
# worlds per CU).  ~0.1 % of the world-steps exceed 1 024 pool words, none 24 contacts (profiles/hull_share_r06_fetch.txt); a world that does -- or in which a hull pair passes the
# bounding-box filter -- is handed off MID-STEP to the standing lane, which runs the kernel on FETCH_CAPACITY above (include/grx_capi.h, grx_fetch_buffers.handoff).
FETCH_FAST_CAPACITY = {"maxcon": 24, "maxefc": 96, "jpool": 1024}
FETCH_SPLIT_PARTS = 4     # default of GRX_FETCH_SPLIT for batches of more than one round of worlds up to FETCH_SPLIT_WIDE (see FetchVecEnv._alloc; profiles/ab_r06_split_fences.txt)
FETCH_SPLIT_WIDE = 12288  # above: 2 parts (16 384 worlds: 2 parts 1.796 M, 4 parts 1.789 M)
HANDOFF_TASKS = ("FetchPickAndPlace",)


def load_fetch_model(task: str, assets_root: Optional[str] = None) -> CompiledModel:
    """Compiled model tables for a Fetch task: compiled from MJCF when an asset tree is given
    (``assets_root`` or $GRX_ASSETS_ROOT = .../gymnasium_robotics/envs/assets), else the packaged blob."""
    xml = FETCH_TASKS[task]["xml"]
    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        return compile_mjcf(os.path.join(assets_root, xml), capacity=FETCH_CAPACITY)
    path = os.path.join(_MODELS_DIR, os.path.splitext(os.path.basename(xml))[0] + ".npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist (no packaged model and no assets_root given)")
    model = load_model(path).with_capacity(**{k: FETCH_CAPACITY[k] for k in ("maxcon", "maxefc", "jpool")})      # (the packaged blobs carry round 2's requests; the tables do not depend on them)
    cap = os.environ.get("GRX_FETCH_CAP")     # experiments: "maxefc,jpool[,maxcon]" other than the packaged capacities (needs a library built with -DGRX_FETCH_ME / -DGRX_FETCH_JP / -DGRX_FETCH_MC to stay on the specialised kernels)
    if cap:
        vals = [int(x) for x in cap.split(",")]
        model = model.with_capacity(maxefc=vals[0], jpool=vals[1], **({"maxcon": vals[2]} if len(vals) > 2 else {}))
    return model


def sample_fetch_reset(cfg, rng, initial_gripper_xpos, height_offset):
    """PCG64 draw order of _reset_sim + _sample_goal for one world (fetch_env.py:153-166,388-391).
    Returns (object_xy or None, goal)."""
    g0 = np.asarray(initial_gripper_xpos, dtype=np.float64)
    oxy = None
    if cfg["has_object"]:
        oxy = g0[:2]
        while np.linalg.norm(oxy - g0[:2]) < 0.1:
            oxy = g0[:2] + rng.uniform(-cfg["obj_range"], cfg["obj_range"], size=2)
    goal = g0[:3] + rng.uniform(-cfg["target_range"], cfg["target_range"], size=3)
    if cfg["has_object"]:
        goal = goal + cfg["target_offset"]
        goal[2] = height_offset
        if cfg["target_in_the_air"] and rng.uniform() < 0.5:
            goal[2] += rng.uniform(0, 0.45)
    return oxy, goal


class FetchVecEnv(GoalVecEnv):
    """``num_envs`` Fetch worlds stepping in lock-step on one GPU.

    autoreset_mode: "next_step" (Gymnasium >= 1.0 default), "same_step" or "disabled".
    output: "numpy" returns float64 numpy arrays like the reference; "torch" returns the fp32
            device tensors the kernel wrote (valid until the next step) -- no PCIe traffic.
    """

    CKPT_SKIP = GoalVecEnv.CKPT_SKIP + ("_ahead",)      # staging rows of the overlapped reset: written and consumed inside one step() call, never state

    def __init__(self, env_id: str = "FetchPickAndPlace-v4", num_envs: int = 1, device: Optional[str] = None,
                 max_episode_steps: Optional[int] = MAX_EPISODE_STEPS, autoreset_mode: str = "next_step",
                 output: str = "numpy", assets_root: Optional[str] = None, model: Optional[CompiledModel] = None,
                 reward_type: Optional[str] = None, seed_offset: int = 0, balance: bool = True, overflow_rerun: bool = True):
        task, rt = parse_env_id(env_id)
        self.balance = balance   # cost-ordered dispatch of the step kernel (see _alloc); results do not depend on it
        self.env_id, self.task_name, self.reward_type = env_id, task, reward_type or rt
        self.cfg = FETCH_TASKS[task]
        self.num_envs = int(num_envs)
        self.max_episode_steps = max_episode_steps
        if autoreset_mode not in ("next_step", "same_step", "disabled"):
            raise ValueError(f"unknown autoreset_mode {autoreset_mode}")
        self.autoreset_mode, self.output = autoreset_mode, output
        self.seed_offset = int(seed_offset)
        if not torch.cuda.is_available():
            raise RuntimeError("FetchVecEnv needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device or "cuda:0")
        self.model = (model or load_fetch_model(task, assets_root)).copy()
        # reset_mocap_welds (utils/mujoco_utils.py:74-80): eq_data[:7] = [0,0,0,0,0,0,1]
        eq = self.model.tables["eq_data"]
        eq[self.model.tables["eq_type"] == 1, :7] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
        self.task = make_fetch_task(self.model, task, self.reward_type)
        self.nq, self.nv, self.nmocap = self.model.dim("nq"), self.model.dim("nv"), self.model.dim("nmocap")
        self.obs_dim = int(self.task.obs_dim)
        self.dt = N_SUBSTEPS * self.model.opt("timestep")
        self._L = _native.lib()
        H, I, F = self.model.pack()
        dev_index = self.device.index or 0
        self._h = _native.acquire_model(H, I, F, dev_index)   # shared with every other environment of the same compiled tables (reference-counted)
        self.lds_bytes = self._L.grx_model_lds_bytes(self._h)
        # the same model with larger row / Jacobian-pool tables (runs on the generic kernel): where the worlds go that overflow the specialised kernel's capacities
        self._h_big = create_rerun_model(self._L, self.model, dev_index, overflow_rerun)
        # Fast step kernel + mid-step hand-off: BUILT, bit-identical to the full kernel's rollout (tests/test_gpu_fetch.py::test_handoff_is_the_full_kernels_rollout), and OFF by
        # default -- measured (profiles/ab_r06_fetch_handoff.txt): +4 % at 4 096 worlds, 0 % at 8 192, -8 % at 16 384.  The third wave per SIMD buys nothing for this kernel (a world
        # runs 1.27x longer at ten per CU than at eight), and the 8 - 19 % of the worlds that live in the hull lane keep their 256-VGPR kernel.  GRX_FETCH_HANDOFF=1 switches it on.
        self._h_fast, self._fast_model = None, None
        if task in HANDOFF_TASKS and self._h_big is not None and self.num_envs >= 64 and os.environ.get("GRX_FETCH_HANDOFF", "0") == "1" and os.environ.get("GRX_FETCH_CAP") is None:
            self._fast_model = self.model.with_capacity(**FETCH_FAST_CAPACITY)
            Hf, If, Ff = self._fast_model.pack()
            self._h_fast = _native.acquire_model(Hf, If, Ff, dev_index)
            self.lds_bytes = self._L.grx_model_lds_bytes(self._h_fast)      # the step kernel's footprint (slots of the cost-ordered dispatch)
        self._alloc(self.num_envs)
        self._env_setup()
        # spaces (envs/robot_env.py:87-100)
        self.single_action_space = Box(-1.0, 1.0, (4,), np.float32)
        self.single_observation_space = Dict(dict(
            desired_goal=Box(-np.inf, np.inf, (3,), np.float64), achieved_goal=Box(-np.inf, np.inf, (3,), np.float64),
            observation=Box(-np.inf, np.inf, (self.obs_dim,), np.float64)))
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self._check_goal_space()
        self._seed_worlds([None] * self.num_envs)
        self._elapsed = np.zeros(self.num_envs, np.int64)
        self._needs_reset = np.zeros(self.num_envs, bool)
        self._has_reset = False
        self.kernel_events = None  # set to [] to collect (start, end) torch.cuda.Event pairs around every step-kernel launch
        self.step_events = None    # ... and [] to collect the pairs around the whole launch group of a step (fast launch + the serialised re-run of the worlds that overflowed its tables)

    def _launch_step(self, bufs):
        def fast(b):
            ev = self.kernel_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _native.check(self._L.grx_fetch_step(self._h_fast or self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, self._stream()))
            if ev is not None:
                e1.record()
                ev.append((e0, e1))

        if self.lane is None:
            fast(bufs)
        else:
            large = lambda b: _native.check(self._L.grx_fetch_step(self._h_big, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, self._stream()))
            # with the fast kernel: the standing lane (hull worlds, worlds near a capacity of the fast tables) runs the model's own kernel on FETCH_CAPACITY
            middle = (lambda b: _native.check(self._L.grx_fetch_step(self._h, ctypes.byref(self.task), ctypes.byref(b), self.num_envs, self._stream()))) if self._h_fast else None
            timed = self.kernel_events is not None and self.step_events is not None      # the fast launch AND the re-run of the worlds that overflowed its tables (behind it, same stream)
            if timed:
                l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                l0.record()
            self.lane.step(self.mask if bufs is self._bufs_masked else None, fast, large, fast_bufs=bufs, launch_lane=middle)
            if timed:
                l1.record()
                self.step_events.append((l0, l1))
        if self.balance:
            self._rebalance()

    # ------------------------------------------------------------------ buffers
    def _alloc(self, n):
        d, f32 = self.device, torch.float32
        z = lambda *s, dtype=f32: torch.zeros(*s, dtype=dtype, device=d)
        self.qpos, self.qvel, self.qacc_ws = z(n, self.nq), z(n, self.nv), z(n, self.nv)
        self.mocap, self.aux, self.goal, self.action = z(n, 7 * self.nmocap), z(n, 8), z(n, 3), z(n, 4)
        self.obs, self.achieved, self.reward = z(n, self.obs_dim), z(n, 3), z(n)
        self.success, self.status = z(n, dtype=torch.uint8), z(n, dtype=torch.int32)
        self.mask = torch.ones(n, dtype=torch.uint8, device=d)
        # Cost-ordered dispatch (include/grx_capi.h, grx_fetch_buffers.order / .cost): every step launch records how long each world took
        # and the next launch starts the expensive worlds first.  A world's cost is strongly correlated from one step to the next (it is
        # in contact or it is not), and with ~2 worlds per resident wave slot the launch otherwise ends with a few slots finishing two
        # expensive worlds while the rest of the chip idles.  Worlds stay inside their XCD's slice (L2 locality of neighbouring rows).
        self.balance = bool(self.balance) and n % 8 == 0 and 1024 <= n <= 65536 * 8   # grx_order_by_cost sorts one XCD slice (n / 8 worlds, <= 65536) per workgroup in LDS
        self._reset_stage = z(n, 6)   # device side of the reset staging buffer (see _reset_worlds)
        self.packed = z(n, self.obs_dim + 8)   # [obs | achieved | desired | reward | success] rows written by the step kernel (cross-rank gather, parallel.py)
        self.final_packed = z(n, self.obs_dim + 8)   # same-step autoreset: row w = the TERMINAL packed row of world w's last finished episode (info["final_obs"], HerReplay.append(final_rows=...))
        self.cost = torch.zeros(n, dtype=torch.int32, device=d) if self.balance else None
        self.cost_ema = torch.zeros(n, dtype=torch.float32, device=d) if self.balance else None
        self.balance_alpha = float(os.environ.get("GRX_BALANCE_ALPHA", 0.1))   # weight of the newest sample in the moving average the order is sorted by (A/B on FetchPickAndPlace: 1.0 -> 2.69 ms, 0.15 -> 2.64 ms per step)
        self.order = None
        # wave slots of one XCD (32 CUs): worlds per CU = what the LDS footprint allows (1 280-byte granules of 160 KB), at most the 8 waves a 2-waves-per-SIMD kernel gets;
        # grx_order_by_cost_slots uses it in the two-worlds-per-slot regime (4096 worlds on 2048 slots), GRX_TAIL_ORDER=0 restores the plain descending order
        self._slots_per_xcd = 32 * min(12 if self._h_fast else 8, (160 * 1024) // (-(-self.lds_bytes // 1280) * 1280)) if (self.balance and os.environ.get("GRX_TAIL_ORDER", "1") != "0") else 0
        if self.balance:
            per = n // 8
            self._slice_base = (torch.arange(8, device=d, dtype=torch.int32) * per).unsqueeze(1)          # [8,1]
            self.order = (self._slice_base + torch.arange(per, device=d, dtype=torch.int32).unsqueeze(0)).t().contiguous().view(-1)   # workgroup b -> slice b & 7, position b >> 3
        # the worlds' caches of separating directions (hull-vs-convex pairs), carried across launches: without it every env.step() starts with one portal
        # search for the arm's permanently near pair (torso / shoulder link, 1.9 cm apart); a stale row is harmless (directions are re-verified)
        self.hullcache = z(n, 90) if os.environ.get("GRX_NO_HULLCACHE") is None else None      # GRX_HULLCACHE_WORDS (csrc/grx_engine.h)
        # No dropped contacts (core.OverflowLane / include/grx_capi.h grx_overflow_lane): a world that exceeds a table capacity of the specialised kernel writes
        # nothing and is stepped on the SAME model with larger tables (generic kernel) -- concurrently with the fast launch once it is in the lane.
        common = (self.qpos, self.qvel, self.qacc_ws, self.mocap, self.aux, self.goal, self.action, self.obs, self.achieved, self.reward, self.success, self.status)
        # mid-step hand-off rows (include/grx_capi.h grx_fetch_buffers.handoff): [substep + 1 | status | ctrl | mocap | qpos | qvel | warm start], 16-word aligned; all zero between steps
        self.handoff, self._handoff_stride = None, 0
        if getattr(self, "_h_fast", None):
            self._handoff_stride = -(-(2 + self.model.dim("nu") + 7 * self.nmocap + self.nq + 2 * self.nv) // 16) * 16
            self.handoff = z(n, self._handoff_stride)
        self._bufs = self._make_bufs(*common, None, self.order, self.cost, self.packed, self.hullcache, self.handoff)
        self._bufs_masked = self._make_bufs(*common, self.mask, self.order, self.cost, self.packed, self.hullcache, self.handoff)
        self._bufs.handoff_stride = self._bufs_masked.handoff_stride = self._handoff_stride
        # SPLIT STEP (include/grx_capi.h grx_fetch_buffers.split_parts): the step launch has P workgroups per world, each running 1 / P of the substeps (the state travels through the
        # world's hand-off row): the launch's tail -- the duration of the LAST workgroup started, a whole world-step otherwise -- shrinks to 1 / P of it.  Bit-identical to the unsplit
        # launch (tests/test_gpu_fetch.py::test_split_step_is_the_plain_step).  GRX_FETCH_SPLIT=P (1: off); not combined with the hull-less fast kernel.
        # Default: 4 parts for batches of more than 2 048 worlds (one round of the chip's wave slots at 8 worlds per CU: below that every workgroup starts at once and there is no tail
        # to shorten), 2 above FETCH_SPLIT_WIDE.  Measured, same call (profiles/ab_r06_split_fences.txt): 4 096 worlds 1.271 M unsplit -> 1.431 / 1.442 / 1.443 / 1.430 M for 2 / 3 / 4 / 5
        # parts, 8 192: 1.540 -> 1.667 / 1.683 / 1.698 / 1.696 M, 16 384: 1.724 -> 1.796 / 1.795 / 1.789 / 1.778 M.  (With agent-scope fences in the hand-off -- the first version,
        # profiles/ab_r06_fetch_split.txt -- every part wrote the XCD's L2 back: 2 parts 1.382 M @4 096, more parts lost.)
        self._split = max(1, int(os.environ.get("GRX_FETCH_SPLIT", str((FETCH_SPLIT_PARTS if n <= FETCH_SPLIT_WIDE else 2) if n > 2048 else 1)))) if self.handoff is None and n >= 64 else 1
        if self._split > 1:
            stride = -(-(2 + self.model.dim("nu") + 7 * self.nmocap + self.nq + 2 * self.nv) // 16) * 16
            self._split_rows, self._split_state = z(n, stride), z(n, 2, dtype=torch.int32)
            for b in (self._bufs, self._bufs_masked):
                b.handoff, b.handoff_stride, b.split_state, b.split_parts = self._split_rows.data_ptr(), stride, self._split_state.data_ptr(), self._split
        self.lane = None
        if self._h_big is not None:
            packed, hullcache, handoff, stride, mk = self.packed, self.hullcache, self.handoff, self._handoff_stride, FetchVecEnv._make_bufs      # (no reference to self: the lane must not keep the environment alive)

            def lane_bufs(m):
                b = mk(*common, m, None, None, packed, hullcache, handoff)
                b.handoff_stride = stride
                return b

            if handoff is not None:
                # hull activity persists for a few steps and 8 % of a stationary batch shows it: a STANDING lane with polling workgroups for the ~1 % of the worlds that enter per step
                self.lane = OverflowLane(n, d, self._fast_model, lane_bufs, mode="lane", handoff=True, poll_grid=int(os.environ.get("GRX_FETCH_POLL", max(64, n // 32))),
                                         ttl=int(os.environ.get("GRX_FETCH_TTL", 4)), margin=0.9)
            else:
                self.lane = OverflowLane(n, d, self.model, lane_bufs, mode="entry")     # overflows are rare events here (0.0007 % of the world-steps)
        # Overlapped same-step reset (include/grx_capi.h, grx_fetch_commit_rows): Fetch episodes end by the time limit only, so the worlds a step will reset are known before
        # it is launched and their reset state depends on nothing the step computes.  The reset kernel runs for them on a side stream, beside the step kernel, into this second
        # set of world rows; ONE small kernel behind the step commits them (parks the terminal rows, copies the staged ones).  The in-line reset (88 us of a 3.3 ms step at
        # 4 096 worlds: a forward pass of ~82 worlds on an otherwise idle chip) leaves the critical path.  GRX_FETCH_AHEAD_RESET=0: the in-line path (A/B, tests).
        self._ahead = None
        if n > 1 and os.environ.get("GRX_FETCH_AHEAD_RESET", "1") != "0":
            sq, sv, sa, sm, sx, sg, so, sh = z(n, self.nq), z(n, self.nv), z(n, self.nv), z(n, 7 * self.nmocap), z(n, 8), z(n, 3), z(n, self.obs_dim), z(n, 3)
            sr, ss, st = z(n), z(n, dtype=torch.uint8), z(n, dtype=torch.int32)
            self._ahead = dict(qpos=sq, qvel=sv, qacc_ws=sa, mocap=sm, aux=sx, goal=sg, obs=so, achieved=sh, reward=sr, success=ss, status=st)
            self._ahead_bufs = self._make_bufs(sq, sv, sa, sm, sx, sg, self.action, so, sh, sr, ss, st, None, None, None, None, None)
            self._ahead_stream = torch.cuda.Stream(device=d)
            self._ahead_late = os.environ.get("GRX_FETCH_AHEAD_ORDER", "after") == "after"      # (A/B tools/ab_fetch_ahead_reset.sh: queued before the step launch the reset workgroups take wave slots from its first, expensive worlds)

    def _rebalance(self):
        """order <- per XCD slice, worlds by decreasing cost of the launch that just ran (in place: the buffer struct keeps its pointer)."""
        _native.check(self._L.grx_order_by_cost_slots(self.cost.data_ptr(), self.cost_ema.data_ptr(), self.balance_alpha, self.num_envs, self._slots_per_xcd, self.order.data_ptr(), self._stream()))

    @staticmethod
    def _make_bufs(*tensors):
        b = _native.FetchBuffersStruct()
        for (name, _), t in zip(_native.FetchBuffersStruct._fields_, tensors):
            setattr(b, name, None if t is None else t.data_ptr())
        return b

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ construction (fetch_env.py:404-428)
    def _env_setup(self):
        T, n, cfg = self.model.tables, self.model.names, self.cfg
        jq = T["jnt_qposadr"].ravel()
        q0 = T["qpos0"].astype(np.float64).copy()
        for name, v in cfg["initial_qpos"].items():
            v = np.atleast_1d(np.asarray(v, dtype=np.float64))
            a = int(jq[n["joint"][name]])
            q0[a: a + len(v)] = v
        one = FetchVecEnv.__new__(FetchVecEnv)  # 1-world scratch buffers sharing the model
        one.__dict__.update(device=self.device, nq=self.nq, nv=self.nv, nmocap=self.nmocap, obs_dim=self.obs_dim, balance=False, _h_big=None)
        one._alloc(1)
        one.qpos[0] = torch.from_numpy(q0).float()
        one.mocap[0] = torch.from_numpy(np.concatenate([T["mocap_pos0"].ravel(), T["mocap_quat0"].ravel()])).float()
        with torch.cuda.device(self.device):
            _native.check(self._L.grx_fetch_forward(self._h, ctypes.byref(self.task), ctypes.byref(one._bufs), 1, 0, self._stream()))
            grip0 = one.obs[0, :3].double().cpu().numpy()
            target = np.array([-0.498, 0.005, -0.431 + cfg["gripper_extra_height"]]) + grip0
            one.mocap[0] = torch.tensor(list(target) + [1.0, 0.0, 1.0, 0.0])
            _native.check(self._L.grx_fetch_forward(self._h, ctypes.byref(self.task), ctypes.byref(one._bufs), 1, 10 * N_SUBSTEPS,
                                                    self._stream()))
            torch.cuda.synchronize(self.device)
        if int(one.status[0]) != 0:
            raise RuntimeError(f"engine reported status {int(one.status[0])} during env setup")
        self.initial_gripper_xpos = one.obs[0, :3].double().cpu().numpy()
        self.height_offset = float(one.achieved[0, 2]) if cfg["has_object"] else 0.0
        self.initial_qpos = one.qpos[0].clone()
        self.initial_qvel = one.qvel[0].clone()
        self._mocap0 = torch.from_numpy(np.concatenate([T["mocap_pos0"].ravel(), T["mocap_quat0"].ravel()])).float().to(self.device)
        self._obj_qadr = int(jq[n["joint"]["object0:joint"]]) if cfg["has_object"] else -1

    # ------------------------------------------------------------------ reset (robot_env.py:154-186)
    def _seed_worlds(self, seeds):
        """One numpy PCG64 per world, seeded like gymnasium.utils.seeding.np_random [3P]; only the raw 128-bit (state, inc) pairs are kept, ON THE DEVICE: the
        streams are advanced by grx_fetch_sample_resets_device (bit-exact with numpy's Generator.uniform), the host never draws."""
        st = np.zeros((self.num_envs, 4), np.uint64)
        mask = (1 << 64) - 1
        for i, sd in enumerate(seeds):
            s = np_random(sd)[0].bit_generator.state["state"]
            st[i] = [s["state"] >> 64, s["state"] & mask, s["inc"] >> 64, s["inc"] & mask]
        self._rng_dev = torch.from_numpy(st.view(np.int64)).to(self.device)

    @property
    def _rng_state(self):
        """host COPY of the worlds' stream positions (inspection / tests: synchronises); assign the whole [N, 4] uint64 array to upload new positions, or use set_world_rng"""
        return self._rng_dev.cpu().numpy().view(np.uint64)

    @_rng_state.setter
    def _rng_state(self, value):
        value = np.ascontiguousarray(value, dtype=np.uint64).reshape(self.num_envs, 4)
        self._rng_dev.copy_(torch.from_numpy(value.view(np.int64)).to(self.device))

    def _stage_reset(self, idx: np.ndarray):
        """The listed worlds' indices go to the device through pinned memory (one asynchronous copy); their PCG64 draws -- the rejection loop of _reset_sim, the goal of
        _sample_goal -- are made ON THE DEVICE from the worlds' device-resident streams (grx_fetch_sample_resets_device, bit-exact with numpy).  _launch_reset then
        runs the compacted, shape-specialised reset kernel (grx_fetch_reset).  Nothing here waits for the device, nothing is drawn on the host."""
        n = len(idx)
        cfg = self.cfg
        stage = torch.empty(n, dtype=torch.int32, pin_memory=True)
        stage.numpy()[:] = np.ascontiguousarray(idx, dtype=np.int32)
        dev = self._reset_stage.view(-1)[: 6 * n]
        idx_dev, samples = dev[:n].view(torch.int32), dev[n:]
        idx_dev.copy_(stage, non_blocking=True)
        toff = np.ascontiguousarray(np.broadcast_to(np.asarray(cfg["target_offset"], dtype=np.float64), (3,)))
        g0 = np.ascontiguousarray(self.initial_gripper_xpos, dtype=np.float64)
        _native.check(self._L.grx_fetch_sample_resets_device(
            self._rng_dev.data_ptr(), idx_dev.data_ptr(), n, int(cfg["has_object"]), int(cfg["target_in_the_air"]), float(cfg["obj_range"]), float(cfg["target_range"]),
            toff.ctypes.data, g0.ctypes.data, float(self.height_offset), samples.data_ptr(), self._stream()))
        return n, idx_dev, samples

    def _launch_reset(self, staged, idx, keep_outcome=False):
        n, idx_dev, samples = staged
        # same-step autoreset: the kernel parks the terminal packed row of every listed world in final_packed before it writes the reset observation
        args = _native.FetchResetArgsStruct(idx_dev.data_ptr(), samples.data_ptr(), self.initial_qpos.data_ptr(), self.initial_qvel.data_ptr(),
                                            self._mocap0.data_ptr(), int(self._obj_qadr), int(keep_outcome), self.final_packed.data_ptr() if keep_outcome else None)
        _native.check(self._L.grx_fetch_reset(self._h, ctypes.byref(self.task), ctypes.byref(self._bufs), ctypes.byref(args), n, self._stream()))
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    def _launch_reset_ahead(self, idx: np.ndarray, after=None):
        """The listed worlds' reset -- index upload, the device-side PCG64 draws, the reset kernel -- on the side stream, into the staged rows (see _alloc): runs beside the step
        kernel the caller launches next.  The side stream first waits for everything already queued on the caller's stream (the previous step's commit read these buffers)."""
        if after is None:
            self._ahead_stream.wait_stream(torch.cuda.current_stream(self.device))
        else:
            self._ahead_stream.wait_event(after)
        with torch.cuda.stream(self._ahead_stream):
            staged = self._stage_reset(idx)
            n, idx_dev, samples = staged
            args = _native.FetchResetArgsStruct(idx_dev.data_ptr(), samples.data_ptr(), self.initial_qpos.data_ptr(), self.initial_qvel.data_ptr(),
                                                self._mocap0.data_ptr(), int(self._obj_qadr), 1, None)
            _native.check(self._L.grx_fetch_reset(self._h, ctypes.byref(self.task), ctypes.byref(self._ahead_bufs), ctypes.byref(args), n, self._stream()))
            done = torch.cuda.Event()
            done.record(self._ahead_stream)
        return staged, done

    def _commit_ahead(self, ahead, idx):
        """behind the step kernel (and the re-run of the worlds that overflowed its tables): the staged reset rows replace the live ones (grx_fetch_commit_rows)"""
        (n, idx_dev, _), done = ahead
        torch.cuda.current_stream(self.device).wait_event(done)
        A = self._ahead
        a = _native.FetchCommitArgsStruct(idx_dev.data_ptr(), n, self.nq, self.nv, 7 * self.nmocap, self.obs_dim,
                                          A["qpos"].data_ptr(), A["qvel"].data_ptr(), A["qacc_ws"].data_ptr(), A["mocap"].data_ptr(), A["aux"].data_ptr(), A["goal"].data_ptr(),
                                          A["obs"].data_ptr(), A["achieved"].data_ptr(), A["status"].data_ptr(),
                                          self.qpos.data_ptr(), self.qvel.data_ptr(), self.qacc_ws.data_ptr(), self.mocap.data_ptr(), self.aux.data_ptr(), self.goal.data_ptr(),
                                          self.obs.data_ptr(), self.achieved.data_ptr(), self.packed.data_ptr(), self.final_packed.data_ptr(), self.status.data_ptr())
        _native.check(self._L.grx_fetch_commit_rows(ctypes.byref(a), self._stream()))
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    def _reset_worlds(self, idx: np.ndarray):
        if len(idx) == 0:
            return None
        staged = self._stage_reset(idx)
        self._launch_reset(staged, idx)
        return staged[1].long()

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            seeds = [seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed)
            self._seed_worlds(seeds)
        with torch.cuda.device(self.device):
            self._reset_worlds(np.arange(self.num_envs))
        self._has_reset = True
        return self._obs_dict(), {}

    # ------------------------------------------------------------------ step (robot_env.py:114-152)
    def step(self, actions):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        if isinstance(actions, torch.Tensor):
            if tuple(actions.shape) != (self.num_envs, 4):
                raise ValueError("Action dimension mismatch")
            self.action.copy_(actions.to(torch.float32), non_blocking=True)
        else:
            a = np.asarray(actions, dtype=np.float32)
            if a.shape != (self.num_envs, 4):
                raise ValueError("Action dimension mismatch")
            self.action.copy_(torch.from_numpy(a), non_blocking=True)
        with torch.cuda.device(self.device):
            pending = np.nonzero(self._needs_reset)[0] if self.autoreset_mode == "next_step" else np.zeros(0, np.int64)
            ahead, will, before = None, np.zeros(0, np.int64), None
            if self._ahead is not None and self.autoreset_mode == "same_step" and self.max_episode_steps is not None:
                will = np.nonzero(self._elapsed + 1 >= self.max_episode_steps)[0]      # the worlds this step truncates: Fetch has no other episode end (compute_terminated)
                if len(will) and self._ahead_late:      # the side stream waits for what is queued NOW, not for the step launch that follows
                    before = torch.cuda.Event()
                    before.record(torch.cuda.current_stream(self.device))
                elif len(will):
                    ahead = (will, self._launch_reset_ahead(will))
            if len(pending):
                self.mask.fill_(1)
                self.mask.index_fill_(0, self._stage_idx(pending), 0)      # (pinned staging + index_fill_: nothing here waits for the running kernel)
                self._launch_step(self._bufs_masked)
            else:
                self._launch_step(self._bufs)
            if before is not None:
                ahead = (will, self._launch_reset_ahead(will, after=before))      # queued behind the step launch: its workgroups take the slots the first finished worlds free
            stepped = ~self._needs_reset if len(pending) else np.ones(self.num_envs, bool)
            self._elapsed[stepped] += 1
            truncated = np.zeros(self.num_envs, bool)
            if self.max_episode_steps is not None:
                truncated = stepped & (self._elapsed >= self.max_episode_steps)
            terminated = np.zeros(self.num_envs, bool)
            info = {}
            if len(pending):  # Gymnasium NEXT_STEP: the reset replaces the step; reward 0, flags False
                tp = self._reset_worlds(pending)
                self.reward.index_fill_(0, tp, 0.0)
                self.packed[:, -2].index_fill_(0, tp, 0.0)
            if self.autoreset_mode == "same_step" and truncated.any():
                done = np.nonzero(truncated)[0]
                # the reset parks the terminal packed rows of these worlds in final_packed (info["final_obs"], the last transition for HER) and
                # leaves reward / success at the finished episode's values (keep_outcome)
                if ahead is not None:
                    if not np.array_equal(ahead[0], done):
                        raise RuntimeError("overlapped reset: the worlds reset ahead of the step are not the ones it truncated")
                    staged = ahead[1][0]
                    if self.output != "torch":
                        info["final_obs"] = self._obs_dict(rows=done)
                    self._commit_ahead(ahead[1], done)
                else:
                    staged = self._stage_reset(done)
                    if self.output != "torch":
                        info["final_obs"] = self._obs_dict(rows=done)
                    self._launch_reset(staged, done, keep_outcome=True)
                if self.output == "torch":
                    fo = self.final_packed[staged[1]]      # (int32 device indices: one gather kernel)
                    info["final_obs"] = {"observation": fo[:, : self.obs_dim], "achieved_goal": fo[:, self.obs_dim: self.obs_dim + 3],
                                         "desired_goal": fo[:, self.obs_dim + 3: self.obs_dim + 6]}
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

    def _obs_dict(self, rows=None, rows_dev=None):
        if self.output == "torch":
            if rows is not None and rows_dev is None:
                rows_dev = torch.from_numpy(rows).to(self.device)
            sel = (lambda t: t) if rows is None else (lambda t: t[rows_dev])
            return {"observation": sel(self.obs), "achieved_goal": sel(self.achieved), "desired_goal": sel(self.goal)}
        sel = (lambda a: a) if rows is None else (lambda a: a[rows])
        return {"observation": sel(self.obs.double().cpu().numpy()), "achieved_goal": sel(self.achieved.double().cpu().numpy()),
                "desired_goal": sel(self.goal.double().cpu().numpy())}

    # ------------------------------------------------------------------ GoalEnv API (fetch_env.py:74-80; core.py:45-114)
    def compute_reward(self, achieved_goal, desired_goal, info=None):
        """Batched reward recompute (HER relabelling): any leading batch shape, last dim 3."""
        as_numpy = not isinstance(achieved_goal, torch.Tensor)
        ag = torch.as_tensor(np.asarray(achieved_goal, dtype=np.float32) if as_numpy else achieved_goal, dtype=torch.float32, device=self.device).contiguous()
        dg = torch.as_tensor(np.asarray(desired_goal, dtype=np.float32) if as_numpy else desired_goal, dtype=torch.float32, device=self.device).contiguous()
        if ag.shape != dg.shape or ag.shape[-1] != 3:
            raise ValueError("achieved_goal and desired_goal must have the same (..., 3) shape")
        out = torch.empty(ag.shape[:-1], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._L.grx_fetch_compute_reward(ag.data_ptr(), dg.data_ptr(), out.numel(), DISTANCE_THRESHOLD,
                                                           int(self.reward_type == "sparse"), out.data_ptr(), self._stream()))
        if not as_numpy:
            return out
        r = out.cpu().numpy()
        return r if self.reward_type == "sparse" else r.astype(np.float64)

    def compute_terminated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(achieved_goal).shape[:-1], bool)  # robot_env.py:106-108

    def compute_truncated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(achieved_goal).shape[:-1], bool)  # robot_env.py:110-112

    # checkpoint / resume: core.GoalVecEnv.get_state / set_state (state rows, the device-resident PCG64 streams, TimeLimit counters, hull caches, dispatch order, lane)

    def close(self):
        if getattr(self, "_h", None):
            _native.release_model(self._h)
            self._h = None
        if getattr(self, "_h_big", None):
            _native.release_model(self._h_big)
            self._h_big = None
        if getattr(self, "_h_fast", None):
            _native.release_model(self._h_fast)
            self._h_fast = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
