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
        out.__dict__.pop("_h", None)   # the native handles now belong to self: do not let the temporary destroy them
        out.__dict__.pop("_h_big", None)
        if getattr(self, "lane", None) is not None and hasattr(self, "_lane_make_bufs"):
            self.lane._make_bufs = self._lane_make_bufs()     # the overflow lane's buffer factory held a weak reference to the temporary

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
        if self.output == "torch":      # the two 16-bit halves of the status words as int16 VIEWS (little endian; the flags use bits 0 - 5): no kernel, no sync; valid until the next step like every torch output
            import torch

            halves = self.status.view(torch.int16).view(-1, 2)
            info["status"], info["status_sticky"] = halves[:, 0], halves[:, 1]
        else:
            st = self.status.cpu().numpy()
            info["status"], info["status_sticky"] = st & 0xFFFF, st >> 16
        return info

    def close(self):
        pass

    # ---- frames: the device rows live in the compiled model's (workspace-centred) world frame, the reference / the oracle / fixtures in the MJCF's (mjcf.CompiledModel.origin)
    def load_world_rows(self, rows=None, **more):
        """Overwrite state rows ([N, width] arrays, one per name: qpos, qvel, qacc_ws, mocap, aux, goal, ...) with values given in the MJCF's WORLD frame -- what `MjData` holds in
        the reference, what the oracle and the golden fixtures record.  World positions inside `qpos` / `mocap` are moved into the model's frame in fp64 and rounded once."""
        import torch

        for k, v in {**(rows or {}), **more}.items():
            t = getattr(self, k)
            t.copy_(torch.from_numpy(np.ascontiguousarray(self.model.rows_from_world(k, v)).astype(np.float32)).to(t.device))

    def world_rows(self, key):
        """the device rows `key` as an fp64 array in the MJCF's world frame (synchronises)"""
        return self.model.rows_to_world(key, getattr(self, key).double().cpu().numpy())

    def _stage_idx(self, idx):
        """index list -> int64 device tensor through pinned memory, enqueued and never waited for (a `.to(device)` of a pageable array is a synchronising copy: issued behind
        a step kernel it makes the host wait for that kernel)"""
        st = self.__dict__.get("_idx_stager")
        if st is None:
            st = self.__dict__["_idx_stager"] = PinnedStager(self.num_envs, 1, self.device)
        return st(np.asarray(idx, dtype=np.int64))

    # ---- per-world PCG64 streams of the families that draw on the device (`_rng_dev`: [N, 4] uint64 bit patterns state_hi, state_lo, inc_hi, inc_lo; the mazes carry a fifth
    # word, numpy's buffered 32-bit half).  The per-world host generators (`np_randoms`) exist only for the tasks whose draws stay on the host.
    def world_rng(self, i):
        """numpy Generator positioned at world i's current stream position (inspection / tests: synchronises)"""
        if getattr(self, "_rng_dev", None) is None or not getattr(self, "_device_draws", True):
            return self.np_randoms[i]
        a = [int(x) for x in self._rng_dev[i].cpu().numpy().view(np.uint64)]
        bg = np.random.PCG64()
        st = bg.state
        st["state"] = {"state": (a[0] << 64) | a[1], "inc": (a[2] << 64) | a[3]}
        if len(a) > 4:
            st["has_uint32"], st["uinteger"] = int(a[4] >> 32), int(a[4] & 0xFFFFFFFF)
        bg.state = st
        return np.random.Generator(bg)

    def set_world_rng(self, i, generator):
        """position world i's stream at `generator`'s (a numpy PCG64 Generator): the device-resident row is overwritten, or the host generator replaced"""
        import torch

        if getattr(self, "_rng_dev", None) is None or not getattr(self, "_device_draws", True):
            self.np_randoms[i] = generator
            return
        s = generator.bit_generator.state
        if s["bit_generator"] != "PCG64":
            raise ValueError("the device streams are numpy PCG64 streams")
        mask = (1 << 64) - 1
        row = [s["state"]["state"] >> 64, s["state"]["state"] & mask, s["state"]["inc"] >> 64, s["state"]["inc"] & mask]
        if self._rng_dev.shape[1] > 4:
            row.append((int(s["has_uint32"]) << 32) | int(s["uinteger"]))
        self._rng_dev[i] = torch.from_numpy(np.array(row, dtype=np.uint64).view(np.int64)).to(self.device)

    # ------------------------------------------------------------------ checkpoint / resume (the reference's own round trip: adroit_hand/adroit_hammer.py:380-402,
    # /root/reference/tests/envs/adroit_hand/test_adroit_hammer.py:10-68 -- get_env_state / set_env_state carry EVERYTHING that determines the future of the episode)
    CKPT_SCALARS = ("_has_reset", "_ar_head", "_step_no", "cap_cur")      # python scalars that are state (everything else that is a scalar is configuration)
    CKPT_SKIP = ("_ezpickle_args", "_ezpickle_kwargs")

    def _ckpt_quiesce(self):
        """nothing of this environment may be in flight on a side stream when the tensors are read or overwritten (families with side-stream work override)"""

    def _ckpt_extra_get(self):
        return {}

    def _ckpt_extra_set(self, extra):
        pass

    def get_state(self):
        """Everything that determines the future of this vector environment at a step boundary, as a dict of clones: every device tensor the environment owns (state rows,
        warm start, mocap / stale-kinematics words, goals, per-world model edits, the DEVICE-RESIDENT PCG64 streams, TimeLimit counters, task bookkeeping, the last outputs,
        sticky status words, dispatch order and measured world costs, hull caches), every host array (`_elapsed`, `_needs_reset`, reset attempts ...), the host-side numpy
        generators of the families that draw on the host, the overflow lane's membership lists and time-to-live counters, and the settle chains in flight of the hand
        families.  `set_state(env.get_state())` followed by the same actions reproduces the uninterrupted rollout bit for bit, across autoresets
        (tests/test_gpu_checkpoint.py).  Synchronises the device; does not change the environment."""
        import copy

        import torch

        self._ckpt_quiesce()
        torch.cuda.synchronize(self.device)
        st = dict(cls=type(self).__name__, env_id=getattr(self, "env_id", None), num_envs=int(self.num_envs), tensors={}, arrays={}, scalars={}, rng={}, dicts={})
        for k, v in self.__dict__.items():
            if k in self.CKPT_SKIP:
                continue
            if isinstance(v, torch.Tensor):
                st["tensors"][k] = v.detach().clone()
            elif isinstance(v, np.ndarray):
                st["arrays"][k] = v.copy()
            elif isinstance(v, dict) and v and all(isinstance(x, torch.Tensor) for x in v.values()):
                st["dicts"][k] = {kk: vv.detach().clone() for kk, vv in v.items()}
            elif isinstance(v, list) and v and all(isinstance(g, np.random.Generator) for g in v):
                st["rng"][k] = [copy.deepcopy(g.bit_generator.state) for g in v]
        for k in self.CKPT_SCALARS:
            if k in self.__dict__:
                st["scalars"][k] = self.__dict__[k]
        lane = getattr(self, "lane", None)
        if lane is not None:
            st["lane"] = lane.get_state()
        st["extra"] = self._ckpt_extra_get()
        return st

    def set_state(self, state):
        """Restore a get_state() checkpoint (of this environment or of another instance of the same id and size) IN PLACE: tensors are copied into the existing buffers, so
        every native buffer struct keeps its pointers.  Also accepts a plain {name: tensor} mapping (state rows only)."""
        import torch

        if "tensors" not in state:      # a plain mapping of state rows
            for k, v in state.items():
                getattr(self, k).copy_(v)
            return
        if state["cls"] != type(self).__name__ or state["num_envs"] != self.num_envs or state["env_id"] != getattr(self, "env_id", None):
            raise ValueError(f"checkpoint of {state['cls']}({state['env_id']!r}, num_envs={state['num_envs']}) does not fit {type(self).__name__}({getattr(self, 'env_id', None)!r}, num_envs={self.num_envs})")
        self._ckpt_quiesce()
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            for k, v in state["tensors"].items():
                cur = self.__dict__.get(k)
                if isinstance(cur, torch.Tensor):
                    # an existing tensor is never re-bound: the native buffer structs hold its data_ptr() (a fresh clone would leave the kernels on the old memory)
                    if cur.shape != v.shape or cur.dtype != v.dtype:
                        raise ValueError(f"checkpoint tensor {k!r} is {tuple(v.shape)} {v.dtype}, this environment holds {tuple(cur.shape)} {cur.dtype}")
                    cur.copy_(v)
                else:      # attributes the checkpointed run had created lazily and this instance has not yet (no native struct can point at them)
                    self.__dict__[k] = v.to(self.device).clone() if v.is_cuda else v.clone()
            self._ckpt_extra_set(state.get("extra", {}))      # (families with lazily created arenas make them here, before their tensors are filled)
            for k, d in state["dicts"].items():
                cur = self.__dict__.get(k)
                for kk, vv in d.items():
                    if isinstance(cur, dict) and kk in cur and cur[kk].shape == vv.shape:
                        cur[kk].copy_(vv)
                    else:
                        self.__dict__.setdefault(k, {})[kk] = vv.clone()
            for k, v in state["arrays"].items():
                cur = self.__dict__.get(k)
                if isinstance(cur, np.ndarray) and cur.shape == v.shape and cur.dtype == v.dtype:
                    cur[...] = v
                else:
                    self.__dict__[k] = v.copy()
            for k, v in state["scalars"].items():
                self.__dict__[k] = v
            for k, states in state["rng"].items():
                gens = self.__dict__.get(k)
                for g, s in zip(gens, states):
                    g.bit_generator.state = s
            lane = getattr(self, "lane", None)
            if lane is not None and "lane" in state:
                lane.set_state(state["lane"])
            torch.cuda.synchronize(self.device)


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
        if slot["event"] is not None and not slot["event"].query():      # sixteen calls later its copy has long landed: the wait is a safety net, never the common path
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



# ---------------------------------------------------------------------------------------------- the overflow lane (include/grx_capi.h, grx_overflow_lane)
# Tables of the LARGE model of every family: 256 rows / 4 080 Jacobian-pool words (row offsets are 14 bits, GRX_ROW_PACK: the kitchen's 8 160 fit as well) / 64 contacts (one lane each: twice the fast kernels' lists); ~32-42 KB of LDS per
# world on the generic kernel.  The reference never truncates a contact list (mujoco.mj_step, envs/robot_env.py:341).
RERUN_CAPACITY = {"maxefc": 256, "jpool": 4080, "maxcon": 64}
# the kitchen (124 colliding geoms, condim-6 finger pads = ten rows per contact, 29-dof spans): measured on 16 384 worlds x 100 steps of random actions -- 26 worlds with a
# truncated list on the tables above, 5 at (320, 6128), none at (400, 8160) (profiles/capacity_r04.txt)
KITCHEN_RERUN_CAPACITY = {"maxefc": 400, "jpool": 8160, "maxcon": 64}
LANE_TTL = 4       # steps a world stays in the lane after the last step in which it came within LANE_MARGIN of a capacity of the fast kernel (profiles/ab_r04_lane_parameters.txt: margin 0.5 - 0.9, ttl 2 - 32, 16 - 48 polling workgroups swept)
LANE_MARGIN = 0.8
LANE_POLL_GRID = 16    # entrants per step that can be re-run while the fast launch is still running (more: the serialised launch behind it takes the rest)


def cost_order_alloc(env, n, device, *bufs):
    """Cost-ordered dispatch for a family whose buffer struct has `order` / `cost` (include/grx_capi.h, grx_kitchen_buffers.order): allocates env.cost / env.cost_ema / env.order
    (the identity order: workgroup j -> XCD slice j & 7, position j >> 3, as grx_world_of_block maps it) and points the given STEP buffer structs at them.  The lane's and the
    compact launches keep their own structs (no order).  False when the batch does not qualify (grx_order_by_cost: a multiple of 8 worlds, at most 8 192 per XCD slice)."""
    import torch

    if n % 8 != 0 or not (1024 <= n <= 65536):
        return False
    per = n // 8
    env.cost, env.cost_ema = torch.zeros(n, dtype=torch.int32, device=device), torch.zeros(n, dtype=torch.float32, device=device)
    env.order = ((torch.arange(8, device=device, dtype=torch.int32) * per).unsqueeze(1) + torch.arange(per, device=device, dtype=torch.int32).unsqueeze(0)).t().contiguous().view(-1)
    for b in bufs:
        b.order, b.cost = env.order.data_ptr(), env.cost.data_ptr()
    return True


def cost_order_update(env, alpha=None):
    """order <- per XCD slice, worlds by decreasing (moving average of the) measured duration of the launches so far; in place, on the environment's stream"""
    import os

    from . import _native

    alpha = float(os.environ.get("GRX_BALANCE_ALPHA", 0.1 if alpha is None else alpha))      # weight of the newest sample in the moving average the order is sorted by
    _native.check(env._L.grx_order_by_cost(env.cost.data_ptr(), env.cost_ema.data_ptr(), alpha, env.num_envs, env.order.data_ptr(), env._stream()))


def create_rerun_model(L, model, device_index, enabled=True, capacity=None):
    """handle of `model` compiled with RERUN_CAPACITY, or None when the lane is switched off (GRX_NO_OVERFLOW_RERUN=1: round-2 behaviour, contacts dropped and flagged)"""
    import ctypes
    import os

    from . import _native

    if not enabled or os.environ.get("GRX_NO_OVERFLOW_RERUN") is not None:
        return None
    cap = dict(capacity or RERUN_CAPACITY)
    if os.environ.get("GRX_RERUN_CAPACITY"):      # "rows,pool,contacts" (experiments: which large tables leave no world of a workload with a truncated list)
        cap = dict(zip(("maxefc", "jpool", "maxcon"), (int(x) for x in os.environ["GRX_RERUN_CAPACITY"].split(","))))
    H, I, F = model.with_capacity(**cap).pack()
    return _native.acquire_model(H, I, F, device_index)


class _LaneBuf:
    """flags [N] u8 + (next_count, entry_count) i32 in ONE tensor (zeroed by one fill), and the two index lists"""

    def __init__(self, n, device):
        import torch

        self.head = torch.zeros((n + 3) // 4 + 2, dtype=torch.int32, device=device)
        self.flags = self.head.view(torch.uint8)[:n]
        self.counts = self.head[-2:]
        self.next_list, self.entry_list = torch.zeros(n, dtype=torch.int32, device=device), torch.zeros(n, dtype=torch.int32, device=device)
        base = self.counts.data_ptr()
        self.next_count_ptr, self.entry_count_ptr = base, base + 4


ENTRY_CAP = 256    # worlds that may overflow for the FIRST time in one step and still be re-run (more: the excess goes on with dropped contacts and the sticky flag says so)


class OverflowLane:
    """Host side of the overflow lane (include/grx_capi.h, grx_overflow_lane): the device-side lists / flags and the launch order of one step.  Nothing here waits
    for the device or reads a flag back: which worlds are in the lane is known to the kernels only.

        lane = OverflowLane(n, device, model, make_bufs)          # make_bufs(mask_tensor_or_None) -> a fresh buffer struct of the family (with a .lane field)
        lane.step(mask, launch_fast, launch_large, fast_bufs)     # launch_fast(bufs), launch_large(bufs): enqueue on torch's CURRENT stream

    step(): the large-table kernel (one workgroup per entry of the lane's current list) runs on a second stream CONCURRENTLY with the fast kernel; when both are done
    it runs once more over the worlds that overflowed in this step (the only serialised part: a launch over an empty list otherwise); then the lists built in this step
    become current.  The size of a list is bounded by the grid of the launch that will walk it (grx_overflow_lane.next_cap / entry_cap), which the host fixes one
    step ahead from the newest counters that have reached it."""

    def __init__(self, n, device, model, make_bufs, ttl=LANE_TTL, mode="lane", lane_first=False, margin=LANE_MARGIN, poll_grid=LANE_POLL_GRID, handoff=False):
        """mode "lane": worlds near a capacity move to a standing lane that runs next to the fast launch (families whose contact-rich states persist: hand + object,
        kitchen, Adroit door / relocate: a few worlds per step and thousand).  mode "entry": no standing lane, an overflowing world is re-run right behind the fast
        launch -- for families where an overflow is a rare event (Fetch: 2 worlds in 100 steps of 4096), whose step is too short to hide the two cross-stream waits
        a standing lane costs per step (+0.19 ms on 3.3 ms, measured)."""
        import os
        self.mode = os.environ.get("GRX_LANE_MODE", mode)
        # handoff (FetchPickAndPlace, include/grx_capi.h grx_fetch_buffers.handoff): three tiers -- the fast kernel hands worlds off MID-STEP to the standing lane's launch (the middle
        # tables, `launch_lane` of step()), which hands what exceeds ITS tables on to the entry launch (the large tables); entries are resumed at their substep, never re-run
        self.handoff = bool(handoff)
        self.entry_cap = max(ENTRY_CAP, n // 16) if self.handoff else ENTRY_CAP
        # lane_first: the standing lane's launch is submitted BEFORE the fast launch (its worlds are the heaviest of the batch: a hand jammed into the door takes 5 - 9 ms
        # against the 11 ms of the whole fast launch, so they should start first).  Measured per family (profiles/ab_r03_lane_first.txt): AdroitDoor +7 %, hand + touch -0.6 %
        # (there the fast kernel then waits for the lane's launch: tools/lane_cost_probe.py) -- hence a per-family switch.
        self.lane_first = bool(int(os.environ.get("GRX_LANE_FIRST", int(lane_first))))
        self.spacer = int(os.environ.get("GRX_LANE_SPACER", 0))      # GPU cycles the main stream idles between the lane's launch and the fast launch (lane_first only)

        import torch

        from . import _native
        from .mjcf.compiler import DIMS

        self.n, self.device = n, device
        self.cur, self.nxt, self.scratch = _LaneBuf(n, device), _LaneBuf(n, device), _LaneBuf(n, device)
        self.ttl = torch.zeros(n, dtype=torch.int8, device=device)
        dims = model.tables["dims"]
        req = lambda k, default: int(dims[DIMS.index(k + "_req")]) or default
        cap = (req("maxefc", 144), req("jpool", 2032), min(req("maxcon", 32), 32))
        # a world moves to the lane when it comes within LANE_MARGIN of a capacity of the fast kernel (and stays `ttl` steps past the last such step): most worlds
        # then enter the lane without ever overflowing, i.e. without the serialised re-run
        margin, ttl = float(os.environ.get("GRX_LANE_MARGIN", margin)), int(os.environ.get("GRX_LANE_TTL", ttl))      # (experiments)
        self.soft = tuple(int(margin * v) for v in cap) + (int(ttl),)
        self._make_bufs = make_bufs
        self.side = torch.cuda.Stream(device=device, priority=int(os.environ.get("GRX_LANE_PRIO", "-1")))     # the lane's worlds start before the fast kernel fills the chip
        self._Lane = _native.OverflowLaneStruct
        # The counters reach the host with a delay (copied to pinned memory after every step, read when the copy has landed, never waited for); `cap_cur` is the
        # capacity the CURRENT list was built under = the grid of the launch that walks it.
        self._pin = [dict(buf=torch.zeros(2, dtype=torch.int32, pin_memory=True), event=None, age=0) for _ in range(4)]
        self._pin_next, self._seen, self._age = 0, 0, 0
        self.cap_cur = 64
        # polling workgroups of the standing launch (grx_overflow_lane.ready / progress / poll_*): an entrant is re-run while the fast launch is still running instead of behind it
        self.poll_grid = int(os.environ.get("GRX_LANE_POLL", poll_grid)) if self.mode == "lane" else 0      # per family: the kitchen, whose fast tables are small by choice, runs 32 (profiles/ab_r05_kitchen_lane_params.txt)
        self._poll = torch.zeros(self.poll_grid + 1, dtype=torch.int32, device=device) if self.poll_grid > 0 else None      # ready[poll_grid], progress
        self._fast_grid = (n + 7) & ~7      # workgroups of a fast launch (csrc/grx_kernels.hip, grx_grid_for)

    def _refresh(self):
        best = None
        for slot in self._pin:
            if slot["event"] is not None and slot["event"].query() and (best is None or slot["age"] > best["age"]):
                best = slot
        if best is not None:
            self._seen = int(best["buf"][0]) + int(best["buf"][1])

    def _fast(self, skip, out, next_cap, join=True):
        L = self._Lane()
        L.skip = None if skip is None else skip.flags.data_ptr()
        L.entry_count, L.entry_list, L.entry_cap = out.entry_count_ptr, out.entry_list.data_ptr(), self.entry_cap
        if join:      # worlds that come close to a capacity move to the lane of the next step
            L.next_flags, L.next_count, L.next_list, L.ttl, L.next_cap = out.flags.data_ptr(), out.next_count_ptr, out.next_list.data_ptr(), self.ttl.data_ptr(), next_cap
            L.soft_maxefc, L.soft_jpool, L.soft_maxcon, L.ttl_init = self.soft
            if self._poll is not None:      # entries are published to the polling workgroups, every workgroup reports its end
                L.ready, L.ready_cap, L.progress = self._poll.data_ptr(), self.poll_grid, self._poll.data_ptr() + 4 * self.poll_grid
        return L

    def _large(self, lst, count_ptr, out, grid, next_cap):
        L = self._Lane()
        L.grid = int(grid)
        L.list, L.count = lst.data_ptr(), count_ptr
        L.next_flags, L.next_count, L.next_list, L.ttl, L.next_cap = out.flags.data_ptr(), out.next_count_ptr, out.next_list.data_ptr(), self.ttl.data_ptr(), int(next_cap)
        L.soft_maxefc, L.soft_jpool, L.soft_maxcon, L.ttl_init = self.soft
        return L

    def step(self, mask, launch_fast, launch_large, fast_bufs, launch_lane=None):
        """mask: the uint8 tensor of the worlds this step covers (None: all); a lane world that is masked out keeps its place.  fast_bufs: the environment's own
        buffer struct for the fast launch (its .lane is set here and cleared again).  launch_lane: the launcher of the STANDING lane's kernel when it is not the large-table
        one (hand-off mode: the middle tables); default launch_large."""
        import os

        import torch

        if self.mode == "entry":       # no standing lane: a world that overflows is re-run on the large tables right behind the fast launch, and that is all
            return self.rerun_only(mask, fast_bufs, launch_fast, launch_large)

        main = torch.cuda.current_stream(self.device)
        cur, nxt = self.cur, self.nxt
        self._refresh()
        next_cap = int(min(self.n, 64 + 2 * self._seen))       # capacity of the list built in this step = grid of the next step's lane launch
        nxt.head.zero_()
        fast_bufs.lane = self._fast(cur, nxt, next_cap)
        # The fast kernel is submitted FIRST and the lane's launch second, on the other stream.  Measured (tools/lane_cost_probe.py): with the lane's launch in front,
        # the fast kernel does not start before the lane's kernel has ENDED (1.1 ms of a 3.1 ms Fetch step) -- on the same stream or on a second one: the completion
        # marker behind the lane's kernel holds up the packet processor for both queues.  Behind the fast kernel the lane's workgroups start when a CU has room and
        # end before the fast launch does.
        launch_lane = launch_lane or launch_large
        b_lane = self._make_bufs(mask)
        b_lane.lane = self._large(cur.next_list, cur.next_count_ptr, nxt, self.cap_cur, next_cap)
        if self.handoff:      # the standing lane's kernel runs the middle tables: what exceeds them is handed on (entries flagged "large": the polling workgroups leave them to the entry launch)
            b_lane.handoff_large = 1
            b_lane.lane.entry_count, b_lane.lane.entry_list, b_lane.lane.entry_cap = nxt.entry_count_ptr, nxt.entry_list.data_ptr(), self.entry_cap
        if self._poll is not None:
            self._poll.zero_()
            L = b_lane.lane
            L.grid = int(self.cap_cur) + self.poll_grid      # the last poll_grid workgroups poll this step's entry list
            L.ready, L.ready_cap, L.progress, L.progress_total = self._poll.data_ptr(), self.poll_grid, self._poll.data_ptr() + 4 * self.poll_grid, self._fast_grid
            L.poll_list, L.poll_grid = nxt.entry_list.data_ptr(), self.poll_grid
        trace = getattr(self, "trace", None)      # diagnostics (tools/lane_dbg_fetch.py): set to [] to collect (step start, lane start, lane end, entry start, entry end) timing events
        ev0 = torch.cuda.Event(enable_timing=trace is not None)
        ev0.record(main)
        if trace is not None:
            _launch_lane = launch_lane

            def launch_lane(b):
                t0 = torch.cuda.Event(enable_timing=True); t0.record()
                _launch_lane(b)
                t1 = torch.cuda.Event(enable_timing=True); t1.record()
                trace.append([ev0, t0, t1])
        if self.lane_first:
            self.side.wait_event(ev0)
            with torch.cuda.stream(self.side):
                launch_lane(b_lane)
                ev1 = torch.cuda.Event()
                ev1.record(self.side)
            if self.spacer > 0:      # the lane's workgroups must be RESIDENT before the fast launch fills every wave slot: a few microseconds of nothing on the main stream
                torch.cuda._sleep(self.spacer)
            launch_fast(fast_bufs)
        else:
            launch_fast(fast_bufs)
            self.side.wait_event(ev0)
            with torch.cuda.stream(self.side):
                launch_lane(b_lane)
                ev1 = torch.cuda.Event()
                ev1.record(self.side)
        main.wait_event(ev1)
        b_entry = self._make_bufs(mask)
        b_entry.lane = self._large(nxt.entry_list, nxt.entry_count_ptr, nxt, self.entry_cap, next_cap)
        if self._poll is not None:      # the entries a polling workgroup has claimed are skipped
            b_entry.lane.ready, b_entry.lane.ready_cap = self._poll.data_ptr(), self.poll_grid
        if trace is not None:
            x0 = torch.cuda.Event(enable_timing=True); x0.record()
        launch_large(b_entry)
        if trace is not None:
            x1 = torch.cuda.Event(enable_timing=True); x1.record()
            trace[-1] += [x0, x1]
        self._keep = (b_lane, b_entry)
        fast_bufs.lane = self._Lane()       # the environment's struct is also used for reset-time launches: no stale skip list
        slot = self._pin[self._pin_next]
        self._pin_next = (self._pin_next + 1) % len(self._pin)
        if os.environ.get("GRX_LANE_DBG") == "nopin":
            pass
        elif slot["event"] is None or slot["event"].query():      # (a slot whose copy is still in flight is skipped: the host never waits)
            done = torch.cuda.Event()
            done.record(main)
            self.side.wait_event(done)
            with torch.cuda.stream(self.side):                  # the read-back rides the side stream: a D2H copy in the main stream sits in the critical path of every step
                slot["buf"].copy_(nxt.counts, non_blocking=True)
                slot["event"] = torch.cuda.Event()
                slot["event"].record(self.side)
            self._age += 1
            slot["age"] = self._age
        self.cur, self.nxt, self.cap_cur = nxt, cur, next_cap

    def rerun_only(self, mask, fast_bufs, launch_fast, launch_large):
        """reset-time step launches (settle chains): every masked world is stepped by the fast kernel whatever lane it is in; a world that overflows is re-run on the
        large tables at once.  Lane membership does not change."""
        sc = self.scratch
        sc.head.zero_()
        fast_bufs.lane = self._fast(None, sc, 0, join=False)
        launch_fast(fast_bufs)
        fast_bufs.lane = self._Lane()
        b = self._make_bufs(mask)
        b.lane = self._large(sc.entry_list, sc.entry_count_ptr, sc, ENTRY_CAP, 0)
        launch_large(b)

    def get_state(self):
        """membership of the lane at a step boundary (the list the NEXT step's lane launch walks, the flags the fast launch skips by, the time-to-live counters) and the
        grid that list was built under; the delayed host view of the counters (`_seen`) rides along so that the capacities of the following steps repeat as well"""
        cur = self.cur
        return dict(head=cur.head.clone(), next_list=cur.next_list.clone(), entry_list=cur.entry_list.clone(), ttl=self.ttl.clone(), cap_cur=int(self.cap_cur), seen=int(self._seen))

    def set_state(self, st):
        import torch

        cur = self.cur
        cur.head.copy_(st["head"]); cur.next_list.copy_(st["next_list"]); cur.entry_list.copy_(st["entry_list"]); self.ttl.copy_(st["ttl"])
        self.cap_cur, self._seen = int(st["cap_cur"]), int(st["seen"])
        for slot in self._pin:      # counter read-backs in flight belong to the run that was replaced
            if slot["event"] is not None:
                slot["event"].synchronize()
            slot["event"], slot["age"] = None, 0
        self._age = 0

    def count(self):
        """worlds in the lane right now (synchronises: diagnostics only)"""
        return 0 if self.mode == "entry" else int(min(self.cur.counts[0].item(), self.cap_cur))

    def entered_last_step(self):
        """world indices that claimed a re-run in the last step (synchronises: tests only)"""
        buf = self.scratch if self.mode == "entry" else self.cur
        k = int(min(buf.counts[1].item(), self.entry_cap))
        return buf.entry_list[:k].cpu().numpy()
