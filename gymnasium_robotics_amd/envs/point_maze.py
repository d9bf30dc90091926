"""Batched PointMaze environments on the MI355X engine (host side, Python).

Vectorised drop-in for PointMazeEnv (/root/reference/gymnasium_robotics/envs/maze/point_maze.py:316-419, ids
PointMaze_{UMaze,Open,Medium,Large}[_Diverse_G|_Diverse_GR][Dense]-v3, __init__.py:960-1078) with the GoalEnv contract.
Per-step work = ONE launch of grx_point_step_kernel; episode-boundary sampling (goal cell, reset cell, xy noise:
maze_v4.py:278-379) stays on the host with one numpy PCG64 per world, so ``reset(seed=s, options=...)`` reproduces the
reference's draws for seed ``s + i`` (pinned by the reference's own golden vectors, tests/test_cpu_maze.py).
"""
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from .. import _native
from ..core import GoalVecEnv, np_random
from ..mjcf import CompiledModel, compile_mjcf, load_model
from ..spaces import Box, Dict, batch_space
from .maze_spec import (redraw_goal, ANT_FRAME_SKIP, ANT_MAZE_HEIGHT, ANT_MAZE_SIZE_SCALING, GOAL_RADIUS, MAPS, POINT_MAZE_HEIGHT, POINT_MAZE_SIZE_SCALING, Maze,
                        parse_ant_maze_id, parse_point_maze_id, sample_maze_reset)

_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models")


# Engine capacities of the ant models: random rollouts peak at 4 contacts / 18 rows / ~200 Jacobian-pool words, the wall-pushing fixture (tests/golden/ant_Large_teacher.npz)
# at 3 / 16.  16 contacts / 64 rows / 512 pool words keep a 3.5x margin and make the working set 12 488 B = 10 allocation granules = TWELVE worlds per CU -- what the
# kernel's 168 VGPRs allow anyway (3 waves per SIMD).  Round 5 measured that the step kernel's throughput is nearly PROPORTIONAL to the resident worlds (a wave's time is set
# by its dependent chains, not by issue slots: profiles/ab_r05_two_worlds_occupancy.txt -- 8 -> 10 worlds per CU: +15 %), so LDS bytes are the lever; rounds 1 - 4 ran
# 96 rows / 1 024 words = 15 176 B = 10 worlds per CU.  A world that exceeds a capacity drops the excess contacts for that substep and raises the sticky status bit.
ANT_CAPACITY = {"maxcon": 16, "maxefc": 64, "jpool": 512, "split_spans": False}


def load_point_maze_model(maze: Maze, layout_name: Optional[str], assets_root: Optional[str] = None, agent: str = "point") -> CompiledModel:
    """Model tables of the agent MJCF (point.xml / ant.xml) + one wall box per wall cell.  Compiled from MJCF when an asset
    tree is available (needed for custom maze maps), else the packaged blob of the registered wall layout.
    ant.xml: the reference takes it from the gymnasium package [3P]; the tree's envs/mujoco/assets/ant.xml is the same model."""
    assets_root = assets_root or os.environ.get("GRX_ASSETS_ROOT")
    if assets_root:
        xml = os.path.join(assets_root, "point", "point.xml") if agent == "point" else os.path.join(assets_root, "..", "mujoco", "assets", "ant.xml")
        return compile_mjcf(xml, mutate=maze.add_walls, capacity=ANT_CAPACITY if agent == "ant" else None)
    if layout_name is None:
        raise OSError("custom maze maps need the MJCF assets (assets_root / $GRX_ASSETS_ROOT)")
    path = os.path.join(_MODELS_DIR, f"{agent}_{layout_name.split('_')[0]}.npz")
    if not os.path.exists(path):
        raise OSError(f"File {path} does not exist")
    model = load_model(path)
    # (the packaged blobs were compiled with round 1's capacity requests: the current ones are applied here, the tables themselves do not depend on them)
    return model.with_capacity(**{k: ANT_CAPACITY[k] for k in ("maxcon", "maxefc", "jpool")}) if agent == "ant" else model


MAZE_SPLIT_PARTS = 5      # default of GRX_MAZE_SPLIT for batches of more than one round of worlds, clamped to frame_skip (see PointMazeVecEnv.__init__; profiles/ab_r06_split_fences.txt: AntMaze_Large @8 192: 3.78 M unsplit -> 4.01 / 4.15 / 4.28 M for 2 / 3 / 5 parts; @4 096: 2.88 -> 3.90 M; @16 384: 4.25 -> 4.45 M)


class PointMazeVecEnv(GoalVecEnv):
    AGENT, N_SUBSTEPS, OBS_SKIP, DEFAULT_MAX_EPISODE_STEPS = "point", 1, 0, 300
    MAZE_GEOMETRY = (POINT_MAZE_SIZE_SCALING, POINT_MAZE_HEIGHT)
    _parse_id = staticmethod(parse_point_maze_id)

    def __init__(self, env_id: Optional[str] = "PointMaze_UMaze-v3", num_envs: int = 1, device: Optional[str] = None, maze_map=None,
                 reward_type: Optional[str] = None, continuing_task: bool = True, reset_target: bool = False,
                 position_noise_range: float = 0.25, max_episode_steps: Optional[int] = -1, autoreset_mode: str = "next_step",
                 output: str = "numpy", assets_root: Optional[str] = None, model: Optional[CompiledModel] = None, seed_offset: int = 0):
        layout, rt, mes = (None, "sparse", self.DEFAULT_MAX_EPISODE_STEPS)
        if maze_map is None:
            layout, rt, mes = self._parse_id(env_id)
            maze_map = MAPS[layout]
        self.reset_target = bool(reset_target)
        self.env_id, self.reward_type = env_id, reward_type or rt
        self.continuing_task, self.position_noise_range = continuing_task, position_noise_range
        self.max_episode_steps = mes if max_episode_steps == -1 else max_episode_steps
        self.autoreset_mode, self.output, self.num_envs, self.seed_offset = autoreset_mode, output, int(num_envs), int(seed_offset)
        self.maze = Maze(maze_map, *self.MAZE_GEOMETRY)
        # a maze whose ONLY reset cell is also its only goal cell: the reference's generate_reset_pos (maze/maze_v4.py:284-297, 400-418) redraws the reset CELL CENTRE until it is
        # farther than half a cell from the noisy goal and would never return; refused here, loudly, instead of spinning (host draws) or being cut short (the device loop is
        # bounded).  Two DISTINCT cells are at least one cell pitch apart, so the centre of the reset cell is >= 0.75 cells from a goal that carries <= 0.25 cells of noise:
        # the reference's loop ends on its first draw there (an 'r' cell next to a 'g' cell is a valid maze).
        ug, ur = np.asarray(self.maze.unique_goal_locations, dtype=np.float64).reshape(-1, 2), np.asarray(self.maze.unique_reset_locations, dtype=np.float64).reshape(-1, 2)
        if len(ur) == 1 and len(ug) == 1 and np.linalg.norm(ur[0] - ug[0]) < 0.5 * self.maze.maze_size_scaling:
            raise ValueError("this maze has a single reset cell that is also its single goal cell: a reset position farther than half a cell from the goal does not exist")
        if not torch.cuda.is_available():      # (after the map checks: a bad map is reported as such on any machine, tests/test_cpu_maze.py)
            raise RuntimeError("PointMazeVecEnv needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device or "cuda:0")
        self.model = model or load_point_maze_model(self.maze, layout, assets_root, self.AGENT)
        self.nq, self.nv, self.nu = self.model.dim("nq"), self.model.dim("nv"), self.model.dim("nu")
        self._L = _native.lib()
        H, I, F = self.model.pack()
        self._h = _native.acquire_model(H, I, F, self.device.index or 0)   # shared with every other environment of the same compiled tables (reference-counted)
        self.task = _native.PointTaskStruct(self.N_SUBSTEPS, int(self.reward_type == "sparse"), int(continuing_task), int(self.AGENT == "ant"),
                                            GOAL_RADIUS, 5.0)
        self.obs_dim = self.nq + self.nv - self.OBS_SKIP
        n, d = self.num_envs, self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=d)
        self.qpos, self.qvel, self.qacc_ws = z(n, self.nq), z(n, self.nv), z(n, self.nv)
        self.goal, self.action, self.obs, self.achieved, self.reward = z(n, 2), z(n, self.nu), z(n, self.obs_dim), z(n, 2), z(n)
        self.success, self.terminated = z(n, dtype=torch.uint8), z(n, dtype=torch.uint8)
        self.status, self.mask = z(n, dtype=torch.int32), torch.ones(n, dtype=torch.uint8, device=d)
        self.packed = z(n, self.obs_dim + 6)   # [obs | achieved | desired | reward | success] rows written by the step kernel (cross-rank gather)
        self._bufs, self._bufs_masked = self._make_bufs(None), self._make_bufs(self.mask)
        # SPLIT STEP (include/grx_capi.h grx_point_buffers.split_parts): P workgroups per world, each running its share of the frame_skip substeps -- a launch whose worlds do not fill a
        # whole number of rounds of wave slots (8 192 ant worlds on 3 072 slots: 2.67) no longer pays for the round it does not fill.  Bit-identical to the plain launch
        # (tests/test_gpu_maze.py::test_split_step_is_the_plain_step).  GRX_MAZE_SPLIT=P (1: off); default: MAZE_SPLIT_PARTS for batches of more than one round of worlds.
        import os
        self._split = max(1, min(self.N_SUBSTEPS, 8, int(os.environ.get("GRX_MAZE_SPLIT", MAZE_SPLIT_PARTS if n > 3072 else 1)))) if n >= 64 else 1
        if self._split > 1:
            self._split_state = z(n, 2, dtype=torch.int32)
            for b in (self._bufs, self._bufs_masked):
                b.split_state, b.split_parts = self._split_state.data_ptr(), self._split
        # reset staging (see _reset_worlds): pinned host rows [start xy | goal xy] + world indices, their device mirrors, the kernel's argument block
        self._stage_host, self._idx_host = torch.empty(n, 4, dtype=torch.float32, pin_memory=True), torch.empty(n, dtype=torch.int32, pin_memory=True)
        self._stage_dev, self._idx_dev, self._stage_event = z(n, 4), z(n, dtype=torch.int32), None
        self._qpos0 = torch.from_numpy(self.model.tables["qpos0"].astype(np.float32)).to(d)
        a = _native.MazeResetArgsStruct()
        a.idx, a.stage, a.qpos0 = self._idx_dev.data_ptr(), self._stage_dev.data_ptr(), self._qpos0.data_ptr()
        a.nq, a.nv, a.obs_dim, a.obs_skip, a.goal_radius = self.nq, self.nv, self.obs_dim, self.OBS_SKIP, GOAL_RADIUS
        for name in ("qpos", "qvel", "qacc_ws", "goal", "obs", "achieved", "reward", "success", "packed"):
            setattr(a, name, getattr(self, name).data_ptr())
        self._reset_args = a
        self.single_action_space = Box(-1.0, 1.0, (self.nu,), np.float32)
        self.single_observation_space = Dict(dict(
            observation=Box(-np.inf, np.inf, (self.obs_dim,), np.float64), achieved_goal=Box(-np.inf, np.inf, (2,), np.float64),
            desired_goal=Box(-np.inf, np.inf, (2,), np.float64)))
        self.action_space = batch_space(self.single_action_space, n)
        self.observation_space = batch_space(self.single_observation_space, n)
        self._check_goal_space()
        # The reset draws (goal cell, reset cell, xy noise: integers / uniform of the world's numpy PCG64 stream) are made ON THE DEVICE from device-resident streams
        # (grx_maze_sample_resets_device, bit-equal to numpy).  With reset_target=True the same stream also feeds MazeEnv.update_goal in the middle of an episode -- host
        # logic that reads which worlds reached their goal -- so that mode keeps per-world numpy generators on the host for both.
        self._device_draws = not self.reset_target
        if self._device_draws:
            self._goal_xy = torch.from_numpy(np.ascontiguousarray(np.asarray(self.maze.unique_goal_locations, dtype=np.float64).reshape(-1, 2))).to(self.device)
            self._reset_xy = torch.from_numpy(np.ascontiguousarray(np.asarray(self.maze.unique_reset_locations, dtype=np.float64).reshape(-1, 2))).to(self.device)
            self._seed_worlds([None] * n)
        else:
            self.np_randoms = [np_random(None)[0] for _ in range(n)]
        self._elapsed = np.zeros(n, np.int64)
        self._needs_reset = np.zeros(n, bool)
        self._has_reset = False
        self.kernel_events = None  # when a list: (start, end) HIP events around every step-kernel launch (benchmarks)

    def _make_bufs(self, mask):
        b = _native.PointBuffersStruct()
        for name in ("qpos", "qvel", "qacc_ws", "goal", "action", "obs", "achieved", "reward", "success", "terminated", "status", "packed"):
            setattr(b, name, getattr(self, name).data_ptr())
        b.mask = None if mask is None else mask.data_ptr()
        return b

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _seed_worlds(self, seeds):
        """one numpy PCG64 per world, seeded like gymnasium.utils.seeding.np_random [3P]; the raw (state, inc) pairs and the empty 32-bit buffer go to the device"""
        st = np.zeros((self.num_envs, 5), np.uint64)
        mask = (1 << 64) - 1
        for i, sd in enumerate(seeds):
            s = np_random(sd)[0].bit_generator.state
            st[i] = [s["state"]["state"] >> 64, s["state"]["state"] & mask, s["state"]["inc"] >> 64, s["state"]["inc"] & mask, (int(s["has_uint32"]) << 32) | int(s["uinteger"])]
        self._rng_dev = torch.from_numpy(st.view(np.int64)).to(self.device)

    def _sample_on_device(self, idx, options):
        """index list through pinned memory, the draws by one kernel into the staging rows grx_maze_reset_rows reads; nothing waits"""
        k = len(idx)
        if self._stage_event is not None:
            self._stage_event.synchronize()
        self._idx_host.numpy()[:k] = idx
        self._idx_dev[:k].copy_(self._idx_host[:k], non_blocking=True)
        self._stage_event = torch.cuda.Event()
        self._stage_event.record()
        fixed = []
        for key, what in (("goal_cell", "Goal"), ("reset_cell", "Reset")):
            cell = (options or {}).get(key)
            if cell is None:
                fixed.append(None)
                continue
            assert self.maze.map_length > cell[0] and self.maze.map_width > cell[1]
            assert self.maze.maze_map[cell[0]][cell[1]] != 1, f"{what} can't be placed in a wall cell, {cell}"
            fixed.append(np.ascontiguousarray(self.maze.cell_rowcol_to_xy(cell), dtype=np.float64))
        _native.check(self._L.grx_maze_sample_resets_device(
            self._rng_dev.data_ptr(), self._idx_dev.data_ptr(), k, self._goal_xy.data_ptr(), int(self._goal_xy.shape[0]), self._reset_xy.data_ptr(), int(self._reset_xy.shape[0]),
            float(self.position_noise_range), float(self.maze.maze_size_scaling), None if fixed[0] is None else fixed[0].ctypes.data, None if fixed[1] is None else fixed[1].ctypes.data,
            self._stage_dev.data_ptr(), self._stream()))

    # ------------------------------------------------------------------ reset (point_maze.py:377-390, maze_v4.py:299-358)
    def _reset_worlds(self, idx, options=None, keep_outcome=False):
        """Host: the reference's draws for the listed worlds (generate_reset_pos / generate_target_goal, maze_v4.py:299-358) into a pinned staging row per
        world.  Device: two asynchronous copies and ONE kernel (grx_maze_reset_rows) that writes state, goal, observation and the packed row -- nothing
        here waits for the step kernel that may still be running, so the draws overlap it."""
        k = len(idx)
        if k == 0:
            return
        if self._device_draws:
            self._sample_on_device(idx, options)
        else:
            if self._stage_event is not None:
                self._stage_event.synchronize()          # the previous reset's copies have left the pinned buffers (normally long ago)
            stage, ih = self._stage_host.numpy(), self._idx_host.numpy()
            for j, w in enumerate(idx):
                goal, start = sample_maze_reset(self.maze, self.np_randoms[w], self.position_noise_range, options)
                stage[j, 0:2], stage[j, 2:4] = start, goal
            ih[:k] = idx
            self._stage_dev[:k].copy_(self._stage_host[:k], non_blocking=True)
            self._idx_dev[:k].copy_(self._idx_host[:k], non_blocking=True)
            self._stage_event = torch.cuda.Event()
            self._stage_event.record()
        a = self._reset_args
        a.keep_outcome = int(keep_outcome)
        _native.check(self._L.grx_maze_reset_rows(ctypes.byref(a), k, self._stream()))
        self._elapsed[idx] = 0
        self._needs_reset[idx] = False

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            seeds = [seed + self.seed_offset + i for i in range(self.num_envs)] if np.isscalar(seed) else list(seed)
            if self._device_draws:
                self._seed_worlds(seeds)
            else:
                self.np_randoms = [np_random(s)[0] for s in seeds]
        with torch.cuda.device(self.device):
            self._reset_worlds(np.arange(self.num_envs), options)
        self._has_reset = True
        return self._obs_dict(), self._info()

    def _info(self, success=None):
        s = self.success if success is None else success
        if self.output == "torch":
            return self._status_info({"success": s.bool()})
        return self._status_info({"success": s.cpu().numpy().astype(bool)})

    # ------------------------------------------------------------------ step (point_maze.py:392-406)
    def step(self, actions):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        a = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions, dtype=np.float32))
        if tuple(a.shape) != (self.num_envs, self.nu):
            raise ValueError("Action dimension mismatch")
        self.action.copy_(a.to(torch.float32), non_blocking=True)
        with torch.cuda.device(self.device):
            pending = np.nonzero(self._needs_reset)[0] if self.autoreset_mode == "next_step" else np.zeros(0, np.int64)
            bufs = self._bufs
            if len(pending):
                self.mask.fill_(1)
                self.mask.index_fill_(0, self._stage_idx(pending), 0)      # (pinned staging + index_fill_: nothing here waits for the running kernel)
                bufs = self._bufs_masked
            ev = self.kernel_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _native.check(self._L.grx_point_step(self._h, ctypes.byref(self.task), ctypes.byref(bufs), self.num_envs, self._stream()))
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
            stepped = ~self._needs_reset
            self._elapsed[stepped] += 1
            terminated = self.terminated.cpu().numpy().astype(bool) & stepped if not self.continuing_task else np.zeros(self.num_envs, bool)
            truncated = np.zeros(self.num_envs, bool)
            if self.max_episode_steps is not None:
                truncated = stepped & (self._elapsed >= self.max_episode_steps)
            if len(pending):
                self._reset_worlds(pending)       # the reset kernel zeroes reward[] and the packed row's reward word of these worlds
            new_goals, final_obs, step_success = None, None, None
            if self.reset_target and self.continuing_task and len(self.maze.unique_goal_locations) > 1:
                # MazeEnv.update_goal (maze_v4.py:400-418): the returned observation still carries the goal that was just reached
                hit = np.nonzero(self.success.cpu().numpy().astype(bool) & stepped)[0]
                if len(hit):
                    ag, dg = self.achieved[hit].double().cpu().numpy(), self.goal[hit].double().cpu().numpy()
                    new_goals = (hit, np.stack([redraw_goal(self.maze, self.np_randoms[w], ag[k], dg[k], self.position_noise_range) for k, w in enumerate(hit)]))
            done = terminated | truncated
            if self.autoreset_mode == "next_step":
                self._needs_reset |= done
            elif self.autoreset_mode == "same_step" and done.any():
                # the step's reward / success / terminal observation belong to the finished episode; the returned observation is the reset one.
                # Everything below is enqueued behind the step kernel without waiting for it (pinned staging, device-side index list).
                rows = np.nonzero(done)[0]
                step_success = self.success.clone()
                if self.output == "torch":
                    if self._stage_event is not None:
                        self._stage_event.synchronize()
                    self._idx_host.numpy()[:len(rows)] = rows
                    ti = self._idx_dev[:len(rows)]
                    ti.copy_(self._idx_host[:len(rows)], non_blocking=True)
                    ti = ti.long()
                    final_obs = {"observation": self.obs[ti], "achieved_goal": self.achieved[ti], "desired_goal": self.goal[ti]}
                else:
                    final_obs = self._obs_dict(rows=rows)
                self._reset_worlds(rows, keep_outcome=True)       # reward[] and the packed row's reward / success keep the finished episode's values
        obs = self._obs_dict()
        if new_goals is not None:
            if self.output == "torch":
                obs = dict(obs, desired_goal=self.goal.clone())
            hit, goals = new_goals
            still = ~done[hit] if self.autoreset_mode == "same_step" else np.ones(len(hit), bool)   # worlds reset in this call keep their reset goal
            if still.any():
                self.goal[torch.from_numpy(hit[still]).to(self.device)] = torch.from_numpy(goals[still].astype(np.float32)).to(self.device)
        info = self._info(step_success)
        if final_obs is not None:
            info["final_obs"] = final_obs
        if self.output == "torch":
            return obs, self.reward, torch.from_numpy(terminated), torch.from_numpy(truncated), info
        return obs, self.reward.double().cpu().numpy(), terminated, truncated, info

    def _obs_dict(self, rows=None):
        if self.output == "torch":
            sel = (lambda t: t) if rows is None else (lambda t: t[torch.from_numpy(rows).to(self.device)])
            return {"observation": sel(self.obs), "achieved_goal": sel(self.achieved), "desired_goal": sel(self.goal)}
        sel = (lambda a: a) if rows is None else (lambda a: a[rows])
        return {"observation": sel(self.obs.double().cpu().numpy()), "achieved_goal": sel(self.achieved.double().cpu().numpy()),
                "desired_goal": sel(self.goal.double().cpu().numpy())}

    # ------------------------------------------------------------------ GoalEnv API (maze_v4.py:381-398)
    def compute_reward(self, achieved_goal, desired_goal, info=None):
        as_numpy = not isinstance(achieved_goal, torch.Tensor)
        ag = torch.as_tensor(np.asarray(achieved_goal, dtype=np.float32) if as_numpy else achieved_goal, dtype=torch.float32, device=self.device).contiguous()
        dg = torch.as_tensor(np.asarray(desired_goal, dtype=np.float32) if as_numpy else desired_goal, dtype=torch.float32, device=self.device).contiguous()
        if ag.shape != dg.shape or ag.shape[-1] != 2:
            raise ValueError("achieved_goal and desired_goal must have the same (..., 2) shape")
        out = torch.empty(ag.shape[:-1], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._L.grx_maze_compute_reward(ag.data_ptr(), dg.data_ptr(), out.numel(), GOAL_RADIUS,
                                                          int(self.reward_type == "sparse"), out.data_ptr(), self._stream()))
        return out.double().cpu().numpy() if as_numpy else out

    def compute_terminated(self, achieved_goal, desired_goal, info=None):
        d = np.linalg.norm(np.asarray(achieved_goal, dtype=np.float64) - np.asarray(desired_goal, dtype=np.float64), axis=-1)      # maze_v4.py:390-397: the fp64 norm
        return (d <= GOAL_RADIUS) if not self.continuing_task else np.zeros(d.shape, bool)

    def compute_truncated(self, achieved_goal, desired_goal, info=None):
        return np.zeros(np.asarray(achieved_goal).shape[:-1], bool)

    def close(self):
        if getattr(self, "_h", None):
            _native.release_model(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AntMazeVecEnv(PointMazeVecEnv):
    """Batched AntMaze (ids AntMaze_*-v5, /root/reference/gymnasium_robotics/envs/maze/ant_maze_v5.py:221-320): the same maze logic
    around gymnasium's Ant [3P] -- free-floating torso + 8 hinges, RK4, 5 substeps of 0.01 s, observation = qpos[2:] | qvel (27)."""

    AGENT, N_SUBSTEPS, OBS_SKIP, DEFAULT_MAX_EPISODE_STEPS = "ant", ANT_FRAME_SKIP, 2, 700
    MAZE_GEOMETRY = (ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT)
    _parse_id = staticmethod(parse_ant_maze_id)

    def __init__(self, env_id: Optional[str] = "AntMaze_UMaze-v5", num_envs: int = 1, **kw):
        super().__init__(env_id, num_envs, **kw)
