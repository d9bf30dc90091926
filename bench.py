"""bench.py -- env-steps/s of the batched env.step() hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): "env-steps/s (whole node) at N parallel envs".  Default workload = BASELINE.json configs[1]:
FetchPickAndPlace-v4, 4096 envs per MI355X, sparse reward + HER reward recompute, uniform random actions.  Episodes are
staggered (world i starts at elapsed step i mod 50), so EVERY timed vector step contains its steady-state share of
autoresets (4096 / 50 = 82 worlds per step: host draws + the reset kernel), not only the 20-substep physics.
Weak scaling: the same number of worlds on every GPU; worlds are tile-sharded over the ranks and the only collective is
ONE RCCL all-gather per step of the packed output rows the step kernel itself wrote.

Before anything is timed the workload is PRE-ROLLED by one full episode horizon of whole steps (--preroll, HER included): a fresh batch is 3 - 8 % faster than the
stationary regime (fewer arm-on-head stragglers, an empty HER ring, fewer worlds in the overflow lanes), and the driver's `--steps 20 --warmup 5` must report the
stationary number.  What still scatters short Fetch windows by +-4 %: the 2 - 5 worlds per 100 steps that exceed the fast kernel's tables and are re-run behind the launch,
1 - 3 ms once each (`roofline.overflow_rerun_ms_per_step`).
`roofline` carries the HBM fraction SURVEY.md 8(d) asks for (tiny by construction: the fused step does ~2e3 FLOP per HBM byte) AND the bound that binds: `roofline.valu` =
wave64 VALU instructions per second against the chip's issue peak (SQ_INSTS_VALU from the SQ pass of tools/collect_profiles.py / the live kernel time), `traffic_ratio` =
PMC bytes / algorithmic bytes.

One "step" = one env.step() of all worlds = ONE launch of the family's step kernel (+ the HER reward kernel for cfg 2).
`--stages K` steps the rank's worlds as K out-of-phase sub-batches on K streams instead (gymnasium_robotics_amd.pipeline; one "step" = one env.step() of every sub-batch): +10 - 25 % from the same
kernels.  The plain 1-GPU line reports that mode BESIDE its value (`sub_batches`: the same worlds and timed region, K = 2, measured right after the plain region) -- `value` itself is always the
plain vector environment.
--workload selects the other BASELINE configs under the same contract and the same JSON schema:
    fetch       cfg 2  FetchPickAndPlace-v4, 4096 worlds / GPU
    hand_touch  cfg 3  HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1, 16384 worlds / GPU
    antmaze     cfg 4  AntMaze_Large_Diverse_GR-v5, 8192 worlds / GPU (65536 over 8)
    adroit      cfg 5b AdroitHandHammer-v2, 16384 worlds / GPU (adroit_door | adroit_pen | adroit_relocate: the other Adroit tasks)
    kitchen     cfg 5a FrankaKitchen-v1, 16384 worlds / GPU, default observation noise
    mixed       cfg 5  FrankaKitchen-v1 + AdroitHandHammer-v2 side by side, 2048 + 2048 worlds / GPU (32768 over 8): two streams, two host threads
    hand_reach         HandReach-v3, 16384 worlds / GPU

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus 8            (spawns 8 ranks itself when not started by torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first HIP call: streams that share a hardware queue serialise (the sub-batch legs keep 4 - 6 streams busy; no effect on the plain lines: profiles/ab_r05_hw_queues.txt)
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# per workload: env id, worlds per GPU, step-kernel name, algorithmic HBM bytes per env-step,
# time limit of the registered id
WORKLOADS = {
    "fetch": dict(env_id="FetchPickAndPlace-v4", worlds=4096, kernel="grx_fetch_step_kernel", algo=715, horizon=50),
    "hand_touch": dict(env_id="HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", worlds=16384, kernel="grx_hand_step_kernel", algo=1635, horizon=100),
    "hand_reach": dict(env_id="HandReach-v3", worlds=16384, kernel="grx_hand_step_kernel", algo=1035, horizon=50),   # r 24+24+24+20, w 72, out 63+15+15+1
    "antmaze": dict(env_id="AntMaze_Large_Diverse_GR-v5", worlds=8192, kernel="grx_point_step_kernel", algo=507, horizon=1000),
    "adroit": dict(env_id="AdroitHandHammer-v2", worlds=16384, kernel="grx_adroit_step_kernel", algo=1098, horizon=200),   # cfg 5b (5 substeps + noslip)
    # the other Adroit tasks; algorithmic bytes = 4 * (r: nq + 2 nv + nu + 7 (+3) | w: nq + 2 nv | out: obs + 1) + 2
    "adroit_door": dict(env_id="AdroitHandDoor-v2", worlds=16384, kernel="grx_adroit_step_kernel", algo=4 * (30 + 60 + 28 + 7 + 90 + 40) + 2, horizon=200),
    "adroit_pen": dict(env_id="AdroitHandPen-v2", worlds=16384, kernel="grx_adroit_step_kernel", algo=4 * (30 + 60 + 24 + 7 + 90 + 46) + 2, horizon=200),
    # cfg 5a: r qpos30 qvel29 warm29 act9 last9 noise59 = 165, w 30 + 29 + 29 + 9 = 97, out obs59 + completed1 = 60 words
    "kitchen": dict(env_id="FrankaKitchen-v1", worlds=16384, kernel="grx_kitchen_step_kernel", algo=4 * (165 + 97 + 60), horizon=280),
    "adroit_relocate": dict(env_id="AdroitHandRelocate-v2", worlds=16384, kernel="grx_adroit_step_kernel", algo=4 * (36 + 72 + 30 + 10 + 108 + 40) + 2, horizon=200),
}
MIXED = ("kitchen", "adroit")   # cfg 5: the two families of the mixed batch, half of the rank's worlds each
MIXED_WORLDS = 4096             # per GPU (32768 over 8)
HER_K = 4  # relabelled goals per transition ("future" strategy with k=4); 28 B per relabelled transition
HBM_PEAK_GBS = 8000.0
PROFILE_TAGS = ("r06",)      # rounds whose profiles/pmc_<tag>_*.json may be quoted -- and only when their build_id is the loaded library's (run_rank)
LONG_WINDOW_STEPS = 100
NORTH_STAR_WORLDS_PER_GPU = 8192      # BASELINE.json north_star: 65 536 FetchPickAndPlace worlds on 8 GPUs
VALU_PEAK_IPS = 256 * 4 * 2.4e9 / 2      # wave64 VALU instructions per second, whole chip (SIMD-32: 2 cycles per wave64 instruction)


def _set_elapsed(env, elapsed):
    """episode positions of the worlds: the envs that keep their TimeLimit counters on the device take them through set_elapsed()"""
    if hasattr(env, "set_elapsed"):
        env.set_elapsed(elapsed)
    else:
        env._elapsed[:] = elapsed


def make_env(workload, n, device, rank):
    w = WORKLOADS[workload]
    kw = dict(num_envs=n, device=device, output="torch", autoreset_mode="same_step", seed_offset=rank * n)
    if workload == "fetch":
        from gymnasium_robotics_amd.envs.fetch import FetchVecEnv as Env
    elif workload == "antmaze":
        from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv as Env
    elif workload == "hand_reach":
        from gymnasium_robotics_amd.envs.hand import HandReachVecEnv as Env
    elif workload == "kitchen":
        from gymnasium_robotics_amd.envs.kitchen import KitchenVecEnv as Env
    elif workload.startswith("adroit"):
        from gymnasium_robotics_amd.envs.adroit import AdroitVecEnv as Env
    else:
        from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv as Env
    if os.environ.get("GRX_BENCH_BALANCE") and workload.startswith("hand"):    # A/B switch of the cost-ordered dispatch for the hand families (off by default there)
        kw["balance"] = os.environ["GRX_BENCH_BALANCE"] == "1"
    return Env(w["env_id"], **kw)


class _DryEnv:
    """--dry-run: a stand-in with the vector envs' attribute surface and row shapes but NO physics (random rows on the CPU), so that everything around the step
    kernels -- launcher, rendezvous, world sharding by seed_offset, the per-step collective on the kernel-written rows, timing, the per-rank report, the JSON
    line -- runs in a GPU-less container over gloo (tests/test_cpu_dist.py) with the exact command line the driver uses on the 8-GPU node."""

    def __init__(self, workload, n, rank, seed_offset=None):
        import types

        w = WORKLOADS[workload]
        self.num_envs, self.max_episode_steps, self.seed_offset = n, w["horizon"], rank * n if seed_offset is None else seed_offset
        self.device = "cpu"
        act = {"fetch": 4, "antmaze": 8, "kitchen": 9, "hand_reach": 20, "hand_touch": 20}.get(workload, 28)
        self.single_action_space = types.SimpleNamespace(shape=(act,))
        width = {"fetch": 33, "antmaze": 33, "hand_reach": 95, "hand_touch": 169, "kitchen": 59}.get(workload, 46)
        if workload in ("fetch", "antmaze", "hand_reach", "hand_touch"):
            self.packed = torch.zeros(n, width)
        else:
            self.obs = torch.zeros(n, width)
        self._elapsed = np.zeros(n, np.int64)
        self.kernel_events = None
        self._gen = torch.Generator().manual_seed(rank)

    def reset(self, seed=None, options=None):
        return None, {}

    def step(self, a):
        rows = getattr(self, "packed", None)
        rows = self.obs if rows is None else rows
        rows.copy_(torch.rand(rows.shape, generator=self._gen))
        rows[:, 0] = torch.arange(self.num_envs, dtype=torch.float32) + self.seed_offset      # world id in column 0: the gathered matrix must be in world order
        self._elapsed += 1
        trunc = torch.from_numpy(self._elapsed >= self.max_episode_steps)
        self._elapsed[trunc.numpy()] = 0
        if self.kernel_events is not None:
            self.kernel_events.append((0.0, 0.0))
        return None, None, torch.zeros(self.num_envs, dtype=torch.bool), trunc, {}

    def clear_status(self):
        pass

    def status_counts(self):
        return {"badnum": 0, "con_overflow": 0, "efc_overflow": 0, "factor": 0, "worlds": self.num_envs}


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle, test infrastructure)
def _oracle_env(workload):
    if workload == "fetch":
        from gymnasium_robotics_amd.envs.fetch import load_fetch_model
        from oracle.fetch_oracle import OracleFetchEnv
        return OracleFetchEnv(load_fetch_model("FetchPickAndPlace"), "FetchPickAndPlace"), 4
    if workload == "antmaze":
        from gymnasium_robotics_amd.envs.maze_spec import ANT_MAZE_HEIGHT, ANT_MAZE_SIZE_SCALING, MAPS, Maze, parse_ant_maze_id
        from gymnasium_robotics_amd.envs.point_maze import load_point_maze_model
        from oracle.maze_oracle import OracleAntMazeEnv
        layout = parse_ant_maze_id(WORKLOADS[workload]["env_id"])[0]
        maze = Maze(MAPS[layout], ANT_MAZE_SIZE_SCALING, ANT_MAZE_HEIGHT)
        return OracleAntMazeEnv(load_point_maze_model(maze, layout, None, "ant"), maze), 8
    if workload == "kitchen":
        from gymnasium_robotics_amd.envs.kitchen_spec import load_kitchen_model
        from oracle.kitchen_oracle import OracleKitchenEnv
        return OracleKitchenEnv(load_kitchen_model()), 9
    if workload.startswith("adroit"):
        from gymnasium_robotics_amd.envs.adroit_spec import load_adroit_model, parse_adroit_id
        from oracle.adroit_oracle import OracleAdroitEnv
        task = parse_adroit_id(WORKLOADS[workload]["env_id"])[0]
        model = load_adroit_model(task)
        return OracleAdroitEnv(model, "dense", task), model.dim("nu")
    if workload == "hand_reach":
        from gymnasium_robotics_amd.envs.hand import load_hand_reach_model
        from oracle.hand_oracle import OracleHandReachEnv
        return OracleHandReachEnv(load_hand_reach_model(None)), 20
    from gymnasium_robotics_amd.envs.hand import load_hand_block_model
    from oracle.manipulate_oracle import OracleHandBlockEnv
    return OracleHandBlockEnv(load_hand_block_model(None, touch=True), "ignore", "xyz", "sparse", "sensordata"), 20


def _cpu_worker(args):
    workload, seconds, seed = args
    sys.path.insert(0, ROOT)
    env, act = _oracle_env(workload)
    rng = np.random.default_rng(seed)
    env.reset(seed=seed)
    horizon = WORKLOADS[workload]["horizon"]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(min(horizon, 25)):
            env.step(rng.uniform(-1, 1, act))
            n += 1
        if n % horizon < 25:
            env.reset()
    return n, time.perf_counter() - t0


def cpu_baseline(workload, seconds=8.0):
    """The oracle (fp64 restatement under oracle/: C physics + Python task layer; MuJoCo cannot be installed here) timed on the host
    cores: one process = one world on one core, then min(8, cores) processes side by side."""
    import multiprocessing as mp

    n1, t1 = _cpu_worker((workload, seconds, 0))
    cores = max(1, min(8, os.cpu_count() or 1))
    multi = None
    if cores > 1:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(workload, seconds, k) for k in range(cores)])
        multi = sum(n / t for n, t in res)
    return {"value": n1 / t1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "value_all_cores": multi, "cores_all": cores,
            "sample": f"{n1} env.step() calls of 1 world on 1 core ({t1:.1f} s), then {cores} independent worlds on {cores} cores for {seconds:.0f} s each; "
                      f"{WORKLOADS[workload]['env_id']}, random actions; oracle = fp64 restatement (not MuJoCo), Python task layer + C physics"}


# ---------------------------------------------------------------------------------------------- cfg 5: two families in one job
def run_rank_mixed(args, rank, world_size, local_rank):
    """BASELINE cfg 5: every rank steps n/2 FrankaKitchen worlds and n/2 AdroitHandHammer worlds.  The two environments are independent (no data
    dependence): ONE host thread enqueues both on their own HIP streams -- the kitchen step is launched (KitchenVecEnv.step_launch: no host sync), then the
    whole Adroit step, then the kitchen step is finished (its completion bits are the one read-back) -- so the two step kernels share the GPU; one vector
    "step" = one env.step() of both halves, followed by one all-gather per family of the rows the kernels wrote."""
    import contextlib

    dry = args.dry_run
    device = "cpu" if dry else f"cuda:{local_rank}"
    sync = (lambda: None) if dry else torch.cuda.synchronize
    if not dry:
        torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(device))
    n = args.worlds_per_gpu or MIXED_WORLDS
    half = n // 2
    envs, streams, gens, gathered = [], [], [], []
    for k, name in enumerate(MIXED):
        env = _DryEnv(name, half, rank) if dry else make_env(name, half, device, rank)
        env.reset(seed=0)
        if args.stagger:
            _set_elapsed(env, np.arange(half) % (env.max_episode_steps or WORKLOADS[name]["horizon"]))
        g = torch.Generator(device=device)
        g.manual_seed(1234 + 2 * rank + k)
        envs.append(env); streams.append(None if dry else torch.cuda.Stream(device=device)); gens.append(g)
        gathered.append(torch.empty(half * world_size, env.obs.shape[1], device=device) if dist else None)
    sync()
    on = (lambda k: contextlib.nullcontext()) if dry else (lambda k: torch.cuda.stream(streams[k]))
    act = lambda k: torch.rand(half, envs[k].single_action_space.shape[0], device=device, generator=gens[k]) * 2 - 1
    preroll = max(WORKLOADS[name]["horizon"] for name in MIXED) if args.preroll < 0 else args.preroll      # steady state before anything is timed (see run_rank)
    if dry:
        preroll = min(preroll, 3)

    def one_step():
        with on(0):     # FrankaKitchen: enqueue only
            (envs[0].step if dry else envs[0].step_launch)(act(0))
        with on(1):     # AdroitHandHammer: its step never waits for the device
            envs[1].step(act(1))
        if not dry:
            with on(0):
                envs[0].step_finish()
        if dist:
            for k in range(2):
                if not dry:
                    torch.cuda.current_stream().wait_stream(streams[k])
                dist.all_gather_into_tensor(gathered[k], envs[k].obs)
            for k in range(2):
                if not dry:
                    streams[k].wait_stream(torch.cuda.current_stream())

    for _ in range(preroll + args.warmup):
        one_step()
    for env in envs:
        env.clear_status()
        env.kernel_events = []
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    if dist:
        dist.barrier()
    elapsed = own_elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = [float(np.mean([0.0 if dry else a.elapsed_time(b) for a, b in env.kernel_events])) for env in envs]
    counts = [env.status_counts() for env in envs]
    dist_report = None
    if dist:
        from gymnasium_robotics_amd.parallel import rank_stats

        dist_report = rank_stats({"kernel_ms_" + MIXED[0]: kern_ms[0], "kernel_ms_" + MIXED[1]: kern_ms[1], "elapsed_s": own_elapsed}, device)
        if dry:
            for k in range(2):
                assert torch.equal(gathered[k][:, 0], torch.arange(half * world_size, dtype=torch.float32)), "gathered rows are not in world order"
    line = None
    if rank == 0:
        w = WORKLOADS[MIXED[0]]     # the dominant kernel: the kitchen step (40 substeps against 5)
        achieved = 0.0 if dry else w["algo"] * half / (kern_ms[0] * 1e-3) / 1e9
        line = {
            "metric": "env-steps/s (whole node)", "value": n * world_size * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mixed batch: {half} FrankaKitchen-v1 (multitask, default noise) + {half} AdroitHandHammer-v2 worlds/GPU x {world_size} GPU, uniform random "
                                   "actions, same-step autoreset at the time limits, the two families on two streams driven by one host thread",
                       "worlds_per_gpu": n, "preroll_steps": preroll, "parallelism": f"world-shard x{world_size}" + (", one RCCL all_gather per family of the kernel-written rows per step" if world_size > 1 else ""),
                       "capacity_overflow_worlds": sum(c["con_overflow"] + c["efc_overflow"] for c in counts), "badnum_worlds": sum(c["badnum"] for c in counts),
                       "kernel_ms": {MIXED[k]: kern_ms[k] for k in range(2)}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "kernel": w["kernel"], "kernel_ms": kern_ms[0], "algorithmic_bytes_per_launch": w["algo"] * half,
                         "note": "duration measured while the Adroit step kernel shares the GPU; the single-family lines (--workload kitchen / adroit) carry the PMC traffic"},
        }
        if dist_report is not None:
            line["dist"] = dist_report
        if dry:
            line["data"] = "DRY RUN: no physics, random rows on the CPU over gloo (plumbing check of the multi-rank command line)"
        if world_size == 1 and not args.no_cpu_baseline and not dry:
            parts = [cpu_baseline(name, seconds=5.0) for name in MIXED]
            mix = lambda key: 2.0 / sum(1.0 / p[key] for p in parts)      # a mixed batch = equal numbers of steps of both families
            line["cpu_baseline"] = {"value": mix("value"), "unit": "env-steps/s", "cores": 1, "kind": "port", "value_all_cores": mix("value_all_cores"),
                                    "cores_all": parts[0]["cores_all"], "sample": "equal-step mix (harmonic mean) of: " + " | ".join(p["sample"] for p in parts)}
    if dist:
        dist.destroy_process_group()
    return line


# ---------------------------------------------------------------------------------------------- out-of-phase sub-batches (--stages K)
def run_rank_stages(args, rank, world_size, local_rank):
    """`--stages K`: the rank's worlds as K sub-batches on K streams (gymnasium_robotics_amd.pipeline.PipelinedVecEnv), walked round-robin by one host thread that
    only enqueues.  One "step" = one env.step() of EVERY stage (all n worlds advance once), with everything the plain workload has in its timed region (resets, the
    overflow re-runs, HER append + relabel per stage, the per-stage gather on more than one rank).  The stages drift out of phase, so the tail of one stage's step
    launch -- a launch ends when its slowest world does -- is filled by the next stage's worlds."""
    from gymnasium_robotics_amd.pipeline import PipelinedVecEnv

    dry = args.dry_run      # no GPU, no physics: _DryEnv stages on the CPU over gloo (the multi-rank plumbing of this command line, tests/test_cpu_dist.py)
    w, K = WORKLOADS[args.workload], args.stages
    device = "cpu" if dry else f"cuda:{local_rank}"
    sync = (lambda: None) if dry else torch.cuda.synchronize
    if not dry:
        torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1 or os.environ.get("GRX_BENCH_FORCE_DIST"):      # (the env var runs the collective leg on a single rank, as in run_rank)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(device))
    n = args.worlds_per_gpu or w["worlds"]
    if dry:
        pe = PipelinedVecEnv(w["env_id"], n, stages=K, device="cpu", seed_offset=rank * n,
                             make_stage=lambda env_id, num_envs, device=None, seed_offset=0, **kw: _DryEnv(args.workload, num_envs, rank, seed_offset=seed_offset))
    else:
        pe = PipelinedVecEnv(w["env_id"], n, stages=K, device=device, seed_offset=rank * n, output="torch", autoreset_mode="same_step")
    m = pe.stage_size
    pe.reset(seed=0)
    her = args.workload == "fetch" and not dry
    gens, replays, gathered = [], [], []
    for k, env in enumerate(pe.stage_envs):
        with pe.on(k):
            if args.stagger:
                _set_elapsed(env, (np.arange(m) + k * m) % (env.max_episode_steps or w["horizon"]))
            g = torch.Generator(device=device)
            g.manual_seed(1234 + rank * K + k)
            gens.append(g)
            rows = getattr(env, "packed", None)
            rows = env.obs if rows is None else rows
            gathered.append(torch.empty(m * world_size, rows.shape[1], device=device) if dist else None)
            if her:
                from gymnasium_robotics_amd.her import HerReplay

                r = HerReplay(env, horizon=w["horizon"], capacity=HER_K * m * 8, seed=rank * K + k, continuous=True)
                r.begin_episode(env.packed)
                r.set_episode_start(-env._elapsed)
                replays.append(r)
    act_dim = pe.single_action_space.shape[0]

    def one_step():
        for k, env in enumerate(pe.stage_envs):
            with pe.on(k):
                a = torch.rand(m, act_dim, device=device, generator=gens[k]) * 2 - 1
                obs, r, term, trunc, info = env.step(a)
                if her:
                    replays[k].append(a, env.packed, term | trunc, final_rows=env.final_packed)
                    replays[k].relabel(HER_K * m, k_future=HER_K)
                if dist:
                    rows = getattr(env, "packed", None)
                    dist.all_gather_into_tensor(gathered[k], env.obs if rows is None else rows)

    preroll = (pe.max_episode_steps or w["horizon"]) if args.preroll < 0 else args.preroll
    if dry:
        preroll = min(preroll, 3)
    for _ in range(preroll + args.warmup):
        one_step()
    for env in pe.stage_envs:
        env.clear_status()
        env.kernel_events = []
        if hasattr(env, "step_events"):
            env.step_events = []
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    if dist:
        dist.barrier()
    elapsed = own_elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([0.0 if dry else a.elapsed_time(b) for env in pe.stage_envs for a, b in env.kernel_events]))
    counts = [env.status_counts() for env in pe.stage_envs]
    dist_report = None
    if dist:
        from gymnasium_robotics_amd.parallel import rank_stats

        dist_report = rank_stats({"kernel_ms": kern_ms, "elapsed_s": own_elapsed}, device)
        if dry:      # stage k of rank r holds the worlds r * n + k * m ... : every gathered matrix must list them rank by rank
            for k in range(K):
                want = torch.cat([torch.arange(m, dtype=torch.float32) + r_ * n + k * m for r_ in range(world_size)])
                assert torch.equal(gathered[k][:, 0], want), f"gathered rows of sub-batch {k} are not in world order"
    line = None
    if rank == 0:
        achieved = w["algo"] * m / (max(kern_ms, 1e-9) * 1e-3) / 1e9
        line = {
            "metric": "env-steps/s (whole node)", "value": n * world_size * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['env_id']}, {n} worlds/GPU x {world_size} GPU as {K} out-of-phase sub-batches of {m} worlds on {K} streams (one host thread, enqueue only; "
                                   f"one step = one env.step() of every sub-batch), uniform random actions, same-step autoreset at the time limit "
                                   f"({'episodes staggered' if args.stagger else 'episodes in lock-step'})"
                                   + (f", sparse reward + on-device HER relabel + replay write per sub-batch ({HER_K} transitions per world and step)" if her else ""),
                       "worlds_per_gpu": n, "stages": K, "preroll_steps": preroll,
                       "parallelism": f"world-shard x{world_size}" + (", one RCCL all_gather of kernel-packed output rows per sub-batch and step" if world_size > 1 else ""),
                       "capacity_overflow_worlds": sum(c["con_overflow"] + c["efc_overflow"] for c in counts), "badnum_worlds": sum(c["badnum"] for c in counts)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "kernel": w["kernel"], "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": w["algo"] * m,
                         "note": f"kernel_ms = mean duration of a sub-batch's step launch ({m} worlds) WHILE the other sub-batches share the GPU: K launches overlap, so K x kernel_ms exceeds "
                                 "ms_per_step; the single-batch line (no --stages) carries the PMC traffic and the VALU issue fraction of the kernel"},
        }
        if dist_report is not None:
            line["dist"] = dist_report
        if dry:
            line["data"] = "DRY RUN: no physics, random rows on the CPU over gloo (plumbing check of the multi-rank command line)"
        if not args.no_cpu_baseline and world_size == 1 and not dry:
            line["cpu_baseline"] = cpu_baseline(args.workload)
    if dist:
        dist.destroy_process_group()
    return line


# ---------------------------------------------------------------------------------------------- one rank
def pmc_summary(kind, workload, n, worlds, live_build, root=None):
    """The committed rocprofv3 PMC summary of `kind` ("hbm_traffic" / "sq_mix") for a workload, or (None, reason).  A summary is attached ONLY when it was measured at this batch size
    AND on the device code that is loaded now (`build_id`, stamped by tools/collect_profiles.py = _native.build_id() of the profiled library): a kernel change without a new
    collect_profiles.py run yields `traffic: null` / `valu: null` and a `traffic_source` that says why -- never an old build's counters (tests/test_cpu_bench_profiles.py)."""
    root = root or ROOT
    for tag in PROFILE_TAGS:
        path = os.path.join(root, "profiles", f"pmc_{tag}_{kind}{'' if workload == 'fetch' else '_' + workload}.json")
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            rel = os.path.relpath(path, root)
            if n != worlds:
                return None, f"{rel}: measured at {worlds} worlds, this run has {n}"
            if d.get("build_id") != live_build:
                return None, f"STALE, not attached: {rel} was measured on build {d.get('build_id') or '(unstamped, before round 6)'}, the loaded libgrx_hip.so is {live_build}"
            need = {"hbm_traffic": "traffic_bytes_per_launch", "sq_mix": "SQ_INSTS_VALU"}.get(kind)
            if need and need not in d:      # a collector pass that found no launch of the step kernel writes a summary without the figure: say so, never fail the bench line over it
                return None, f"INCOMPLETE, not attached: {rel} holds no {need} (the counter pass collected no sample)"
            return d, rel
    return None, None


def run_rank(args, rank, world_size, local_rank):
    if args.workload == "mixed":
        return run_rank_mixed(args, rank, world_size, local_rank)
    if args.stages > 1:
        return run_rank_stages(args, rank, world_size, local_rank)
    w = WORKLOADS[args.workload]
    dry = args.dry_run
    device = "cpu" if dry else f"cuda:{local_rank}"
    sync = (lambda: None) if dry else torch.cuda.synchronize
    if not dry:
        torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1 or os.environ.get("GRX_BENCH_FORCE_DIST"):      # the env var runs the collective leg on a single rank (1-GPU boxes: tools/ab_dist.sh)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(device))
    n = args.worlds_per_gpu or w["worlds"]
    env = _DryEnv(args.workload, n, rank) if dry else make_env(args.workload, n, device, rank)
    env.reset(seed=0)
    if args.stagger:   # steady state: every step resets its share of the worlds (world i is i mod horizon steps into its episode)
        _set_elapsed(env, np.arange(n) % (env.max_episode_steps or w["horizon"]))
    act_dim = env.single_action_space.shape[0]
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    out_rows = getattr(env, "packed", None)
    if out_rows is None:
        out_rows = env.obs   # plain (non-goal) environments: the observation rows are the per-step output
    gathered = torch.empty(n * world_size, out_rows.shape[1], device=device) if dist else None
    her = args.workload == "fetch" and not dry
    # HER "future" relabelling on the device (gymnasium_robotics_amd/her.py): the packed rows of the last `horizon` steps stay in an HBM ring; every
    # step ONE kernel gathers HER_K relabelled transitions per world (goal substitution + reward recompute + replay write)
    replay = None
    if her:
        from gymnasium_robotics_amd.her import HerReplay

        replay = HerReplay(env, horizon=w["horizon"], capacity=HER_K * n * 8, seed=rank, continuous=True)
        replay.begin_episode(env.packed)
        replay.set_episode_start(-env._elapsed)   # staggered: world i is elapsed[i] steps into its episode

    act_buf = torch.empty(n, act_dim, device=device)

    def one_step():
        a = act_buf.uniform_(-1.0, 1.0, generator=gen)      # uniform random actions: one kernel
        obs, r, term, trunc, info = env.step(a)
        if her:
            replay.append(a, env.packed, term | trunc, final_rows=env.final_packed)   # the reset kernel parked the terminal rows there: the last transition of every episode is relabelled too
            replay.relabel(HER_K * n, k_future=HER_K)
        if dist:   # the step kernel wrote the packed [obs | achieved | desired | reward | success] rows: one collective, no pack kernels.
            # Stream-ordered, not pipelined: on one rank the collective costs 0.01 ms of the 3.3 ms step; an asynchronous gather from a staging copy, waited for two
            # steps later, measured 0.2 - 0.9 ms SLOWER per step (cross-stream dependencies in both directions every step; tools/ab_dist.sh)
            dist.all_gather_into_tensor(gathered, out_rows)

    # Steady state before anything is timed: a fresh batch is not what a training run sees -- the worlds of a Fetch batch that end up with the upper arm
    # resting on the head link (the stragglers that end a launch) accumulate over an episode, and the HER ring holds fewer steps than its horizon -- so the first ~25 steps
    # after reset() are ~10 % faster than the stationary regime.  One full horizon of untimed steps of the WHOLE step (HER append + relabel and the collective included) puts
    # every world through every phase of its episode; `--steps 20 --warmup 5` then reports what `--steps 100` reports.
    preroll = (env.max_episode_steps or w["horizon"]) if args.preroll < 0 else args.preroll
    if dry:
        preroll = min(preroll, 3)
    for _ in range(preroll + args.warmup):
        one_step()
    env.clear_status()
    env.kernel_events = []   # HIP events (torch's current stream = the launch stream) around every step-kernel launch of the timed region
    if hasattr(env, "step_events"):
        env.step_events = []   # ... and around the whole launch group of a step (fast kernel + the overflow lane's launches on the side stream, joined before the entry launch)
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    if dist:
        dist.barrier()
    elapsed = own_elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([0.0 if dry else a.elapsed_time(b) for a, b in env.kernel_events]))
    lane_ms = float(np.mean([a.elapsed_time(b) for a, b in env.step_events])) if (not dry and getattr(env, "step_events", None)) else None
    dist_report = None
    if dist:   # the rank count as the collective library reports it + every rank's kernel / wall time (a slow rank, a rank that fell back to another device: visible in the line)
        from gymnasium_robotics_amd.parallel import rank_stats

        dist_report = rank_stats({"kernel_ms": kern_ms, "elapsed_s": own_elapsed}, device)
        if dry:
            assert gathered is not None and torch.equal(gathered[:, 0], torch.arange(n * world_size, dtype=torch.float32)), "gathered rows are not in world order"
    counts = env.status_counts()

    # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE
    # cannot be read from inside the process); tools/collect_profiles.py writes the summary bench.py quotes
    # A summary is attached ONLY while it was measured on the device code that is loaded now (`build_id` = digest of the library's .hip_fatbin, _native.build_id): a kernel
    # change without a new collect_profiles.py run yields `traffic: null` / `valu: null` and says why, never an old build's counters.
    traffic, traffic_src, valu, live_build = None, None, None, None
    if not dry:
        from gymnasium_robotics_amd import _native

        live_build = _native.build_id()

    def _summary(kind):
        return pmc_summary(kind, args.workload, n, w["worlds"], live_build)

    if not dry:
        d, traffic_src = _summary("hbm_traffic")
        if d is not None:
            traffic = d["traffic_bytes_per_launch"]
        # The bound that actually binds: VALU issue.  SQ_INSTS_VALU per launch comes from the SQ pass of tools/collect_profiles.py (same command, separate --pmc run);
        # peak = 256 CU x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction = 1228.8 G wave-instructions/s (MI355X_MICROARCH.md constants table: v_fma_f32 wave64 = 2 cyc).
        sq, sq_src = _summary("sq_mix")
        if sq is not None:
            ips = sq["SQ_INSTS_VALU"] / (max(kern_ms, 1e-9) * 1e-3)
            valu = {"insts_per_s": ips, "peak": VALU_PEAK_IPS, "frac": ips / VALU_PEAK_IPS, "unit": "wave64 VALU instructions/s", "insts_per_launch": sq["SQ_INSTS_VALU"],
                    "valu_active_frac_of_wave_cycles": sq.get("SQ_ACTIVE_INST_VALU_frac"), "wait_frac_of_wave_cycles": sq.get("SQ_WAIT_ANY_frac"), "source": sq_src, "build_id": sq.get("build_id")}
        elif sq_src:
            valu = {"frac": None, "source": sq_src}
    # `long_window`: the same environment stepped on for 100 more timed steps right behind the driver's window -- a 20-step window holds 0 - 3 overflow re-runs of 1 - 3 ms each
    # and scatters by +-4 %; the long figure is the one to compare builds by.  One GPU only (no collective bookkeeping), skipped when the window already is that long.
    long_window = None
    if not dry and world_size == 1 and not dist and args.steps < LONG_WINDOW_STEPS and args.long_window:
        sync()
        t1 = time.perf_counter()
        for _ in range(LONG_WINDOW_STEPS):
            one_step()
        sync()
        dt = time.perf_counter() - t1
        long_window = {"steps": LONG_WINDOW_STEPS, "value": n * LONG_WINDOW_STEPS / dt, "ms_per_step": dt / LONG_WINDOW_STEPS * 1e3,
                       "note": "the same environment, stepped on right behind the timed region (everything the timed region contains); not `value`"}
    line = None
    if rank == 0:
        value = n * world_size * args.steps / elapsed
        # FetchPickAndPlace since round 6: the step is a GROUP of concurrent launches (fast kernel + the standing lane's kernel for the ~8 % hull worlds + the entry launch); the
        # N worlds' algorithmic bytes are priced against the whole group's duration, not against the fast kernel alone (which steps ~90 % of them); `kernel_ms` stays the fast
        # kernel's own duration -- the figure rocprofv3's per-kernel average must agree with
        handoff = (not dry) and getattr(env, "_h_fast", None) is not None and lane_ms is not None
        achieved = w["algo"] * n / (max(lane_ms if handoff else kern_ms, 1e-9) * 1e-3) / 1e9 if not dry else 0.0
        line = {
            "metric": "env-steps/s (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['env_id']}, {n} worlds/GPU x {world_size} GPU, uniform random actions, same-step autoreset at the time limit "
                                   f"({'episodes staggered: every step resets its share of the worlds' if args.stagger else 'episodes in lock-step'})"
                                   + (f", sparse reward + on-device HER relabel + replay write ({HER_K} transitions per world and step, future k={HER_K})" if her else ""),
                       "worlds_per_gpu": n, "preroll_steps": preroll, "parallelism": f"world-shard x{world_size}" + (", one RCCL all_gather of kernel-packed output rows per step" if world_size > 1 else ""),
                       "capacity_overflow_worlds": counts["con_overflow"] + counts["efc_overflow"], "badnum_worlds": counts["badnum"],
                       "status_note": "worlds (of rank 0) whose sticky status flagged a dropped contact / bad number at least once in the timed region"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "build_id": live_build, "traffic": traffic, "traffic_source": traffic_src, "traffic_ratio": (traffic / (w["algo"] * n)) if traffic else None,      # PMC bytes / algorithmic bytes per launch: scratch + model-table re-reads
                         "valu": valu, "kernel": w["kernel"], "kernel_ms": kern_ms, "achieved_over": "kernel_plus_overflow_lane_ms (fast kernel + hull lane + entry launch, concurrent)" if handoff else "kernel_ms",
                         "kernel_plus_overflow_lane_ms": lane_ms,      # families with an overflow lane: fast launch + the lane's concurrent and serialised launches (`achieved` is the fast launch's)
                         # Fetch: the few worlds per 100 steps that exceed the fast tables are re-run behind the launch, 1 - 3 ms ONCE each: a 20-step window holds 0 - 3 of them, so
                         # short runs scatter by +-4 % around the long-run mean
                         "overflow_rerun_ms_per_step": (lane_ms - kern_ms) if lane_ms is not None else None,
                         "algorithmic_bytes_per_launch": w["algo"] * n,
                         "note": "fused path is instruction-issue / latency bound (~2e3 FLOP/B), HBM fraction is tiny by construction; see DESIGN.md 6"},
        }
        # the ant family has no overflow lane and tables smaller than a full contact list (envs/point_maze.py ANT_CAPACITY: 16 contacts / 64 rows / 512 pool words against a measured
        # peak of 4 / 18 / ~200): a world that exceeds them DROPS contacts, which the reference never does -- such a run is not a measurement
        if not dry and args.workload == "antmaze" and line["config"]["capacity_overflow_worlds"] > 0:
            raise SystemExit(f"bench.py: {line['config']['capacity_overflow_worlds']} ant worlds exceeded the engine tables (dropped contacts): raise ANT_CAPACITY")
        if long_window is not None:
            line["long_window"] = long_window
        if dist_report is not None:
            line["dist"] = dist_report
        if dry:
            line["data"] = "DRY RUN: no physics, random rows on the CPU over gloo (plumbing check of the multi-rank command line)"
    if dist:
        dist.destroy_process_group()
    # The north star's share of one GPU (BASELINE.json: FetchPickAndPlace-v4, 65 536 worlds on 8 GPUs = 8 192 per GPU), in the driver's own line: the same workload, timed region,
    # warm-up and step count at 8 192 worlds, measured in the same process right after the plain region.  Eight ranks without a step collective would run eight of these; the
    # product of 8 is a projection and is NOT reported.
    if line is not None and world_size == 1 and not dist and not dry and args.north_star_share and args.workload == "fetch" and n == w["worlds"] and not args.no_cpu_baseline:
        try:
            import copy

            env = replay = None      # (the closure one_step sees the same cells: the 4 096-world buffers are released before the next leg allocates)
            torch.cuda.empty_cache()
            a3 = copy.copy(args)
            a3.worlds_per_gpu, a3.no_cpu_baseline, a3.sub_batches, a3.north_star_share = NORTH_STAR_WORLDS_PER_GPU, True, False, False
            l3 = run_rank(a3, rank, world_size, local_rank)
            line["north_star_share"] = {"worlds": NORTH_STAR_WORLDS_PER_GPU, "value": l3["value"], "ms_per_step": l3["ms_per_step"], "kernel_ms": l3["roofline"]["kernel_ms"],
                                        "long_window": l3.get("long_window"), "capacity_overflow_worlds": l3["config"]["capacity_overflow_worlds"],
                                        "note": "one GPU's share of BASELINE's 65 536-world target (8 192 worlds), plain env.step(), same timed region / warm-up / steps as `value`; "
                                                "what eight GPUs reach together is not measured here"}
        except Exception as e:      # the extra leg must never cost the line
            line["north_star_share"] = {"error": repr(e)}
    # The same worlds as two out-of-phase sub-batches (run_rank_stages), measured in the same process right after the plain line: reported BESIDE
    # `value`, never as it -- `value` stays env.step() over all worlds of one vector environment, the configuration BASELINE.json names.
    if line is not None and world_size == 1 and not dist and not dry and args.sub_batches and not args.no_cpu_baseline and args.workload != "hand_touch":      # (like cpu_baseline: a reporting extra of the full line; the A/B and profiling tools pass --no-cpu-baseline)
        try:
            import copy

            env = replay = None
            torch.cuda.empty_cache()
            a2 = copy.copy(args)
            a2.stages, a2.no_cpu_baseline = 2, True
            l2 = run_rank_stages(a2, rank, world_size, local_rank)
            line["sub_batches"] = {"stages": 2, "value": l2["value"], "ms_per_step": l2["ms_per_step"], "capacity_overflow_worlds": l2["config"]["capacity_overflow_worlds"],
                                   "note": "the same worlds, timed region and step count as `value`, stepped as 2 out-of-phase sub-batches on 2 streams (python bench.py --stages 2; "
                                           "gymnasium_robotics_amd.pipeline.PipelinedVecEnv): for callers that act on half a batch at a time"}
        except Exception as e:      # the extra leg must never cost the line
            line["sub_batches"] = {"error": repr(e)}
    if line is not None and world_size == 1 and not args.no_cpu_baseline and not dry:      # (last: its worker processes load the host)
        line["cpu_baseline"] = cpu_baseline(args.workload)
    return line


def _spawned(local_rank, args, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus), RANK=str(local_rank), LOCAL_RANK=str(local_rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    line = run_rank(args, local_rank, args.gpus, local_rank)
    if line is not None:
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--worlds-per-gpu", type=int, default=0)
    ap.add_argument("--preroll", type=int, default=-1, help="untimed steady-state steps before --warmup (-1: one full episode horizon of the workload, 0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stagger", dest="stagger", action="store_false")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["mixed"], default="fetch")
    ap.add_argument("--no-sub-batches", dest="sub_batches", action="store_false", help="skip the extra leg that reports the same worlds stepped as 2 out-of-phase sub-batches beside `value` (1 GPU, and only when the CPU baseline leg runs too)")
    ap.add_argument("--no-north-star-share", dest="north_star_share", action="store_false", help="skip the extra leg that reports the same Fetch workload at 8 192 worlds (one GPU's share of BASELINE's 65 536) beside `value`")
    ap.add_argument("--no-long-window", dest="long_window", action="store_false", help="skip the 100 extra timed steps reported as `long_window` beside a shorter timed region")
    ap.add_argument("--stages", type=int, default=1, help="K > 1: the rank's worlds as K out-of-phase sub-batches on K streams (gymnasium_robotics_amd.pipeline; not for --workload mixed)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU, no physics: random rows on the CPU over gloo -- checks the multi-rank plumbing of this exact command line")
    args = ap.parse_args()

    if "WORLD_SIZE" in os.environ:   # started by torchrun: one rank per process already
        ws = int(os.environ["WORLD_SIZE"])
        if ws != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but torchrun started {ws} ranks")
        line = run_rank(args, int(os.environ.get("RANK", "0")), ws, int(os.environ.get("LOCAL_RANK", "0")))
        if line is not None:
            print(json.dumps(line), flush=True)
        return
    if args.gpus > 1:   # plain `python bench.py --gpus N`: become the launcher (one process per GPU, RCCL over xGMI)
        avail = args.gpus if args.dry_run else torch.cuda.device_count()
        if avail < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} requested but only {avail} HIP device(s) are visible")
        import torch.multiprocessing as mp

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)
        return
    line = run_rank(args, 0, 1, 0)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
