"""bench.py -- env-steps/s of the batched Fetch hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): "env-steps/s (whole node) at N parallel envs".  Workload at N=1 = BASELINE.json
configs[1]: FetchPickAndPlace-v4, 4096 envs on one MI355X, sparse reward + HER reward recompute,
random actions, episodes auto-reset at 50 steps (SAME_STEP, so every timed vector step runs the full
20-substep physics for every world).  Weak scaling: 4096 worlds per GPU; worlds are tile-sharded over the
ranks and the only collective is the RCCL all-gather of the per-step outputs.

One "step" = one env.step() of all worlds = ONE launch of grx_fetch_step_kernel (+ the HER reward kernel).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENV_ID = "FetchPickAndPlace-v4"
WORLDS_PER_GPU = 4096
# Other BASELINE.json configs can be timed with --workload (the driver's default run is cfg 2 = "fetch").  Per config: env id, worlds
# per GPU, action dim, step-kernel name, algorithmic HBM bytes per env-step
WORKLOADS = {
    "fetch": ("FetchPickAndPlace-v4", 4096, 4, "grx_fetch_step_kernel", 715),
    "hand_touch": ("HandManipulateBlockRotateXYZ_ContinuousTouchSensors-v1", 16384, 20, "grx_hand_step_kernel", 1635),   # cfg 3
    "hand_reach": ("HandReach-v3", 16384, 20, "grx_hand_step_kernel", 1035),   # r 24+24+24+20, w 72, out 63+15+15+1
    "antmaze": ("AntMaze_Large_Diverse_GR-v5", 8192, 8, "grx_point_step_kernel", 507),                                      # cfg 4 (8192 per GPU x 8)
}
HER_K = 4  # relabelled goals per transition ("future" strategy with k=4)
# algorithmic HBM bytes per env-step of the fused kernel (SURVEY.md §8(d) cfg 2):
# read qpos22+qvel21+warm21+mocap7+act4 = 75 words, write 22+21+21+7 = 71, outputs obs25+ag3+dg3+r1 = 32 -> 178*4 + 3 flag bytes
ALGO_BYTES_PER_ENV_STEP = 715
HER_BYTES_PER_TRANSITION = 28
HBM_PEAK_GBS = 8000.0


def cpu_baseline(seconds: float = 12.0):
    """Oracle (fp64 restatement, oracle/) timed on one host core on the same workload, single world."""
    from gymnasium_robotics_amd.envs.fetch import load_fetch_model
    from oracle.fetch_oracle import OracleFetchEnv

    env = OracleFetchEnv(load_fetch_model("FetchPickAndPlace"), "FetchPickAndPlace")
    rng = np.random.default_rng(0)
    env.reset(seed=0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for t in range(50):
            env.step(rng.uniform(-1, 1, 4))
            n += 1
        env.reset()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} env.step() calls of 1 world, {ENV_ID}, random actions, 50-step episodes ({dt:.1f} s of CPU work); "
                      "oracle = fp64 restatement (MuJoCo is not installable here), Python task layer + C physics"}


def other_workload(args):
    """Same contract as the default run for the other single-kernel families: random actions, same-step autoreset at the family's
    time limit, one RCCL all-gather of the per-step outputs when N > 1.  No HER leg, no CPU baseline."""
    env_id, n_default, act_dim, kernel, algo = WORKLOADS[args.workload]
    world_size, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(f"cuda:{local_rank}"))
    device = f"cuda:{local_rank}"
    n = args.worlds_per_gpu if args.worlds_per_gpu != WORLDS_PER_GPU else n_default
    if args.workload == "antmaze":
        from gymnasium_robotics_amd.envs.point_maze import AntMazeVecEnv as Env
    else:
        from gymnasium_robotics_amd.envs.hand import HandBlockVecEnv, HandReachVecEnv
        Env = HandReachVecEnv if args.workload == "hand_reach" else HandBlockVecEnv
    env = Env(env_id, num_envs=n, device=device, output="torch", autoreset_mode="same_step", seed_offset=rank * n)
    env.reset(seed=0)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    gdim = env.goal.shape[1]
    out_dim = env.obs_dim + 2 * gdim + 2
    packed = torch.empty(n, out_dim, device=device)
    gathered = torch.empty(n * world_size, out_dim, device=device) if dist else None
    events = []

    def one_step(timed):
        a = torch.rand(n, act_dim, device=device, generator=gen) * 2 - 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        obs, r, term, trunc, info = env.step(a)
        e1.record()
        if timed:
            events.append((e0, e1))
        if dist:
            packed[:, : env.obs_dim] = obs["observation"]
            packed[:, env.obs_dim: env.obs_dim + gdim] = obs["achieved_goal"]
            packed[:, env.obs_dim + gdim: env.obs_dim + 2 * gdim] = obs["desired_goal"]
            packed[:, -2] = r
            packed[:, -1] = (info["is_success"] if "is_success" in info else info["success"]).float()
            dist.all_gather_into_tensor(gathered, packed)

    for _ in range(args.warmup):
        one_step(False)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # env.step() here = the step kernel (+ masked reset launches at episode ends): events bracket the whole call
    step_ms = float(np.median([a.elapsed_time(b) for a, b in events]))
    if rank == 0:
        value = n * world_size * args.steps / elapsed
        achieved = algo * n / (step_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "env-steps/s (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{env_id}, {n} worlds/GPU x {world_size} GPU, uniform random actions, same-step autoreset at the time limit",
                       "worlds_per_gpu": n, "parallelism": f"world-shard x{world_size}" + (", RCCL all_gather of outputs" if world_size > 1 else ""),
                       "status_flagged_worlds": int((env.status != 0).sum().item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel, "kernel_ms": step_ms, "algorithmic_bytes_per_launch": algo * n,
                         "note": "median env.step() device time (step kernel; episode-end steps add masked reset launches); issue-bound, see DESIGN.md"},
        }))
    if dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--worlds-per-gpu", type=int, default=WORLDS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="fetch")
    args = ap.parse_args()
    if args.workload != "fetch":
        return other_workload(args)

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(f"cuda:{local_rank}"))
    n_gpus = world_size
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from gymnasium_robotics_amd.envs.fetch import FetchVecEnv

    n = args.worlds_per_gpu
    env = FetchVecEnv(ENV_ID, num_envs=n, device=device, output="torch", autoreset_mode="same_step", seed_offset=rank * n)
    env.reset(seed=0)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    out_dim = env.obs_dim + 3 + 3 + 1 + 1  # obs, achieved, desired, reward, success
    packed = torch.empty(n, out_dim, device=device)
    gathered = torch.empty(n * n_gpus, out_dim, device=device) if dist else None
    perm = torch.randperm(n, device=device, generator=gen)

    def one_step():
        a = torch.rand(n, 4, device=device, generator=gen) * 2 - 1
        obs, r, term, trunc, info = env.step(a)
        # HER relabel: reward recompute for HER_K substituted goals per transition
        ag = obs["achieved_goal"].unsqueeze(0).expand(HER_K, n, 3).contiguous()
        dg = torch.stack([obs["desired_goal"][torch.roll(perm, k)] for k in range(HER_K)])
        env.compute_reward(ag, dg, None)
        if dist:
            packed[:, : env.obs_dim] = obs["observation"]
            packed[:, env.obs_dim: env.obs_dim + 3] = obs["achieved_goal"]
            packed[:, env.obs_dim + 3: env.obs_dim + 6] = obs["desired_goal"]
            packed[:, -2] = r
            packed[:, -1] = info["is_success"].float()
            dist.all_gather_into_tensor(gathered, packed)

    for _ in range(args.warmup):
        one_step()
    env.kernel_events = []  # HIP events (torch's current stream = the launch stream) around every grx_fetch_step_kernel launch
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    status_max = int(env.status.max().item())
    # dominant kernel: average duration of the grx_fetch_step_kernel launches of the timed region
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in env.kernel_events]))

    # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
    # (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process); the committed summary is profiles/pmc_r01_hbm_traffic.*
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_r01_hbm_traffic.json")) as f:
            traffic = json.load(f)["traffic_bytes_per_launch"] if n == WORLDS_PER_GPU else None
    except OSError:
        pass
    if rank == 0:
        total_steps = n * n_gpus * args.steps
        value = total_steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "env-steps/s (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{ENV_ID}, {n} worlds/GPU x {n_gpus} GPU, 20 fused substeps, sparse reward + HER recompute (k={HER_K}), "
                                   "uniform random actions, same-step autoreset at 50 steps", "worlds_per_gpu": n,
                       "parallelism": f"world-shard x{n_gpus}" + (", RCCL all_gather of outputs" if n_gpus > 1 else ""),
                       "status_max": status_max},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": "profiles/pmc_r01_hbm_traffic.txt (rocprofv3 PMC, bytes per launch)",
                         "kernel": "grx_fetch_step_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n,
                         "note": "fused path is VALU/LDS/latency bound (~2e3 FLOP/B), HBM fraction is tiny by construction; see DESIGN.md"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
