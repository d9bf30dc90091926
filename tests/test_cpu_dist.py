"""world_size=2 gloo test of the multi-GPU data path: tile-sharding + the one all-gather per step."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gymnasium_robotics_amd.parallel import all_gather_outputs, pack_outputs, shard_range, unpack_outputs

    n_total, obs_dim = 8, 25
    lo, hi = shard_range(n_total, rank, world)
    # per-world synthetic outputs that encode the global world index, as each rank's env would produce for its tile
    idx = torch.arange(lo, hi, dtype=torch.float32)
    obs = {"observation": idx[:, None] + torch.arange(obs_dim)[None] * 0.01, "achieved_goal": idx[:, None].repeat(1, 3),
           "desired_goal": idx[:, None].repeat(1, 3) + 0.5}
    packed = pack_outputs(obs, -idx, (idx.long() % 2 == 0))
    full = all_gather_outputs(packed)
    o, r, s = unpack_outputs(full, obs_dim, 3)
    ok = (full.shape == (n_total, obs_dim + 8) and torch.equal(r, -torch.arange(n_total, dtype=torch.float32))
          and torch.equal(o["achieved_goal"][:, 0], torch.arange(n_total, dtype=torch.float32))
          and torch.equal(s, torch.arange(n_total) % 2 == 0))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29517, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_shard_range_errors():
    from gymnasium_robotics_amd.parallel import shard_range

    assert shard_range(4096 * 8, 3, 8) == (3 * 4096, 4 * 4096)
    try:
        shard_range(10, 0, 3)
        assert False
    except ValueError:
        pass


def _env_worker(rank, world, port, n_total, steps, ret):
    """each rank owns a tile of the worlds of ONE logical vector env (seed_offset = first world of the tile) and contributes its packed rows"""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_vec_env import EmuFetchVecEnv

    from gymnasium_robotics_amd.parallel import all_gather_outputs, shard_range

    lo, hi = shard_range(n_total, rank, world)
    env = EmuFetchVecEnv("FetchPickAndPlace-v4", num_envs=hi - lo, seed_offset=lo)
    rng = np.random.default_rng(7)
    acts = rng.uniform(-1, 1, (steps, n_total, 4)).astype(np.float32)   # the same global action stream on every rank
    out = [all_gather_outputs(env.reset(seed=11))]
    for t in range(steps):
        out.append(all_gather_outputs(env.step(acts[t, lo:hi])))
    ret[rank] = torch.stack(out).numpy()
    dist.destroy_process_group()


def test_sharded_env_equals_the_unsharded_one():
    """SURVEY.md 8(e) with a real env API object (worlds stepped by the lane emulator): 2 ranks x 2 worlds, every step one all-gather of the
    packed [obs | achieved | desired | reward | success] rows.  Every rank must see, in world order, exactly the rows a single 4-world env
    produces for the same seed and actions (world i is seeded seed + i wherever it lives)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu_vec_env import EmuFetchVecEnv

    n_total, steps = 4, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_env_worker, args=(2, 29531, n_total, steps, ret), nprocs=2, join=True)
    env = EmuFetchVecEnv("FetchPickAndPlace-v4", num_envs=n_total)
    rng = np.random.default_rng(7)
    acts = rng.uniform(-1, 1, (steps, n_total, 4)).astype(np.float32)
    ref = [env.reset(seed=11)] + [env.step(acts[t]) for t in range(steps)]
    ref = torch.stack(ref).numpy()
    assert ret[0].shape == ref.shape == (steps + 1, n_total, 33)
    assert np.array_equal(ret[0], ref) and np.array_equal(ret[1], ref)
    assert len({tuple(row[25:28]) for row in ref[0]}) == n_total      # four different worlds (different object positions), in seed order


@pytest.mark.parametrize("workload,ranks,extra", [("antmaze", 8, ["--worlds-per-gpu", "64"]), ("mixed", 2, ["--worlds-per-gpu", "64"]), ("fetch", 2, ["--worlds-per-gpu", "32"]),
                                                  ("fetch", 2, ["--worlds-per-gpu", "32", "--stages", "2"])])      # (the last: out-of-phase sub-batches, one gather per sub-batch)
def test_bench_multi_rank_command_line_dry_run(workload, ranks, extra):
    """The command the driver runs on the 8-GPU node (`python bench.py --gpus N ...`), with --dry-run: no GPU, no physics (random rows on the CPU), gloo instead
    of RCCL -- everything else is the real code path: the launcher, the rendezvous on 127.0.0.1, world sharding, the per-step all-gather of the rows (checked to
    arrive in world order), barrier + max-over-ranks timing, the per-rank report and the one JSON line of rank 0.  The first time this meets hardware it cannot fail
    on plumbing (BASELINE configs[3]: AntMaze on 8 ranks; configs[4]: the mixed batch)."""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--gpus", str(ranks), "--workload", workload, "--steps", "3", "--warmup", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["dist"]["backend"] == "gloo" and line["dist"]["nranks"] == ranks and len(line["dist"]["elapsed_s_per_rank"]) == ranks
    assert line["data"].startswith("DRY RUN")


def test_bench_under_torchrun_dry_run():
    """the DRIVER's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`):
    bench.py must take rank / world size from the environment instead of spawning, and still print exactly one JSON line (rank 0)"""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--dry-run", "--gpus", "2", "--workload", "antmaze", "--steps", "3", "--warmup", "1", "--worlds-per-gpu", "64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dist"]["nranks"] == 2 and line["value"] > 0
    # a rank count that disagrees with --gpus is refused, loudly
    bad = subprocess.run(cmd[:-10] + [os.path.join(root, "bench.py"), "--dry-run", "--gpus", "4", "--workload", "antmaze", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=root)
    assert bad.returncode != 0


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 2` on a box with fewer HIP devices (this container has none) must fail loudly, not fall back to anything"""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode != 0 and "HIP device(s) are visible" in (out.stderr + out.stdout)
