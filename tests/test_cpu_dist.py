"""world_size=2 gloo test of the multi-GPU data path: tile-sharding + the one all-gather per step."""
import os

import numpy as np
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
