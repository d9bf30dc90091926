"""The RCCL leg of the multi-GPU path on the one GPU a test box has: a single-rank `nccl` (= RCCL on ROCm) process group, the per-step collective of
bench.py / parallel.all_gather_outputs on the packed rows a real step kernel wrote.  (World sharding itself is covered by the 2-rank gloo tests in
tests/test_cpu_dist.py; 8-GPU runs are the driver's.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(180)
def test_single_rank_rccl_all_gather_of_kernel_packed_rows():
    import torch
    import torch.distributed as dist

    import gymnasium_robotics_amd as grx
    from gymnasium_robotics_amd.parallel import all_gather_outputs, unpack_outputs

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        env = grx.make_vec("FetchPush-v4", num_envs=64, device="cuda:0", output="torch", autoreset_mode="disabled")
        env.reset(seed=0)
        obs, r, term, trunc, info = env.step(torch.zeros(64, 4, device="cuda:0"))
        gathered = all_gather_outputs(env.packed)
        torch.cuda.synchronize()
        assert gathered.shape == env.packed.shape and torch.equal(gathered, env.packed)
        o, rew, succ = unpack_outputs(gathered, env.obs_dim, 3)
        assert torch.equal(o["observation"], obs["observation"]) and torch.equal(rew, r) and np.array_equal(succ.cpu().numpy(), info["is_success"].cpu().numpy() > 0.5)
    finally:
        dist.destroy_process_group()
