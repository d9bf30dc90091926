"""World sharding across the GPUs of one node.

Worlds are independent, so the data path has no collective inside ``step``: rank r of G owns the
contiguous tile ``[r*N/G, (r+1)*N/G)`` and runs its own ``FetchVecEnv``.  The only exchange is the
all-gather of the per-step outputs (RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    if n_total % world_size:
        raise ValueError("number of worlds must be divisible by the number of ranks")
    n = n_total // world_size
    return rank * n, (rank + 1) * n


def pack_outputs(obs: Dict[str, torch.Tensor], reward: torch.Tensor, success: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """[n, obs+ag+dg+2] fp32 row per world: observation | achieved_goal | desired_goal | reward | success"""
    o, ag, dg = obs["observation"], obs["achieved_goal"], obs["desired_goal"]
    n, w = o.shape[0], o.shape[1] + ag.shape[1] + dg.shape[1] + 2
    if out is None:
        out = torch.empty(n, w, dtype=torch.float32, device=o.device)
    k = 0
    for t in (o, ag, dg):
        out[:, k: k + t.shape[1]] = t
        k += t.shape[1]
    out[:, k] = reward
    out[:, k + 1] = success.to(torch.float32)
    return out


def unpack_outputs(packed: torch.Tensor, obs_dim: int, goal_dim: int):
    o = packed[:, :obs_dim]
    ag = packed[:, obs_dim: obs_dim + goal_dim]
    dg = packed[:, obs_dim + goal_dim: obs_dim + 2 * goal_dim]
    return {"observation": o, "achieved_goal": ag, "desired_goal": dg}, packed[:, -2], packed[:, -1] > 0.5


def all_gather_outputs(packed: torch.Tensor, gathered: torch.Tensor = None, group=None) -> torch.Tensor:
    """One collective per step: every rank receives the [N_total, w] output matrix (rank-major = world order)."""
    ws = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty(packed.shape[0] * ws, packed.shape[1], dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed.contiguous(), group=group)
    return gathered


def rank_stats(values: Dict[str, float], device, group=None) -> Dict[str, object]:
    """What rank 0 needs to report about the whole job next to its own numbers: the rank count AS THE BACKEND SEES IT (RCCL / gloo, not the launcher's
    argument), the backend's name, and every rank's value of each entry of `values` (kernel time, elapsed time, ...), rank-major.  One small all-gather."""
    ws = dist.get_world_size(group)
    keys = sorted(values)
    mine = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=device)
    allv = torch.empty(ws * len(keys), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allv, mine, group=group)
    allv = allv.view(ws, len(keys)).cpu()
    out = {"backend": dist.get_backend(group), "nranks": ws}
    for j, k in enumerate(keys):
        out[k + "_per_rank"] = [float(x) for x in allv[:, j]]
    return out
