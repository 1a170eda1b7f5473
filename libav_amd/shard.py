"""Stream sharding + the control plane used at N > 1 GPUs.

Pictures of different streams share nothing, so streams are dealt round-robin to ranks and no
sample data ever crosses xGMI.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests) carries only: the broadcast of the stream table from rank 0, the barrier around the
timed region, and the reduction of the per-rank counters.
"""
import torch
import torch.distributed as dist


def make_stream_table(n_streams, base_seed):
    """rank-0 side: one row per stream: (stream id, seed)"""
    return torch.tensor([[s, base_seed + s] for s in range(n_streams)], dtype=torch.int64)


def broadcast_stream_table(table, n_streams, device):
    """Every rank returns the same [n_streams, 2] int64 table (rank 0 provides it)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return table
    t = table.to(device) if dist.get_rank() == 0 else torch.zeros((n_streams, 2), dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return t.cpu()


def my_streams(table, rank, world):
    """Static round-robin deal: stream s -> rank s mod world."""
    return [(int(s), int(seed)) for s, seed in table.tolist() if s % world == rank]


def reduce_counters(elapsed_s, units, device):
    """(max over ranks of elapsed, sum over ranks of units)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
