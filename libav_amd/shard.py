"""Stream sharding + the control plane used at N > 1 GPUs.

Pictures of different streams share nothing, so no sample data ever crosses xGMI: every rank owns the
DPBs of the streams it decodes in its local HBM.  torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests) carries only the control plane:

* the broadcast of the stream table from rank 0,
* the barrier around the timed region,
* the closing reduction of the per-rank counters (max time, summed units),
* and the WORK QUEUE: rank 0 owns one counter of (stream, GOP) items; a rank takes the next
  `batch` items with one atomic fetch-and-add on it (`WorkQueue.next`) when it has room for more
  work, so a slow or busy GPU simply takes fewer items.  The counter lives in the job's rendezvous
  store (served by rank 0's process); nothing but the item index travels.  The reference's analogue
  is the row-progress protocol of its frame threads (libavcodec/pthread_frame.c:502-541): a shared
  counter per unit of work, consumers wait on / advance it, no data is copied.

`my_streams` is the static deal (stream s -> rank s mod world) used where the work per rank must be
fixed in advance (weak-scaling bench with identical ranks).
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


def make_work_items(stream_gops):
    """(stream, gop) items of a job, interleaved across streams so that the GOPs of one stream are handed out in
    order (gop g of a stream is never taken before gop g-1 of it has been: a closed GOP does not need its
    predecessor's pictures, but a consumer that wants in-order output per stream gets it for free).
    stream_gops: list of GOP counts per stream (uneven lengths allowed)."""
    items, g = [], 0
    while True:
        row = [(s, g) for s, n in enumerate(stream_gops) if g < n]
        if not row:
            return items
        items += row
        g += 1


class WorkQueue:
    """One shared counter of work items, owned by rank 0, pulled by every rank.

    All ranks construct the queue collectively (same `n_items`, same order of construction); `next()` is a
    one-sided atomic fetch-and-add and needs no matching call on other ranks."""
    _generation = 0

    def __init__(self, n_items, batch=1):
        self.n_items, self.batch = int(n_items), int(batch)
        self._local = 0
        self._store = None
        WorkQueue._generation += 1
        self._key = "mi355_work_queue_%d" % WorkQueue._generation
        if dist.is_initialized() and dist.get_world_size() > 1:
            self._store = dist.distributed_c10d._get_default_store()
            if dist.get_rank() == 0:
                self._store.add(self._key, 0)          # create the counter before anybody pulls
            dist.barrier()

    def next(self):
        """-> range of item indices this rank now owns, or None when the queue is drained"""
        if self._store is not None:
            start = self._store.add(self._key, self.batch) - self.batch
        else:
            start, self._local = self._local, self._local + self.batch
        if start >= self.n_items:
            return None
        return range(start, min(start + self.batch, self.n_items))


def reduce_counters(elapsed_s, units, device):
    """(max over ranks of elapsed, sum over ranks of units)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_counts(value, device):
    """every rank's integer `value`, as a list indexed by rank (diagnostics of the queue's deal)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.int64, device=device)
    t[dist.get_rank()] = int(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
