"""Multi-GPU sharding of the channel axis (host logic only).

Channels are independent reference chains, so a node's G GPUs take contiguous channel ranges and
exchange nothing on the data path; torch.distributed (RCCL on GPUs, gloo in the CPU tests) is only
used to line the ranks up for timing and to collect per-rank summaries.
"""


def channel_range(n_channels, world_size, rank):
    """Contiguous range [lo, hi) of rank's channels; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    q, r = divmod(n_channels, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def max_over_ranks(dist, value, device=None):
    """MAX all-reduce of a python float (elapsed time): the slowest rank defines the step time."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, values, device=None):
    """SUM all-reduce of a short list of python ints (per-rank check counters): every rank gets the job's totals."""
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def gather_channel_rows(dist, local_rows, n_channels, world_size, device=None):
    """Concatenate per-rank row blocks (e.g. n_bits per channel) back into channel order on every rank."""
    import torch
    sizes = [channel_range(n_channels, world_size, r) for r in range(world_size)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=device)
    pad[: local_rows.shape[0]] = local_rows
    outs = [torch.zeros_like(pad) for _ in range(world_size)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
