"""Multi-GPU sharding of the likelihood path: one process per GPU, contiguous window blocks per rank.

Windows (indeed pairs) are independent, so there is no exchange step inside the path; the only
collective is the gather of per-pair records (log-likelihood + off-haplotype flags) to rank 0 that the
downstream genotype reduction consumes, in window order.  Backend "nccl" is RCCL over
xGMI on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist


def window_block(n_windows: int, rank: int, world: int):
    """Contiguous, balanced block [w0, w1) of rank `rank`: concatenating ranks keeps window order."""
    base, rem = divmod(n_windows, world)
    w0 = rank * base + min(rank, rem)
    return w0, w0 + base + (1 if rank < rem else 0)


def gather_records(local: torch.Tensor, dst: int = 0, group=None, sizes=None):
    """Gather a 1-D per-pair tensor from every rank to `dst`, ragged sizes allowed.

    `sizes` (per-rank element counts) may be passed when every rank already knows them (e.g. equal
    shards); otherwise they are exchanged first.  Returns the concatenation in rank (= window) order on
    `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        szt = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(szt, n, group=group)
        sizes = [int(s.item()) for s in szt]
    m = max(sizes)
    if local.numel() < m:
        padded = torch.zeros(m, dtype=local.dtype, device=local.device)
        padded[:local.numel()] = local
    else:
        padded = local
    bufs = [torch.empty(m, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
