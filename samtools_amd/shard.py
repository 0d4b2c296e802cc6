"""Window sharding across the GPUs of one node (BASELINE.json north_star).

Reference positions split into non-overlapping windows; every rank (one process per GPU) runs the
whole hot path on its own windows and the per-window pileup text is collected on rank 0 with ONE
gather per step (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).  The reference's
own precedent for position sharding is bam_consensus.c:2759-2790 (span jobs) and bedcov.c:297-308
(per-interval iterators).  Nothing here computes pileups: it only decides who owns which columns
and moves finished text.
"""
from typing import List, Sequence, Tuple


def plan_windows(contig_lengths: Sequence[int], window_cols: int) -> List[Tuple[int, int, int]]:
    """All windows (tid, beg, end) of a genome in output order (contig order, then position)."""
    out = []
    for tid, n in enumerate(contig_lengths):
        beg = 0
        while beg < n:
            end = min(n, beg + window_cols)
            out.append((tid, beg, end))
            beg = end
    return out


def windows_of_rank(windows: Sequence[Tuple[int, int, int]], rank: int, world: int) -> List[int]:
    """Indices of the windows rank `rank` owns: contiguous blocks (so a rank's text is one contiguous
    piece of the final output), sizes differing by at most one window."""
    n = len(windows)
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return list(range(lo, hi))


def halo_columns(max_ref_span: int) -> int:
    """Reads starting up to this many columns before a window can still touch it or rewrite (through
    mate-overlap resolution) the qualities of a read that does: 2 x the longest reference span.

    Two refinements found with long ref skips (host_pump.h): a shard must also receive (i) the earlier
    mate of every read that is live in it, even when that mate ends before the shard starts (HTSlib may rewrite bases of
    the later mate beyond the earlier mate's end), and (ii) the records after its last column up to the first one that is
    certainly pushed (it releases the shard's last columns and may be an overlap mate).  The drivers' pumps implement both;
    a region-restricted reader per rank must widen its query accordingly."""
    return 2 * int(max_ref_span)


def exchange_sizes(n_local: int, device, group=None) -> List[int]:
    """Every rank's text size (one 8-byte all_gather)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    return [int(s.item()) for s in sizes]


class PendingGather:
    """An in-flight gather: wait() returns the per-rank byte tensors on the destination rank (None elsewhere)."""

    def __init__(self, work, recv, sizes):
        self._work, self._recv, self._sizes = work, recv, sizes

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._recv is None:
            return None
        return [r[:s] for r, s in zip(self._recv, self._sizes)]


def gather_text(local, dst: int = 0, group=None, sizes: Sequence[int] = None, recv=None, async_op: bool = False):
    """Collect every rank's byte tensor (uint8, 1-D, on the backend's device) on `dst` in rank order.

    ONE gather of the text, padded to the largest piece; when `sizes` (every rank's byte count) is not supplied it is
    obtained with one 8-byte all_gather first.  `local` may be longer than its entry in `sizes` (a reusable buffer).
    `recv` optionally supplies the destination's receive buffers (world tensors of >= max(sizes) bytes) so that a
    steady-state loop allocates nothing.  With async_op=True a PendingGather is returned at once: the copy runs on
    the backend's own stream and overlaps whatever the caller launches next (the next window's kernels)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = exchange_sizes(local.numel(), local.device, group)
    sizes = [int(x) for x in sizes]
    cap = max(max(sizes), 1)
    if local.numel() >= cap:
        buf = local[:cap]
    else:
        buf = torch.zeros(cap, dtype=torch.uint8, device=local.device)
        buf[:local.numel()] = local
    if rank == dst:
        if recv is None:
            recv = [torch.empty(cap, dtype=torch.uint8, device=local.device) for _ in range(world)]
        recv = [r[:cap] for r in recv]
    else:
        recv = None
    work = dist.gather(buf, recv, dst=dst, group=group, async_op=async_op)
    pending = PendingGather(work if async_op else None, recv, sizes)
    return pending if async_op else pending.wait()
