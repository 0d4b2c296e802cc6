"""Reference-column sharding across the GPUs of one node (BASELINE.json north_star).

Reference positions split into non-overlapping column blocks; every rank (one process per GPU) runs the
whole hot path on its own block and the per-block pileup text is collected on rank 0 with ONE gather per
step: an 8-byte size all-gather plus grouped point-to-point transfers of exactly the bytes each rank
produced (ncclGroupStart / ncclSend / ncclRecv over xGMI when the backend is "nccl" = RCCL, gloo in the
CPU tests).  The reference's own precedent for position sharding is the span-job loop of
bam_consensus.c:2759-2810 and the per-interval iterators of bedcov.c:297-308; what a rank must read
besides the reads that touch its block (the halo) follows host_pump.h.

Two layers use this module:
  * bench.py --gpus N: ONE device-resident input, each rank piles its block (`block_of`, `read_range`);
  * `python -m samtools_amd.shard mpileup|depth <args>` under torchrun: the product drivers
    (sta_main_mpileup / sta_main_depth) run with STA_SHARD=rank/world, which restricts a rank to its block
    of the (region-clipped) genome, and the text is gathered and written by rank 0 (`run_sharded_cli`).
Nothing here computes pileups: it only decides who owns which columns and moves finished text.
"""
import os
import time
import sys
import tempfile
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


def block_of(rank: int, world: int, n_cols: int) -> Tuple[int, int]:
    """Columns [beg, end) of a linear coordinate space of n_cols that rank `rank` owns: equal contiguous blocks
    (the same rule the C drivers apply to STA_SHARD=rank/world, driver_shard.h)."""
    beg = n_cols * rank // world
    end = n_cols * (rank + 1) // world
    return beg, end


def halo_columns(max_ref_span: int) -> int:
    """Reads starting up to this many columns before (or after) a block can still touch it, rewrite -- through
    mate-overlap resolution -- the qualities of a read that does, or be the push that releases the block's last
    columns: 2 x the longest reference span (host_pump.h keep_mates / surely_pushed)."""
    return 2 * int(max_ref_span)


def read_range(abs_pos, blk_beg: int, blk_end: int, halo: int) -> Tuple[int, int]:
    """Index range [lo, hi) of the position-sorted reads a rank stages for block [blk_beg, blk_end): everything starting
    inside the block or within `halo` columns of either side."""
    import numpy as np
    lo = int(np.searchsorted(abs_pos, blk_beg - halo, side="left"))
    hi = int(np.searchsorted(abs_pos, blk_end + halo, side="left"))
    return lo, hi


def exchange_sizes(n_local: int, device, group=None) -> List[int]:
    """Every rank's text size (one 8-byte all_gather)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    return [int(s.item()) for s in sizes]


def wait_all(works) -> None:
    for w in works or ():
        w.wait()


def gather_text_v(local, n_local: int, dst: int = 0, group=None, sizes: Sequence[int] = None, recv=None):
    """Variable-size gather of byte tensors onto `dst`, nothing padded: `dst` posts one receive per peer straight into
    its slice of ONE contiguous buffer (rank order = output order) and every other rank posts one send of exactly its
    n_local bytes; the operations are issued as one group (dist.batch_isend_irecv -> ncclGroupStart/End on RCCL).
    Returns the list of in-flight works (wait with wait_all); on `dst`, `recv` (world-total bytes, allocated here when
    None) holds the concatenated text once they completed -- read it as recv[:sum(sizes)].  dst's own piece is copied
    locally."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = exchange_sizes(n_local, local.device, group)
    sizes = [int(x) for x in sizes]
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # the 1-GPU test box: several ranks share one device and talk over gloo, which moves host memory
        host = local[:sizes[rank]].cpu()
        works = gather_text_v(host, sizes[rank], dst=dst, group=group, sizes=sizes)
        wait_all(works)
        if rank == dst:
            if recv is None:
                recv = torch.empty(max(sum(sizes), 1), dtype=torch.uint8, device=local.device)
            recv[:sum(sizes)].copy_(gather_text_v.last_recv[:sum(sizes)])
            gather_text_v.last_recv = recv
        return []
    ops = []
    if rank == dst:
        if recv is None:
            recv = torch.empty(max(sum(sizes), 1), dtype=torch.uint8, device=local.device)
        off = 0
        for r in range(world):
            piece = recv[off:off + sizes[r]]
            if r == rank:
                piece.copy_(local[:sizes[r]])
            elif sizes[r]:
                ops.append(dist.P2POp(dist.irecv, piece, r, group))
            off += sizes[r]
    elif sizes[rank]:
        ops.append(dist.P2POp(dist.isend, local[:sizes[rank]], dst, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    gather_text_v.last_recv = recv if rank == dst else None
    return works


def stream_text(local, n_local: int, write, dst: int = 0, group=None, sizes: Sequence[int] = None, chunk_bytes: int = 64 << 20):
    """The gather without the staging tensor: `dst` hands its own text to `write` (a callable taking a bytes-like object), then takes the
    other ranks' blocks in rank order, `chunk_bytes` at a time through TWO receive buffers -- the next chunk is in flight while the
    previous one is written -- so rank 0 holds 2 x chunk_bytes instead of the whole job's text (2.2 GB per step at the bench's 8-GPU
    shape).  Every other rank sends its block in the same chunks.  Same bytes over the same links as gather_text_v; the output is
    byte-identical (tests/test_shard_gloo.py).  Precedent for writing blocks as they complete: the region jobs of
    bam_consensus.c:2759-2790."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = exchange_sizes(n_local, local.device, group)
    sizes = [int(x) for x in sizes]
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"      # (the 1-GPU test box: ranks share a device and talk over gloo)
    src = local[:sizes[rank]].cpu() if via_host else local[:sizes[rank]]
    if rank != dst:
        works = []
        for off in range(0, sizes[rank], chunk_bytes):
            works.append(dist.isend(src[off:min(off + chunk_bytes, sizes[rank])], dst, group=group))
        wait_all(works)
        return
    write(src.cpu().numpy().tobytes() if src.is_cuda else src.numpy().tobytes())
    bufs = [torch.empty(chunk_bytes, dtype=torch.uint8, device=src.device) for _ in range(2)]
    jobs = [(r, off, min(chunk_bytes, sizes[r] - off)) for r in range(world) if r != dst for off in range(0, sizes[r], chunk_bytes)]
    inflight = None
    for k, (r, off, n) in enumerate(jobs):
        w = dist.irecv(bufs[k & 1][:n], r, group=group)
        if inflight is not None:
            pw, pk, pn = inflight
            pw.wait()
            write(bufs[pk & 1][:pn].cpu().numpy().tobytes())
        inflight = (w, k, n)
    if inflight is not None:
        pw, pk, pn = inflight
        pw.wait()
        write(bufs[pk & 1][:pn].cpu().numpy().tobytes())


def gather_text(local, dst: int = 0, group=None, sizes: Sequence[int] = None):
    """Blocking form: every rank's byte tensor on `dst`, in rank order, as one contiguous tensor (None elsewhere)."""
    import torch.distributed as dist

    if sizes is None:
        sizes = exchange_sizes(local.numel(), local.device, group)
    works = gather_text_v(local, local.numel(), dst=dst, group=group, sizes=sizes)
    wait_all(works)
    if dist.get_rank(group) != dst:
        return None
    return gather_text_v.last_recv[:sum(sizes)]


def run_sharded_cli(argv: Sequence[str], out=None) -> int:
    """`mpileup ...` / `depth ...` across the ranks of the current process group: every rank runs the product driver
    restricted to its block (STA_SHARD), rank 0 writes the concatenated text to `out` (default stdout, or the -o file
    the command names).  Returns the worst exit status over the ranks."""
    import torch
    import torch.distributed as dist
    from . import _capi

    rank, world = dist.get_rank(), dist.get_world_size()
    sub, args = argv[0], list(argv[1:])
    if sub not in ("mpileup", "depth"):
        raise SystemExit("samtools_amd.shard: only mpileup and depth shard over reference columns")
    # the command's own -o/--output is where rank 0 finally writes (all four getopt spellings); every rank's block is
    # captured in memory (sta_main_capture) and goes from there straight into the gather
    final, rest, k = None, [], 0
    while k < len(args):
        a = args[k]
        if a in ("-o", "--output") and k + 1 < len(args):
            final = args[k + 1]; k += 2; continue
        if a.startswith("--output="):
            final = a[len("--output="):]; k += 1; continue
        if a.startswith("-o") and len(a) > 2 and not a.startswith("--"):
            final = a[2:]; k += 1; continue
        rest.append(a); k += 1
    args = rest
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    os.environ["STA_SHARD"] = "%d/%d" % (rank, world)
    t0 = time.perf_counter()
    # On GPUs (RCCL) the block's text never leaves device memory on its way into the gather: the driver emits every window behind the
    # previous one in a device buffer (sta_main_capture_device), which is copied device to device into the tensor the collective
    # sends -- rank 0's download of the gathered text is the only PCIe trip.  STA_SHARD_HOST_CAPTURE=1 (and the CPU / gloo form of the
    # tests) takes the text through host memory as before.
    dev_capture = use_cuda and not os.environ.get("STA_SHARD_HOST_CAPTURE") and int(os.environ.get("STA_DEV_THREADS", "1")) == 1
    try:
        if dev_capture:
            rc, head, n_dev = _capi.main_capture_device(sub, args)
            local = torch.empty(len(head) + n_dev, dtype=torch.uint8, device=dev)
            if head:
                local[:len(head)].copy_(torch.frombuffer(bytearray(head), dtype=torch.uint8))
            _capi.capture_device_take(local.data_ptr() + len(head) if n_dev else None, n_dev)
            n_local = len(head) + n_dev
        else:
            rc, data = _capi.main_capture(sub, args)
            local = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev) if data else torch.zeros(0, dtype=torch.uint8, device=dev)
            n_local = len(data)
    finally:
        os.environ.pop("STA_SHARD", None)
    t_drv = time.perf_counter() - t0
    rcs = torch.tensor([rc], dtype=torch.int64, device=dev)
    dist.all_reduce(rcs, op=dist.ReduceOp.MAX)
    worst = int(rcs.item())
    t1 = time.perf_counter()
    # STA_SHARD_PWRITE=1 with -o FILE: the text stays sharded -- after the 8-byte size all-gather every rank writes ITS block at its offset of
    # the file (pwrite), nothing but the sizes crosses the links.  The gather funnels every rank's text through one xGMI link each into rank 0
    # (76.8 GB/s): fine under BAQ (a 16 M-column step is 23.7 ms for 1.29 GB), seven times too slow for `-B` and `depth`.
    if os.environ.get("STA_SHARD_PWRITE") and final and out is None:
        sizes = exchange_sizes(n_local, dev)
        if worst == 0:
            if rank == 0:
                with open(final, "wb") as fh:
                    fh.truncate(sum(sizes))
            dist.barrier()
            data = memoryview(local.cpu().numpy()) if n_local else memoryview(b"")
            fd = os.open(final, os.O_WRONLY)
            try:
                off, done = sum(sizes[:rank]), 0
                while done < n_local:
                    done += os.pwrite(fd, data[done:done + (256 << 20)], off + done)
            finally:
                os.close(fd)
            dist.barrier()
        elif rank == 0:
            sys.stderr.write("samtools_amd.shard: a rank failed (worst exit status %d): no output written\n" % worst)
        if os.environ.get("STA_SHARD_TIMING"):
            sys.stderr.write("[shard %d/%d] driver %.3f s, %d bytes (%s capture); written in place %.3f s\n" % (rank, world, t_drv, n_local, "device" if dev_capture else "host", time.perf_counter() - t1))
        return worst
    # STA_SHARD_STREAM=1: rank 0 writes every block as it arrives (two 64 MiB receive buffers) instead of staging the whole job's text
    stream = bool(os.environ.get("STA_SHARD_STREAM"))
    if stream:
        sizes = exchange_sizes(n_local, dev)
        sink = None
        if rank == 0 and worst == 0:
            sink = out if out is not None else (open(final, "wb") if final else sys.stdout.buffer)
        stream_text(local, n_local, (sink.write if sink is not None else (lambda b: None)), dst=0, sizes=sizes,
                    chunk_bytes=int(os.environ.get("STA_SHARD_STREAM_CHUNK", str(64 << 20))))
        if rank == 0 and sink is not None:
            sink.flush()
            if final and out is None:
                sink.close()
        whole = None
    else:
        whole = gather_text(local, dst=0)
    t_gather = time.perf_counter() - t1
    if os.environ.get("STA_SHARD_TIMING"):
        sys.stderr.write("[shard %d/%d] driver %.3f s, %d bytes (%s capture); gather %.3f s\n" % (rank, world, t_drv, n_local, "device" if dev_capture else "host", t_gather))
    if rank == 0:
        if worst != 0:
            # a failed block would leave a silent hole in the concatenation: nothing is written (the ranks' own messages
            # are on stderr already)
            sys.stderr.write("samtools_amd.shard: a rank failed (worst exit status %d): no output written\n" % worst)
            return worst
        if stream:
            return worst
        buf = whole.cpu().numpy().tobytes()
        if out is not None:
            out.write(buf)
        elif final:
            with open(final, "wb") as fh:
                fh.write(buf)
        else:
            sys.stdout.buffer.write(buf)
            sys.stdout.buffer.flush()
    return worst


def main() -> int:
    """torchrun entry: python -m torch.distributed.run --nproc-per-node N -m samtools_amd.shard mpileup -f ref.fa in.bam"""
    import torch
    import torch.distributed as dist

    backend = os.environ.get("STA_SHARD_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        # one process per GPU; STA_SHARD_ONE_DEVICE=1 puts every rank on device 0 (the 1-GPU test box)
        local = 0 if os.environ.get("STA_SHARD_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ["STA_DEVICE"] = str(local)
    dist.init_process_group(backend)
    try:
        return run_sharded_cli(sys.argv[1:])
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
