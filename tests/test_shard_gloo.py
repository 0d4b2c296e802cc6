"""Multi-rank path on CPU: window ownership and the one-gather collection of per-window text
(samtools_amd/shard.py), world_size 2 over gloo.  The engine itself is not called here (no GPU):
each rank fabricates the text its windows would produce, and rank 0 must reassemble the exact
single-process output."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from samtools_amd import shard  # noqa: E402


def fake_window_text(tid, beg, end):
    # variable-length rows, like pileup text
    return b"".join(b"chr%d\t%d\t%s\n" % (tid, p + 1, b"." * (p % 7)) for p in range(beg, end))


def _worker(rank, world, port, contigs, wcols, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wins = shard.plan_windows(contigs, wcols)
    mine = shard.windows_of_rank(wins, rank, world)
    text = b"".join(fake_window_text(*wins[i]) for i in mine)
    local = torch.frombuffer(bytearray(text), dtype=torch.uint8) if text else torch.zeros(0, dtype=torch.uint8)
    whole = shard.gather_text(local, dst=0)
    # the steady-state form bench.py uses: sizes exchanged once, a reusable (larger) send buffer, one reusable receive buffer,
    # grouped point-to-point transfers of the TRUE sizes (nothing padded), two gathers in flight
    sizes = shard.exchange_sizes(local.numel(), local.device)
    padded = torch.zeros(local.numel() + 5, dtype=torch.uint8)
    padded[:local.numel()] = local
    recv = [torch.empty(max(sum(sizes), 1), dtype=torch.uint8) for _ in range(2)] if rank == 0 else [None, None]
    pend = [shard.gather_text_v(padded, local.numel(), dst=0, sizes=sizes, recv=recv[k]) for k in range(2)]
    for p in pend:
        shard.wait_all(p)
    # the streaming form (STA_SHARD_STREAM): rank 0 writes the blocks as they arrive, through two small receive buffers -- chunks far
    # smaller than a block here, so that blocks span many chunks and the last chunk of each is short
    parts = []
    shard.stream_text(padded, local.numel(), parts.append, dst=0, sizes=sizes, chunk_bytes=97)
    if rank == 0:
        assert b"".join(parts) == bytes(whole.numpy().tobytes())
    else:
        assert not parts
    if rank == 0:
        assert sum(sizes) == whole.numel()
        for k in range(2):
            assert bytes(recv[k][:sum(sizes)].numpy().tobytes()) == bytes(whole.numpy().tobytes())
        q.put(bytes(whole.numpy().tobytes()))
    else:
        assert whole is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("contigs,wcols", [([1000, 37, 512], 128), ([90], 100), ([300, 300], 64)])
@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_reassemble_single_process_output(contigs, wcols, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, contigs, wcols, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = b"".join(fake_window_text(*w) for w in shard.plan_windows(contigs, wcols))
    assert got == want


def test_window_ownership_is_a_partition():
    wins = shard.plan_windows([1000, 1, 999, 4096], 256)
    for world in (1, 2, 3, 8):
        owned = [i for r in range(world) for i in shard.windows_of_rank(wins, r, world)]
        assert owned == list(range(len(wins)))          # every window exactly once, in output order
        sizes = [len(shard.windows_of_rank(wins, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert shard.halo_columns(151) == 302


def test_blocks_partition_the_columns_and_read_ranges_cover_them():
    import numpy as np
    rng = np.random.default_rng(5)
    pos = np.sort(rng.integers(0, 100000, 5000))
    for world in (1, 2, 3, 8):
        blocks = [shard.block_of(r, world, 100003) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == 100003
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        for b, e in blocks:
            lo, hi = shard.read_range(pos, b, e, halo=300)
            assert np.all(pos[lo:hi] >= b - 300) and np.all(pos[lo:hi] < e + 300)
            assert (lo == 0 or pos[lo - 1] < b - 300) and (hi == len(pos) or pos[hi] >= e + 300)
