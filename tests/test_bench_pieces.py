"""bench.py steps windows larger than a workload's parity-test size (`bench_cols`): the synthetic input is then assembled from pieces of that
size (piece k: seed + k), and a rank of a sharded run builds only its own pieces and the adjacent piece of each neighbour's window.  What a rank
stages must be exactly the reads the whole input holds for its block (+ halo), whichever pieces it built."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def test_rank_pieces_equal_the_whole_input():
    import bench
    from synth import synth_ref
    from samtools_amd import shard
    bench.WORKLOADS["_pieces"] = dict(bench._wl("mpileup", 30, 1 << 15, ["mpileup"], baq=True), bench_cols=1 << 17)
    try:
        W, world = 1 << 17, 3
        ref = synth_ref(world * W, seed=1)
        full = bench.make_reads("_pieces", ref, W, 42)
        one = bench.make_reads("_pieces", synth_ref(W, seed=1), W, 42)
        assert one["n"] > 25000 and full["n"] > 3 * 25000
        halo = shard.halo_columns(150 + 3)
        for rank in range(world):
            rd = bench.make_reads("_pieces", ref, W, 42, chunks=(rank - 1, rank, rank + 1))
            assert rd["n"] < full["n"]                                   # (it did not build everything)
            b, e = shard.block_of(rank, world, world * W)
            lo, hi = shard.read_range(rd["_abs_pos"], b, e, halo=halo)
            lo2, hi2 = shard.read_range(full["_abs_pos"], b, e, halo=halo)
            q1 = bench.slice_reads(np, rd, lo, hi, max(0, b - halo))
            q2 = bench.slice_reads(np, full, lo2, hi2, max(0, b - halo))
            assert hi - lo == hi2 - lo2 > 20000
            for k in q1:
                if isinstance(q1[k], np.ndarray):
                    assert np.array_equal(q1[k], q2[k]), (rank, k)
    finally:
        del bench.WORKLOADS["_pieces"]
