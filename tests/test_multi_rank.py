"""N>1 path on CPU: world_size-2 gloo processes, each rendering its shard with the CPU build
of the kernel core, host gather to rank 0 -- must equal the single-process render bit for bit.
(On the GPU box the same sharding code drives Context.render; bench.py --gpus N.)"""
import os
import socket
import sys

import numpy as np
import pytest

import orc

W, H, SPP = 40, 40, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(orc.ROOT, "tests"))
    import importlib
    rt = orc.rt()
    sh = importlib.import_module("raytracing-1w_amd.sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = rt.Scene.reference(5, build_seed=1)

    def render_fn(tile, sample_offset, spp, out_sum):
        img, _ = orc.flat_render(sc, W, H, spp, tile=tile, sample_offset=sample_offset, out_sum=out_sum, chunk=SPP // world,
                                 threads=2)
        return img

    if mode == "frame":
        # what bench.py --gpus N does: ONE row-interleaved tile per rank (rt1w_render_params.strip_rows/strip_period), every
        # rank writes its strips into one shared host frame, nothing is stitched afterwards
        y0, rows, srows, period = sh.interleaved_tile(H, world, rank)
        fr = sh.SharedFrame(f"rt1w_test_{port}", W, H, rank == 0) if rank == 0 else None
        dist.barrier()
        if rank != 0:
            fr = sh.SharedFrame(f"rt1w_test_{port}", W, H, False)
        packed, _ = orc.flat_render(sc, W, H, SPP, tile=(0, y0, W, rows), strips=(srows, period), chunk=SPP // world, threads=2)
        pos = 0
        for (sy, n) in sh.row_strips(H, world, rank):
            fr.array[sy:sy + n] = packed[pos:pos + n]
            pos += n
        assert pos == rows
        dist.barrier()
        if rank == 0:
            q.put(fr.array.copy())
        dist.barrier()
        fr.close()
    elif mode == "rows":
        _, packed = sh.render_rows(render_fn, W, H, SPP, world, rank)
        parts = sh.gather_to_rank0(packed)
        if rank == 0:
            q.put(sh.stitch_rows(W, H, world, parts))
    else:
        off, n = sh.sample_range(SPP, world, rank)
        sums = render_fn((0, 0, W, H), off, n, True)
        parts = sh.gather_to_rank0(sums)
        if rank == 0:
            q.put(rt.resolve(sh.combine_sample_sums(parts), SPP))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rows", "samples", "frame"])
def test_two_rank_sharded_render_equals_single(rt, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = rt.Scene.reference(5, build_seed=1)
    full, _ = orc.flat_render(sc, W, H, SPP, chunk=SPP // 2)
    assert np.array_equal(got, full)


def test_row_strips_partition_every_row_once():
    import importlib
    orc.rt()
    sh = importlib.import_module("raytracing-1w_amd.sharding")
    for height in (1, 15, 16, 17, 600, 2160):
        for world in (1, 2, 3, 8):
            rows = sorted((y, n) for r in range(world) for (y, n) in sh.row_strips(height, world, r))
            covered = np.zeros(height, dtype=int)
            for y, n in rows:
                covered[y:y + n] += 1
            assert np.all(covered == 1)
            for r in range(world):                   # the one-launch form of the same partition
                y0, n_rows, srows, period = sh.interleaved_tile(height, world, r)
                mine = [y for (sy, n) in sh.row_strips(height, world, r) for y in range(sy, sy + n)]
                assert n_rows == len(mine)
                assert [y0 + (t // srows) * period + t % srows for t in range(n_rows)] == mine
    with pytest.raises(ValueError):
        sh.sample_range(10, 4, 0)
    assert [sh.sample_range(1000, 8, r) for r in (0, 7)] == [(0, 125), (875, 125)]
