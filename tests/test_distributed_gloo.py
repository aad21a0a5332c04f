"""world_size-2 (and 3) CPU test of the N>1 path: interleaved row-block strips -> one gather ->
de-interleave, over torch.distributed/gloo.  The per-rank strips are produced here by the CPU oracle
(the checker standing in for the GPU kernel, which cannot run on this box); what is under test is the
product's partition + exchange code in ray_tracing_amd/multi_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, W, H, rb, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlibs import DATA_DIR, Oracle, synthetic_skybox
    from ray_tracing_amd import multi_gpu as mg
    o = Oracle(); o.set_skybox(synthetic_skybox(32, seed=7)); o.load_scene(os.path.join(DATA_DIR, "scene_0.txt"))
    assert mg.collective_for() == "gather"
    ok = True
    out = torch.empty((world, mg.strip_rows(H, rb, world), W, 3), dtype=torch.float32) if rank == 0 else None
    # two consecutive frames with different seeds through the same buffers (a stale strip would show), the
    # first with the blocking call and rank r holding strip r, the second with the asynchronous call and the strips
    # handed out rotated by one (the root has the last, never the longest), as TiledFrame does both
    for frame_no, seed in enumerate((3, 4)):
        first = frame_no
        rows = mg.owned_rows(H, rb, mg.strip_of_rank(rank, world) if first else rank, world)
        if first and rank == 0:
            assert (rows >= 0).sum() == min((mg.owned_rows(H, rb, s, world) >= 0).sum() for s in range(world))
        strip = torch.zeros((len(rows), W, 3), dtype=torch.float32)
        for lr, j in enumerate(rows):
            if j >= 0:
                strip[lr] = torch.from_numpy(o.render_counter(W, H, 2, 4, seed=seed, rows=(int(j), int(j) + 1), threads=1)[j])
        got, work = mg.gather_strips(strip, rank, world, dst=0, out=out, async_op=frame_no == 1)
        if work is not None:
            work.wait()
        if rank == 0:
            frame = mg.assemble(got, H, rb, world, first=first)
            full = o.render_counter(W, H, 2, 4, seed=seed, threads=2)
            ok = ok and bool((frame.numpy().view(np.uint32) == full.view(np.uint32)).all())
        else:
            assert got is None
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,rb,W", [(2, 36, 8, 48), (3, 29, 4, 48),
                                         (8, 1080, 8, 12)])      # the 8-GPU partition of a 1080-row frame: 135 blocks, strips of 136 rows
def test_gather_and_deinterleave_over_gloo(world, H, rb, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, rb, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
