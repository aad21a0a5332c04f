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


def test_phase_attribution_of_a_pipelined_frame_loop():
    """ray_tracing_amd.attribute_phases / judge_phases (the Python twin of rt_multi_profile_collect): synthetic timelines, no GPU."""
    import ray_tracing_amd as rt
    # a serial loop: every frame starts when the previous one is in host memory -- 1 idle, 4 render, 2 gather, 1 de-interleave, 2 copy
    serial = [(10 * k + 1, 10 * k + 5, 10 * k + 7, 10 * k + 8, 10 * k + 10) for k in range(6)]
    a = rt.attribute_phases(serial)
    assert a["frames"] == 5 and a["step_ms"] == 10
    assert [a[k] for k in rt.PHASES] == [1, 4, 2, 1, 2]
    # a pipelined loop: renders back to back (4 ms each); gather, de-interleave and copy of frame k run beside the render of frame
    # k + 1: the device renders ALL the time, and that is what the step is made of
    pipe = [(4 * k, 4 * k + 4, 4 * k + 5, 4 * k + 5.5, 4 * k + 6.5) for k in range(8)]
    b = rt.attribute_phases(pipe)
    assert b["step_ms"] == 4 and abs(sum(b[k] for k in rt.PHASES) - 4) < 1e-9
    assert abs(b["render_ms"] - 4 * 6 / 7 - 2.5 / 7) < 0.3 and b["idle_ms"] == 0        # (the last frame's tail is not covered by a next render)
    # a deep pipeline whose renders take 1 of 4 ms: the rest of the step the rendered strips wait for the gather
    waiting = [(4 * k, 4 * k + 1, 4 * k + 12, 4 * k + 12, 4 * k + 12) for k in range(40)]
    c = rt.attribute_phases(waiting)
    assert c["step_ms"] == 4 and abs(c["render_ms"] - 1) < 0.2 and abs(c["gather_ms"] - 3) < 0.2
    # the critical rank is the one with the least slack; the bound is its largest share
    root = dict(a, render_ms=1, gather_ms=6, idle_ms=0, deinterleave_ms=1, copy_ms=2)
    fast = dict(a, render_ms=2, gather_ms=8, idle_ms=0, deinterleave_ms=0, copy_ms=0)
    slow = dict(a, render_ms=9, gather_ms=1, idle_ms=0, deinterleave_ms=0, copy_ms=0)
    assert rt.judge_phases([root, fast, slow]) == {"critical_rank": 2, "step_bound": "render"}
    starved = dict(slow, render_ms=2, idle_ms=8, gather_ms=0)
    assert rt.judge_phases([root, fast, starved, slow]) == {"critical_rank": 3, "step_bound": "render"}
    assert rt.judge_phases([dict(root, gather_ms=1, copy_ms=7), fast]) == {"critical_rank": 0, "step_bound": "copy"}
    assert rt.judge_phases([dict(root, gather_ms=7, render_ms=1), dict(fast, gather_ms=9, render_ms=1)]) == {"critical_rank": 0, "step_bound": "gather"}
    assert rt.judge_phases([]) == {"critical_rank": None, "step_bound": None}
