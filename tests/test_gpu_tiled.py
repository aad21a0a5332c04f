"""The frame loop bench.py runs (ray_tracing_amd/multi_gpu.py TiledFrame): strip render -> gather ->
de-interleave -> pinned host frame, two frames in flight, everything ordered on its own non-default stream.
Every step uses a DIFFERENT seed, so a stale strip / frame (a stream-ordering bug) cannot look correct."""
import os
import socket

import numpy as np
import pytest
import torch

import ray_tracing_amd as rt
from rtlibs import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    r.set_tuning(poison_frame=True)
    r.set_skybox(rt.load_skybox()); r.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); r.set_camera()
    yield r
    r.close()


def test_pipelined_frames_world_1(gpu):
    from ray_tracing_amd.multi_gpu import TiledFrame
    W, H, spp, nb = 320, 180, 8, 4
    want = {s: gpu.render(W, H, spp, nb, seed=s) for s in (1, 2, 3, 4, 5)}
    t = TiledFrame(gpu, W, H, spp, nb, seed=1, rank=0, world=1, device=torch.device("cuda", 0))
    for s in (1, 2, 3):
        got = t.render_now(seed=s).numpy()
        assert (bits(got) == bits(want[s])).all(), s
    for s in (1, 2, 3, 4, 5):            # five frames back to back, two in flight
        t.step(seed=s)
    t.flush()
    assert (bits(t.host_frame.numpy()) == bits(want[5])).all()
    # the legacy null stream is a distinct, explicit choice (handle 0 -> RT_STREAM_LEGACY), not the context's stream
    strip = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")     # filled on the null stream ...
    gpu.render_device(gpu.params(W, H, spp, nb, seed=2), strip.data_ptr(), 0)   # ... rendered on the null stream: ordered
    assert (bits(strip.cpu().numpy()) == bits(want[2])).all()                  # .cpu() syncs the null stream


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _join_all(procs, seconds):
    """Wait for the rank processes; one that is still running after `seconds` is killed (it must not outlive the test or hold
    the suite up -- its own faulthandler timer has dumped its stacks by then) and the test fails."""
    import time
    deadline = time.time() + seconds
    for p in procs:
        p.join(max(1.0, deadline - time.time()))
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.kill(); p.join(10)
    assert not stuck, f"{len(stuck)} rank process(es) did not finish within {seconds} s"
    for p in procs:
        assert p.exitcode == 0, p.exitcode


def _nccl_worker(rank, world, port, q, backend="nccl", share_gpu=False, size=(320, 180, 8)):
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)      # a collective that never completes ends the worker with its stacks on stderr
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from ray_tracing_amd.multi_gpu import TiledFrame
    device = 0 if share_gpu else rank
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    g = rt.Renderer(device)
    g.set_tuning(poison_frame=True)
    g.set_skybox(rt.load_skybox()); g.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); g.set_camera()
    (W, H, spp), nb = size, 4
    t = TiledFrame(g, W, H, spp, nb, rank=rank, world=world, device=dev)
    ok = True
    for s in (1, 2, 3):
        got = t.render_now(seed=s)
        if rank == 0:
            ok = ok and bool((bits(got.numpy()) == bits(g.render(W, H, spp, nb, seed=s))).all())
    # pipelined: up to three frames in flight, three strip buffers and two frame buffers in rotation -- runs of 4, 5, 6 and 9
    # frames end on every combination of them; a different seed per frame, so that a stale or overwritten buffer shows
    seed = 4
    for run in (4, 5, 6, 9):
        for _ in range(run):
            t.step(seed=seed); seed += 1
        t.flush()
        if rank == 0:
            ok = ok and bool((bits(t.host_frame.numpy()) == bits(g.render(W, H, spp, nb, seed=seed - 1))).all())
    if rank == 0:
        q.put(ok)
    dist.barrier()
    g.close()
    dist.destroy_process_group()


def test_consecutive_launches_overlap_on_the_two_streams(gpu):
    """rt_stream(ctx, 0 / 1): launches alternate between two of the context's streams and rotate through its scratch sets, so
    consecutive ones are on the GPU together.  Every launch has its own seed and destination; each frame must equal the
    one rendered alone.  Large enough (a few hundred microseconds a launch) for the launches to really overlap."""
    W, H, spp, nb = 1920, 1080, 16, 4
    seeds = [11, 12, 13, 14, 15, 16, 17]
    streams = [gpu.stream(0), gpu.stream(1)]
    assert streams[0] and streams[1] and streams[0] != streams[1]
    want = {}
    for s in seeds[:3] + seeds[-1:]:
        alone = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        gpu.render_device(gpu.params(W, H, spp, nb, seed=s), alone.data_ptr()); gpu.synchronize()
        want[s] = alone.cpu().numpy()
    out = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0") for _ in seeds]
    torch.cuda.synchronize()
    for k, s in enumerate(seeds):                       # nothing waits in between: up to two launches run together
        gpu.render_device(gpu.params(W, H, spp, nb, seed=s), out[k].data_ptr(), streams[k & 1])
    gpu.synchronize(); torch.cuda.synchronize()
    for k, s in enumerate(seeds):
        if s in want:
            assert (bits(out[k].cpu().numpy()) == bits(want[s])).all(), s
    # strips of different ranks on the two streams, as a host that renders for two ranks would issue them
    rows = rt.strip_rows(H, 8, 8)
    strips = [torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(8)]
    torch.cuda.synchronize()
    for r in range(8):
        gpu.render_device(gpu.params(W, H, spp, nb, seed=11, row_block=8, rank=r, world=8), strips[r].data_ptr(), streams[r & 1])
    gpu.synchronize(); torch.cuda.synchronize()
    frame = np.empty((H, W, 3), dtype=np.float32)
    from ray_tracing_amd.multi_gpu import owned_rows
    for r in range(8):
        g = owned_rows(H, 8, r, 8)
        frame[g[g >= 0]] = strips[r].cpu().numpy()[g >= 0]
    assert (bits(frame) == bits(want[11])).all()


@pytest.mark.parametrize("world,size", [(2, (320, 180, 8)), (3, (320, 180, 8)),
                                        (5, (160, 1080, 4))])     # 1080 rows over five ranks: 135 blocks, strips of 216 rows (the pool allows six
                                                                    # processes on a GPU: five ranks + this test process; world 8 runs over gloo on CPU tensors,
                                                                    # tests/test_distributed_gloo.py)
def test_pipelined_frames_ranks_sharing_one_gpu_over_gloo(world, size):
    """The N-rank frame loop of bench.py on the ONE GPU this pool's boxes have: every rank is its own process with
    its own context, streams and strips on GPU 0, the strips travel through gloo (which moves device tensors) instead
    of RCCL (which refuses two ranks on one device).  Everything but the transport is what an N-GPU run executes:
    per-rank interleaved strips, asynchronous gather, double buffering, de-interleave, host copy -- with a different
    seed per step, so a stale strip or a mis-ordered stream cannot look correct."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q, "gloo", True, size)) for r in range(world)]
    for p in procs:
        p.start()
    _join_all(procs, 300)
    assert q.get(timeout=5) is True


def _one_rank_rccl_worker(port, q):
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)      # a collective that never completes ends the worker with its stacks on stderr
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    import torch.distributed as dist
    from ray_tracing_amd.multi_gpu import TiledFrame
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(rt.load_skybox()); g.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); g.set_camera()
    W, H, spp, nb = 640, 360, 8, 4
    t = TiledFrame(g, W, H, spp, nb, rank=0, world=1, device=dev, force_collective=True)
    ok = t.primitive == "gather" and t.post is not None
    seed = 1
    for run in (1, 2, 3, 4, 7, 12):                 # frames in flight: asynchronous RCCL gathers, nothing blocks the host
        for _ in range(run):
            t.step(seed=seed); seed += 1
        t.flush()
        ok = ok and bool((bits(t.host_frame.numpy()) == bits(g.render(W, H, spp, nb, seed=seed - 1))).all())
    q.put(ok)
    g.close()
    dist.destroy_process_group()


def test_pipelined_frames_one_rank_over_rccl():
    """The N > 1 frame loop with the REAL backend on the one GPU of a test box: a one-rank RCCL group runs the gather
    (asynchronous, on RCCL's own stream), the wait for it on the post stream, the de-interleave and the rotation of
    three strip and two frame buffers exactly as N ranks would; a different seed per frame."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_rccl_worker, args=(_free_port(), q))
    p.start()
    _join_all([p], 240)
    assert q.get(timeout=5) is True


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL gather between ranks)")
def test_pipelined_frames_two_ranks_over_rccl():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    _join_all(procs, 300)
    assert q.get(timeout=5) is True


def test_native_multi_gpu_entry_world_1(gpu):
    """rt_multi_*: the single-process, native-RCCL entry of the C ABI.  With one device it must equal rt_render()
    bit for bit and never load RCCL; duplicate / out-of-range device lists are argument errors, not crashes."""
    import ctypes as C
    W, H, spp, nb = 200, 77, 5, 6
    want = gpu.render(W, H, spp, nb, seed=4)
    m = rt.MultiRenderer([0])
    assert m.size() == 1
    m.set_tuning(poison_frame=1)
    m.set_skybox(rt.load_skybox()); m.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); m.set_camera()
    assert (bits(m.render(W, H, spp, nb, seed=4)) == bits(want)).all()
    m.compile_scene()
    assert (bits(m.render(W, H, spp, nb, seed=4)) == bits(want)).all()
    # the multi-device path itself -- RCCL loaded with dlopen, ncclCommInitAll, one grouped ncclGather, de-interleave,
    # copy to the host -- on a one-rank communicator (all a 1-GPU box can run); row_block 4: 77 rows = 19 blocks + 1 row
    m.set_tuning(poison_frame=1, force_collective=1)
    for rb in (8, 4):
        assert (bits(m.render(W, H, spp, nb, seed=4, row_block=rb)) == bits(want)).all()
    m.close()
    L = rt.lib()
    h = C.c_void_p()
    assert L.rt_multi_create(C.byref(h), (C.c_int * 2)(0, 0), 2) == -1 and b"twice" in L.rt_last_error()
    assert L.rt_multi_create(C.byref(h), (C.c_int * 1)(99), 1) == -1
    assert L.rt_multi_create(C.byref(h), None, 0) == -1


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (ncclGather between them)")
def test_native_multi_gpu_entry_two_devices(gpu):
    W, H, spp, nb = 320, 181, 8, 4
    want = gpu.render(W, H, spp, nb, seed=9)
    m = rt.MultiRenderer([0, 1])
    m.set_tuning(poison_frame=1)
    m.set_skybox(rt.load_skybox()); m.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); m.set_camera()
    for _ in range(2):
        assert (bits(m.render(W, H, spp, nb, seed=9)) == bits(want)).all()
    m.close()


def _check_per_rank(line, ranks):
    """The N > 1 bench line says where every rank's step went (round 6): one entry per rank, five disjoint shares of the rank's timed
    window -- strip render, de-interleave, copy, waiting for the gather, idle -- that sum to its step, the rank with the least slack,
    and what bounds the step there."""
    pr = line["per_rank"]
    assert len(pr) == ranks and [r["rank"] for r in pr] == list(range(ranks))
    for r in pr:
        assert r["frames"] == line["steps"] - 1, r                      # one interval fewer than frames recorded in the timed region
        phases = [r[k] for k in ("idle_ms", "render_ms", "gather_ms", "deinterleave_ms", "copy_ms")]
        assert all(v >= 0 for v in phases) and r["step_ms"] > 0
        assert abs(sum(phases) - r["step_ms"]) <= 0.10 * r["step_ms"], r
        if r["rank"] > 0:                                               # only the root assembles the frame and copies it out
            assert r["deinterleave_ms"] == 0 and r["copy_ms"] == 0
    assert pr[0]["copy_ms"] > 0
    # (a run of four steps with six frames in flight may have rendered all of a rank's strips before its first frame ended)
    assert sum(r["render_ms"] for r in pr) > 0 or line["steps"] < 8
    # (every rank's step is the same frame rate seen from its own end of the pipeline -- over the two or three intervals of these short
    # runs, with ranks that share one GPU over gloo, the ends lie up to a frame apart: not asserted)
    assert line["critical_rank"] in range(ranks) and line["step_bound"] in ("render", "gather", "copy", "host")


def test_bench_self_launch_four_ranks_sharing_the_gpu():
    """`python bench.py --gpus 4 --share-gpu`: the self-launched N-rank bench (a child torch.distributed.run, one process per
    rank) end to end on the one GPU of the box -- strips over gloo -- with its built-in check of the last frame."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--share-gpu", "--steps", "3", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"] + (["--leave-early"] if os.environ.get("RT_TEST_LEAVE_EARLY") else []),   # (scripts/repro_verify_race.py)
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-1500:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["verified"] is True and line["verification"]["equals_blocking_rt_render"] is True
    assert "SHARING ONE GPU" in line["config"]["partition"]
    _check_per_rank(line, 4)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (ncclGather between them)")
def test_native_multi_gpu_frames_in_flight_two_devices(gpu):
    """rt_multi_frame_submit / rt_multi_frame_wait over two real devices: three strip buffers per device, grouped ncclGather
    on the collective streams, a different seed per frame."""
    W, H, spp, nb = 320, 181, 8, 4
    seeds = list(range(30, 41))
    want = {s: gpu.render(W, H, spp, nb, seed=s) for s in seeds}
    m = rt.MultiRenderer([0, 1])
    m.set_skybox(rt.load_skybox()); m.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); m.set_camera()
    from ray_tracing_amd.frames import FrameLoop
    for depth in (2, 3, 4):
        loop = FrameLoop(m, W, H, spp, nb, depth=depth)
        got = []
        loop.run(seeds, on_frame=lambda k, a: got.append(a.copy()))
        for k, s in enumerate(seeds):
            assert (bits(got[k]) == bits(want[s])).all(), (depth, k)
        loop.close()
    m.progressive_begin(W, H, init_scale=8, max_bounces=4, seed=1)
    gpu.progressive_begin(W, H, init_scale=8, max_bounces=4, seed=1)
    for _ in range(6):
        m.progressive_pass(); gpu.progressive_pass()
    assert (bits(m.progressive_resolve()) == bits(gpu.progressive_resolve())).all()
    m.close()


def _bench_line(args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-1500:])
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_bench_native_multi_eight_contexts_on_the_one_gpu():
    """`bench.py --gpus 8 --native-multi --one-device`: the ONE-process host of the N-GPU path (rt_multi_frame_submit /
    rt_multi_frame_wait) with eight contexts, strips of 136 rows and the frame queue three deep, on the one GPU of the box;
    its built-in check compares the last frame with a blocking render and oracle rows."""
    line = _bench_line(["--gpus", "8", "--native-multi", "--one-device", "--steps", "4", "--warmup", "2"])
    assert line["n_gpus"] == 8 and line["verified"] is True
    assert line["collective"]["contexts"] == 8 and line["collective"]["ranks_seen"] == 0      # no communicator on one device
    assert "one host process" in line["config"]["frame_loop"]
    _check_per_rank(line, 8)


def test_bench_native_multi_one_rank_communicator():
    """`bench.py --gpus 1 --native-multi --force-collective`: the same host over RCCL itself; the line carries what the
    communicator reports (ncclCommCount, ncclCommCuDevice, ncclGetVersion), not what the script asked for."""
    line = _bench_line(["--gpus", "1", "--native-multi", "--force-collective", "--steps", "4", "--warmup", "2"])
    assert line["verified"] is True
    c = line["collective"]
    assert c["ranks_seen"] == 1 and c["devices_seen"] == [0] and c["rccl_version"] > 20000
    _check_per_rank(line, 1)


def test_bench_one_rank_process_group_over_rccl():
    """`bench.py --force-collective --torch-loop`: the one-process-per-GPU host (multi_gpu.TiledFrame over torch.distributed, backend
    nccl = RCCL) on a one-rank process group -- gather, de-interleave, copy stream -- with the phases of its step in the line."""
    line = _bench_line(["--gpus", "1", "--force-collective", "--torch-loop", "--steps", "4", "--warmup", "2"])
    assert line["verified"] is True and line["collective"]["backend"] == "nccl"
    _check_per_rank(line, 1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (ncclGather between them)")
def test_bench_native_multi_two_devices():
    line = _bench_line(["--gpus", "2", "--native-multi", "--steps", "4", "--warmup", "2"])
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert line["collective"]["ranks_seen"] == 2 and sorted(line["collective"]["devices_seen"]) == [0, 1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL gather between ranks)")
def test_bench_two_ranks_over_rccl():
    line = _bench_line(["--gpus", "2", "--steps", "4", "--warmup", "2"])
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert line["collective"]["ranks_seen"] == 2 and sorted(line["collective"]["devices_seen"]) == [0, 1] and line["collective"]["backend"] == "nccl"
