"""Frames in flight behind the C ABI (rt_frame_submit / rt_frame_wait, rt_multi_frame_*): the reference's workers keep
rendering while its main thread presents (main.c:354-408 vs 450-482).  Every pipelined frame must be the frame the
blocking rt_render() produces for the same parameters, bit for bit, whatever the number of slots in rotation, the kind
of host memory, or the path (one context; the N-GPU path over RCCL itself on a one-rank communicator)."""
import os
import subprocess
import time

import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import bits

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_cli")


@pytest.fixture(scope="module")
def sky():
    return rt.load_skybox()


def _reference_frames(scene, sky, W, H, spp, nb, seeds, compiled=False):
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_scene(scene); g.set_skybox(sky); g.set_camera()
    if compiled:
        g.compile_scene()
    out = {s: g.render(W, H, spp, nb, seed=s) for s in set(seeds)}
    g.close()
    return out


def _run_pipeline(q, W, H, spp, nb, seeds, slots, pinned=True, row_block=8):
    """Submit frame k+slots-1 ... wait frame k: `slots` frames in flight.  Returns the frames in submission order."""
    bufs = [rt.HostFrame(W, H) if pinned else np.empty((H, W, 3), np.float32) for _ in range(slots)]
    view = [b.array if pinned else b for b in bufs]
    got = []
    for k, seed in enumerate(seeds):
        s = k % slots
        if k >= slots:
            assert q.frame_wait(s)
            got.append(view[s].copy())
            view[s][:] = np.nan                # a frame that is not delivered cannot pass as the one before it
        q.frame_submit(rt.Renderer.params(W, H, spp, nb, seed=seed, row_block=row_block), s, bufs[s])
    n = len(seeds)
    for k in range(max(0, n - slots), n):
        assert q.frame_wait(k % slots)
        got.append(view[k % slots].copy())
    for b in bufs:
        if pinned:
            b.free()
    return got


@pytest.mark.parametrize("slots,pinned", [(2, True), (3, True), (4, True), (2, False)])
def test_pipelined_frames_equal_blocking_renders(sky, scene_paths, slots, pinned):
    W, H, spp, nb = 320, 180, 8, 4
    seeds = [11, 12, 13, 11, 14, 15, 16, 12, 17]
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    got = _run_pipeline(g, W, H, spp, nb, seeds, slots, pinned)
    for k, seed in enumerate(seeds):
        assert (bits(got[k]) == bits(want[seed])).all(), (k, seed)
    # a blocking render in between shifts which scratch set the next frame gets: still the same frames
    assert (bits(g.render(W, H, spp, nb, seed=13)) == bits(want[13])).all()
    got = _run_pipeline(g, W, H, spp, nb, seeds[:5], slots, pinned)
    for k, seed in enumerate(seeds[:5]):
        assert (bits(got[k]) == bits(want[seed])).all(), (k, seed)
    g.close()


def test_frame_slot_rules(sky, scene_paths):
    L = rt.lib()
    g = rt.Renderer(0)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    W, H = 64, 36
    buf = rt.HostFrame(W, H)
    p = g.params(W, H, 2, 2, seed=1)
    import ctypes as C
    assert L.rt_frame_wait(g._ctx, 0) == -3 and b"nothing was submitted" in L.rt_last_error()
    assert L.rt_frame_poll(g._ctx, 1) == -3
    assert L.rt_frame_submit(g._ctx, C.byref(p), rt.FRAME_SLOTS, C.c_void_p(buf.ptr)) == -1
    assert L.rt_frame_submit(g._ctx, C.byref(p), 0, None) == -1
    g.frame_submit(p, 0, buf)
    assert L.rt_frame_submit(g._ctx, C.byref(p), 0, C.c_void_p(buf.ptr)) == -3 and b"not been waited for" in L.rt_last_error()
    t0 = time.time()
    while g.frame_poll(0) is None:             # never blocks; becomes True once the frame is there
        assert time.time() - t0 < 30
    assert L.rt_frame_wait(g._ctx, 0) == -3    # the poll that saw the frame released the slot
    want = g.render(W, H, 2, 2, seed=1)
    assert (bits(buf.array) == bits(want)).all()
    ps = g.params(W, H, 2, 2, seed=1, world=2)
    assert L.rt_frame_submit(g._ctx, C.byref(ps), 0, C.c_void_p(buf.ptr)) == -1     # strips: rt_multi_frame_submit
    buf.free(); g.close()


@pytest.mark.parametrize("compiled", [False, True])
def test_c1_pipelined_frames_equal_blocking_render(sky, scene_paths, compiled):
    """The benchmark's own loop at its own workload (C1), a different seed per frame."""
    W, H, spp, nb = 1920, 1080, 64, 4
    seeds = [0, 1, 2, 0, 3]
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds, compiled)
    g = rt.Renderer(0)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    if compiled:
        g.compile_scene()
    got = _run_pipeline(g, W, H, spp, nb, seeds, 2)
    for k, seed in enumerate(seeds):
        assert (bits(got[k]) == bits(want[seed])).all(), (k, seed)
    g.close()


@pytest.mark.parametrize("slots", [2, 3, 4])
def test_multi_frames_over_a_one_rank_rccl_communicator(sky, scene_paths, slots):
    """rt_multi_frame_submit's N > 1 path -- three strip buffers per device, the grouped ncclGather on a stream of its own,
    de-interleave, copy stream -- on the one GPU of the box (rt_tuning.force_collective: a one-rank communicator)."""
    W, H, spp, nb = 200, 77, 4, 4              # 77 rows: the last row block is partial
    seeds = list(range(20, 31))
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    m = rt.MultiRenderer([0])
    m.set_tuning(force_collective=1, poison_frame=1)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    got = _run_pipeline(m, W, H, spp, nb, seeds, slots, row_block=4)
    for k, seed in enumerate(seeds):
        assert (bits(got[k]) == bits(want[seed])).all(), (k, seed)
    # the blocking call is submit + wait on the same path
    assert (bits(m.render(W, H, spp, nb, seed=25, row_block=4)) == bits(want[25])).all()
    # a larger frame afterwards: every buffer is replaced while nothing is in flight
    W2, H2 = 333, 130
    want2 = _reference_frames(scene_paths[0], sky, W2, H2, spp, nb, [1, 2, 3])
    got = _run_pipeline(m, W2, H2, spp, nb, [1, 2, 3], slots)
    for k, seed in enumerate([1, 2, 3]):
        assert (bits(got[k]) == bits(want2[seed])).all(), (k, seed)
    m.close()


def test_multi_single_device_is_the_plain_frame_queue(sky, scene_paths):
    W, H, spp, nb = 160, 90, 4, 4
    seeds = [5, 6, 7, 8]
    want = _reference_frames(scene_paths[1], sky, W, H, spp, nb, seeds)
    m = rt.MultiRenderer([0])
    m.set_scene(scene_paths[1]); m.set_skybox(sky); m.set_camera()
    got = _run_pipeline(m, W, H, spp, nb, seeds, 2)
    for k, seed in enumerate(seeds):
        assert (bits(got[k]) == bits(want[seed])).all(), (k, seed)
    m.close()


@pytest.mark.parametrize("multi", [False, True])
def test_cancel_reaches_a_frame_in_flight(sky, scene_paths, multi):
    """rt_cancel() while a submitted frame renders: its wait reports the cut, the frame after it is whole.  (Compiled kernel:
    the one that leaves no room on the GPU for anything a request might need to run there.)"""
    W, H, spp, nb = 1920, 1080, 1024, 8        # ~100 ms of GPU work
    if multi:
        q = rt.MultiRenderer([0])
        q.set_tuning(force_collective=1)
        ctx = q.context(0)
    else:
        q = ctx = rt.Renderer(0)
    q.set_scene(scene_paths[0]); q.set_skybox(sky); q.set_camera()
    q.compile_scene()
    a, b = rt.HostFrame(W, H), rt.HostFrame(320, 180)
    q.frame_submit(rt.Renderer.params(W, H, spp, nb, seed=1), 0, a)
    time.sleep(0.02)
    ctx.cancel()
    t0 = time.time()
    assert q.frame_wait(0) is False            # RT_CANCELLED
    assert time.time() - t0 < 0.02
    q.frame_submit(rt.Renderer.params(320, 180, 4, 4, seed=2), 1, b)
    assert q.frame_wait(1) is True
    want = _reference_frames(scene_paths[0], sky, 320, 180, 4, 4, [2])[2]
    assert (bits(b.array) == bits(want)).all()
    a.free(); b.free(); q.close()


def test_cancelled_progressive_pass_is_not_counted(sky, scene_paths, oracle):
    """rt_cancel() that hits a progressive pass WITHOUT rt_progressive_invalidate(): the pass is not published (main.c:382)
    and the count the resolve divides by must not include it either -- the weight sum lives on the device beside the
    accumulation buffer (round-2 advisor finding: the frame came out darkened).  With init_scale 1 the published passes
    are the samples of a counter-mode frame, so the resolve after k whole passes + one cut pass is the oracle's k-spp
    frame."""
    import torch
    W, H, nb, seed = 1280, 720, 10, 3
    g = rt.Renderer(0)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    g.progressive_begin(W, H, init_scale=1, max_bounces=nb, seed=seed)
    g.progressive_pass()
    g.synchronize()
    # A deterministic cancel point: ~90 ms of other work of this context sits in front of the second pass on its stream, so the
    # pass is still QUEUED when the request is made a few microseconds later -- and a queued launch that a request covers stops
    # at its first pixel fetch (rt_hip.h).  (The long launch is covered too and stops within a millisecond; nobody looks at it.)
    scratch = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    g.render_device(g.params(1920, 1080, 1024, 8, seed=9), scratch.data_ptr())
    g.progressive_pass()
    g.cancel()
    assert g.was_cancelled()                    # the most recent launch -- the second pass -- was cut short
    st = g.progressive_state()
    assert st["passes"] == 2 and st["count"] == 1.0      # two enqueued, one published
    frame = g.progressive_resolve()
    oracle.set_skybox(sky); oracle.load_scene(scene_paths[0]); oracle.set_camera()
    rows = list(range(0, H, 97))
    want = oracle.render_counter_rows(W, H, 1, nb, rows, seed=seed, threads=min(os.cpu_count() or 1, 64))
    for r, v in want.items():
        assert (bits(frame[r]) == bits(v)).all(), r
    g.progressive_pass()                                  # the ladder goes on: the next pass is published again
    assert g.progressive_state()["count"] == 2.0
    g.close()


def test_cli_frame_loop(tmp_path, scene_paths):
    """rt_cli --frames K: the plain-C presenter loop (two frames in flight); its last frame is the frame of seed + K - 1."""
    import json
    W, H, spp, nb, seed, K = 160, 90, 4, 4, 40, 5
    for extra in ([], ["--force-collective"], ["--compile"]):
        out = tmp_path / "last.ppm"
        cmd = [CLI, "--scene", scene_paths[0], "--skybox", os.path.join(rt.DATA_DIR, "skybox"), "--width", str(W), "--height", str(H),
               "--spp", str(spp), "--bounces", str(nb), "--seed", str(seed), "--frames", str(K), "--out", str(out)] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["frames"] == K and line["msamples_per_s"] > 0
        raw = out.read_bytes()
        head = f"P6\n{W} {H}\n255\n".encode()
        img = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
        want = _reference_frames(scene_paths[0], rt.load_skybox(), W, H, spp, nb, [seed + K - 1])[seed + K - 1]
        assert (img == (want * np.float32(255)).astype(np.uint8)[::-1]).all(), extra


def _read_device_frame(ptr, event, W, H):
    """What an on-GPU consumer does: order itself behind the event, then read the device frame (here: copy it out)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipEventSynchronize(C.c_void_p(event)) == 0
    out = np.empty((H, W, 3), np.float32)
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0      # hipMemcpyDeviceToHost
    return out


@pytest.mark.parametrize("group", [False, True])
def test_device_resident_frames(sky, scene_paths, group):
    """rt_frame_submit_device / rt_multi_frame_submit_device: the frame stays in device memory, the caller gets the buffer and
    an event behind it (the reference's presenter re-uploads the frame at once, gpu_and_windowing.c:371-376: a consumer on the
    GPU needs no copy).  Every frame, read back by the test itself, is the frame rt_frame_submit delivers; slots rotate."""
    W, H, spp, nb = 320, 181, 8, 4
    seeds = [21, 22, 23, 24, 25, 26, 27]
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    if group:
        q = rt.MultiRenderer([0]); q.set_tuning(force_collective=1)
    else:
        q = rt.Renderer(0)
    q.set_scene(scene_paths[0]); q.set_skybox(sky); q.set_camera()
    slots, pending = 3, {}
    for k, s in enumerate(seeds):
        j = k % slots
        if k >= slots:                 # the consumer of the frame three back is done with the buffer: release the slot
            ptr, ev, seed_ = pending.pop(j)
            got = _read_device_frame(ptr, ev, W, H)
            assert q.frame_wait(j)
            assert (bits(got) == bits(want[seed_])).all(), (k, seed_)
        ptr, ev = q.frame_submit_device(rt.Renderer.params(W, H, spp, nb, seed=s), j)
        assert ptr and ev
        pending[j] = (ptr, ev, s)
    for j, (ptr, ev, seed_) in sorted(pending.items()):
        got = _read_device_frame(ptr, ev, W, H)
        assert q.frame_wait(j)
        assert (bits(got) == bits(want[seed_])).all(), seed_
    q.close()


@pytest.mark.parametrize("n", [2, 5, 8])
def test_group_of_contexts_on_one_device(sky, scene_paths, n):
    """rt_multi_create_on_one_device: the n-device code of rt_multi -- n contexts, n strips per frame on their own streams, three
    strip buffers each, the rotated hand-out, the de-interleave, frames in flight, the ladder -- on the ONE GPU of the box (the
    gather is the n device copies it amounts to there: RCCL refuses two ranks on one device).  Bit-identical to rt_render()."""
    from ray_tracing_amd.frames import FrameLoop
    W, H, spp, nb = 320, 181, 8, 4
    seeds = list(range(30, 41))
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    m = rt.MultiRenderer([0], on_one_device=n)
    assert m.size() == n and m.collective_info()["ranks"] == 0
    m.set_tuning(poison_frame=1)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    assert (bits(m.render(W, H, spp, nb, seed=30)) == bits(want[30])).all()
    m.compile_scene()
    for depth in (2, 3, 4):
        loop = FrameLoop(m, W, H, spp, nb, depth=depth)
        got = []
        loop.run(seeds, on_frame=lambda k, a: got.append(a.copy()))
        for k, s in enumerate(seeds):
            assert (bits(got[k]) == bits(want[s])).all(), (n, depth, k)
        loop.close()
    g = rt.Renderer(0)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    m.progressive_begin(W, H, init_scale=8, max_bounces=4, seed=1)
    g.progressive_begin(W, H, init_scale=8, max_bounces=4, seed=1)
    for _ in range(6):
        m.progressive_pass(); g.progressive_pass()
    assert (bits(m.progressive_resolve()) == bits(g.progressive_resolve())).all()
    g.close(); m.close()


def test_group_on_one_device_at_the_benchmark_size(sky, scene_paths):
    """C1 through eight contexts on the one GPU (strips of 136 rows, the last one a row block short), frames in flight."""
    from ray_tracing_amd.frames import FrameLoop
    W, H, spp, nb = 1920, 1080, 64, 4
    seeds = [3, 4, 5, 6, 7]
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    m = rt.MultiRenderer([0], on_one_device=8)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera(); m.compile_scene()
    loop = FrameLoop(m, W, H, spp, nb, depth=3)
    got = []
    loop.run(seeds, on_frame=lambda k, a: got.append(a.copy()))
    for k, s in enumerate(seeds):
        assert (bits(got[k]) == bits(want[s])).all(), k
    loop.close(); m.close()


def test_group_profile_says_where_every_devices_step_went(sky, scene_paths):
    """rt_multi_profile_enable / _collect (include/rt_hip.h rt_multi_phases): per device, five disjoint shares of the time between its
    first and last frame end -- strip render, de-interleave, copy, waiting for the gather, idle -- that sum to its step; frames stay
    bit-identical with the events in the streams; collect resets the log; too small an array is refused."""
    import ctypes as C
    from ray_tracing_amd.frames import FrameLoop
    W, H, spp, nb, n = 640, 360, 16, 4, 4
    seeds = list(range(50, 62))
    want = _reference_frames(scene_paths[0], sky, W, H, spp, nb, seeds)
    m = rt.MultiRenderer([0], on_one_device=n)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    m.profile_phases(True)
    assert all(p["frames"] == 0 and p["step_ms"] == 0 for p in m.collect_phases())        # nothing submitted yet
    loop = FrameLoop(m, W, H, spp, nb, depth=3)
    got = []
    loop.run(seeds, on_frame=lambda k, a: got.append(a.copy()))
    for k, s in enumerate(seeds):
        assert (bits(got[k]) == bits(want[s])).all(), k
    phases = m.collect_phases()
    assert len(phases) == n
    for i, p in enumerate(phases):
        assert p["frames"] == len(seeds) - 1 and p["step_ms"] > 0, p
        shares = [p[k] for k in rt.PHASES]
        assert all(v >= 0 for v in shares) and abs(sum(shares) - p["step_ms"]) <= 1e-6 * max(p["step_ms"], 1.0), p
        assert (p["copy_ms"] > 0 and p["deinterleave_ms"] > 0) if i == 0 else (p["copy_ms"] == 0 and p["deinterleave_ms"] == 0), p
    assert sum(p["render_ms"] for p in phases) > 0
    verdict = rt.judge_phases(phases)
    assert verdict["critical_rank"] in range(n) and verdict["step_bound"] in ("render", "gather", "copy", "host")
    assert all(p["frames"] == 0 for p in m.collect_phases())                               # the log was reset
    small = (rt.MultiPhases * (n - 1))()
    assert rt.lib().rt_multi_profile_collect(m._m, small, n - 1) == -1                     # RT_ERR_ARGUMENT: room for every device is needed
    m.profile_phases(False)
    loop.run(seeds[:4])
    assert all(p["frames"] == 0 for p in m.collect_phases())                               # off: nothing is recorded
    loop.close(); m.close()
