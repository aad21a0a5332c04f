"""The behaviour of the HIP runtime and of the GPU's workgroup scheduler that the library's host side is TUNED to, as regression
tests.  None of this is documented by ROCm; each fact was found by a probe script in rounds 3-5 and the scheduler was shaped
around it (docs/lab/r05.md sections 3, 10, 12).  A driver, runtime or firmware change that alters one of them costs 10-50 % in
frame latency while every parity test stays green -- so each test names the knob to revisit when it goes red.

What is protected is the reference's "workers render while the main thread presents" (main.c:354-408 beside :450-482): frames
delivered when they are done, consecutive launches overlapping, interactive passes that answer at once.

Thresholds are generous (the boxes are shared): a test fails when the mechanism is gone, not when a box is busy."""
import time

import numpy as np
import pytest

import ray_tracing_amd as rt

pytestmark = pytest.mark.gpu
W, H = 1920, 1080


@pytest.fixture(scope="module")
def c1(scene_paths):
    g = rt.Renderer(0)
    g.set_skybox(rt.load_skybox()); g.set_scene(scene_paths[0]); g.set_camera()
    g.compile_scene()
    yield g
    g.close()


def test_a_small_device_to_host_copy_waits_behind_a_persistent_launch_and_a_64KB_one_does_not(scene_paths):
    """rt_internal.h RT_CTL_COPY_BYTES = 65536.  The runtime performs device-to-host copies of up to 16 KB with a KERNEL, and a
    kernel needs a workgroup slot: beside a persistent launch that holds them all it waits until the launch drains (7.5 ms
    beside a 9 ms launch; profiles/r05/copy_beside_kernel.txt).  From 64 KB on a DMA engine does the copy (0.04 ms).  A launch's
    control words therefore travel in 64 KB -- or every frame of a frame loop is delivered one launch late."""
    import torch
    g = rt.Renderer(0)
    g.set_skybox(rt.load_skybox()); g.set_scene(scene_paths[1]); g.compile_scene()
    dev = torch.device("cuda", 0)
    buf = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    src = torch.zeros((65536,), dtype=torch.float32, device=dev)
    host = torch.empty((65536,), dtype=torch.float32, pin_memory=True)
    copy = torch.cuda.Stream(dev, priority=-1)
    p = rt.Renderer.params(W, H, 256, 8, seed=1)                    # a whole-chip launch of ~9 ms (C2)
    g.render_device(p, buf.data_ptr()); g.synchronize()

    def copy_beside_launch(n_floats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.render_device(p, buf.data_ptr())
        time.sleep(0.001)                                           # the launch is running
        t1 = time.perf_counter()
        with torch.cuda.stream(copy):
            host[:n_floats].copy_(src[:n_floats], non_blocking=True)
        copy.synchronize()
        waited = (time.perf_counter() - t1) * 1e3
        g.synchronize()
        return waited, (time.perf_counter() - t0) * 1e3

    copy_beside_launch(16384); copy_beside_launch(16)               # first use of the stream and of both copy paths
    big = sorted(copy_beside_launch(16384) for _ in range(3))[1]    # 64 KB
    small = sorted(copy_beside_launch(16) for _ in range(3))[1]     # 64 bytes
    g.close()
    print(f"beside a {big[1]:.1f} ms launch: a 64 KB copy returned after {big[0]:.3f} ms, a 64-byte copy after {small[0]:.3f} ms")
    assert big[1] > 4.0, "the launch this test hides a copy behind has become too short to tell anything: use more samples per pixel"
    assert big[0] < 1.0, (f"a 64 KB device-to-host copy took {big[0]:.2f} ms beside a running launch: it no longer goes to a DMA engine -- "
                          "RT_CTL_COPY_BYTES (csrc/rt_internal.h) must grow, or the control words must ride in the frame's own copy")
    assert small[0] > 0.5 * big[1], (f"a 64-byte device-to-host copy now returns after {small[0]:.2f} ms beside a {small[1]:.1f} ms launch: the runtime no "
                                     "longer does small copies with a kernel that queues behind the launch -- good news: the 64 KB work-around "
                                     "(RT_CTL_COPY_BYTES, 64 KB of pinned memory per slot and ticket) can go")


def test_a_depth_2_frame_loop_delivers_every_frame_when_it_is_done(c1):
    """rt_frame_submit / rt_frame_wait with two frames in flight: the first delivery arrives within
    1.3 x a lone frame -- not one launch late, as it did while the control words travelled in a 64-byte copy -- and no interval
    between two deliveries exceeds 1.5 x the median.  Knobs: RT_CTL_COPY_BYTES, workgroups_per_cu_for() (rt_api.cpp)."""
    g = c1
    frames = [rt.HostFrame(W, H) for _ in range(2)]                 # page-locked (rt_host_alloc): the copy runs beside the next render
    p_of = lambda k: rt.Renderer.params(W, H, 64, 4, seed=k)        # noqa: E731
    for k in range(4):                                              # every stream, scratch set and slot used once
        g.frame_submit(p_of(k), k % 2, frames[k % 2]); g.frame_wait(k % 2)
    lone = []
    for k in range(5):
        g.synchronize()
        t0 = time.perf_counter()
        g.frame_submit(p_of(k), 0, frames[0]); g.frame_wait(0)
        lone.append((time.perf_counter() - t0) * 1e3)
    lone = sorted(lone)[len(lone) // 2]
    best = None
    for attempt in range(3):                                        # (a shared box: the mechanism has to show once)
        g.synchronize()
        n = 24
        t0 = time.perf_counter()
        g.frame_submit(p_of(0), 0, frames[0])
        stamps = []
        for k in range(n):
            if k + 1 < n:
                g.frame_submit(p_of(k + 1), (k + 1) % 2, frames[(k + 1) % 2])
            g.frame_wait(k % 2)
            stamps.append((time.perf_counter() - t0) * 1e3)
        first = stamps[0]
        steps = np.diff(stamps)[1:]
        res = (first / lone, float(steps.max() / np.median(steps)), float(np.median(steps)))
        print(f"lone frame {lone:.3f} ms; depth 2: first delivery after {first:.3f} ms ({res[0]:.2f} x), median interval {res[2]:.3f} ms, longest {steps.max():.3f} ({res[1]:.2f} x)")
        if best is None or max(res[0] / 1.3, res[1] / 1.5) < max(best[0] / 1.3, best[1] / 1.5):
            best = res
        if best[0] <= 1.3 and best[1] <= 1.5:
            break
    assert best[0] <= 1.3, (f"the first frame of a depth-2 loop was delivered after {best[0]:.2f} x a lone frame: deliveries wait for the NEXT launch "
                            "again -- see RT_CTL_COPY_BYTES (the control words' copy must not need a workgroup slot) and rt_frame_submit's copy stream")
    assert best[1] <= 1.5, (f"an interval between two delivered frames was {best[1]:.2f} x the median in all three attempts: consecutive launches no longer "
                            "hand the chip over smoothly -- see workgroups_per_cu_for() and lone_launch_ahead() in rt_api.cpp")
    for f in frames:
        f.free()
    assert best[2] <= 1.08 * lone, f"a frame loop with two in flight ({best[2]:.3f} ms per frame) is slower than frames one at a time ({lone:.3f}): the overlap of consecutive launches is gone"


def test_eight_strips_of_a_frame_cost_little_more_than_the_frame(c1):
    """scripts/strip_loop_probe.py at N = 8 as a test: one rank's strip of a C1 frame with RT_LAUNCH_SETS launches in flight on the
    context's streams takes <= frame / 8 / 0.90 (round 5: 0.935).  What it rests on: five render streams that get hardware queues
    of their own (rt_create makes them, in order), one workgroup slot per CU for small launches far ahead
    (workgroups_per_cu_for: RT_SMALL_LAUNCH_PIXELS_PER_STREAM), launches that clear nothing they do not have to."""
    g = c1
    S = rt.LAUNCH_SETS
    eff, per_step = 0.0, {}
    for attempt in range(3):                                        # (a shared box: the mechanism has to show once)
        per_step = _strip_steps(g, S)
        eff = max(eff, per_step[1] / 8 / per_step[8])
        if eff >= 0.90:
            break
    print(f"frame {per_step[1]:.3f} ms per step, strip of one of eight ranks {per_step[8]:.3f} ms: {eff:.3f} of frame / 8")
    assert eff >= 0.90, (f"a strip of one of eight ranks takes {per_step[8]:.3f} ms against frame / 8 = {per_step[1] / 8:.3f} ({eff:.3f}): the fixed cost per "
                         "launch has grown -- see workgroups_per_cu_for() (one slot per CU for small launches far ahead), the order in which rt_create makes "
                         "the render streams, and what a launch clears (rt_launch_trace)")


def _strip_steps(g, S):
    import torch
    per_step = {}
    for world in (1, 8):
        rank = world // 2
        p_of = lambda k: rt.Renderer.params(W, H, 64, 4, seed=k, row_block=8, rank=rank, world=world)   # noqa: E731
        bufs = [torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0") for _ in range(S)]
        for k in range(2 * S):
            g.render_device(p_of(k), bufs[k % S].data_ptr(), stream=g.stream(k % S))
        g.synchronize()
        best = None
        for rep in range(4):
            torch.cuda.synchronize()
            n = 40
            t0 = time.perf_counter()
            for k in range(n):
                g.render_device(p_of(k), bufs[k % S].data_ptr(), stream=g.stream(k % S))
            g.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            best = dt if best is None or dt < best else best
        per_step[world] = best
    return per_step


INTERACTIVE_PROBE = r"""
import sys, time
import ray_tracing_amd as rt
g = rt.Renderer(0)                       # the context first, as every host of the library does: its render streams are made now
g.set_skybox(rt.load_skybox()); g.set_scene(sys.argv[1]); g.set_camera(); g.compile_scene()
g.progressive_begin(1920, 1080, init_scale=1, max_bounces=10, seed=3)
for _ in range(16):
    g.progressive_pass()
g.synchronize()
best = None
for rep in range(4):
    n = 128
    t0 = time.perf_counter()
    for _ in range(n):
        g.progressive_pass()
    g.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    best = dt if best is None or dt < best else best
print(best)
g.close()
"""


def test_three_interactive_passes_overlap_on_three_streams(c1, scene_paths):
    """rt_progressive_pass: pass n renders on stream n mod 3; only the publish steps wait for each other.  A 1080p pass of one sample
    per pixel is ramp-up and tail from end to end -- 0.254 ms one after the other, 0.175 ms overlapped (round 5).  Knobs:
    RT_PASSES_IN_FLIGHT and the one-slot-per-CU rule for single passes (rt_api.cpp progressive_launch), stream creation order.

    Measured in a FRESH PROCESS whose first act is rt_create(): the hardware queue a stream is given depends on the streams that
    exist -- or have existed -- in the process when it is made (docs/lab/r05.md section 10), and this test process has made dozens.
    The same loop on this process's long-lived context is printed beside it: round 6 saw 0.231 ms there against 0.179 in a fresh
    process -- the overlap is a property of a host that creates its context first, which is what the library documents."""
    import os
    import subprocess
    import sys
    from rtlibs import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", INTERACTIVE_PROBE, scene_paths[0]], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    fresh = float(p.stdout.strip().splitlines()[-1])
    g = c1
    g.progressive_begin(W, H, init_scale=1, max_bounces=10, seed=3)
    for _ in range(16):
        g.progressive_pass()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(128):
        g.progressive_pass()
    g.synchronize()
    here = (time.perf_counter() - t0) / 128 * 1e3
    print(f"{fresh:.4f} ms per 1080p interactive pass of one sample per pixel in a fresh process (round 5: 0.175; one stream: 0.254); "
          f"{here:.4f} ms on this test process's context, made after many other streams")
    assert fresh <= 0.22, (f"an interactive pass takes {fresh:.3f} ms: consecutive passes no longer overlap (0.254 ms on one stream) -- see RT_PASSES_IN_FLIGHT, "
                           "the render streams' creation order in rt_create, and the reuse of the camera rays (rt_primary_passes_run)")
