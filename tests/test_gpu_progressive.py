"""Progressive accumulation (the reference's worker()/update_frame() protocol, main.c:354-408,450-482)
on the GPU against the oracle's restatement of the same protocol; the oracle's scale ladder itself is
pinned to the compiled reference in tests/test_oracle_vs_ref.py."""
import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import bits, oracle_progressive, synthetic_skybox

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,init_scale,passes", [(64, 32, 8, 6), (96, 48, 4, 5), (50, 30, 4, 4), (40, 24, 1, 3), (128, 64, 16, 7)])
def test_progressive_matches_oracle(oracle, scene_paths, W, H, init_scale, passes):
    sky = synthetic_skybox(32, seed=7)
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(sky); oracle.set_skybox(sky)
    oracle.set_camera()
    for si in (0, 1):
        g.set_scene(scene_paths[si]); oracle.load_scene(scene_paths[si])
        g.progressive_begin(W, H, init_scale=init_scale, max_bounces=10, seed=5)
        weights = [g.progressive_pass() for _ in range(passes)]
        frame = g.progressive_resolve()
        want, _, count, next_scale = oracle_progressive(oracle, W, H, init_scale, passes, 10, 5)
        st = g.progressive_state()
        assert st["passes"] == passes and st["next_scale"] == next_scale
        assert np.float32(st["count"]) == np.float32(count) == np.float32(sum(np.float32(w) for w in weights))
        assert (bits(frame) == bits(want)).all(), (si, W, H, init_scale)
    g.close()


def test_invalidate_restarts_the_ladder(oracle, scene_paths):
    sky = synthetic_skybox(32, seed=7)
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(sky); g.set_scene(scene_paths[0]); oracle.set_skybox(sky); oracle.load_scene(scene_paths[0])
    g.progressive_begin(64, 32, init_scale=8, max_bounces=4, seed=1)
    gen0 = g.progressive_state()["generation"]
    for _ in range(5):
        g.progressive_pass()
    cam = dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0)
    g.set_camera(**cam); oracle.set_camera(**cam)
    g.progressive_invalidate()                      # invalidate_accumulation(), main.c:115-124
    st = g.progressive_state()
    assert st == dict(next_scale=8, count=0.0, generation=gen0 + 1, passes=0)
    with pytest.raises(rt.RtError):                 # update_frame() would wait for a pass (main.c:462)
        g.progressive_resolve()
    for _ in range(3):
        g.progressive_pass()
    want, _, _, _ = oracle_progressive(oracle, 64, 32, 8, 3, 4, 1)
    assert (bits(g.progressive_resolve()) == bits(want)).all()
    oracle.set_camera(); g.close()


def test_full_resolution_passes_equal_rt_render(scene_paths):
    """With init_scale 1 the accumulated passes are the samples of rt_render(): same frame, bit for bit."""
    sky = synthetic_skybox(32, seed=7)
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(sky); g.set_scene(scene_paths[0])
    g.progressive_begin(160, 90, init_scale=1, max_bounces=4, seed=9)
    for _ in range(6):
        g.progressive_pass()
    assert (bits(g.progressive_resolve()) == bits(g.render(160, 90, 6, 4, seed=9))).all()
    g.close()


@pytest.mark.parametrize("compiled", [False, True])
def test_cancel_gives_up_a_frame_in_flight(scene_paths, compiled):
    """rt_cancel() (main.c:316-317: a worker abandons its pass when the frame is invalidated): a long launch is cut
    short from another thread, reports RT_CANCELLED, and the next launch renders the complete, correct frame.
    With the generic and with the compiled kernel: the latter leaves no register of any SIMD free, and a request that has to
    run something on the GPU to be delivered (rounds 1-2: a small copy, which the runtime does with a kernel) waited for the
    launch it was to stop (profiles/r03/cancel_probe.txt)."""
    import threading
    import time
    import torch
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(rt.load_skybox()); g.set_scene(scene_paths[0]); g.set_camera()
    if compiled:
        g.compile_scene()
    W, H, spp, nb = 1920, 1080, 1024, 8                     # ~100 ms of GPU work
    strip = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    p = g.params(W, H, spp, nb)
    t0 = time.perf_counter()
    g.render_device(p, strip.data_ptr())
    g.synchronize()
    full = time.perf_counter() - t0
    assert not g.was_cancelled()
    t0 = time.perf_counter()
    g.render_device(p, strip.data_ptr())
    threading.Timer(0.005, g.cancel).start()                # from another thread, 5 ms into the launch
    g.synchronize()
    cut = time.perf_counter() - t0
    assert g.was_cancelled()
    assert cut < 0.5 * full and cut < 0.03, (cut, full)     # 5 ms + the request's way to every wave (3 ms at 1024 samples per pixel) + the paths in flight
    print(f"full launch {full * 1e3:.1f} ms, cancelled after 5 ms: returned after {cut * 1e3:.1f} ms")
    # the request is forgotten by the next launch; rt_render() reports a cancelled frame as such
    a = g.render(320, 180, 8, 4, seed=3)
    s = g.render(320, 180, 8, 4, seed=3, kernel=rt.KERNEL_SIMPLE)
    assert (a.view(np.uint32) == s.view(np.uint32)).all()
    g.cancel()                                              # nothing in flight: must not leak into the next launch
    a2 = g.render(320, 180, 8, 4, seed=3)
    assert (a2.view(np.uint32) == s.view(np.uint32)).all()
    g.close()


def test_cancel_reaches_both_launches_in_flight(scene_paths):
    """Two launches on the context's two streams are on the GPU together (rt_stream); rt_cancel() stops both, and
    the next launches on either stream -- each takes the next of the scratch sets -- are complete and correct."""
    import threading
    import time
    import torch
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(rt.load_skybox()); g.set_scene(scene_paths[0]); g.set_camera()
    W, H, spp, nb = 1920, 1080, 512, 8                      # ~60 ms of GPU work each
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    streams = [g.stream(0), g.stream(1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.render_device(g.params(W, H, spp, nb, seed=1), bufs[0].data_ptr(), streams[0]); g.synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(2):
        g.render_device(g.params(W, H, spp, nb, seed=k), bufs[k].data_ptr(), streams[k])
    threading.Timer(0.005, g.cancel).start()
    g.synchronize()
    cut = time.perf_counter() - t0
    assert g.was_cancelled()
    assert cut < one, (cut, one)                            # two launches, stopped: less than ONE takes
    want = g.render(320, 180, 8, 4, seed=3, kernel=rt.KERNEL_SIMPLE)
    small = [torch.zeros((180, 320, 3), dtype=torch.float32, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    for k in range(4):
        g.render_device(g.params(320, 180, 8, 4, seed=3), small[k].data_ptr(), streams[k & 1])
    g.synchronize()
    assert not g.was_cancelled()
    for k in range(4):
        assert (small[k].cpu().numpy().view(np.uint32) == want.view(np.uint32)).all(), k
    g.close()


def test_kept_pixel_lists_change_no_bit(oracle, scene_paths):
    """Once the scale ladder has reached full resolution, a pass differs from the pass before last (same scratch set)
    in its sample number only, and rt_progressive_pass keeps rt_primary_pass's output -- pixel lists and the sky pixels of
    the low-resolution frame -- instead of recomputing it (never with poison_frame, which would wipe those sky pixels:
    hence no poisoning here).  Whatever that output depends on must invalidate it: camera, scene, skybox, frame size, an
    ordinary rt_render() through the same scratch sets in between."""
    sky = synthetic_skybox(32, seed=7)
    sky2 = synthetic_skybox(16, seed=9)
    g = rt.Renderer(0)
    g.set_skybox(sky); oracle.set_skybox(sky)
    g.set_scene(scene_paths[0]); oracle.load_scene(scene_paths[0])
    g.set_camera(); oracle.set_camera()
    W, H = 96, 54

    def check(init_scale, passes, what):
        g.progressive_begin(W, H, init_scale=init_scale, max_bounces=6, seed=3)
        for _ in range(passes):
            g.progressive_pass()
        want, _, _, _ = oracle_progressive(oracle, W, H, init_scale, passes, 6, 3)
        assert (bits(g.progressive_resolve()) == bits(want)).all(), what

    before = rt.lib().rt_primary_passes_run(g._ctx)
    check(1, 7, "seven full-resolution passes: two of them on kept lists")
    assert rt.lib().rt_primary_passes_run(g._ctx) - before == rt.LAUNCH_SETS          # one per scratch set
    check(4, 8, "ladder 4, 2, 1 and five more passes")
    cam = dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0)
    g.set_camera(**cam); oracle.set_camera(**cam)
    check(1, 5, "camera moved")
    g.set_scene(scene_paths[1]); oracle.load_scene(scene_paths[1])
    check(1, 5, "scene replaced")
    g.set_skybox(sky2); oracle.set_skybox(sky2)
    g.set_scene(scene_paths[2]); oracle.load_scene(scene_paths[2])
    check(1, 5, "skybox and scene replaced (scene_2 is mostly sky)")
    # an ordinary frame through the same scratch sets between two passes
    g.progressive_begin(W, H, init_scale=1, max_bounces=6, seed=3)
    for k in range(6):
        g.progressive_pass()
        if k in (2, 3):
            g.render(64, 40, 2, 3, seed=k)
    want, _, _, _ = oracle_progressive(oracle, W, H, 1, 6, 6, 3)
    assert (bits(g.progressive_resolve()) == bits(want)).all()
    # a camera move in the middle of the full-resolution passes (main.c:522-570 -> invalidate_accumulation())
    g.progressive_begin(W, H, init_scale=1, max_bounces=6, seed=3)
    for _ in range(4):
        g.progressive_pass()
    g.set_camera(); oracle.set_camera()
    g.progressive_invalidate()
    for _ in range(4):
        g.progressive_pass()
    want, _, _, _ = oracle_progressive(oracle, W, H, 1, 4, 6, 3)
    assert (bits(g.progressive_resolve()) == bits(want)).all()
    # the ladder restarted from half resolution with nothing else changed: the pass at scale 2 overwrites part of the
    # low-resolution frame, so the full-resolution lists of the other scratch set have lost their sky pixels
    g.progressive_begin(W, H, init_scale=1, max_bounces=6, seed=3)
    for _ in range(4):
        g.progressive_pass()
    g.progressive_begin(W, H, init_scale=2, max_bounces=6, seed=3)
    for _ in range(4):
        g.progressive_pass()
    want, _, _, _ = oracle_progressive(oracle, W, H, 2, 4, 6, 3)
    assert (bits(g.progressive_resolve()) == bits(want)).all()
    oracle.set_camera(); oracle.set_skybox(sky); g.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_ladders_reassemble_to_the_single_device_ladder(oracle, scene_paths, world):
    """rt_progressive_begin_rank(): `world` contexts (here all on the one GPU of the box, as the ranks of a one-process-per-GPU
    host would each have) accumulate the rows of their own 16-row blocks through the whole scale ladder with nothing
    exchanged; their resolved rows, put back in frame order, are the single-device ladder's frame and the oracle's."""
    from ray_tracing_amd.multi_gpu import frame_index
    sky = synthetic_skybox(32, seed=7)
    oracle.set_skybox(sky); oracle.set_camera()
    for (W, H, init_scale, passes, si) in [(96, 70, 8, 6, 0), (64, 130, 16, 7, 1), (50, 33, 4, 4, 0), (40, 24, 1, 3, 0)]:
        oracle.load_scene(scene_paths[si])
        want, _, count, next_scale = oracle_progressive(oracle, W, H, init_scale, passes, 10, 5)
        ranks = []
        for r in range(world):
            g = rt.Renderer(0)
            g.set_tuning(poison_frame=True)
            g.set_skybox(sky); g.set_scene(scene_paths[si])
            g.progressive_begin(W, H, init_scale=init_scale, max_bounces=10, seed=5, rank=r, world=world)
            ranks.append(g)
        for _ in range(passes):
            for g in ranks:
                g.progressive_pass()
        rows = np.concatenate([g.progressive_resolve() for g in ranks])          # what one gather delivers
        frame = rows[frame_index(H, 16, world)]
        st = ranks[0].progressive_state()
        assert st["passes"] == passes and st["next_scale"] == next_scale and np.float32(st["count"]) == np.float32(count)
        assert (bits(frame) == bits(want)).all(), (world, W, H, init_scale)
        for g in ranks:
            g.close()


def test_group_ladder_over_a_one_rank_rccl_communicator(oracle, scene_paths):
    """rt_multi_progressive_*: the group's ladder, its ONE gather per displayed frame and the de-interleave on the one GPU of
    the box (rt_tuning.force_collective); invalidation with a moved camera restarts it (main.c:115-124)."""
    sky = synthetic_skybox(32, seed=7)
    oracle.set_skybox(sky); oracle.load_scene(scene_paths[0]); oracle.set_camera()
    m = rt.MultiRenderer([0])
    m.set_tuning(force_collective=1, poison_frame=1)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    W, H = 96, 70
    m.progressive_begin(W, H, init_scale=8, max_bounces=10, seed=5)
    for p in range(1, 7):
        m.progressive_pass()
        if p in (1, 4, 6):           # frames are displayed while the ladder runs (update_frame(), main.c:450-482)
            want, _, count, next_scale = oracle_progressive(oracle, W, H, 8, p, 10, 5)
            assert (bits(m.progressive_resolve()) == bits(want)).all(), p
    st = m.progressive_state()
    assert st["passes"] == 6 and st["next_scale"] == next_scale and np.float32(st["count"]) == np.float32(count)
    cam = dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0)
    m.set_camera(**cam); oracle.set_camera(**cam)
    m.progressive_invalidate()
    with pytest.raises(rt.RtError):
        m.progressive_resolve()
    for _ in range(3):
        m.progressive_pass()
    want, _, _, _ = oracle_progressive(oracle, W, H, 8, 3, 10, 5)
    assert (bits(m.progressive_resolve()) == bits(want)).all()
    oracle.set_camera(); m.close()


@pytest.mark.parametrize("compiled", [False, True])
@pytest.mark.parametrize("W,H,init_scale,calls", [(64, 32, 8, [20]), (96, 50, 4, [1, 12, 1, 9]), (50, 30, 1, [8, 1, 30]), (131, 67, 2, [40])])
def test_batched_passes_are_the_passes_one_by_one(oracle, scene_paths, compiled, W, H, init_scale, calls):
    """rt_progressive_passes(): the passes at full resolution go several to a launch (the kernel adds a pixel's samples, in
    pass order, onto the sums so far); sums, count, ladder state and the resolved frame are those of one call of
    rt_progressive_pass() per pass -- and the oracle's ladder."""
    sky = synthetic_skybox(32, seed=7)
    g = rt.Renderer(0)
    g.set_skybox(sky); oracle.set_skybox(sky); oracle.set_camera()
    for si in (0, 1):
        g.set_scene(scene_paths[si]); oracle.load_scene(scene_paths[si])
        if compiled:
            g.compile_scene()
        g.progressive_begin(W, H, init_scale=init_scale, max_bounces=10, seed=5)
        done = 0
        for n in calls:
            g.progressive_passes(n)
            done += n
            if n == calls[0] or done == sum(calls):       # a frame is shown between the calls too
                want, _, count, next_scale = oracle_progressive(oracle, W, H, init_scale, done, 10, 5)
                st = g.progressive_state()
                assert st["passes"] == done and st["next_scale"] == next_scale and np.float32(st["count"]) == np.float32(count)
                assert (bits(g.progressive_resolve()) == bits(want)).all(), (si, W, H, init_scale, done)
    g.close()


def test_batched_passes_at_a_frame_size_with_many_samples(oracle, scene_paths):
    """A batch as a presenter would ask for it: 640x360, the ladder from 1/8 and then 150 passes at full resolution in one call
    (one launch of 147 samples per pixel after the three low-resolution ones) against rt_render()-style single passes."""
    sky = synthetic_skybox(64, seed=3)
    a, b = rt.Renderer(0), rt.Renderer(0)
    for g in (a, b):
        g.set_skybox(sky); g.set_scene(scene_paths[0]); g.compile_scene()
        g.progressive_begin(640, 360, init_scale=8, max_bounces=6, seed=77)
    a.progressive_passes(150)
    for _ in range(150):
        b.progressive_pass()
    assert a.progressive_state() == b.progressive_state()
    assert (bits(a.progressive_resolve()) == bits(b.progressive_resolve())).all()
    a.progressive_passes(300)                    # more than RT_PROGRESSIVE_BATCH: two launches
    for _ in range(300):
        b.progressive_pass()
    assert a.progressive_state() == b.progressive_state()
    assert (bits(a.progressive_resolve()) == bits(b.progressive_resolve())).all()
    a.close(); b.close()


@pytest.mark.parametrize("world", [2, 3])
def test_batched_passes_on_ranks_and_on_a_group(oracle, scene_paths, world):
    """The same on the ranks of a host with one process per GPU (rt_progressive_begin_rank) and on a device group
    (rt_multi_progressive_passes, here `world` contexts on the one GPU of the box)."""
    from ray_tracing_amd.multi_gpu import frame_index
    sky = synthetic_skybox(32, seed=7)
    oracle.set_skybox(sky); oracle.set_camera(); oracle.load_scene(scene_paths[0])
    W, H, init_scale, passes = 96, 70, 4, 30
    want, _, count, next_scale = oracle_progressive(oracle, W, H, init_scale, passes, 10, 5)
    ranks = []
    for r in range(world):
        g = rt.Renderer(0)
        g.set_skybox(sky); g.set_scene(scene_paths[0])
        g.progressive_begin(W, H, init_scale=init_scale, max_bounces=10, seed=5, rank=r, world=world)
        ranks.append(g)
    for g in ranks:
        g.progressive_passes(12)         # two at 1/4 and 1/2, ten at full resolution in one launch
    for g in ranks:
        g.progressive_passes(passes - 12)
    rows = np.concatenate([g.progressive_resolve() for g in ranks])
    st = ranks[0].progressive_state()
    assert st["passes"] == passes and st["next_scale"] == next_scale and np.float32(st["count"]) == np.float32(count)
    assert (bits(rows[frame_index(H, 16, world)]) == bits(want)).all()
    for g in ranks:
        g.close()
    m = rt.MultiRenderer([0], on_one_device=world)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    m.progressive_begin(W, H, init_scale=init_scale, max_bounces=10, seed=5)
    m.progressive_passes(passes)
    assert (bits(m.progressive_resolve()) == bits(want)).all()
    m.close()


def test_cancelled_batch_publishes_none_of_its_passes(oracle, scene_paths):
    """A batch that rt_cancel() cuts short is not published (main.c:382) -- none of its passes: sums and count stay those of the
    passes before it, and the ladder goes on with the next call."""
    import torch
    sky = synthetic_skybox(32, seed=7)
    W, H, nb, seed = 640, 360, 10, 3
    g = rt.Renderer(0)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    g.progressive_begin(W, H, init_scale=1, max_bounces=nb, seed=seed)
    g.progressive_passes(10)
    g.synchronize()
    before = g.progressive_resolve()
    # (the deterministic cancel point of test_gpu_frames.py: the batch is still queued behind ~90 ms of other work when the request is made)
    scratch = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    g.render_device(g.params(1920, 1080, 1024, 8, seed=9), scratch.data_ptr())
    g.progressive_passes(20)
    g.cancel()
    assert g.was_cancelled()
    st = g.progressive_state()
    assert st["passes"] == 30 and st["count"] == 10.0
    assert (bits(g.progressive_resolve()) == bits(before)).all()
    g.progressive_passes(9)
    assert g.progressive_state()["count"] == 19.0
    g.close()


def test_overlapped_passes_and_what_a_host_does_between_them(oracle, scene_paths):
    """Round 5: consecutive single passes render on three of the context's streams into the scratch sets' own low-resolution
    frames and only their publish steps wait for each other; a pass that keeps its set's pixel lists clears nothing, because the
    last workgroup of the set's previous launch left the counters clean.  A presenter that resolves after every pass, renders
    something else in between, cancels passes in flight and goes on must get the oracle's ladder bit for bit."""
    import torch
    sky = synthetic_skybox(32, seed=7)
    g = rt.Renderer(0)
    g.set_skybox(sky); g.set_scene(scene_paths[0]); oracle.set_skybox(sky); oracle.load_scene(scene_paths[0])
    oracle.set_camera()
    W, H, nb, seed = 160, 96, 6, 9
    g.progressive_begin(W, H, init_scale=1, max_bounces=nb, seed=seed)
    before = rt.lib().rt_primary_passes_run(g._ctx)
    shown = {}
    for k in range(1, 13):
        g.progressive_pass()
        frame = g.progressive_resolve()                     # update_frame() after every worker iteration (main.c:450-482)
        if k in (1, 6, 12): shown[k] = frame.copy()
    assert rt.lib().rt_primary_passes_run(g._ctx) - before == rt.LAUNCH_SETS      # the other seven kept their sets' lists
    other = g.render(320, 180, 8, 4, seed=3)               # takes a scratch set and a stream of its own
    assert (bits(other) == bits(g.render(320, 180, 8, 4, seed=3, kernel=rt.KERNEL_SIMPLE))).all()
    for _ in range(5):
        g.progressive_pass()
    shown[17] = g.progressive_resolve().copy()
    for k, frame in shown.items():
        want, _, count, _ = oracle_progressive(oracle, W, H, 1, k, nb, seed)
        assert (bits(frame) == bits(want)).all(), k
    # passes in flight are cancelled (some of them cut short: their last workgroups leave "cancelled" behind), the ladder starts
    # again -- on the same lists, nothing changed but the sample numbers -- and the sets' next launches must come out whole
    big = rt.Renderer(0)
    big.set_skybox(rt.load_skybox()); big.set_scene(scene_paths[0]); big.compile_scene()
    big.progressive_begin(1920, 1080, init_scale=1, max_bounces=10, seed=2)
    for _ in range(2 * rt.LAUNCH_SETS):
        big.progressive_pass()
    big.synchronize()
    for _ in range(12):
        big.progressive_pass()
    big.progressive_invalidate()                            # rt_cancel() + clear (main.c:115-124)
    runs = rt.lib().rt_primary_passes_run(big._ctx)
    for _ in range(2 * rt.LAUNCH_SETS):
        big.progressive_pass()
    assert rt.lib().rt_primary_passes_run(big._ctx) == runs                     # every pass on kept lists
    st = big.progressive_state()
    assert st["passes"] == 2 * rt.LAUNCH_SETS and st["count"] == float(2 * rt.LAUNCH_SETS)
    got = big.progressive_resolve()
    ref = rt.Renderer(0)
    ref.set_skybox(rt.load_skybox()); ref.set_scene(scene_paths[0]); ref.compile_scene()
    ref.set_tuning(poison_frame=True)                       # (never keeps lists)
    ref.progressive_begin(1920, 1080, init_scale=1, max_bounces=10, seed=2)
    for _ in range(2 * rt.LAUNCH_SETS):
        ref.progressive_pass()
    assert (bits(got) == bits(ref.progressive_resolve())).all()
    rc, rep = big.last_launch_report()
    assert rc == 0 and rep["stamp"] == rep["launch_id"] and not rep["cancelled"] and rep["pixels_written"] == rep["pixels_listed"] > 0
    for r in (g, big, ref): r.close()
