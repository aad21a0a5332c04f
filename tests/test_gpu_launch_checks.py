"""Every delivered frame is a complete frame (include/rt_hip.h): a launch accounts for itself -- camera-ray blocks finished, object
pixels listed / fetched / written, the stamp of its last wave -- and every call that delivers a frame refuses one whose launch
did not.  The reference publishes a column whole or not at all (main.c:377-396).

* a normal launch leaves the expected numbers (one- and 64-list launches, interactive passes, culled scenes);
* a launch that loses its tail (testing aid rt_tuning.test_drop_pixels: the waves see every list shorter) is refused with
  RT_ERR_DEVICE by rt_render, rt_frame_wait / rt_frame_poll, rt_multi_frame_wait, rt_launch_check_wait (the hook of hosts that
  enqueue launches themselves, i.e. multi_gpu.TiledFrame) and by the interactive ladder, which does not publish the pass;
* csrc/rt_lit.h is audited in production (rt_tuning.audit_known_taps): frames unchanged, audited taps counted, none disagrees on
  the shipped scenes and on random ones -- and a table that IS wrong (testing aid) is caught."""
import numpy as np
import pytest
import torch

import ray_tracing_amd as rt
from ray_tracing_amd import multi_gpu
from rtlibs import bits, large_scene, LARGE_SCENE_CAMERA, make_scene

pytestmark = pytest.mark.gpu
ERR_DEVICE = -2


@pytest.fixture(scope="module")
def sky():
    return rt.load_skybox()


def _renderer(scene, sky, compiled=False, **tuning):
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True, **tuning)
    g.set_scene(scene); g.set_skybox(sky); g.set_camera()
    if compiled:
        g.compile_scene()
    return g


@pytest.mark.parametrize("W,H,spp,compiled", [(320, 180, 8, False), (1920, 1080, 2, True), (256, 144, 1, False), (640, 360, 33, True)])
def test_a_launch_accounts_for_itself(sky, scene_paths, W, H, spp, compiled):
    g = _renderer(scene_paths[0], sky, compiled)
    frame = g.render(W, H, spp, 4, seed=3)
    rc, r = g.last_launch_report()
    assert rc == 0, rt.lib().rt_last_error()
    assert r["launch_checked"] == 1 and r["stamp"] == r["launch_id"] != 0 and r["cancelled"] == 0
    assert r["primary_blocks_done"] == r["primary_blocks_expected"] == ((W + 7) // 8) * ((H + 7) // 8)
    assert r["pixels_listed"] == r["pixels_fetched"] == r["pixels_written"] > 0
    assert r["waves_left"] > 0 and r["taps_audited"] == 0
    # the object pixels are exactly the pixels the launch says it wrote: the others are sky, finished by the camera-ray pass
    s = g.render(W, H, spp, 4, seed=4)
    differs = (bits(frame) != bits(s)).any(axis=2).sum()
    assert differs <= r["pixels_listed"]
    # the cross-check kernel does not account for itself, and says so
    g.render(W, H, 1, 2, kernel=rt.KERNEL_SIMPLE)
    rc, r = g.last_launch_report()
    assert rc == 0 and r["launch_checked"] == 0
    g.close()


def test_launch_numbers_are_not_reused(sky, scene_paths):
    g = _renderer(scene_paths[0], sky)
    ids = []
    for k in range(3):
        g.render(64, 36, 2, 2, seed=k)
        ids.append(g.last_launch_report()[1]["launch_id"])
    g.reserve(640, 360)                      # allocates, announces nothing
    g.render(64, 36, 2, 2)
    ids.append(g.last_launch_report()[1]["launch_id"])
    assert ids == list(range(ids[0], ids[0] + 4)), ids
    g.close()


def test_short_launch_is_refused_by_every_delivering_call(sky, scene_paths):
    W, H, spp, nb = 320, 180, 8, 4
    L = rt.lib()
    good = _renderer(scene_paths[0], sky)
    want = good.render(W, H, spp, nb, seed=5)
    good.close()
    g = _renderer(scene_paths[0], sky, test_drop_pixels=3)
    p = g.params(W, H, spp, nb, seed=5)
    out = np.empty((H, W, 3), np.float32)
    # rt_render
    assert L.rt_render(g._ctx, p, out.ctypes.data) == ERR_DEVICE
    msg = L.rt_last_error().decode()
    assert "incomplete" in msg and "written" in msg, msg
    rc, r = g.last_launch_report()
    assert rc == ERR_DEVICE and r["stamp"] == r["launch_id"] and 0 < r["pixels_written"] < r["pixels_listed"]
    # the frame queue: wait and poll
    host = rt.HostFrame(W, H)
    g.frame_submit(p, 0, host)
    assert L.rt_frame_wait(g._ctx, 0) == ERR_DEVICE
    g.frame_submit(p, 1, host)
    g.synchronize()
    rc = L.rt_frame_poll(g._ctx, 1)
    while rc == rt.PENDING:
        rc = L.rt_frame_poll(g._ctx, 1)
    assert rc == ERR_DEVICE
    # a host that enqueues launches itself: rt_launch_check_*
    d = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    g.render_device(p, d.data_ptr())
    g.launch_check_submit(2)
    with pytest.raises(rt.RtError, match="incomplete"):
        g.launch_check_wait(2)
    # ... and with the knob off again the same context delivers the frame
    g.set_tuning(test_drop_pixels=0)
    assert (bits(g.render(W, H, spp, nb, seed=5)) == bits(want)).all()
    g.render_device(p, d.data_ptr())
    g.launch_check_submit(2)
    assert g.launch_check_wait(2)
    g.frame_submit(p, 0, host)
    assert g.frame_wait(0) and (bits(host.array) == bits(want)).all()
    host.free()
    g.close()


def test_short_launch_is_refused_by_the_device_group(sky, scene_paths):
    W, H, spp, nb = 320, 180, 4, 3
    L = rt.lib()
    m = rt.MultiRenderer([0], on_one_device=3)
    m.set_tuning(poison_frame=1, test_drop_pixels=2)
    m.set_scene(scene_paths[0]); m.set_skybox(sky); m.set_camera()
    p = rt.Renderer.params(W, H, spp, nb, seed=2)
    host = rt.HostFrame(W, H)
    m.frame_submit(p, 0, host)
    assert L.rt_multi_frame_wait(m._m, 0) == ERR_DEVICE
    assert "incomplete" in L.rt_last_error().decode()
    out = np.empty((H, W, 3), np.float32)
    assert L.rt_multi_render(m._m, p, out.ctypes.data) == ERR_DEVICE
    m.set_tuning(poison_frame=1)
    m.frame_submit(p, 0, host)
    assert m.frame_wait(0)
    g = _renderer(scene_paths[0], sky)
    assert (bits(host.array) == bits(g.render(W, H, spp, nb, seed=2))).all()
    g.close(); host.free(); m.close()


def test_tiled_frame_loop_refuses_a_short_launch(sky, scene_paths):
    W, H, spp, nb = 320, 180, 4, 3
    g = _renderer(scene_paths[0], sky, test_drop_pixels=1)
    t = multi_gpu.TiledFrame(g, W, H, spp, nb, seed=1, device=torch.device("cuda:0"))
    with pytest.raises(rt.RtError, match="incomplete"):
        for k in range(6):
            t.step(seed=k)
        t.flush()
    torch.cuda.synchronize()
    g.set_tuning(test_drop_pixels=0)
    t = multi_gpu.TiledFrame(g, W, H, spp, nb, seed=1, device=torch.device("cuda:0"))
    for k in range(6):
        t.step(seed=k)
    t.flush()
    assert (bits(t.host_frame.numpy()) == bits(g.render(W, H, spp, nb, seed=5))).all()
    g.close()


def test_the_ladder_does_not_publish_a_short_pass(sky, scene_paths):
    W, H = 256, 144
    L = rt.lib()
    g = _renderer(scene_paths[0], sky)
    g.set_tuning(poison_frame=False)
    g.progressive_begin(W, H, init_scale=2, max_bounces=4, seed=9)
    g.progressive_pass(); g.progressive_pass()
    before = g.progressive_state()
    assert before["count"] == 1.25
    g.set_tuning(test_drop_pixels=2)
    g.progressive_pass()                                   # incomplete: neither added nor counted, and reported
    out = np.empty((H, W, 3), np.float32)
    assert L.rt_progressive_resolve(g._ctx, out.ctypes.data) == ERR_DEVICE
    assert "not published" in L.rt_last_error().decode()
    g.set_tuning(test_drop_pixels=0)
    g.progressive_invalidate()                             # the host starts over: the error goes with the sums
    g.progressive_passes(12)
    got = g.progressive_resolve()
    f = _renderer(scene_paths[0], sky)
    f.set_tuning(poison_frame=False)
    f.progressive_begin(W, H, init_scale=2, max_bounces=4, seed=9)
    for _ in range(12):
        f.progressive_pass()
    assert (bits(got) == bits(f.progressive_resolve())).all()
    # a batch of passes in one launch is gated the same way
    g.set_tuning(test_drop_pixels=2)
    g.progressive_passes(16)
    assert L.rt_progressive_resolve(g._ctx, out.ctypes.data) == ERR_DEVICE
    g.close(); f.close()


@pytest.mark.parametrize("scene,compiled", [(0, True), (0, False), (1, True), (2, False)])
def test_audited_taps_agree_on_the_shipped_scenes(sky, scene_paths, scene, compiled):
    W, H, spp, nb = 640, 360, 16, 6
    g = _renderer(scene_paths[scene], sky, compiled)
    want = g.render(W, H, spp, nb, seed=7)
    for k in (-1, 3):
        g.set_tuning(audit_known_taps=k)
        got = g.render(W, H, spp, nb, seed=7)
        rc, r = g.last_launch_report()
        assert rc == 0, rt.lib().rt_last_error()
        assert (bits(got) == bits(want)).all()              # the audit changes no frame
        assert r["taps_disagreeing"] == 0
        if scene == 0:                                      # (a sphere emitter: rt_lit.h has answers to audit)
            assert r["taps_audited"] > 1000, r
    g.close()


def test_audited_taps_agree_on_random_scenes(sky):
    """The fuzz legs of tests/test_gpu_fuzz.py with the audit on: spheres and boxes around a sphere emitter, three scales."""
    rng = np.random.default_rng(5)
    audited = 0
    g = rt.Renderer(0)
    g.set_skybox(sky)
    g.set_tuning(poison_frame=True, audit_known_taps=-1)
    for case in range(24):
        scale = (0.5, 2.0, 8.0)[case % 3]
        objs = [dict(type="sphere", center=tuple(rng.uniform(-1, 1, 3) * scale + np.array([0, 3 * scale, 0])), radius=0.4 * scale,
                     albedo=(1, 1, 1), emission_power=4.0, emission_color=(1, 0.9, 0.8))]
        for _ in range(int(rng.integers(3, 14))):
            if rng.random() < 0.5:
                objs.append(dict(type="sphere", center=tuple(rng.uniform(-3, 3, 3) * scale), radius=float(rng.uniform(0.2, 1.2)) * scale,
                                 albedo=tuple(rng.uniform(0.1, 1, 3)), roughness=float(rng.random()), metallic=float(rng.random() < 0.3)))
            else:
                objs.append(dict(type="cube", origin=tuple(rng.uniform(-3, 2, 3) * scale), size=tuple(rng.uniform(0.1, 2.5, 3) * scale),
                                 albedo=tuple(rng.uniform(0.1, 1, 3)), roughness=float(rng.random()), reflectance=float(rng.random())))
        g.set_scene(make_scene(objs))
        g.set_camera(pos=(5 * scale, 4 * scale, 5 * scale), front=(-1, -0.7, -1))
        g.render(192, 108, 8, 5, seed=case)
        rc, r = g.last_launch_report()
        assert rc == 0, (case, rt.lib().rt_last_error())
        assert r["taps_disagreeing"] == 0, (case, r)
        audited += r["taps_audited"]
    assert audited > 10000, audited
    # large (culled) scenes classify per pixel when there are enough samples per pixel
    g.set_scene(large_scene(128, seed=17)); g.set_camera(**LARGE_SCENE_CAMERA)
    g.render(320, 180, 32, 4, seed=1)
    rc, r = g.last_launch_report()
    assert rc == 0 and r["taps_disagreeing"] == 0, r
    g.close()


def test_a_wrong_table_is_caught_by_the_audit(sky, scene_paths):
    W, H, spp, nb = 320, 180, 8, 5
    L = rt.lib()
    g = rt.Renderer(0)
    g.set_skybox(sky); g.set_camera()
    g.set_tuning(poison_frame=True, test_corrupt_lit_table=1, audit_known_taps=2)
    g.set_scene(scene_paths[0])                             # (the table is built here)
    out = np.empty((H, W, 3), np.float32)
    assert L.rt_render(g._ctx, g.params(W, H, spp, nb), out.ctypes.data) == ERR_DEVICE
    assert "contradict" in L.rt_last_error().decode()
    rc, r = g.last_launch_report()
    assert rc == ERR_DEVICE and 0 < r["taps_disagreeing"] <= r["taps_audited"]
    g.set_tuning(test_corrupt_lit_table=0)
    g.set_scene(scene_paths[0])
    g.render(W, H, spp, nb)
    assert g.last_launch_report()[1]["taps_disagreeing"] == 0
    g.close()


@pytest.mark.parametrize("compiled", [False, True])
def test_the_background_audit_looks_at_one_launch_in_61(sky, scene_paths, compiled):
    """rt_tuning.audit_known_taps = 0, the default (round 6; until then the default was "never"): every 61st launch of a context is rendered
    by the audit variant -- the embedded one of a shipped scene that is compiled, the generic kernel's otherwise; nothing is ever built
    for it -- with one answer of csrc/rt_lit.h in 8 re-traced and compared.  Frames are unchanged, the other launches audit nothing,
    RT_AUDIT_OFF audits none at all."""
    W, H, spp, nb = 160, 90, 4, 4
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera()
    if compiled:
        g.compile_scene()
    want = {s: g.render(W, H, spp, nb, seed=s, kernel=rt.KERNEL_SIMPLE) for s in (3, 4)}      # (the cross-check kernel does not account for itself: these launches count all the same)
    audited = []
    for k in range(2, 130):                                 # launches 2 ... 129 of the context: numbers 60 and 121 are the audit's
        got = g.render(W, H, spp, nb, seed=3 + (k & 1))
        assert (bits(got) == bits(want[3 + (k & 1)])).all(), k
        rc, r = g.last_launch_report()
        assert rc == 0 and r["taps_disagreeing"] == 0
        if r["taps_audited"]:
            audited.append(k)
    assert audited == [60, 121], audited
    g.set_tuning(audit_known_taps=rt.AUDIT_OFF)
    for k in range(130, 200):
        g.render(W, H, spp, nb, seed=3)
        assert g.last_launch_report()[1]["taps_audited"] == 0
    g.close()


def test_the_background_audit_catches_a_wrong_table(sky, scene_paths):
    """... and a table that is wrong (the fault injection of rt_hip_testing.h) does not survive a context's first 61 launches with the
    DEFAULT tuning: the launch the background audit picks is refused, RT_ERR_DEVICE, with the disagreeing taps counted."""
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True, test_corrupt_lit_table=1)
    g.set_scene(scene_paths[0]); g.set_skybox(sky); g.set_camera(); g.compile_scene()
    refused = []
    for k in range(61):
        try:
            g.render(160, 90, 4, 4, seed=k)
        except rt.RtError as e:
            refused.append((k, str(e)))
    assert len(refused) == 1 and refused[0][0] == 60 and "contradict the answer rt_lit.h gave" in refused[0][1], refused
    rc, r = g.last_launch_report()
    assert rc == ERR_DEVICE and 0 < r["taps_disagreeing"] <= r["taps_audited"]
    g.close()


def test_the_background_audit_compiles_nothing(sky, scene_paths, oracle):
    """A compiled scene that is NOT among the kernels embedded in the library (one coordinate of scene_0 changed: hiprtc compiled it): the
    launch the background audit picks is rendered by the GENERIC kernel's audit variant -- no second hiprtc build in front of a frame --,
    audits its taps, and the frame is the compiled kernel's, bit for bit."""
    import ctypes as C
    from rtlibs import scene_objects
    rc, buf = rt.parse_scene_file(scene_paths[0])
    assert rc == 0
    objs, n = scene_objects(buf)
    objs[6]["geom"][1] = np.float32(1.25)                  # the first sphere, lifted a little
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(sky); g.set_scene(buf); g.set_camera()
    g.compile_scene()
    assert g.compiled_scene_info().startswith("hiprtc")

    def cached():
        a, b = C.c_int(), C.c_int()
        rt.lib().rt_compiled_scene_counts(C.byref(a), C.byref(b))
        return a.value + b.value
    before = cached()
    W, H, spp, nb = 160, 90, 4, 4
    want = g.render(W, H, spp, nb, seed=5)
    audited = []
    for k in range(1, 62):
        got = g.render(W, H, spp, nb, seed=5)
        assert (bits(got) == bits(want)).all(), k
        rc, r = g.last_launch_report()
        assert rc == 0 and r["taps_disagreeing"] == 0
        if r["taps_audited"]:
            audited.append(k)
    assert audited == [60] and cached() == before          # launch 60 audited, and nothing was compiled for it
    g.close()
    oracle.set_skybox(sky); oracle.set_scene(buf); oracle.set_camera()
    assert (bits(want) == bits(oracle.render_counter(W, H, spp, nb, seed=5))).all()
