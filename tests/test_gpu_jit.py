"""rt_compile_scene(): the scene-specialised (hiprtc-compiled) trace kernel must produce the same bits as
the generic kernels and the oracle, must be dropped when the scene changes, and must refuse what it
cannot specialise while leaving rendering intact."""
import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import bits, make_scene, synthetic_skybox

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    r.set_tuning(poison_frame=True)      # a pixel a launch fails to write must not pass as a leftover of an earlier frame
    yield r
    r.close()


@pytest.mark.parametrize("scene_i,bounces", [(0, 4), (0, 10), (1, 8), (2, 8)])
def test_compiled_scene_is_bit_identical(gpu, oracle, scene_paths, scene_i, bounces):
    sky = rt.load_skybox()
    gpu.set_skybox(sky); oracle.set_skybox(sky)
    gpu.set_scene(scene_paths[scene_i]); oracle.load_scene(scene_paths[scene_i])
    gpu.set_camera(); oracle.set_camera()
    assert not gpu.scene_is_compiled()
    generic = gpu.render(192, 108, 6, bounces, seed=4)
    gpu.compile_scene()
    assert gpu.scene_is_compiled()
    compiled = gpu.render(192, 108, 6, bounces, seed=4)
    want = oracle.render_counter(192, 108, 6, bounces, seed=4)
    assert (bits(compiled) == bits(generic)).all()
    assert (bits(compiled) == bits(want)).all()


def test_compiled_scene_full_size_and_strips(gpu, scene_paths):
    import torch
    sky = rt.load_skybox()
    gpu.set_skybox(sky); gpu.set_scene(scene_paths[0]); gpu.set_camera()
    W, H, spp, nb = 1920, 1080, 4, 4
    generic = gpu.render(W, H, spp, nb, seed=0)
    gpu.compile_scene()
    compiled = gpu.render(W, H, spp, nb, seed=0)
    assert (bits(compiled) == bits(generic)).all()
    world, rb = 4, 8
    rows = rt.strip_rows(H, rb, world)
    strips = torch.zeros((world, rows, W, 3), dtype=torch.float32, device="cuda:0")
    for rank in range(world):
        gpu.render_device(gpu.params(W, H, spp, nb, seed=0, row_block=rb, rank=rank, world=world), strips[rank].data_ptr())
    gpu.synchronize()
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    gpu.deinterleave_device(strips.data_ptr(), frame.data_ptr(), W, H, rb, world)
    gpu.synchronize()
    assert (bits(frame.cpu().numpy()) == bits(generic)).all()


def test_set_scene_drops_the_compiled_kernel(gpu, oracle, scene_paths):
    sky = synthetic_skybox(32, seed=7)
    gpu.set_skybox(sky); oracle.set_skybox(sky)
    gpu.set_scene(scene_paths[0]); gpu.compile_scene()
    assert gpu.scene_is_compiled()
    gpu.set_scene(scene_paths[1]); oracle.load_scene(scene_paths[1])
    assert not gpu.scene_is_compiled()
    assert (bits(gpu.render(96, 54, 3, 6, seed=2)) == bits(oracle.render_counter(96, 54, 3, 6, seed=2))).all()


def test_random_scene_and_edge_geometry(gpu, oracle):
    """Random boxes/spheres sharing many planes (grid-aligned), an emissive cube, camera on a slab plane."""
    sky = synthetic_skybox(32, seed=5)
    rng = np.random.default_rng(9)
    objs = []
    for k in range(40):
        if k % 4 == 3:
            objs.append(dict(type="sphere", center=rng.integers(-3, 6, 3).astype(float), radius=float(rng.choice([0.5, 1.0])),
                             albedo=rng.uniform(0, 1, 3), roughness=rng.uniform(0, 1), metallic=float(k % 8 == 3)))
        else:
            objs.append(dict(type="cube", origin=rng.integers(-4, 6, 3).astype(float), size=rng.choice([0.0, 0.5, 1.0, 2.0], 3),
                             albedo=rng.uniform(0, 1, 3), roughness=rng.uniform(0, 1), reflectance=rng.uniform(0, 1),
                             metallic=float(k % 5 == 0), emission_power=4.0 if k == 6 else 0.0))
    scene = make_scene(objs)
    cam = dict(pos=(6, 4, 8), front=(-0.6, -0.35, -1), up=(0, 1, 0), fov=1.0)     # pos components sit on slab planes
    for r in (gpu, oracle):
        r.set_skybox(sky); r.set_scene(scene); r.set_camera(**cam)
    gpu.compile_scene()
    g = gpu.render(131, 77, 3, 7, seed=13)
    c = oracle.render_counter(131, 77, 3, 7, seed=13)
    assert (bits(g) == bits(c)).all()
    gpu.set_camera(); oracle.set_camera()


def test_refuses_what_it_cannot_specialise(gpu, oracle):
    sky = synthetic_skybox(16, seed=1)
    gpu.set_skybox(sky); oracle.set_skybox(sky)
    big = make_scene([dict(type="sphere", center=(i % 10, i // 10, 0), radius=0.3) for i in range(65)])
    gpu.set_scene(big); oracle.set_scene(big)
    with pytest.raises(rt.RtError):
        gpu.compile_scene()
    assert not gpu.scene_is_compiled()
    assert (bits(gpu.render(40, 30, 2, 3, seed=1)) == bits(oracle.render_counter(40, 30, 2, 3, seed=1))).all()
    empty = make_scene([])
    gpu.set_scene(empty)
    with pytest.raises(rt.RtError):
        gpu.compile_scene()


def test_slab_planes_outside_the_tuned_window(gpu, oracle):
    """Plane coordinates the shared-reciprocal slab test does not accept -- a non-zero coordinate below 2^-76
    and a negative zero -- must send the scene through the reference-order kernel (still bit-exact) and make
    rt_compile_scene refuse it."""
    sky = synthetic_skybox(16, seed=3)
    for origin in ((1e-30, 0.0, 0.0), (-0.0, 0.0, 0.0)):
        scene = make_scene([dict(type="cube", origin=origin, size=(2.0, 1.0, 2.0), roughness=0.5),
                            dict(type="cube", origin=(-3, -0.1, -3), size=(9, 0.1, 9), roughness=1.0),
                            dict(type="sphere", center=(1.0, 3.0, 1.0), radius=0.7, emission_power=3.0)])
        cam = dict(pos=(0.0, 2.0, 5.0), front=(0.1, -0.3, -1.0), up=(0, 1, 0), fov=1.0)    # pos.x on the plane x = +-0
        for r in (gpu, oracle):
            r.set_skybox(sky); r.set_scene(scene); r.set_camera(**cam)
        with pytest.raises(rt.RtError):
            gpu.compile_scene()
        g = gpu.render(64, 48, 3, 5, seed=21)
        c = oracle.render_counter(64, 48, 3, 5, seed=21)
        assert (bits(g) == bits(c)).all()
    gpu.set_camera(); oracle.set_camera()


def test_ray_origins_on_slab_planes_stay_exact(gpu, oracle):
    """Camera exactly on several slab planes of the scene (numerators +0 for every primary ray), including two
    planes of one box whose entry parameters are zeros of opposite sign."""
    sky = synthetic_skybox(16, seed=4)
    scene = make_scene([dict(type="cube", origin=(1.0, 0.0, -2.0), size=(1.0, 1.0, 1.0), roughness=0.3),       # x in [1,2], z in [-2,-1]
                        dict(type="cube", origin=(0.0, -0.5, -4.0), size=(1.0, 1.0, 3.0), metallic=1.0),       # hi.x = 1, hi.z = -1
                        dict(type="cube", origin=(-4, -0.6, -6), size=(9, 0.1, 9), roughness=1.0),
                        dict(type="sphere", center=(1.5, 3.0, -1.5), radius=0.6, emission_power=4.0)])
    for pos, front in (((1.0, 0.5, -1.0), (0.3, -0.1, -1.0)), ((1.0, 0.5, -1.0), (-0.3, 0.1, 1.0)), ((2.0, 1.0, 1.0), (-0.4, -0.3, -1.0))):
        for r in (gpu, oracle):
            r.set_skybox(sky); r.set_scene(scene); r.set_camera(pos=pos, front=front, up=(0, 1, 0), fov=1.2)
        generic = gpu.render(80, 60, 3, 6, seed=8)
        c = oracle.render_counter(80, 60, 3, 6, seed=8)
        assert (bits(generic) == bits(c)).all()
        gpu.compile_scene()
        assert (bits(gpu.render(80, 60, 3, 6, seed=8)) == bits(c)).all()
    gpu.set_camera(); oracle.set_camera()


def test_compiled_scenes_are_kept_and_shared(scene_paths):
    """rt_jit.cpp keeps every compiled scene for the life of the process (a module is never unloaded: DESIGN.md section 10) and
    hands it to any context that asks for the same scene, options and device: the second compilation costs nothing, contexts
    that share a module render the same bits, and a context that moves to another scene and back gets both kernels right --
    also after the context that compiled first is gone, and after a frame loop has run the kernel on both streams."""
    import time
    from ray_tracing_amd.frames import FrameLoop
    sky = rt.load_skybox()
    a = rt.Renderer(0)
    a.set_skybox(sky); a.set_scene(scene_paths[1]); a.set_camera()
    want1 = a.render(160, 90, 5, 6, seed=9)                     # generic kernel
    a.compile_scene()
    assert (bits(a.render(160, 90, 5, 6, seed=9)) == bits(want1)).all()
    b = rt.Renderer(0)
    b.set_skybox(sky); b.set_scene(scene_paths[1]); b.set_camera()
    t0 = time.perf_counter(); b.compile_scene(); shared = time.perf_counter() - t0
    assert shared < 0.05, shared                                # (a compilation takes 0.4 s)
    assert (bits(b.render(160, 90, 5, 6, seed=9)) == bits(want1)).all()
    loop = FrameLoop(a, 640, 360, 16, 6); loop.run(range(6)); loop.close()     # both streams, frames in flight
    a.close()                                                   # the module stays: b still runs it
    assert (bits(b.render(160, 90, 5, 6, seed=9)) == bits(want1)).all()
    b.set_scene(scene_paths[2]); assert not b.scene_is_compiled()
    want2 = b.render(160, 90, 5, 6, seed=9)
    b.compile_scene()
    assert (bits(b.render(160, 90, 5, 6, seed=9)) == bits(want2)).all()
    b.set_scene(scene_paths[1])
    t0 = time.perf_counter(); b.compile_scene(); back = time.perf_counter() - t0
    assert back < 0.05, back
    assert (bits(b.render(160, 90, 5, 6, seed=9)) == bits(want1)).all()
    m = rt.MultiRenderer([0])                                   # what used to fault after an unload: other code is loaded now (RCCL's)
    m.set_tuning(force_collective=1)
    m.set_scene(scene_paths[1]); m.set_skybox(sky); m.set_camera()
    assert (bits(m.render(160, 90, 5, 6, seed=9)) == bits(want1)).all()
    m.close(); b.close()


@pytest.mark.parametrize("scene_i,bounces", [(0, 4), (1, 8), (2, 8)])
def test_embedded_kernel_equals_hiprtc_equals_generic(oracle, scene_paths, scene_i, bounces):
    """The kernels of the shipped scenes are compiled when the library is built and embedded in it (csrc/Makefile): the one a
    default rt_compile_scene() loads, the one hiprtc makes at run time (any extra option bypasses the embedded table) and the
    generic kernel render the same bits -- and the oracle's."""
    import time
    sky = rt.load_skybox()
    W, H, spp, seed = 256, 144, 8, 12
    frames, info = {}, {}
    for name, flags in (("generic", None), ("embedded", ""), ("hiprtc", "-DRT_FORCE_HIPRTC")):
        g = rt.Renderer(0)
        g.set_tuning(poison_frame=True, jit_flags=flags or None)
        g.set_skybox(sky); g.set_scene(scene_paths[scene_i]); g.set_camera()
        if flags is not None:
            t0 = time.perf_counter()
            g.compile_scene()
            info[name] = (g.compiled_scene_info(), time.perf_counter() - t0)
        frames[name] = g.render(W, H, spp, bounces, seed=seed)
        g.close()
    assert info["embedded"][0].startswith("embedded"), info
    assert info["hiprtc"][0].startswith("hiprtc"), info
    assert info["embedded"][1] < 0.1, info                 # a module load, not a compilation
    oracle.set_skybox(sky); oracle.load_scene(scene_paths[scene_i]); oracle.set_camera()
    want = oracle.render_counter(W, H, spp, bounces, seed=seed)
    for name in ("generic", "embedded", "hiprtc"):
        assert (bits(frames[name]) == bits(want)).all(), name


def test_an_edited_shipped_scene_is_not_mistaken_for_the_embedded_one(oracle, scene_paths):
    """The embedded table is keyed by the packed scene, not by a file name: one coordinate changed -> hiprtc compiles it."""
    from rtlibs import scene_objects
    sky = rt.load_skybox()
    rc, buf = rt.parse_scene_file(scene_paths[0])
    assert rc == 0
    objs, n = scene_objects(buf)
    objs[6]["geom"][1] = np.float32(1.25)                  # the first sphere, lifted a little
    g = rt.Renderer(0)
    g.set_tuning(poison_frame=True)
    g.set_skybox(sky); g.set_scene(buf); g.set_camera()
    g.compile_scene()
    assert g.compiled_scene_info().startswith("hiprtc")
    got = g.render(192, 108, 4, 4, seed=2)
    g.close()
    oracle.set_skybox(sky); oracle.set_scene(buf); oracle.set_camera()
    assert (bits(got) == bits(oracle.render_counter(192, 108, 4, 4, seed=2))).all()


def test_compiled_scene_cache_is_bounded():
    """A host that edits its scene and recompiles makes a new cache key per edit: the cache keeps at most its cap, evicted
    modules are parked (still loaded: the contexts that use them go on), nothing is unloaded."""
    import ctypes as C
    L = rt.lib()
    L.rt_compiled_scene_counts.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rt_compiled_scene_cache_cap.argtypes = [C.c_int]

    def counts():
        a, b = C.c_int(), C.c_int()
        L.rt_compiled_scene_counts(C.byref(a), C.byref(b))
        return a.value, b.value
    old_cap = L.rt_compiled_scene_cache_cap(4)
    try:
        sky = synthetic_skybox(16, seed=1)
        g = rt.Renderer(0)
        g.set_tuning(poison_frame=True)
        g.set_skybox(sky); g.set_camera()
        cached0, parked0 = counts()
        first = None
        for k in range(7):
            g.set_scene(make_scene([dict(type="sphere", center=(0, 0, 0), radius=1.0 + 0.125 * k, albedo=(.5, .5, .5)),
                                    dict(type="cube", origin=(-4, -2, -4), size=(8, .5, 8), albedo=(.8, .2, .2))]))
            g.compile_scene()
            if first is None:
                first = g.render(96, 54, 2, 3, seed=1)
        cached, parked = counts()
        assert cached <= 4 and parked >= parked0 + 3, (cached0, parked0, cached, parked)     # seven new keys into a cache of four
        # the first scene's module was evicted; compiling it again works (a new module) and renders the same bits
        g.set_scene(make_scene([dict(type="sphere", center=(0, 0, 0), radius=1.0, albedo=(.5, .5, .5)),
                                dict(type="cube", origin=(-4, -2, -4), size=(8, .5, 8), albedo=(.8, .2, .2))]))
        g.compile_scene()
        assert (bits(g.render(96, 54, 2, 3, seed=1)) == bits(first)).all()
        g.close()
    finally:
        L.rt_compiled_scene_cache_cap(old_cap)
