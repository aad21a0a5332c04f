"""Live comparison of the oracle with the compiled reference (oracle/_ref), run wherever that build
exists (the build container; the libraries also travel to the GPU box).  Skipped otherwise -- the
committed fixtures in tests/golden/ pin the same behaviour."""
import ctypes as C

import numpy as np
import pytest

from rtlibs import Ref, bits, ref_available, synthetic_skybox

pytestmark = pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def refs():
    r, rb = Ref(), Ref(bounce_patch=True)
    sky = synthetic_skybox(32, seed=7)
    r.set_skybox(sky); rb.set_skybox(sky)
    return r, rb, sky


def test_stream_scale_ladder(refs, oracle, scene_paths):
    """worker()'s progressive scale ladder (init_scale, halved after each publish) in stream mode."""
    r, _, sky = refs
    oracle.set_skybox(sky)
    r.L.ref_render_stream_scaled.argtypes = [C.c_int] * 4 + [C.c_void_p] * 2
    for path in scene_paths:
        r.load_scene(path); oracle.load_scene(path)
        for (W, H, passes, s0) in [(64, 36, 4, 4), (80, 40, 5, 8), (50, 30, 3, 4), (64, 64, 6, 16)]:
            r.set_rng(0)
            fr = np.zeros((H, W, 3), np.float32); ac = np.zeros((H, W, 3), np.float32)
            r.L.ref_render_stream_scaled(W, H, passes, s0, fr.ctypes.data, ac.ctypes.data)
            fo, ao, so = oracle.render_stream(W, H, passes=passes, init_scale=s0, max_bounces=10, state=0)
            assert (bits(ao) == bits(ac)).all() and (bits(fo) == bits(fr)).all() and so == r.get_rng()


def test_counter_frames_all_bounce_limits(refs, oracle, scene_paths):
    _, rb, sky = refs
    oracle.set_skybox(sky)
    for path in scene_paths:
        rb.load_scene(path); oracle.load_scene(path)
        for nb in (1, 2, 3, 4, 6, 8, 10):
            rb.set_bounce_limit(nb)
            assert (bits(oracle.render_counter(72, 40, 3, nb, seed=11)) == bits(rb.render_counter(72, 40, 3, seed=11))).all()
    rb.set_bounce_limit(10)


def test_random_rays_and_directions(refs, oracle, scene_paths):
    r, _, sky = refs
    oracle.set_skybox(sky)
    rng = np.random.default_rng(3)
    for path in scene_paths:
        r.load_scene(path); oracle.load_scene(path)
        for k in range(1500):
            o = rng.uniform(-2, 8, 3).astype(np.float32)
            d = rng.normal(size=3).astype(np.float32)
            if k % 5 == 0:
                d[rng.integers(3)] = 0
            if k % 9 == 0:
                o = np.round(o)
            i1, h1 = oracle.trace_ray(o, d); i2, h2 = r.trace_ray(o, d)
            assert i1 == i2 and (bits(h1) == bits(h2)).all()
    for k in range(1500):
        d = r.normalize(rng.normal(size=3).astype(np.float32))
        assert (bits(oracle.sample_cubemap(d)) == bits(r.sample_cubemap(d))).all()


def test_bounce_patch_is_the_only_difference(refs, scene_paths):
    r, rb, _ = refs
    rb.set_bounce_limit(10)
    for path in scene_paths:
        r.load_scene(path); rb.load_scene(path)
        assert (bits(r.render_counter(48, 27, 2, seed=4)) == bits(rb.render_counter(48, 27, 2, seed=4))).all()
