"""Scenes of more than 64 objects: the cluster cull of csrc/rt_cull.h must not change a bit.  The culled
kernels against the same kernels with every object tested (rt_tuning.test_every_object) and against the oracle, on random
scenes that mix small and large objects, cameras inside and outside the scene, and the largest scene the reference takes."""
import os

import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import LARGE_SCENE_CAMERA, bits, large_scene, stick_scene, synthetic_skybox

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    r.set_tuning(poison_frame=True)
    yield r
    r.close()


CAMERAS = [LARGE_SCENE_CAMERA,                                                        # outside, looking in
           dict(pos=(0.3, 0.2, -0.4), front=(0.2, -0.1, 1), up=(0, 1, 0), fov=1.2),    # in the middle of the objects
           dict(pos=(-9, 9, 2), front=(1, -1, 0), up=(0, 1, 0), fov=0.8)]              # axis-aligned components in the view direction


@pytest.mark.parametrize("n", [32, 47, 64, 65, 100, 256, 777, 1024])
def test_culled_equals_every_object_equals_oracle(gpu, oracle, n):
    sky = synthetic_skybox(32, seed=n)
    scene = large_scene(n, seed=n)
    gpu.set_skybox(sky); gpu.set_scene(scene)
    oracle.set_skybox(sky); oracle.set_scene(scene)
    W, H, spp, nb = 96, 54, 3, 5
    for cam in CAMERAS:
        gpu.set_camera(**cam); oracle.set_camera(**cam)
        gpu.set_tuning(test_every_object=False)
        culled = gpu.render(W, H, spp, nb, seed=n)
        gpu.set_tuning(test_every_object=True)
        plain = gpu.render(W, H, spp, nb, seed=n)
        gpu.set_tuning(test_every_object=False)
        assert (bits(culled) == bits(plain)).all(), (n, cam)
        want = oracle.render_counter(W, H, spp, nb, seed=n, threads=min(os.cpu_count() or 1, 32))
        assert (bits(culled) == bits(want)).all(), (n, cam)
    gpu.set_camera(); oracle.set_camera()


def test_cull_fuzz_against_every_object(gpu):
    """Random large scenes at random scales and cameras: the culled frame is the every-object frame (GPU against GPU: cheap enough
    for many cases; the oracle pins a subset above)."""
    rng = np.random.default_rng(2024)
    sky = synthetic_skybox(16, seed=5)
    gpu.set_skybox(sky)
    for case in range(24):
        n = int(rng.integers(32, 400))
        extent = float(rng.choice([0.5, 3.0, 10.0, 30.0, 300.0, 5000.0]))      # (beyond 64 the cull's margin grows with the scene)
        gpu.set_scene(large_scene(n, seed=1000 + case, extent=extent, floor=bool(case & 1), light=bool(case & 2)))
        pos = rng.uniform(-1.5 * extent, 1.5 * extent, 3)
        front = -pos + rng.uniform(-0.3 * extent, 0.3 * extent, 3)
        gpu.set_camera(pos=tuple(pos), front=tuple(front), up=(0, 1, 0), fov=float(rng.uniform(0.5, 1.4)))
        W, H, spp, nb = 128, 72, 2, 6
        gpu.set_tuning(test_every_object=False)
        culled = gpu.render(W, H, spp, nb, seed=case)
        gpu.set_tuning(test_every_object=True)
        plain = gpu.render(W, H, spp, nb, seed=case)
        gpu.set_tuning(test_every_object=False)
        assert (bits(culled) == bits(plain)).all(), (case, n, extent)
    gpu.set_camera()


def test_groups_of_clusters_fuzz_against_every_object(gpu):
    """Round 6: from 60 clusters on (473 objects) the cluster boxes are reached through GROUPS of eight clusters whose (ray, group) pairs are
    dealt to the lanes (csrc/rt_device.h rt_group).  The boundary (59 / 60 clusters), cluster counts that leave the last group short (500
    objects: 63 clusters, the last group has 7; 1001: 126 clusters, 6), every scale of the margin, cameras inside and outside, sky-only
    and floorless scenes: the culled frame is the every-object frame."""
    rng = np.random.default_rng(6)
    sky = synthetic_skybox(16, seed=6)
    gpu.set_skybox(sky)
    sizes = [472, 473, 480, 481, 500, 633, 777, 1001, 1017, 1024]
    for case, n in enumerate(sizes + [int(x) for x in rng.integers(473, 1025, 8)]):
        extent = float([10.0, 0.5, 3.0, 30.0, 300.0, 5000.0][case % 6])
        gpu.set_scene(large_scene(n, seed=2000 + case, extent=extent, floor=bool(case & 1), light=bool(case & 2)))
        pos = rng.uniform(-1.5 * extent, 1.5 * extent, 3)
        front = -pos + rng.uniform(-0.3 * extent, 0.3 * extent, 3)
        gpu.set_camera(pos=tuple(pos), front=tuple(front), up=(0, 1, 0), fov=float(rng.uniform(0.5, 1.4)))
        W, H, spp, nb = 128, 72, 2, 6
        gpu.set_tuning(test_every_object=False)
        culled = gpu.render(W, H, spp, nb, seed=case)
        gpu.set_tuning(test_every_object=True)
        plain = gpu.render(W, H, spp, nb, seed=case)
        gpu.set_tuning(test_every_object=False)
        assert (bits(culled) == bits(plain)).all(), (case, n, extent)
    gpu.set_camera()


@pytest.mark.parametrize("n", [200, 600, 1024])
def test_objects_that_span_the_scene(gpu, oracle, n):
    """Thin boxes that run the whole length of the scene: every cluster's and group's box covers most of it, a ray passes most groups, and
    a wave with more than RT_GROUP_PAIRS_MAX (ray, group) pairs asks every cluster box instead of dealing them (rt_kernels.hip step 1)."""
    sky = synthetic_skybox(32, seed=n)
    scene = stick_scene(n, seed=n)
    gpu.set_skybox(sky); gpu.set_scene(scene)
    oracle.set_skybox(sky); oracle.set_scene(scene)
    W, H, spp, nb = 96, 54, 2, 5
    for cam in CAMERAS[:2]:
        gpu.set_camera(**cam); oracle.set_camera(**cam)
        gpu.set_tuning(test_every_object=False)
        culled = gpu.render(W, H, spp, nb, seed=n)
        gpu.set_tuning(test_every_object=True)
        plain = gpu.render(W, H, spp, nb, seed=n)
        gpu.set_tuning(test_every_object=False)
        assert (bits(culled) == bits(plain)).all(), (n, cam)
        want = oracle.render_counter(W, H, spp, nb, seed=n, threads=min(os.cpu_count() or 1, 32))
        assert (bits(culled) == bits(want)).all(), (n, cam)
    gpu.set_camera(); oracle.set_camera()


@pytest.mark.parametrize("n", [473, 500])
def test_groups_boundary_against_the_oracle(gpu, oracle, n):
    """... and the oracle itself at the first cluster counts that use the groups (60 clusters; 63 with a short last group)."""
    sky = synthetic_skybox(32, seed=n)
    scene = large_scene(n, seed=n)
    gpu.set_skybox(sky); gpu.set_scene(scene)
    oracle.set_skybox(sky); oracle.set_scene(scene)
    W, H, spp, nb = 96, 54, 2, 5
    for cam in CAMERAS[:2]:
        gpu.set_camera(**cam); oracle.set_camera(**cam)
        got = gpu.render(W, H, spp, nb, seed=n)
        want = oracle.render_counter(W, H, spp, nb, seed=n, threads=min(os.cpu_count() or 1, 32))
        assert (bits(got) == bits(want)).all(), (n, cam)
    gpu.set_camera(); oracle.set_camera()


def test_camera_far_outside_tests_every_object(gpu, oracle):
    """A camera farther out than twice the scene's extent: its waves take the plain loop (the margins are proved for origins
    within that); the frame is the oracle's either way."""
    sky = synthetic_skybox(16, seed=9)
    scene = large_scene(128, seed=4, extent=2.0)
    cam = dict(pos=(40, 25, 40), front=(-1, -0.6, -1), up=(0, 1, 0), fov=0.15)
    for r in (gpu, oracle):
        r.set_skybox(sky); r.set_scene(scene); r.set_camera(**cam)
    got = gpu.render(96, 54, 2, 4, seed=3)
    assert (bits(got) == bits(oracle.render_counter(96, 54, 2, 4, seed=3))).all()
    gpu.set_camera(); oracle.set_camera()


def test_largest_scene_at_full_hd_rows_against_the_oracle(gpu, oracle):
    """1024 objects at 1920x1080: the culled frame equals the every-object frame, and rows of it the oracle's."""
    sky = synthetic_skybox(32, seed=3)
    scene = large_scene(1024, seed=17)
    cam = CAMERAS[0]
    for r in (gpu, oracle):
        r.set_skybox(sky); r.set_scene(scene); r.set_camera(**cam)
    W, H, spp, nb = 1920, 1080, 2, 5
    culled = gpu.render(W, H, spp, nb, seed=1)
    gpu.set_tuning(test_every_object=True)
    plain = gpu.render(W, H, spp, nb, seed=1)
    gpu.set_tuning(test_every_object=False)
    assert (bits(culled) == bits(plain)).all()
    rows = [0, 137, 540, 811, 1079]
    want = oracle.render_counter_rows(W, H, spp, nb, rows, seed=1, threads=min(os.cpu_count() or 1, 64))
    for r_, v in want.items():
        assert (bits(culled[r_]) == bits(v)).all(), r_
    gpu.set_camera(); oracle.set_camera()


def test_a_compiled_scene_keeps_its_compiled_kernel(gpu, oracle):
    """32 ... 64 objects: rt_set_scene builds clusters and the generic path is the culled one; a scene the host has compiled
    (rt_compile_scene, up to 64 objects) renders with its compiled kernel -- the same frame either way, and the oracle's."""
    n = 48
    sky = synthetic_skybox(32, seed=n)
    scene = large_scene(n, seed=n)
    gpu.set_skybox(sky); gpu.set_scene(scene); gpu.set_camera(**LARGE_SCENE_CAMERA)
    oracle.set_skybox(sky); oracle.set_scene(scene); oracle.set_camera(**LARGE_SCENE_CAMERA)
    W, H, spp, nb = 96, 54, 3, 5
    culled = gpu.render(W, H, spp, nb, seed=n)
    gpu.compile_scene()
    assert gpu.scene_is_compiled()
    compiled = gpu.render(W, H, spp, nb, seed=n)
    want = oracle.render_counter(W, H, spp, nb, seed=n, threads=min(os.cpu_count() or 1, 32))
    assert (bits(culled) == bits(want)).all() and (bits(compiled) == bits(want)).all()
    gpu.set_scene(scene)                      # drops the compiled kernel
    gpu.set_camera(); oracle.set_camera()


@pytest.mark.parametrize("n", [40, 100, 600])
def test_interactive_ladder_on_a_large_scene(gpu, oracle, n):
    """The culled kernels under the interactive protocol: passes of one sample per pixel (every lane takes a pixel for itself),
    the low-resolution steps, and several full-resolution passes in one launch (rt_progressive_passes) -- 40 objects (with the
    lit-taps table), 100, and 600 (one workgroup of twelve waves per CU) -- against the oracle's ladder."""
    from rtlibs import oracle_progressive
    sky = synthetic_skybox(32, seed=n)
    scene = large_scene(n, seed=n)
    gpu.set_skybox(sky); gpu.set_scene(scene); gpu.set_camera(**LARGE_SCENE_CAMERA)
    oracle.set_skybox(sky); oracle.set_scene(scene); oracle.set_camera(**LARGE_SCENE_CAMERA)
    W, H, init_scale, nb, seed = 80, 48, 4, 6, 11
    gpu.progressive_begin(W, H, init_scale=init_scale, max_bounces=nb, seed=seed)
    for _ in range(4):
        gpu.progressive_pass()
    want, _, _, _ = oracle_progressive(oracle, W, H, init_scale, 4, nb, seed)
    assert (bits(gpu.progressive_resolve()) == bits(want)).all(), n
    gpu.progressive_passes(12)
    want, _, count, _ = oracle_progressive(oracle, W, H, init_scale, 16, nb, seed)
    st = gpu.progressive_state()
    assert st["passes"] == 16 and np.float32(st["count"]) == np.float32(count)
    assert (bits(gpu.progressive_resolve()) == bits(want)).all(), n
    gpu.set_camera(); oracle.set_camera()


@pytest.mark.parametrize("extent", [500.0, 20000.0])
def test_scenes_with_large_coordinates_are_culled_too(gpu, oracle, extent):
    """Coordinates beyond 64: the conservative margins are proportional to the scene's extent (rt_cull.h), the frame is the
    every-object frame and the oracle's."""
    n = 200
    sky = synthetic_skybox(32, seed=n)
    scene = large_scene(n, seed=n, extent=extent)
    cam = dict(pos=(1.4 * extent, 0.9 * extent, 1.4 * extent), front=(-1, -0.5, -1), up=(0, 1, 0), fov=1.0)
    gpu.set_skybox(sky); gpu.set_scene(scene); gpu.set_camera(**cam)
    oracle.set_skybox(sky); oracle.set_scene(scene); oracle.set_camera(**cam)
    W, H, spp, nb = 96, 54, 3, 5
    gpu.set_tuning(test_every_object=False)
    culled = gpu.render(W, H, spp, nb, seed=n)
    gpu.set_tuning(test_every_object=True)
    plain = gpu.render(W, H, spp, nb, seed=n)
    gpu.set_tuning(test_every_object=False)
    want = oracle.render_counter(W, H, spp, nb, seed=n, threads=min(os.cpu_count() or 1, 32))
    assert (bits(culled) == bits(plain)).all() and (bits(culled) == bits(want)).all(), extent
    gpu.set_camera(); oracle.set_camera()
