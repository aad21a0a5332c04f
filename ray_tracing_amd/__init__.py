"""ray_tracing_amd -- MI355X (gfx950) back end for the path-tracing hot path of cozis/ray_tracing.

The product is ``librt_hip.so`` (hand-written HIP kernels behind a C ABI, see include/rt_hip.h).
This module is only a thin ctypes binding over that ABI for tests, bench.py and Python hosts; it
contains no rendering logic and no CPU fallback: if the library is missing it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")
ROOT = os.path.dirname(_HERE)
DATA_DIR = os.path.join(ROOT, "data")

MAX_OBJECTS = 1024
SCENE_BYTES = 68 * MAX_OBJECTS + 4          # sizeof(Scene), reference scene.h:33-36
FACE_NAMES = ["front", "back", "left", "right", "top", "bottom"]   # CubeFace order

KERNEL_AUTO, KERNEL_SIMPLE, KERNEL_WAVEFRONT = 0, 1, 2


class RtError(RuntimeError):
    pass


class Vector3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", Vector3), ("front", Vector3), ("up", Vector3), ("fov", C.c_float)]


class CameraBasis(C.Structure):
    _fields_ = [("pos", Vector3), ("lower_left_corner", Vector3), ("horizontal", Vector3), ("vertical", Vector3)]


class Cubemap(C.Structure):
    _fields_ = [("data", C.c_void_p * 6), ("w", C.c_int), ("h", C.c_int), ("chan", C.c_int)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_bounces", C.c_int),
                ("seed", C.c_uint64), ("row_block", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("kernel", C.c_int)]


class Tuning(C.Structure):
    """rt_tuning (include/rt_hip.h): `size` is set by rt_default_tuning() and checked by rt_set_tuning()."""
    _fields_ = [("size", C.c_size_t), ("dequeue_shards", C.c_int), ("workgroups_per_cu", C.c_int), ("jit_waves_per_simd", C.c_int),
                ("audit_known_taps", C.c_int), ("jit_flags", C.c_char_p)]


class TestKnobs(C.Structure):
    """rt_test_knobs (include/rt_hip_testing.h): the test suite's knobs and fault injections."""
    __test__ = False
    _fields_ = [("size", C.c_size_t), ("force_collective", C.c_int), ("poison_frame", C.c_int), ("trace_known_taps", C.c_int),
                ("test_every_object", C.c_int), ("test_drop_pixels", C.c_int), ("test_corrupt_lit_table", C.c_int)]


TEST_KNOB_NAMES = tuple(n for n, _ in TestKnobs._fields_ if n != "size")


class MultiPhases(C.Structure):
    """rt_multi_phases: where a device's step of the N-GPU frame loop goes (include/rt_hip.h)."""
    _fields_ = [("frames", C.c_int), ("step_ms", C.c_double), ("idle_ms", C.c_double), ("render_ms", C.c_double), ("gather_ms", C.c_double),
                ("deinterleave_ms", C.c_double), ("copy_ms", C.c_double)]


PHASES = ("idle_ms", "render_ms", "gather_ms", "deinterleave_ms", "copy_ms")


def attribute_phases(frames):
    """Where a device's time goes per step of a PIPELINED frame loop -- the Python twin of rt_multi_profile_collect() (rt_multi.cpp), for
    hosts that record their own events (multi_gpu.TiledFrame).  frames: per frame, the times (ms, one clock) at which its render
    stream reached the launch, its strip was rendered, its gather was done, its frame was de-interleaved, its copy to the host was
    done (a rank that does not assemble the frame repeats the gather's time).

    Several frames are in flight, so a frame's own phases say little about the step: frame f was rendered while frame f - 3 was
    gathered.  Instead every instant between the first and the last frame END is given to what the device was doing then, whichever
    frame it was for, by priority: a strip render in progress (from the stream reaching the launch to the end of its trace kernel)
    > a de-interleave > a copy to the host > a rendered strip waiting for its gather > nothing (idle: no launch was there to run).
    The five shares are disjoint and sum to the window; divided by the intervals in it they are ms per step."""
    out = {"frames": 0, "step_ms": 0.0, **{k: 0.0 for k in PHASES}}
    frames = [tuple(float(x) for x in f) for f in frames]
    if len(frames) < 2:
        return out
    ends = [f[4] for f in frames]
    t0, t1 = ends[0], max(ends)
    if not t1 > t0:
        return out
    # (class, from, to) of every activity; classes in priority order
    order = ("render_ms", "deinterleave_ms", "copy_ms", "gather_ms")
    spans = []
    for b, r, g, a, c in frames:
        spans += [("render_ms", b, r), ("gather_ms", r, g), ("deinterleave_ms", g, a), ("copy_ms", a, c)]
    cuts = sorted({t0, t1} | {t for _, a, b in spans for t in (a, b) if t0 < t < t1})
    for lo, hi in zip(cuts, cuts[1:]):
        mid = 0.5 * (lo + hi)
        active = {k for k, a, b in spans if a <= mid < b}
        key = next((k for k in order if k in active), "idle_ms")
        out[key] += hi - lo
    n = len(frames) - 1
    out["frames"] = n
    out["step_ms"] = (t1 - t0) / n
    for k in PHASES:
        out[k] /= n
    return out


def judge_phases(per_rank):
    """From every rank's shares: the critical rank -- the one whose device has the least slack: the smallest share of waiting for the
    gather or for work --, and what bounds the step there: `render` (its strips, their wait for workgroup slots included), `copy`
    (de-interleave + the frame's way to the host: rank 0 only), `host` (idle: no launch was there to run) -- or `gather` when even
    that rank mostly waits for the collective."""
    ranks = [(i, r) for i, r in enumerate(per_rank or []) if r.get("frames")]
    if not ranks:
        return {"critical_rank": None, "step_bound": None}
    slack = lambda r: r["gather_ms"] + r["idle_ms"]      # noqa: E731
    critical, c = min(ranks, key=lambda ir: (slack(ir[1]), ir[0]))
    shares = {"render": c["render_ms"], "gather": c["gather_ms"], "copy": c["deinterleave_ms"] + c["copy_ms"], "host": c["idle_ms"]}
    return {"critical_rank": critical, "step_bound": max(shares, key=shares.get)}


class LaunchReport(C.Structure):
    """rt_launch_report: what a launch left in its control words, and what it was expected to leave."""
    _fields_ = [("launch_checked", C.c_int), ("launch_id", C.c_uint), ("stamp", C.c_uint), ("cancelled", C.c_uint), ("waves_left", C.c_uint),
                ("primary_blocks_expected", C.c_uint), ("primary_blocks_done", C.c_uint),
                ("pixels_listed", C.c_ulonglong), ("pixels_fetched", C.c_ulonglong), ("pixels_written", C.c_ulonglong),
                ("taps_audited", C.c_ulonglong), ("taps_disagreeing", C.c_ulonglong)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


STREAM_LEGACY = C.c_void_p(-1).value      # RT_STREAM_LEGACY: the device's legacy null stream


class MouseState(C.Structure):
    _fields_ = [("first_mouse", C.c_int), ("yaw", C.c_float), ("pitch", C.c_float),
                ("last_x", C.c_float), ("last_y", C.c_float)]


# every symbol include/rt_hip.h and include/rt_hip_testing.h declare (tests check the library exports all of them)
EXPORTS = [
    "rt_abi_version", "rt_multi_profile_enable", "rt_multi_profile_collect", "rt_default_test_knobs", "rt_set_test_knobs", "rt_multi_set_test_knobs", "rt_default_params", "rt_create", "rt_destroy", "rt_last_error", "rt_set_scene", "rt_set_skybox",
    "rt_set_camera", "rt_set_tuning", "rt_default_tuning", "rt_compile_scene", "rt_scene_is_compiled", "rt_compiled_scene_info", "rt_compiled_scene_counts", "rt_compiled_scene_cache_cap", "rt_spec_stats_read", "rt_spec_symbol_read", "rt_render", "rt_render_device", "rt_stream", "rt_reserve", "rt_strip_rows", "rt_deinterleave_device", "rt_deinterleave_rotated_device", "rt_strip_of_rank",
    "rt_frame_submit", "rt_frame_submit_device", "rt_frame_wait", "rt_frame_poll", "rt_host_alloc", "rt_host_free",
    "rt_multi_frame_submit_device", "rt_multi_collective_info", "rt_multi_create_on_one_device",
    "rt_multi_frame_submit", "rt_multi_frame_wait", "rt_multi_frame_poll", "rt_profile_collect_span", "rt_profile_collect_split",
    "rt_progressive_begin_rank", "rt_progressive_resolve_device", "rt_multi_progressive_begin", "rt_multi_progressive_pass", "rt_multi_progressive_passes", "rt_progressive_passes",
    "rt_multi_progressive_resolve", "rt_multi_progressive_invalidate", "rt_multi_progressive_state",
    "rt_synchronize", "rt_cancel", "rt_was_cancelled", "rt_last_launch_report", "rt_launch_check_submit", "rt_launch_check_wait", "rt_primary_passes_run", "rt_progressive_begin", "rt_progressive_pass", "rt_progressive_resolve",
    "rt_progressive_invalidate", "rt_progressive_state", "rt_selftest", "rt_profile_enable", "rt_profile_collect", "rt_parse_scene_file",
    "rt_parse_scene_string", "rt_load_cubemap", "rt_free_cubemap", "rt_decode_jpeg_file",
    "rt_camera_default", "rt_camera_basis_for", "rt_mouse_state_default", "rt_move_camera",
    "rt_multi_create", "rt_multi_destroy", "rt_multi_size", "rt_multi_context", "rt_multi_set_scene", "rt_multi_set_skybox",
    "rt_multi_set_camera", "rt_multi_set_tuning", "rt_multi_compile_scene", "rt_multi_render",
    "rt_rotate_camera", "rt_path_seed", "rt_set_frame_sink", "rt_move_frame_to_the_gpu", "rt_write_ppm", "rt_write_png", "rt_screenshot",
]

_lib = None


def lib():
    """Load librt_hip.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C ray_tracing_amd/csrc`")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  If torch is going to share this process (bench.py, multi-GPU tests) it must be
    # loaded first so that librt_hip.so binds to the runtime torch uses; two runtimes in one process
    # leave the second one without devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.rt_last_error.restype = C.c_char_p
    L.rt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.rt_destroy.argtypes = [C.c_void_p]
    L.rt_set_scene.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_set_skybox.argtypes = [C.c_void_p, C.POINTER(Cubemap)]
    L.rt_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
    L.rt_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
    L.rt_default_tuning.argtypes = [C.POINTER(Tuning)]
    if hasattr(L, "rt_set_test_knobs"):
        L.rt_set_test_knobs.argtypes = [C.c_void_p, C.POINTER(TestKnobs)]
        L.rt_default_test_knobs.argtypes = [C.POINTER(TestKnobs)]
    L.rt_compile_scene.argtypes = [C.c_void_p]
    L.rt_scene_is_compiled.argtypes = [C.c_void_p]
    if hasattr(L, "rt_compiled_scene_info"):
        L.rt_compiled_scene_info.argtypes = [C.c_void_p]
        L.rt_compiled_scene_info.restype = C.c_char_p
    L.rt_default_params.argtypes = [C.POINTER(RenderParams), C.c_int, C.c_int, C.c_int, C.c_int]
    L.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
    L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p]
    L.rt_strip_rows.argtypes = [C.c_int, C.c_int, C.c_int]
    if hasattr(L, "rt_reserve"):
        L.rt_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int]
    if hasattr(L, "rt_stream"):
        L.rt_stream.argtypes = [C.c_void_p, C.c_int]
        L.rt_stream.restype = C.c_void_p
    L.rt_deinterleave_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    L.rt_deinterleave_rotated_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.rt_strip_of_rank.argtypes = [C.c_int, C.c_int]
    L.rt_synchronize.argtypes = [C.c_void_p]
    if hasattr(L, "rt_cancel"):
        L.rt_cancel.argtypes = [C.c_void_p]
        L.rt_was_cancelled.argtypes = [C.c_void_p]
        L.rt_primary_passes_run.argtypes = [C.c_void_p]; L.rt_primary_passes_run.restype = C.c_longlong
    if hasattr(L, "rt_multi_create"):       # (scripts/ab.py also loads older builds of the library)
        L.rt_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
        L.rt_multi_destroy.argtypes = [C.c_void_p]
        L.rt_multi_size.argtypes = [C.c_void_p]
        L.rt_multi_context.argtypes = [C.c_void_p, C.c_int]
        L.rt_multi_context.restype = C.c_void_p
        L.rt_multi_set_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_multi_set_skybox.argtypes = [C.c_void_p, C.POINTER(Cubemap)]
        L.rt_multi_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.rt_multi_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
        if hasattr(L, "rt_multi_set_test_knobs"):
            L.rt_multi_set_test_knobs.argtypes = [C.c_void_p, C.POINTER(TestKnobs)]
        L.rt_multi_compile_scene.argtypes = [C.c_void_p]
        L.rt_multi_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]
    if hasattr(L, "rt_frame_submit"):
        L.rt_frame_submit.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.c_void_p]
        if hasattr(L, "rt_frame_submit_device"):
            L.rt_frame_submit_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
            L.rt_multi_frame_submit_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
            L.rt_multi_collective_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.rt_multi_create_on_one_device.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        if hasattr(L, "rt_multi_profile_enable"):
            L.rt_multi_profile_enable.argtypes = [C.c_void_p, C.c_int]
            L.rt_multi_profile_collect.argtypes = [C.c_void_p, C.POINTER(MultiPhases), C.c_int]
        L.rt_frame_wait.argtypes = [C.c_void_p, C.c_int]
        L.rt_frame_poll.argtypes = [C.c_void_p, C.c_int]
        L.rt_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.rt_host_free.argtypes = [C.c_void_p]
        L.rt_multi_frame_submit.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.c_void_p]
        L.rt_multi_frame_wait.argtypes = [C.c_void_p, C.c_int]
        L.rt_multi_frame_poll.argtypes = [C.c_void_p, C.c_int]
        L.rt_profile_collect_span.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        if hasattr(L, "rt_profile_collect_split"):
            L.rt_profile_collect_split.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "rt_progressive_begin_rank"):
        L.rt_progressive_begin_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.rt_progressive_resolve_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_multi_progressive_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.rt_multi_progressive_pass.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        if hasattr(L, "rt_multi_progressive_passes"):
            L.rt_multi_progressive_passes.argtypes = [C.c_void_p, C.c_int]
        L.rt_multi_progressive_resolve.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_multi_progressive_invalidate.argtypes = [C.c_void_p]
        L.rt_multi_progressive_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    if hasattr(L, "rt_last_launch_report"):
        L.rt_last_launch_report.argtypes = [C.c_void_p, C.POINTER(LaunchReport)]
        L.rt_launch_check_submit.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rt_launch_check_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(LaunchReport)]
    L.rt_progressive_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.rt_progressive_pass.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    if hasattr(L, "rt_progressive_passes"):
        L.rt_progressive_passes.argtypes = [C.c_void_p, C.c_int]
    L.rt_progressive_resolve.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_progressive_invalidate.argtypes = [C.c_void_p]
    L.rt_progressive_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.rt_selftest.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.rt_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.rt_profile_collect.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.rt_parse_scene_file.argtypes = [C.c_char_p, C.c_void_p]
    L.rt_parse_scene_string.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    L.rt_load_cubemap.argtypes = [C.POINTER(Cubemap), C.POINTER(C.c_char_p)]
    L.rt_free_cubemap.argtypes = [C.POINTER(Cubemap)]
    L.rt_decode_jpeg_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rt_camera_default.argtypes = [C.POINTER(Camera)]
    L.rt_camera_basis_for.argtypes = [C.POINTER(Camera), C.c_float, C.POINTER(CameraBasis)]
    L.rt_mouse_state_default.argtypes = [C.POINTER(MouseState)]
    L.rt_move_camera.argtypes = [C.POINTER(Camera), C.c_int, C.c_float]
    L.rt_rotate_camera.argtypes = [C.POINTER(Camera), C.POINTER(MouseState), C.c_double, C.c_double]
    L.rt_path_seed.restype = C.c_uint64
    L.rt_path_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    L.rt_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.rt_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.rt_screenshot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RtError(f"{what} failed ({rc}): {lib().rt_last_error().decode(errors='replace')}")


# ---- host-side mirror of the reference loaders (no GPU needed) -----------------------------------

def parse_scene_file(path):
    """scene.c:611 parse_scene_file().  Returns (status, raw Scene buffer as uint8[69636])."""
    buf = np.zeros(SCENE_BYTES, dtype=np.uint8)
    rc = lib().rt_parse_scene_file(os.fsencode(path), buf.ctypes.data_as(C.c_void_p))
    return rc, buf


def parse_scene_string(text):
    if isinstance(text, str):
        text = text.encode()
    buf = np.zeros(SCENE_BYTES, dtype=np.uint8)
    rc = lib().rt_parse_scene_string(text, len(text), buf.ctypes.data_as(C.c_void_p))
    return rc, buf


def decode_jpeg(path):
    p = C.POINTER(C.c_uint8)()
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    _check(lib().rt_decode_jpeg_file(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h), C.byref(ch)),
           f"rt_decode_jpeg_file({path})")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, ch.value)).copy()
    finally:
        C.CDLL(None).free(p)


_skybox_cache = {}


def load_skybox(directory=None):
    """gpu_and_windowing.c:24 load_cubemap() on <dir>/{front,back,left,right,top,bottom}.jpg
    -> uint8 array (6, h, w, chan) in CubeFace order."""
    directory = directory or os.path.join(DATA_DIR, "skybox")
    if directory not in _skybox_cache:
        _skybox_cache[directory] = np.stack([decode_jpeg(os.path.join(directory, n + ".jpg")) for n in FACE_NAMES])
    return _skybox_cache[directory]


def default_camera():
    cam = Camera()
    lib().rt_camera_default(C.byref(cam))
    return cam


def camera_basis(cam, aspect):
    b = CameraBasis()
    lib().rt_camera_basis_for(C.byref(cam), aspect, C.byref(b))
    return b


def strip_rows(height, row_block, world):
    return lib().rt_strip_rows(height, row_block, world)


# ---- the GPU path ---------------------------------------------------------------------------------

FRAME_SLOTS = 8          # RT_FRAME_SLOTS
LAUNCH_SETS = 5          # RT_LAUNCH_SETS
CHECK_TICKETS = 8        # RT_CHECK_TICKETS
PENDING, CANCELLED = 2, 1
AUDIT_OFF = -2           # RT_AUDIT_OFF: rt_tuning.audit_known_taps, "never" (0 is the background audit: one launch in 61)


class HostFrame:
    """Page-locked host memory for one frame (rt_host_alloc), viewed as a float32 array (height, width, 3)."""

    def __init__(self, width, height):
        self._p = C.c_void_p()
        _check(lib().rt_host_alloc(C.byref(self._p), width * height * 12), "rt_host_alloc")
        self.ptr = self._p.value
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_float)), shape=(height, width, 3))

    def free(self):
        if self._p:
            self.array = None
            lib().rt_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _FrameQueue:
    """rt_frame_submit / rt_frame_wait / rt_frame_poll of a Renderer or MultiRenderer."""

    def frame_submit(self, params, slot, host):
        """host: a HostFrame (or any writable float32 array of the frame's size; page-locked memory lets the copy overlap)."""
        ptr = host.ptr if isinstance(host, HostFrame) else host.ctypes.data
        _check(getattr(lib(), self._prefix + "frame_submit")(self._handle(), C.byref(params), slot, C.c_void_p(ptr)), self._prefix + "frame_submit")

    def frame_submit_device(self, params, slot):
        """The frame stays in device memory: returns (device pointer of height x width x 3 floats, hipEvent_t handle recorded behind
        it); frame_wait(slot) releases the slot."""
        d, e = C.c_void_p(), C.c_void_p()
        _check(getattr(lib(), self._prefix + "frame_submit_device")(self._handle(), C.byref(params), slot, C.byref(d), C.byref(e)), self._prefix + "frame_submit_device")
        return d.value, e.value

    def frame_wait(self, slot):
        """Blocks until the slot's frame is in host memory; True if it is complete, False if rt_cancel() cut it short."""
        rc = getattr(lib(), self._prefix + "frame_wait")(self._handle(), slot)
        if rc < 0:
            _check(rc, self._prefix + "frame_wait")
        return rc == 0

    def frame_poll(self, slot):
        """None while the frame is on its way; otherwise as frame_wait()."""
        rc = getattr(lib(), self._prefix + "frame_poll")(self._handle(), slot)
        if rc < 0:
            _check(rc, self._prefix + "frame_poll")
        return None if rc == PENDING else rc == 0


class Renderer(_FrameQueue):
    """One rt_context = one GPU.  Mirrors the life cycle of the reference's main(): set scene,
    skybox and camera once, then render frames."""
    _prefix = "rt_"

    def _handle(self):
        return self._ctx

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        _check(lib().rt_create(C.byref(self._ctx), device), "rt_create")
        self.device = device
        self._keep = {}

    def close(self):
        if self._ctx and not getattr(self, "_borrowed", False):
            lib().rt_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        """scene: path to a scene_*.txt, or a raw Scene buffer (uint8[69636])."""
        if isinstance(scene, (str, bytes, os.PathLike)):
            rc, buf = parse_scene_file(scene)
            _check(rc, f"rt_parse_scene_file({scene})")
        else:
            buf = np.ascontiguousarray(scene, dtype=np.uint8)
            assert buf.nbytes == SCENE_BYTES
        _check(lib().rt_set_scene(self._ctx, buf.ctypes.data_as(C.c_void_p)), "rt_set_scene")
        return buf

    def compile_scene(self):
        """JIT a trace kernel specialised for the current scene (rt_compile_scene); raises RtError if it cannot."""
        _check(lib().rt_compile_scene(self._ctx), "rt_compile_scene")

    def scene_is_compiled(self):
        return bool(lib().rt_scene_is_compiled(self._ctx))

    def compiled_scene_info(self):
        """Where the compiled kernel came from: 'embedded, compiled with the library by ...' or 'hiprtc x.y at run time'."""
        return lib().rt_compiled_scene_info(self._ctx).decode()

    def set_skybox(self, faces):
        """faces: uint8 (6, h, w, chan) in CubeFace order."""
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        cm = Cubemap()
        for i in range(6):
            cm.data[i] = faces[i].ctypes.data
        cm.h, cm.w, cm.chan = faces.shape[1], faces.shape[2], faces.shape[3]
        _check(lib().rt_set_skybox(self._ctx, C.byref(cm)), "rt_set_skybox")

    def set_camera(self, pos=None, front=None, up=None, fov=None):
        cam = default_camera()
        if pos is not None:
            cam.pos = Vector3(*pos)
        if front is not None:
            cam.front = Vector3(*front)
        if up is not None:
            cam.up = Vector3(*up)
        if fov is not None:
            cam.fov = fov
        _check(lib().rt_set_camera(self._ctx, C.byref(cam)), "rt_set_camera")

    def set_tuning(self, dequeue_shards=0, workgroups_per_cu=0,
                   jit_waves_per_simd=0, jit_flags=None, poison_frame=None, trace_known_taps=None, test_every_object=None,
                   audit_known_taps=None, test_drop_pixels=None, test_corrupt_lit_table=None):
        """rt_set_tuning() (scheduling knobs, 0 / None = automatic: they never change a frame) and, for the arguments that are the test
        suite's -- poison_frame, trace_known_taps, test_every_object, test_drop_pixels, test_corrupt_lit_table --,
        rt_set_test_knobs() (include/rt_hip_testing.h).  Those and audit_known_taps stay as last set until set again."""
        t = Tuning()
        lib().rt_default_tuning(C.byref(t))
        t.dequeue_shards = dequeue_shards
        t.workgroups_per_cu, t.jit_waves_per_simd = workgroups_per_cu, jit_waves_per_simd
        t.jit_flags = jit_flags.encode() if jit_flags else None
        if audit_known_taps is not None:
            self._audit = int(audit_known_taps)
        t.audit_known_taps = getattr(self, "_audit", 0)
        _check(lib().rt_set_tuning(self._ctx, C.byref(t)), "rt_set_tuning")
        kept = getattr(self, "_test_knobs", None) or {}
        for name, value in (("poison_frame", poison_frame), ("trace_known_taps", trace_known_taps), ("test_every_object", test_every_object),
                            ("test_drop_pixels", test_drop_pixels), ("test_corrupt_lit_table", test_corrupt_lit_table)):
            if value is not None:
                kept[name] = int(value)
        self._test_knobs = kept
        if kept:
            k = TestKnobs()
            lib().rt_default_test_knobs(C.byref(k))
            for name, value in kept.items():
                setattr(k, name, value)
            _check(lib().rt_set_test_knobs(self._ctx, C.byref(k)), "rt_set_test_knobs")

    @staticmethod
    def params(width, height, spp, max_bounces, seed=0, row_block=8, rank=0, world=1, kernel=KERNEL_AUTO):
        p = RenderParams()
        lib().rt_default_params(C.byref(p), width, height, spp, max_bounces)
        p.seed, p.row_block, p.rank, p.world, p.kernel = seed, row_block, rank, world, kernel
        return p

    def render(self, width, height, spp, max_bounces, seed=0, kernel=KERNEL_AUTO):
        """Whole frame -> host float32 array (height, width, 3), row 0 = bottom (as the reference's `frame`)."""
        p = self.params(width, height, spp, max_bounces, seed=seed, kernel=kernel)
        out = np.empty((height, width, 3), dtype=np.float32)
        rc = lib().rt_render(self._ctx, C.byref(p), out.ctypes.data_as(C.c_void_p))
        if rc == 1:
            raise RtError("rt_render: cancelled (rt_cancel): the frame is incomplete")
        _check(rc, "rt_render")
        return out

    @staticmethod
    def _stream_arg(stream):
        """None -> the context's own stream (NULL in the C ABI); a hipStream_t handle as int otherwise, where the
        handle 0 -- torch's default stream -- is the device's legacy null stream and is passed as RT_STREAM_LEGACY
        (NULL already means "the context's stream")."""
        if stream is None:
            return None
        return C.c_void_p(STREAM_LEGACY if stream == 0 else stream)

    def stream(self, which=0):
        """rt_stream(): the context's stream (0) or its second, low-priority one (1) as a hipStream_t handle (int)."""
        h = lib().rt_stream(self._ctx, which)
        if not h:
            raise RtError(f"rt_stream({which}) failed: {lib().rt_last_error().decode(errors='replace')}")
        return int(h)

    def reserve(self, width, height):
        _check(lib().rt_reserve(self._ctx, width, height), "rt_reserve")

    def render_device(self, params, device_ptr, stream=None):
        """Enqueue one strip render into device memory (no sync)."""
        _check(lib().rt_render_device(self._ctx, C.byref(params), C.c_void_p(device_ptr), self._stream_arg(stream)),
               "rt_render_device")

    def deinterleave_device(self, strips_ptr, frame_ptr, width, height, row_block, world, stream=None, first=0):
        """first: position of strip 0 in the gathered buffer (1 when the strips were handed out by rt_strip_of_rank)."""
        _check(lib().rt_deinterleave_rotated_device(self._ctx, C.c_void_p(strips_ptr), C.c_void_p(frame_ptr), width, height,
                                                    row_block, world, first, self._stream_arg(stream)),
               "rt_deinterleave_rotated_device")

    # -- progressive accumulation (reference worker()/update_frame() protocol)
    def progressive_begin(self, width, height, init_scale=8, max_bounces=10, seed=0, rank=0, world=1):
        """world > 1: this context accumulates only the rows of its row blocks (rt_progressive_begin_rank) and
        progressive_resolve() returns those rows, (rt_strip_rows(height, 16, world), width, 3)."""
        if world == 1:
            _check(lib().rt_progressive_begin(self._ctx, width, height, init_scale, max_bounces, seed), "rt_progressive_begin")
            self._prog_shape = (height, width, 3)
        else:
            _check(lib().rt_progressive_begin_rank(self._ctx, width, height, init_scale, max_bounces, seed, rank, world), "rt_progressive_begin_rank")
            self._prog_shape = (strip_rows(height, 16, world), width, 3)

    def progressive_pass(self):
        w = C.c_float()
        _check(lib().rt_progressive_pass(self._ctx, C.byref(w)), "rt_progressive_pass")
        return w.value

    def progressive_passes(self, count):
        """`count` passes in as few launches as the ladder allows (rt_progressive_passes): the same frame as `count` progressive_pass() calls."""
        _check(lib().rt_progressive_passes(self._ctx, int(count)), "rt_progressive_passes")

    def progressive_resolve(self):
        out = np.empty(self._prog_shape, dtype=np.float32)
        _check(lib().rt_progressive_resolve(self._ctx, out.ctypes.data_as(C.c_void_p)), "rt_progressive_resolve")
        return out

    def progressive_invalidate(self):
        _check(lib().rt_progressive_invalidate(self._ctx), "rt_progressive_invalidate")

    def progressive_state(self):
        s, c, g, n = C.c_int(), C.c_float(), C.c_uint32(), C.c_int()
        _check(lib().rt_progressive_state(self._ctx, C.byref(s), C.byref(c), C.byref(g), C.byref(n)), "rt_progressive_state")
        return dict(next_scale=s.value, count=c.value, generation=g.value, passes=n.value)

    def selftest(self, which, seed=1, blocks=4096, iters=256):
        out = (C.c_ulonglong * 8)()
        _check(lib().rt_selftest(self._ctx, which, seed, blocks, iters, out), "rt_selftest")
        return list(out)

    def synchronize(self):
        _check(lib().rt_synchronize(self._ctx), "rt_synchronize")

    def cancel(self):
        """rt_cancel(): ask every launch enqueued so far -- running or queued -- to stop (one store; callable from any thread)."""
        _check(lib().rt_cancel(self._ctx), "rt_cancel")

    def was_cancelled(self):
        rc = lib().rt_was_cancelled(self._ctx)
        if rc < 0:
            _check(rc, "rt_was_cancelled")
        return rc == 1

    def last_launch_report(self):
        """rt_last_launch_report(): (status, dict) of the most recent launch -- status 0 complete, 1 cancelled, < 0 incomplete
        (rt_last_error() has the text); waits for the launch."""
        r = LaunchReport()
        rc = lib().rt_last_launch_report(self._ctx, C.byref(r))
        return rc, r.as_dict()

    def launch_check_submit(self, ticket, stream=None):
        """rt_launch_check_submit(): the control words of the most recent launch are copied on `stream` (ordered behind it by the caller)."""
        _check(lib().rt_launch_check_submit(self._ctx, ticket, self._stream_arg(stream)), "rt_launch_check_submit")

    def launch_check_wait(self, ticket):
        """rt_launch_check_wait(): True if the ticket's launch is complete, False if rt_cancel() cut it short; raises RtError if it
        did not account for every pixel."""
        rc = lib().rt_launch_check_wait(self._ctx, ticket, None)
        if rc < 0:
            _check(rc, "rt_launch_check_wait")
        return rc == 0

    def profile(self, on=True):
        """on = 2: also time the camera-ray pass apart (profile_collect_split)."""
        _check(lib().rt_profile_enable(self._ctx, int(on)), "rt_profile_enable")

    def profile_collect(self):
        ms, n = C.c_double(), C.c_int()
        _check(lib().rt_profile_collect(self._ctx, C.byref(ms), C.byref(n)), "rt_profile_collect")
        return ms.value, n.value

    def profile_collect_split(self):
        """(summed per-launch kernel ms, launches, span ms, the part of the first that was the camera-ray pass)"""
        ms, n, span, prim = C.c_double(), C.c_int(), C.c_double(), C.c_double()
        _check(lib().rt_profile_collect_split(self._ctx, C.byref(ms), C.byref(n), C.byref(span), C.byref(prim)), "rt_profile_collect_split")
        return ms.value, n.value, span.value, prim.value

    def profile_collect_span(self):
        """(summed per-launch kernel ms, launches, ms from the first launch's start to the end of the last one's trace kernel)"""
        ms, n, span = C.c_double(), C.c_int(), C.c_double()
        _check(lib().rt_profile_collect_span(self._ctx, C.byref(ms), C.byref(n), C.byref(span)), "rt_profile_collect_span")
        return ms.value, n.value, span.value


class MultiRenderer(_FrameQueue):
    """rt_multi_*: one frame on several GPUs of this node from one process (native RCCL gather, no torch)."""
    _prefix = "rt_multi_"

    def _handle(self):
        return self._m

    def context(self, i):
        """Device i's rt_context as a Renderer view (owned by the group: do not close it)."""
        r = Renderer.__new__(Renderer)
        r._ctx = C.c_void_p(lib().rt_multi_context(self._m, i))
        r.device, r._keep, r._borrowed = i, {}, True
        return r

    def __init__(self, devices, on_one_device=None):
        """devices: the GPUs of the group.  on_one_device=n (testing aid): n contexts on devices[0], the gather done by copies."""
        devices = list(devices)
        self._m = C.c_void_p()
        if on_one_device:
            _check(lib().rt_multi_create_on_one_device(C.byref(self._m), devices[0], on_one_device), "rt_multi_create_on_one_device")
            return
        ids = (C.c_int * len(devices))(*devices)
        _check(lib().rt_multi_create(C.byref(self._m), ids, len(devices)), "rt_multi_create")

    def collective_info(self):
        """What the group's RCCL communicator reports: dict(ranks=ncclCommCount, devices=[ncclCommCuDevice ...], version=ncclGetVersion)."""
        ranks, ver = C.c_int(), C.c_int()
        devs = (C.c_int * 64)()
        _check(lib().rt_multi_collective_info(self._m, C.byref(ranks), devs, C.byref(ver)), "rt_multi_collective_info")
        return {"ranks": ranks.value, "devices": [devs[i] for i in range(ranks.value)], "version": ver.value}

    def profile_phases(self, on=True):
        """rt_multi_profile_enable(): timed events around every phase of every frame, per device."""
        _check(lib().rt_multi_profile_enable(self._m, 1 if on else 0), "rt_multi_profile_enable")

    def collect_phases(self):
        """rt_multi_profile_collect(): per device, dict(frames, step_ms, idle_ms, render_ms, gather_ms, deinterleave_ms, copy_ms)."""
        n = self.size()
        a = (MultiPhases * n)()
        _check(lib().rt_multi_profile_collect(self._m, a, n), "rt_multi_profile_collect")
        return [{k: getattr(a[i], k) for k, _ in MultiPhases._fields_} for i in range(n)]

    def close(self):
        if self._m:
            lib().rt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return lib().rt_multi_size(self._m)

    def set_scene(self, scene):
        if isinstance(scene, (str, bytes, os.PathLike)):
            rc, buf = parse_scene_file(scene)
            _check(rc, f"rt_parse_scene_file({scene})")
        else:
            buf = np.ascontiguousarray(scene, dtype=np.uint8)
        _check(lib().rt_multi_set_scene(self._m, buf.ctypes.data_as(C.c_void_p)), "rt_multi_set_scene")

    def set_skybox(self, faces):
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        cm = Cubemap()
        for i in range(6):
            cm.data[i] = faces[i].ctypes.data
        cm.h, cm.w, cm.chan = faces.shape[1], faces.shape[2], faces.shape[3]
        _check(lib().rt_multi_set_skybox(self._m, C.byref(cm)), "rt_multi_set_skybox")

    def set_camera(self, pos=None, front=None, up=None, fov=None):
        cam = default_camera()
        if pos is not None:
            cam.pos = Vector3(*pos)
        if front is not None:
            cam.front = Vector3(*front)
        if up is not None:
            cam.up = Vector3(*up)
        if fov is not None:
            cam.fov = fov
        _check(lib().rt_multi_set_camera(self._m, C.byref(cam)), "rt_multi_set_camera")

    def set_tuning(self, **kw):
        """rt_multi_set_tuning() for rt_tuning's fields, rt_multi_set_test_knobs() for rt_test_knobs' (include/rt_hip_testing.h)."""
        t = Tuning()
        lib().rt_default_tuning(C.byref(t))
        k = TestKnobs()
        lib().rt_default_test_knobs(C.byref(k))
        for name, v in kw.items():
            if name == "jit_flags" and isinstance(v, str):
                v = v.encode()
            setattr(k if name in TEST_KNOB_NAMES else t, name, v)
        _check(lib().rt_multi_set_tuning(self._m, C.byref(t)), "rt_multi_set_tuning")
        if any(name in TEST_KNOB_NAMES for name in kw):
            _check(lib().rt_multi_set_test_knobs(self._m, C.byref(k)), "rt_multi_set_test_knobs")

    def compile_scene(self):
        _check(lib().rt_multi_compile_scene(self._m), "rt_multi_compile_scene")

    # -- the interactive protocol on the group (rt_multi_progressive_*)
    def progressive_begin(self, width, height, init_scale=8, max_bounces=10, seed=0):
        _check(lib().rt_multi_progressive_begin(self._m, width, height, init_scale, max_bounces, seed), "rt_multi_progressive_begin")
        self._prog_shape = (height, width, 3)

    def progressive_pass(self):
        w = C.c_float()
        _check(lib().rt_multi_progressive_pass(self._m, C.byref(w)), "rt_multi_progressive_pass")
        return w.value

    def progressive_passes(self, count):
        _check(lib().rt_multi_progressive_passes(self._m, int(count)), "rt_multi_progressive_passes")

    def progressive_resolve(self):
        out = np.empty(self._prog_shape, dtype=np.float32)
        _check(lib().rt_multi_progressive_resolve(self._m, out.ctypes.data_as(C.c_void_p)), "rt_multi_progressive_resolve")
        return out

    def progressive_invalidate(self):
        _check(lib().rt_multi_progressive_invalidate(self._m), "rt_multi_progressive_invalidate")

    def progressive_state(self):
        s, c, g, n = C.c_int(), C.c_float(), C.c_uint32(), C.c_int()
        _check(lib().rt_multi_progressive_state(self._m, C.byref(s), C.byref(c), C.byref(g), C.byref(n)), "rt_multi_progressive_state")
        return dict(next_scale=s.value, count=c.value, generation=g.value, passes=n.value)

    def render(self, width, height, spp, max_bounces, seed=0, row_block=8):
        p = Renderer.params(width, height, spp, max_bounces, seed=seed, row_block=row_block)
        out = np.empty((height, width, 3), dtype=np.float32)
        _check(lib().rt_multi_render(self._m, C.byref(p), out.ctypes.data_as(C.c_void_p)), "rt_multi_render")
        return out
