"""ctypes bindings used by the tests (and bench.py's cpu_baseline leg / smoke()) for

  * oracle/liboracle.so          -- the repo's CPU restatement (the parity oracle)
  * oracle/_ref/libref*.so       -- the unmodified reference behind a headless harness (optional)

TEST INFRASTRUCTURE: nothing under ray_tracing_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DATA_DIR = os.path.join(ROOT, "data")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

MAX_OBJECTS = 1024
FACE_NAMES = ["front", "back", "left", "right", "top", "bottom"]  # CubeFace order


class Vector3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", Vector3), ("front", Vector3), ("up", Vector3), ("fov", C.c_float)]


class CameraBasis(C.Structure):
    _fields_ = [("pos", Vector3), ("lower_left_corner", Vector3), ("horizontal", Vector3), ("vertical", Vector3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "object_tests", "rng_draws", "sky_fetches", "flops",
                                          "box_tests", "box_flops", "sphere_tests", "sphere_flops", "sky_samples", "sky_sample_flops", "first_ray_flops")]


# numpy view of the reference's Object (scene.h:24-31): 68 bytes
OBJECT_DTYPE = np.dtype([
    ("type", "<i4"),
    ("geom", "<f4", (6,)),      # sphere: center[3], radius | cube: origin[3], size[3]
    ("albedo", "<f4", (3,)),
    ("roughness", "<f4"), ("reflectance", "<f4"), ("metallic", "<f4"), ("emission_power", "<f4"),
    ("emission_color", "<f4", (3,)),
])
assert OBJECT_DTYPE.itemsize == 68
SCENE_BYTES = 68 * MAX_OBJECTS + 4


def scene_buffer():
    return np.zeros(SCENE_BYTES, dtype=np.uint8)


def scene_objects(buf):
    """(objects view, num_objects) of a raw Scene buffer."""
    n = int(buf[68 * MAX_OBJECTS:].view("<i4")[0])
    return buf[:68 * MAX_OBJECTS].view(OBJECT_DTYPE), n


def scene_signature(buf):
    """The defined fields of a Scene as a plain list (ignores union padding the parser leaves dirty)."""
    objs, n = scene_objects(buf)
    out = []
    for o in objs[:n]:
        ngeom = 4 if o["type"] == 1 else 6
        out.append((int(o["type"]), o["geom"][:ngeom].view("<u4").tolist(), o["albedo"].view("<u4").tolist(),
                    [int(np.float32(o[k]).view("<u4")) for k in ("roughness", "reflectance", "metallic", "emission_power")],
                    o["emission_color"].view("<u4").tolist()))
    return out


def make_scene(objects):
    """Build a raw Scene buffer from dicts: {type:'cube'|'sphere', center/radius | origin/size, material...}."""
    buf = scene_buffer()
    objs = buf[:68 * MAX_OBJECTS].view(OBJECT_DTYPE)
    for i, d in enumerate(objects):
        o = objs[i]
        if d["type"] == "sphere":
            o["type"] = 1
            o["geom"][:3] = d.get("center", (0, 0, 0))
            o["geom"][3] = d.get("radius", 1)
        else:
            o["type"] = 0
            o["geom"][:3] = d.get("origin", (0, 0, 0))
            o["geom"][3:] = d.get("size", (1, 1, 1))
        o["albedo"] = d.get("albedo", (0.44, 0.68, 0.84))
        o["roughness"] = d.get("roughness", 0)
        o["reflectance"] = d.get("reflectance", 0.2)
        o["metallic"] = d.get("metallic", 0)
        o["emission_power"] = d.get("emission_power", 0)
        o["emission_color"] = d.get("emission_color", (1, 1, 1))
    buf[68 * MAX_OBJECTS:].view("<i4")[0] = len(objects)
    return buf


def large_scene(n, seed, extent=10.0, floor=True, light=True):
    """n objects: small spheres and boxes scattered over [-extent, extent]^3, optionally a floor slab and a sphere emitter
    (the generator of test_max_objects, parametrised; also bench.py's L* workloads)."""
    rng = np.random.default_rng(seed)
    objs = []
    for k in range(n):
        if floor and k == 0:
            objs.append(dict(type="cube", origin=(-extent, -extent - 0.5, -extent), size=(2 * extent, 0.5, 2 * extent), albedo=(.6, .6, .6), roughness=1.0))
        elif light and k == n // 2:
            objs.append(dict(type="sphere", center=(0.0, extent * 0.8, 0.0), radius=1.0, albedo=(1, 1, 1), emission_power=4.0))
        elif k % 2:
            objs.append(dict(type="sphere", center=rng.uniform(-extent, extent, 3), radius=rng.uniform(0.1, 0.6), albedo=rng.uniform(0, 1, 3),
                             roughness=rng.uniform(0, 1), metallic=float(k % 7 == 0)))
        else:
            objs.append(dict(type="cube", origin=rng.uniform(-extent, extent - 1, 3), size=rng.uniform(0.1, 1, 3), albedo=rng.uniform(0, 1, 3),
                             metallic=float(k % 8 == 0), roughness=rng.uniform(0, 1), reflectance=rng.uniform(0, 1)))
    return make_scene(objs)


def stick_scene(n, seed, extent=10.0):
    """n objects that SPAN the scene: thin boxes running the whole length of one axis, at random places, and a few large spheres -- the
    adversary of a spatial hierarchy (every cluster's and every group's box covers most of the scene, a ray passes most of them)."""
    rng = np.random.default_rng(seed)
    objs = []
    for k in range(n):
        if k == n // 2:
            objs.append(dict(type="sphere", center=(0.0, extent * 0.8, 0.0), radius=1.0, albedo=(1, 1, 1), emission_power=4.0))
        elif k % 9 == 0:
            objs.append(dict(type="sphere", center=rng.uniform(-extent, extent, 3), radius=rng.uniform(0.5, 2.0), albedo=rng.uniform(0, 1, 3), roughness=rng.uniform(0, 1)))
        else:
            axis = int(rng.integers(0, 3))
            origin = rng.uniform(-extent, extent, 3); size = rng.uniform(0.03, 0.12, 3)
            origin[axis] = -extent; size[axis] = 2 * extent
            objs.append(dict(type="cube", origin=origin, size=size, albedo=rng.uniform(0, 1, 3), roughness=rng.uniform(0, 1), metallic=float(k % 8 == 0)))
    return make_scene(objs)


LARGE_SCENE_CAMERA = dict(pos=(14, 9, 14), front=(-1, -0.5, -1), up=(0, 1, 0), fov=1.0)      # outside large_scene(), looking in


def synthetic_skybox(size=64, seed=7):
    """Deterministic 6 x size x size x 3 u8 skybox (distinct per face, varying per texel)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(6, size, size, 3), dtype=np.uint8)


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "rt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    """The CPU restatement.  One global context per loaded library."""

    def __init__(self, counters=False):
        build_oracle()
        name = "liboracle_count.so" if counters else "liboracle.so"
        L = self.L = C.CDLL(os.path.join(ORACLE_DIR, name))
        L.orc_path_seed.restype = C.c_uint64
        L.orc_path_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_random_float.restype = C.c_float
        L.orc_random_float.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_random_direction.argtypes = [C.POINTER(C.c_uint64), C.c_void_p]
        L.orc_camera_ray.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_camera_basis.argtypes = [C.c_float, C.POINTER(CameraBasis)]
        L.orc_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sample_cubemap.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pixel.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
        L.orc_render_stream.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
        L.orc_render_counter.argtypes = [C.c_int] * 4 + [C.c_uint64] + [C.c_int] * 3 + [C.c_void_p]
        L.orc_render_counter_rows.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_void_p] + [C.c_int] * 2 + [C.c_void_p]
        L.orc_time_columns.argtypes = [C.c_int] * 5
        L.orc_parse_scene_file.argtypes = [C.c_char_p, C.c_void_p]
        L.orc_parse_scene_string.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_set_scene.argtypes = [C.c_void_p]
        L.orc_set_skybox.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_set_camera.argtypes = [C.POINTER(Camera)]
        L.orc_default_camera.argtypes = [C.POINTER(Camera)]
        L.orc_counters_get.argtypes = [C.POINTER(Counters)]
        self._keep = {}
        self.set_camera()

    # -- inputs
    def parse_scene_file(self, path):
        buf = scene_buffer()
        rc = self.L.orc_parse_scene_file(os.fsencode(path), _fp(buf))
        return rc, buf

    def parse_scene_string(self, text):
        if isinstance(text, str):
            text = text.encode()
        buf = scene_buffer()
        rc = self.L.orc_parse_scene_string(text, len(text), _fp(buf))
        return rc, buf

    def set_scene(self, buf):
        assert buf.nbytes == SCENE_BYTES
        self.L.orc_set_scene(_fp(buf))

    def load_scene(self, path):
        rc, buf = self.parse_scene_file(path)
        assert rc == 0, path
        self.set_scene(buf)
        return buf

    def set_skybox(self, faces):
        """faces: uint8 array (6, h, w, chan), CubeFace order."""
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        self._keep["sky"] = faces
        ptrs = (C.c_void_p * 6)(*[faces[i].ctypes.data for i in range(6)])
        self._keep["sky_ptrs"] = ptrs
        self.L.orc_set_skybox(ptrs, faces.shape[2], faces.shape[1], faces.shape[3])

    def set_camera(self, pos=None, front=None, up=None, fov=None):
        cam = Camera()
        self.L.orc_default_camera(C.byref(cam))
        if pos is not None:
            cam.pos = Vector3(*pos)
        if front is not None:
            cam.front = Vector3(*front)
        if up is not None:
            cam.up = Vector3(*up)
        if fov is not None:
            cam.fov = fov
        self.L.orc_set_camera(C.byref(cam))
        self.camera = cam

    # -- building blocks
    def path_seed(self, seed, p, s):
        return self.L.orc_path_seed(seed, p, s)

    def random_floats(self, state, n):
        st = C.c_uint64(state)
        out = np.array([self.L.orc_random_float(C.byref(st)) for _ in range(n)], dtype=np.float32)
        return out, st.value

    def random_direction(self, state):
        st = C.c_uint64(state)
        out = np.zeros(3, np.float32)
        self.L.orc_random_direction(C.byref(st), _fp(out))
        return out, st.value

    def camera_ray(self, px, py, aspect):
        out = np.zeros(6, np.float32)
        self.L.orc_camera_ray(px, py, aspect, _fp(out))
        return out

    def camera_basis(self, aspect):
        b = CameraBasis()
        self.L.orc_camera_basis(aspect, C.byref(b))
        return b

    def trace_ray(self, o, d):
        o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
        out = np.zeros(7, np.float32)
        obj = self.L.orc_trace_ray(_fp(o), _fp(d), _fp(out))
        return obj, out

    def sample_cubemap(self, d):
        d = np.asarray(d, np.float32)
        out = np.zeros(3, np.float32)
        self.L.orc_sample_cubemap(_fp(d), _fp(out))
        return out

    def pixel(self, u, v, aspect, max_bounces, state):
        st = C.c_uint64(state)
        out = np.zeros(3, np.float32)
        self.L.orc_pixel(u, v, aspect, max_bounces, C.byref(st), _fp(out))
        return out, st.value

    # -- frames
    def render_stream(self, W, H, passes=1, init_scale=1, max_bounces=10, state=0):
        st = C.c_uint64(state)
        frame = np.zeros((H, W, 3), np.float32)
        accum = np.zeros((H, W, 3), np.float32)
        self.L.orc_render_stream(W, H, passes, init_scale, max_bounces, C.byref(st), _fp(frame), _fp(accum))
        return frame, accum, st.value

    def render_counter(self, W, H, spp, max_bounces, seed=0, rows=None, threads=None):
        r0, r1 = rows if rows is not None else (0, H)
        if threads is None:
            threads = min(os.cpu_count() or 1, 32)
        frame = np.zeros((H, W, 3), np.float32)
        self.L.orc_render_counter(W, H, spp, max_bounces, seed, r0, r1, threads, _fp(frame))
        return frame

    def render_counter_rows(self, W, H, spp, max_bounces, rows, seed=0, threads=None):
        """Only the listed frame rows (dealt dynamically to the threads); returns {row: [W, 3] array}."""
        rows = np.ascontiguousarray(sorted(set(int(r) for r in rows)), dtype=np.int32)
        assert len(rows) and rows[0] >= 0 and rows[-1] < H
        if threads is None:
            threads = min(os.cpu_count() or 1, 32)
        frame = np.zeros((H, W, 3), np.float32)
        self.L.orc_render_counter_rows(W, H, spp, max_bounces, seed, rows.ctypes.data_as(C.c_void_p), len(rows), threads, _fp(frame))
        return {int(r): frame[int(r)] for r in rows}

    def time_columns(self, W, H, passes, max_bounces, threads):
        self.L.orc_time_columns(W, H, passes, max_bounces, threads)

    def counters(self):
        c = Counters()
        self.L.orc_counters_get(C.byref(c))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def counters_reset(self):
        self.L.orc_counters_reset()


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref.so")) and \
        os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_bounce.so"))


class Ref:
    """The compiled reference (oracle/_ref).  bounce_patch=True loads the build whose only change is
    that main.c:156's literal 10 reads a runtime variable."""

    def __init__(self, bounce_patch=False):
        name = "libref_bounce.so" if bounce_patch else "libref.so"
        L = self.L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", name))
        L.ref_random_float.restype = C.c_float
        L.ref_get_rng.restype = C.c_uint64
        L.ref_set_rng.argtypes = [C.c_uint64]
        L.ref_path_seed.restype = C.c_uint64
        L.ref_path_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.ref_camera_ray.argtypes = [C.c_float] * 3 + [C.c_void_p]
        L.ref_trace_ray.argtypes = [C.c_void_p] * 3
        L.ref_sample_cubemap.argtypes = [C.c_void_p] * 2
        L.ref_normalize.argtypes = [C.c_void_p] * 2
        L.ref_pixel.argtypes = [C.c_float] * 3 + [C.c_void_p]
        L.ref_render_stream.argtypes = [C.c_int] * 3 + [C.c_void_p] * 2
        L.ref_render_counter_rows.argtypes = [C.c_int] * 3 + [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        L.ref_time_columns.argtypes = [C.c_int] * 4
        L.ref_load_scene.argtypes = [C.c_char_p]
        L.ref_parse_scene_into.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        L.ref_set_scene.argtypes = [C.c_void_p]
        L.ref_load_skybox.argtypes = [C.c_char_p]
        L.ref_set_skybox_raw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_skybox_face.restype = C.c_void_p
        L.ref_skybox_face.argtypes = [C.c_int]
        L.ref_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.ref_sizeof_scene.restype = C.c_size_t
        L.ref_sizeof_object.restype = C.c_size_t
        self.bounce_patch = bounce_patch
        self._keep = {}

    def set_bounce_limit(self, n):
        assert self.bounce_patch or n == 10
        self.L.ref_set_bounce_limit(n)

    def parse_scene_file(self, path):
        buf = scene_buffer()
        rc = self.L.ref_parse_scene_into(os.fsencode(path), _fp(buf), buf.nbytes)
        return rc, buf

    def set_scene(self, buf):
        self.L.ref_set_scene(_fp(buf))

    def load_scene(self, path):
        assert self.L.ref_load_scene(os.fsencode(path)) == 0

    def load_skybox(self, directory):
        self.L.ref_load_skybox(os.fsencode(directory))

    def skybox_faces(self):
        w, h, ch = C.c_int(), C.c_int(), C.c_int()
        self.L.ref_skybox_dims(C.byref(w), C.byref(h), C.byref(ch))
        n = w.value * h.value * ch.value
        faces = [np.ctypeslib.as_array(C.cast(self.L.ref_skybox_face(i), C.POINTER(C.c_uint8)), shape=(n,)).copy()
                 for i in range(6)]
        return np.stack(faces).reshape(6, h.value, w.value, ch.value)

    def set_skybox(self, faces):
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        self._keep["sky"] = faces
        ptrs = (C.c_void_p * 6)(*[faces[i].ctypes.data for i in range(6)])
        self._keep["sky_ptrs"] = ptrs
        self.L.ref_set_skybox_raw(ptrs, faces.shape[2], faces.shape[1], faces.shape[3])

    def set_camera(self, pos=(5, 5, 5), front=(-1, -1, -1), up=(0, 1, 0), fov=30.0):
        a = [np.asarray(v, np.float32) for v in (pos, front, up)]
        self.L.ref_set_camera(_fp(a[0]), _fp(a[1]), _fp(a[2]), fov)

    def set_rng(self, s):
        self.L.ref_set_rng(s)

    def get_rng(self):
        return self.L.ref_get_rng()

    def random_floats(self, state, n):
        self.set_rng(state)
        out = np.array([self.L.ref_random_float() for _ in range(n)], dtype=np.float32)
        return out, self.get_rng()

    def random_direction(self, state):
        self.set_rng(state)
        out = np.zeros(3, np.float32)
        self.L.ref_random_direction(_fp(out))
        return out, self.get_rng()

    def path_seed(self, seed, p, s):
        return self.L.ref_path_seed(seed, p, s)

    def camera_ray(self, px, py, aspect):
        out = np.zeros(6, np.float32)
        self.L.ref_camera_ray(px, py, aspect, _fp(out))
        return out

    def trace_ray(self, o, d):
        o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
        out = np.zeros(7, np.float32)
        obj = self.L.ref_trace_ray(_fp(o), _fp(d), _fp(out))
        return obj, out

    def sample_cubemap(self, d):
        d = np.asarray(d, np.float32)
        out = np.zeros(3, np.float32)
        self.L.ref_sample_cubemap(_fp(d), _fp(out))
        return out

    def normalize(self, d):
        d = np.asarray(d, np.float32)
        out = np.zeros(3, np.float32)
        self.L.ref_normalize(_fp(d), _fp(out))
        return out

    def pixel(self, u, v, aspect, state):
        self.set_rng(state)
        out = np.zeros(3, np.float32)
        self.L.ref_pixel(u, v, aspect, _fp(out))
        return out, self.get_rng()

    def render_stream(self, W, H, passes=1, state=0):
        self.set_rng(state)
        frame = np.zeros((H, W, 3), np.float32)
        accum = np.zeros((H, W, 3), np.float32)
        self.L.ref_render_stream(W, H, passes, _fp(frame), _fp(accum))
        return frame, accum, self.get_rng()

    def render_counter(self, W, H, spp, seed=0, rows=None):
        r0, r1 = rows if rows is not None else (0, H)
        frame = np.zeros((H, W, 3), np.float32)
        self.L.ref_render_counter_rows(W, H, spp, seed, r0, r1, _fp(frame))
        return frame

    def time_columns(self, W, H, passes, threads):
        self.L.ref_time_columns(W, H, passes, threads)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_progressive(o, W, H, init_scale, passes, max_bounces, seed):
    """Drive Oracle.progressive passes like worker(): returns (frame, accum, count, next_scale)."""
    import ctypes as C
    o.L.orc_progressive_pass.restype = C.c_int
    o.L.orc_progressive_pass.argtypes = [C.c_int] * 5 + [C.c_uint64, C.c_void_p, C.POINTER(C.c_float)]
    o.L.orc_resolve.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
    accum = np.zeros((H, W, 3), np.float32)
    count = C.c_float(0)
    s = init_scale
    for p in range(passes):
        s = o.L.orc_progressive_pass(W, H, s, p, max_bounces, seed, accum.ctypes.data_as(C.c_void_p), C.byref(count))
    frame = np.zeros((H, W, 3), np.float32)
    o.L.orc_resolve(W, H, accum.ctypes.data_as(C.c_void_p), count, frame.ctypes.data_as(C.c_void_p))
    return frame, accum, count.value, s
