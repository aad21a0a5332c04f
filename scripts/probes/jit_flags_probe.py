"""Development aid: C1 kernel time with different extra hiprtc options for the scene-specialised kernel (rt_tuning.jit_flags),
interleaved in one process.  usage: jit_flags_probe.py "<flags A>" "<flags B>" ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("NOTORCH"): sys.modules["torch"] = None     # the system's HIP runtime and compiler instead of torch's bundled ones
import ray_tracing_amd as rt
flags = [""] + sys.argv[1:]
W, H, spp, nb = 1920, 1080, 64, 4
rs = []
sky = rt.load_skybox()
for f in flags:
    g = rt.Renderer(0); g.set_skybox(sky); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
    g.set_tuning(jit_flags=f or None)
    try:
        g.compile_scene()
    except rt.RtError as e:
        print(f"{f!r}: {e}"); g = None
    if g: g.profile(True)
    rs.append(g)
import ctypes as C
_p = C.c_void_p(); rt.lib().hipMalloc = None
hip = C.CDLL("libamdhip64.so.7" if os.environ.get("NOTORCH") else os.path.join(os.path.dirname(__import__("torch").__file__), "lib", "libamdhip64.so"))
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
assert hip.hipMalloc(C.byref(_p), H * W * 12) == 0
class _S:
    def data_ptr(self): return _p.value
strip = _S()
times = [[] for _ in flags]
for it in range(8):
    for k, g in enumerate(rs):
        if not g: continue
        g.render_device(g.params(W, H, spp, nb), strip.data_ptr()); g.synchronize()
        ms, n = g.profile_collect()
        if it: times[k].append(ms)
for f, t in zip(flags, times):
    if t: print(f"{f or '(default)':60s} {statistics.median(t):.3f} ms  (min {min(t):.3f})", flush=True)
