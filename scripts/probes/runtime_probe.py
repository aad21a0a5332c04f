"""Development aid: the C ABI's frame queue on C1 in a Python process WITHOUT torch (librt_hip.so then binds to the system's
HIP runtime, as rt_cli does) or with it (torch's bundled runtime is loaded first).  usage: runtime_probe.py [notorch] [generic]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "notorch":
    sys.modules["torch"] = None          # `import torch` raises ImportError
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera(); g.reserve(1920, 1080)
if 'generic' not in sys.argv: g.compile_scene()
loop = FrameLoop(g, 1920, 1080, 64, 4, depth=2)
loop.run(range(3))
for rep in range(3):
    g.profile(True)
    t0 = time.perf_counter(); loop.run(range(30)); dt = (time.perf_counter() - t0) / 30 * 1e3
    ms, n, span = g.profile_collect_span(); g.profile(False)
    print(("generic kernel, " if "generic" in sys.argv else "compiled kernel, ") + f"{'no torch (system HIP runtime)' if 'notorch' in sys.argv else 'torch loaded first (its bundled HIP runtime)'}: {dt:.3f} ms per step, kernels {ms / n:.3f} ms per launch, span {span / n:.3f}", flush=True)
if "generic" not in sys.argv:
    import ctypes as C
    buf = C.create_string_buffer(1 << 20); n = C.c_size_t()
    L = rt.lib(); L.rt_spec_symbol_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    if L.rt_spec_symbol_read(g._ctx, b"", buf, len(buf), C.byref(n)) == 0:
        out = os.path.join(ROOT, "gpurun_out", "spec_" + ("system" if "notorch" in sys.argv else "torch") + ".co")
        os.makedirs(os.path.dirname(out), exist_ok=True); open(out, "wb").write(buf.raw[:n.value]); print("  code object ->", out, n.value, "bytes")
for l in open("/proc/self/maps"):
    if any(k in l for k in ("hiprtc", "comgr", "amdhip64", "hsa-runtime")) and "r-xp" in l: print("  ", l.split()[-1])
