"""Site counters of the compiled kernel over a run of single interactive passes (1080p, 10 bounces): rounds per wave and pass, lanes
at work in them.  usage: progressive_stats.py [workgroups per CU [passes]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracing_amd as rt
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
W, H, nb = 1920, 1080, 10
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
g.set_tuning(jit_flags="-DRT_STATS", workgroups_per_cu=wg); g.compile_scene()
out = (C.c_ulonglong * 128)()
rt.lib().rt_spec_stats_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
read = lambda: rt.lib().rt_spec_stats_read(g._ctx, out, 1)
g.progressive_begin(W, H, init_scale=1, max_bounces=nb, seed=1)
for _ in range(6): g.progressive_pass()
g.synchronize(); read()
for _ in range(passes): g.progressive_pass()
g.synchronize(); read()
waves = 256 * 4 * (wg if wg else 4)       # (0, the library's own choice, varies per launch: rounds per wave are then per 4 096)
site = lambda k: (out[2 * k] / passes, out[2 * k + 1] / max(1, out[2 * k]))
for k, name in ((7, "rounds"), (8, "rounds that shade"), (12, "tap batches"), (21, "pixel fetches"), (24, "rounds with lanes that got no pixel")):
    n, lanes = site(k)
    print(f"{name:40s} {n:10.0f} per pass = {n / waves:6.2f} per wave of {waves}; lanes {lanes:5.1f}")
