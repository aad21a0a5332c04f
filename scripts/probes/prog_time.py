"""Development aid: time of the reference's interactive protocol (rt_progressive_pass: 1 sample per pixel per pass,
scale ladder from 1/8 resolution) at 1920x1080, scene_0, 10 bounces.  usage: prog_time.py [lib.so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracing_amd as rt
if len(sys.argv) > 1: rt.LIB_PATH = os.path.abspath(sys.argv[1])
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
g.progressive_begin(1920, 1080, init_scale=8, max_bounces=10, seed=0)
for _ in range(8): g.progressive_pass()
g.synchronize()
t = time.perf_counter()
N = 200
for _ in range(N): g.progressive_pass()
g.synchronize()
dt = (time.perf_counter() - t) / N
print(f"{rt.LIB_PATH}: full-resolution 1-spp pass {dt * 1e3:.3f} ms = {1920 * 1080 / dt / 1e6:.0f} Msamples/s")
