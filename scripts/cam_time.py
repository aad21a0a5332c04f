"""Development aid: C1 kernel time with the default camera (5,5,5: on several slab planes) vs a camera
moved by a hair (no exact-zero slab numerators)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.profile(True)
if not os.environ.get("RT_NO_JIT"): g.compile_scene()
ts = {0: [], 1: []}
for it in range(8):
    for k, pos in enumerate([None, (5.0000123, 5.0000234, 5.0000345)]):
        g.set_camera(pos=pos)
        g.render(1920, 1080, 64, 4)
        ms, n = g.profile_collect()
        if it: ts[k].append(ms)
print("default camera %.3f ms   nudged camera %.3f ms" % (statistics.median(ts[0]), statistics.median(ts[1])))
