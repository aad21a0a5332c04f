"""How long a blocking rt_render() of a C1 frame takes from the caller's side, into pageable memory (what a host that malloc()s
its frame gets) -- against the kernel time of the same launch.  usage: blocking_render_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
g.render(1920, 1080, 64, 4, seed=0)
g.profile(True)
ts = []
for k in range(8):
    t = time.perf_counter(); g.render(1920, 1080, 64, 4, seed=k); ts.append((time.perf_counter() - t) * 1e3)
ms, n = g.profile_collect()
print("rt_render() 1920x1080x64 spp into a numpy array (pageable): %s ms per call; kernels %.3f ms per launch" % (" ".join("%.2f" % x for x in ts), ms / n))
g.close()
