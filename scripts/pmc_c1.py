"""Render C1 a few times (for rocprofv3 --pmc passes).  usage: pmc_c1.py <kernel> [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
if kernel == 0 and not os.environ.get("RT_NO_JIT"):
    g.compile_scene()
for _ in range(3):
    g.render(1920, 1080, spp, 4, kernel=kernel)
g.close()
