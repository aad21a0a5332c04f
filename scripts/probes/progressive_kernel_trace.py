import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
g.progressive_begin(1920, 1080, init_scale=8, max_bounces=10, seed=1)
for _ in range(40): g.progressive_pass()
g.synchronize()
g.close()
