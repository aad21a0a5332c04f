"""Blocking renders of scene_0 at 1 and at 64 samples per pixel (1080p, 10 bounces), to be run under rocprofv3 --pmc: what a round of
the compiled kernel costs when every lane takes a pixel for itself (one sample per pixel) against the eight streams."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
for spp in (1, 1, 1, 64, 64, 64):
    g.render(1920, 1080, spp, 10, seed=3)
g.close()
