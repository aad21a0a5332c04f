"""Development aid for rocprofv3 --pmc runs: three C1 frames with the library given in RT_LIB_FILE (path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
if os.environ.get("RT_LIB_FILE"): rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB_FILE"])
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
for _ in range(3): g.render(1920, 1080, 64, 4)
