"""Render one BASELINE.json config a few times and exit (the program rocprofv3 --pmc / --stats passes wrap).
usage: render_cfg.py <C1|C2|C3|C4strip|L256|L1024> [frames] [generic]     (RT_LIB_FILE=<path> picks another build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
if os.environ.get("RT_LIB_FILE"): rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB_FILE"])
CFG = {"C1": (0, 1920, 1080, 64, 4, 1, 0), "C2": (1, 1920, 1080, 256, 8, 1, 0), "C3": (2, 3840, 2160, 64, 8, 1, 0),
       "C4strip": (0, 3840, 2160, 1024, 8, 8, 3), "C1strip8": (0, 1920, 1080, 64, 4, 8, 3),
       "L256": (-256, 1920, 1080, 16, 5, 1, 0), "L512": (-512, 1920, 1080, 16, 5, 1, 0), "L1024": (-1024, 1920, 1080, 16, 5, 1, 0)}      # bench.py's synthetic large scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C1"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
generic = len(sys.argv) > 3 and sys.argv[3] == "generic"
scene, W, H, spp, nb, world, rank = CFG[name]
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox())
if scene < 0:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from rtlibs import LARGE_SCENE_CAMERA, large_scene
    g.set_scene(large_scene(-scene, seed=17)); g.set_camera(**LARGE_SCENE_CAMERA)
    generic = True
else:
    g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt")
if os.environ.get("RT_JIT_FLAGS"): g.set_tuning(jit_flags=os.environ["RT_JIT_FLAGS"])     # e.g. -gline-tables-only for PC sampling
if os.environ.get("RT_TRACE_KNOWN_TAPS"): g.set_tuning(trace_known_taps=True)              # no tap classification (rt_lit.h): every tap is traced
if not generic:
    g.compile_scene()
strip = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
p = g.params(W, H, spp, nb, row_block=8, rank=rank, world=world)
for _ in range(frames):
    g.render_device(p, strip.data_ptr())
g.synchronize()
g.close()
