"""Where the culled trace of a large scene spends its steps (needs `make -C ray_tracing_amd/csrc stats`): per trace of one wave,
the cluster boxes tested, the per-lane cluster walks (steps = the busiest lane's), the member boxes tested and the exact tests.
usage: stats_large.py [objects]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_amd as rt
from rtlibs import LARGE_SCENE_CAMERA, large_scene
rt.LIB_PATH = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_hip_stats.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(large_scene(n, seed=17)); g.set_camera(**LARGE_SCENE_CAMERA)
out = (C.c_ulonglong * 128)()
rt.lib().rt_stats_read(out, 1)
g.render(480, 270, 4, 5)
rt.lib().rt_stats_read(out, 1)
site = lambda k: (out[2 * k], out[2 * k + 1])
traces, tl = site(9)
print(f"{n} objects: {traces} culled traces of a wave, {tl / max(traces, 1):.1f} lanes active on average")
for k, name in ((32, "cluster boxes tested, every one for every ray (wave-uniform steps; scenes of few clusters)"),
                (43, "group boxes tested (wave-uniform steps)"), (44, "dealt steps of (ray, group) pairs (64 pairs a step)"),
                (45, "cluster boxes on the groups' grids (steps of EIGHT boxes; one per dealt step)"), (33, "cluster walk steps (a step = every lane that still has a cluster takes its next one)"),
                (34, "member boxes tested (steps of EIGHT boxes; one per walk step)"), (35, "exact tests (batches of up to 64 queued (ray, object) candidates)")):
    e, l = site(k)
    print(f"  {name:90s} {e / max(traces, 1):8.1f} per trace, {l / max(e, 1):5.1f} lanes active")
