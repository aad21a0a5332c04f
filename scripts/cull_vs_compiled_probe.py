"""Scenes of 16 ... 64 objects (tests/rtlibs.py large_scene): the compiled kernel (every object, unrolled), the generic kernel
(every object, loop) and the cluster cull (a build with RT_CULL_MIN_OBJECTS lowered: scripts/patches/cull_from_16.py).
usage: cull_vs_compiled_probe.py <librt_hip.so> <librt_hip.so with the cull from 16 objects>"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ray_tracing_amd as rt
from rtlibs import LARGE_SCENE_CAMERA, large_scene

def load(path):
    rt._lib = None
    rt.LIB_PATH = os.path.abspath(path)
    L = rt.lib()
    r = rt.Renderer(0)
    r._L = L
    return r

W, H, spp, nb = 1920, 1080, 8, 5
sky = rt.load_skybox()
plain, culled = load(sys.argv[1]), load(sys.argv[2])
for g in (plain, culled):
    g.set_skybox(sky); g.profile(True)
d = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
def timed(g, L):
    rt._lib = L
    t = []
    for it in range(5):
        g.render_device(g.params(W, H, spp, nb, seed=1), d.data_ptr()); g.synchronize()
        ms, _ = g.profile_collect()
        if it: t.append(ms)
    return statistics.median(t), d.cpu().numpy().copy()
for n in (16, 24, 32, 36, 40, 44, 48, 56, 64):
    scene = large_scene(n, seed=17)
    rt._lib = plain._L
    plain.set_scene(scene); plain.set_camera(**LARGE_SCENE_CAMERA)
    generic_ms, f0 = timed(plain, plain._L)
    rt._lib = plain._L; plain.compile_scene()
    compiled_ms, f1 = timed(plain, plain._L)
    rt._lib = culled._L
    culled.set_scene(scene); culled.set_camera(**LARGE_SCENE_CAMERA)
    culled_ms, f2 = timed(culled, culled._L)
    same = bool((f0.view(np.uint32) == f1.view(np.uint32)).all() and (f0.view(np.uint32) == f2.view(np.uint32)).all())
    print(f"{n:3d} objects: generic {generic_ms:7.3f} ms   compiled {compiled_ms:7.3f} ms   culled {culled_ms:7.3f} ms   identical {same}", flush=True)
