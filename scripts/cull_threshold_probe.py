"""Where does the cluster cull start to pay?  Frames of n objects (tests/rtlibs.py large_scene), culled vs every object tested,
interleaved on one device (HIP events over camera-ray pass + trace kernel).  usage: cull_threshold_probe.py"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ray_tracing_amd as rt
from rtlibs import LARGE_SCENE_CAMERA, large_scene
W, H, spp, nb = 1920, 1080, 8, 5
sky = rt.load_skybox()
g = rt.Renderer(0); g.set_skybox(sky); g.profile(True)
for n in (65, 80, 100, 128, 192, 256, 512, 1024):
    g.set_scene(large_scene(n, seed=17)); g.set_camera(**LARGE_SCENE_CAMERA)
    t = {False: [], True: []}; frames = {}
    for it in range(6):
        for every in (False, True):
            g.set_tuning(test_every_object=every)
            d = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            g.render_device(g.params(W, H, spp, nb, seed=1), d.data_ptr()); g.synchronize()
            ms, _ = g.profile_collect()
            if it: t[every].append(ms)
            frames[every] = d.cpu().numpy()
    a, b = statistics.median(t[False]), statistics.median(t[True])
    same = bool((frames[False].view(np.uint32) == frames[True].view(np.uint32)).all())
    print(f"{n:5d} objects: culled {a:8.3f} ms   every object {b:8.3f} ms   ratio {b / a:5.2f}   identical {same}", flush=True)
