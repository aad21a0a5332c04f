"""Interleaved A/B timing of two builds of librt_hip.so on synthetic large scenes (tests/rtlibs.py large_scene), one process, one
device; frames must be identical.  usage: ab_large.py libA.so libB.so [rounds [objects,objects,...]]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ray_tracing_amd as rt
from rtlibs import LARGE_SCENE_CAMERA, large_scene, stick_scene
libs, rounds = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 7
sky, rs = None, []
for p in libs:
    rt._lib = None; rt.LIB_PATH = os.path.abspath(p)
    L = rt.lib(); r = rt.Renderer(0); r._L = L
    if sky is None: sky = rt.load_skybox()
    r.set_skybox(sky); r.profile(True); rs.append(r)
W, H, spp, nb = 1920, 1080, 8, 5
for n in ([int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (40, 64, 128, 256, 512, 1024)):
    scene = stick_scene(n, seed=17) if os.environ.get("AB_SCENE") == "sticks" else large_scene(n, seed=17)      # AB_SCENE=sticks: objects that span the scene
    for r in rs:
        rt._lib = r._L; r.set_scene(scene); r.set_camera(**LARGE_SCENE_CAMERA)
    t = [[], []]; frames = [None, None]
    for it in range(rounds + 1):
        for k, r in enumerate(rs):
            rt._lib = r._L
            d = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); torch.cuda.synchronize()
            r.render_device(r.params(W, H, spp, nb, seed=1), d.data_ptr()); r.synchronize()
            ms, _ = r.profile_collect()
            if it: t[k].append(ms)
            frames[k] = d.cpu().numpy()
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"{n:5d} objects: A {a:8.3f} ms   B {b:8.3f} ms   B/A {b / a:.4f}   identical={bool((frames[0].view(np.uint32) == frames[1].view(np.uint32)).all())}", flush=True)
