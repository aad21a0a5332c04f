import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb = 1920, 1080, 64, 4
for d in (2, 3, 4):
    loop = FrameLoop(g, W, H, spp, nb, depth=d)
    loop.run(range(6)); torch.cuda.synchronize()
    for K in (20, 40):
        t0 = time.perf_counter()
        st = loop.run(range(100, 100 + K)); torch.cuda.synchronize()
        t1 = time.perf_counter()
        iv = [(b - a) * 1e3 for a, b in zip([t0] + st[:-1], st)]
        print(f"depth {d}, K {K}: {(t1 - t0) / K * 1e3:.3f} ms per step; intervals: " + " ".join(f"{x:.1f}" for x in iv), flush=True)
    loop.close()
g.close()
