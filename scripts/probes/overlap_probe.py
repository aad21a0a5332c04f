"""Development aid: frames per second of one context when consecutive launches go to one stream or alternate between
the context's two streams (rt_stream: the tail of one launch then runs under the start of the next).
usage: overlap_probe.py [world ...]     (C1; the strip of rank 3 of `world`)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb = 1920, 1080, int(os.environ.get("SPP", 64)), 4
N = 40
streams = [g.stream(0), g.stream(1)]
for world in [int(a) for a in sys.argv[1:]] or [8, 1]:
    rows = rt.strip_rows(H, 8, world)
    bufs = [torch.empty((rows, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for mode in ("one stream", "two streams"):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(N):
                i = (k & 1) if mode == "two streams" else 0
                g.render_device(g.params(W, H, spp, nb, seed=k, row_block=8, rank=3 % world, world=world), bufs[k & 1].data_ptr(), streams[i])
            g.synchronize(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N * 1e3
        print(f"world {world}: {mode}: {dt:.3f} ms per frame", flush=True)
