"""Development aid: resident workgroups per CU (rt_tuning.workgroups_per_cu; 0 = the library's choice) against the step time of
launches that follow each other on the context's two streams -- whole C1 frames and the strips of one of 8 / 4 / 2 ranks,
device only -- and against the step time of the C ABI's frame queue (frames to pinned host memory, what bench.py times) on
C1 / C2 / C3.  Fewer workgroups per launch leave room for the next launch to be resident beside this one's tail.
usage: wg_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
dev = torch.device("cuda", 0)
CFG = {"C1": ("scene_0.txt", 1920, 1080, 64, 4), "C2": ("scene_1.txt", 1920, 1080, 256, 8), "C3": ("scene_2.txt", 3840, 2160, 64, 8)}
sky = rt.load_skybox()

def renderer(scene):
    g = rt.Renderer(0)
    g.set_skybox(sky); g.set_scene(f"{rt.DATA_DIR}/{scene}"); g.set_camera(); g.compile_scene()
    return g

def strips(g, world, wg, n, W, H, spp, nb):
    g.set_tuning(workgroups_per_cu=wg)
    streams = [g.stream(0), g.stream(1)]
    rows = rt.strip_rows(H, 8, world)
    out = [torch.empty((rows, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    for k in range(4):
        g.render_device(g.params(W, H, spp, nb, seed=k, row_block=8, rank=0, world=world), out[k & 1].data_ptr(), streams[k & 1])
    g.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        g.render_device(g.params(W, H, spp, nb, seed=k, row_block=8, rank=0, world=world), out[k & 1].data_ptr(), streams[k & 1])
    g.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def frames(g, wg, n, W, H, spp, nb):
    g.set_tuning(workgroups_per_cu=wg)
    loop = FrameLoop(g, W, H, spp, nb)
    loop.run(range(3))
    t0 = time.perf_counter()
    loop.run(range(n))
    dt = (time.perf_counter() - t0) / n * 1e3
    loop.close()
    return dt

WGS = (4, 3, 2, 0)
g = renderer("scene_0.txt")
for rep in range(2):
    for world, n in ((1, 20), (2, 30), (4, 40), (8, 60)):
        print(f"C1 strip of world {world}, device only: " + "  ".join(f"{wg} wg/CU {strips(g, world, wg, n, 1920, 1080, 64, 4):.3f} ms" for wg in WGS), flush=True)
g.close()
for cfg in ("C1", "C2", "C3"):
    scene, W, H, spp, nb = CFG[cfg]
    g = renderer(scene)
    for rep in range(3):
        print(f"{cfg} frame queue (to host): " + "  ".join(f"{wg} wg/CU {frames(g, wg, 20, W, H, spp, nb):.3f} ms" for wg in WGS), flush=True)
    g.close()
