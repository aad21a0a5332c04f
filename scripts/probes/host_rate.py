"""Development aid: can the host keep a 1 ms frame loop fed?  A strip-sized frame (the 135 rows one of eight ranks
renders of a 1080p frame, here as a frame of its own) through TiledFrame: GPU time per step vs host time per step()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd.multi_gpu import TiledFrame
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
for H in (135, 270, 1080):
    t = TiledFrame(g, 1920, H, 64, 4, device=torch.device("cuda", 0))
    for _ in range(5): t.step()
    t.flush()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): t.step()
    t1 = time.perf_counter()
    t.flush()
    t2 = time.perf_counter()
    print(f"1920x{H}: host {1e3 * (t1 - t0) / N:.3f} ms per step() call, {1e3 * (t2 - t0) / N:.3f} ms per frame end to end")
