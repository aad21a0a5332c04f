"""Development aid: the C ABI's frame queue (rt_frame_submit / rt_frame_wait) against the torch-side loop
(multi_gpu.TiledFrame) in ONE process on ONE device: ms per step, per-launch kernel time, span per launch.
usage: frame_loop_probe.py [lib.so ...]   (default: the tree's library)"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
from ray_tracing_amd.multi_gpu import TiledFrame

libs = sys.argv[1:] or [rt.LIB_PATH]
cfg = os.environ.get("PROBE_CFG", "C1")
W, H, spp, nb, scene = {"C1": (1920, 1080, 64, 4, 0), "C3": (3840, 2160, 64, 8, 2), "C2": (1920, 1080, 256, 8, 1)}[cfg]
K = int(os.environ.get("PROBE_STEPS", "30"))
sky = None
rs = []
for p in libs:
    rt._lib = None; rt.LIB_PATH = os.path.abspath(p)
    L = rt.lib()
    r = rt.Renderer(0); r._L = L
    if sky is None: sky = rt.load_skybox()
    r.set_skybox(sky); r.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); r.set_camera(); r.compile_scene(); r.reserve(W, H)
    rs.append((p, r))

def timed(r, run):
    run(3); torch.cuda.synchronize()
    r.profile(True)
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / K
    ms, n, span = r.profile_collect_span(); r.profile(False)
    return dt, ms / n, span / n

for rep in range(3):
    for p, r in rs:
        rt._lib = r._L
        for depth in (2, 3):
            loop = FrameLoop(r, W, H, spp, nb, depth=depth)
            dt, per, span = timed(r, lambda n: loop.run(range(n)))
            print(f"{os.path.relpath(p, ROOT):45s} C frame queue depth {depth}: {dt:.3f} ms/step  per-launch {per:.3f}  span/launch {span:.3f}", flush=True)
            loop.close()
        if os.environ.get("PROBE_GROUP"):
            m = rt.MultiRenderer([0]); m.set_tuning(force_collective=1)
            m.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); m.set_skybox(sky); m.set_camera(); m.compile_scene()
            for depth in (2, 3):
                loop = FrameLoop(m, W, H, spp, nb, depth=depth)
                dt, per, span = timed(m.context(0), lambda n: loop.run(range(n)))
                print(f"{os.path.relpath(p, ROOT):45s} group queue, 1-rank RCCL, depth {depth}: {dt:.3f} ms/step  per-launch {per:.3f}  span/launch {span:.3f}", flush=True)
                loop.close()
            m.close()
        t = TiledFrame(r, W, H, spp, nb, device=torch.device("cuda", 0))
        def run(n):
            for k in range(n): t.step(seed=k)
            t.flush()
        dt, per, span = timed(r, run)
        print(f"{os.path.relpath(p, ROOT):45s} TiledFrame (torch)       : {dt:.3f} ms/step  per-launch {per:.3f}  span/launch {span:.3f}", flush=True)
        del t
