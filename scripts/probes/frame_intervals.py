"""Development aid: the sequence of delivery intervals of a twenty-frame run through the C ABI frame queue (C1), by frames in flight and
by rt_tuning.workgroups_per_cu (0: the library decides per launch from what is ahead of it)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
g = rt.Renderer(0); g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera(); g.compile_scene(); g.reserve(1920,1080)
for depth, wg in ((2, 0), (3, 0), (3, 2), (4, 2)):
    g.set_tuning(workgroups_per_cu=wg)
    loop = FrameLoop(g, 1920, 1080, 64, 4, depth=depth)
    loop.run(range(0, 8)); torch.cuda.synchronize()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = loop.run(range(10, 30)); torch.cuda.synchronize(); t1 = time.perf_counter()
        iv = np.diff([t0] + st) * 1e3
        print(f"depth {depth}, workgroups_per_cu {wg}: total {(t1 - t0) * 1e3:.2f} ms = {(t1 - t0) * 50:.3f} per step; intervals: " + " ".join(f"{x:.2f}" for x in iv) + f" | tail after last stamp {(t1 - st[-1]) * 1e3:.2f}")
    loop.close()
g.close()
