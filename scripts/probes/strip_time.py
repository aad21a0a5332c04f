"""Kernel time of one rank's strip for world = 1, 2, 4, 8 on a single GPU (what each GPU of an N-GPU
run executes), with the automatic schedule."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb = 1920, 1080, 64, 4
g.profile(True)
for world in (1, 2, 4, 8):
    rows = rt.strip_rows(H, 8, world)
    strip = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda:0")
    for mode in ("auto",):
        ts = []
        for rank in (0, world - 1):
            p = g.params(W, H, spp, nb, row_block=8, rank=rank, world=world)
            for it in range(4):
                torch.cuda.synchronize()
                import time; t = time.perf_counter()
                g.render_device(p, strip.data_ptr()); g.synchronize()
                dt = (time.perf_counter() - t) * 1e3
                ms, n = g.profile_collect()
                if it: ts.append((ms, dt))
        print(f"world {world} schedule={mode:4s}: primary pass + trace kernel {min(t[0] for t in ts):.3f} ms, wall {min(t[1] for t in ts):.3f} ms", flush=True)
