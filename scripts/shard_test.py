"""Development aid: C1 and world-8 strip kernel time with 1 vs 64 dequeue counters (RT_SHARDS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb = 1920, 1080, 64, 4
for world in (1, 8):
    rows = rt.strip_rows(H, 8, world)
    strip = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda:0")
    p = g.params(W, H, spp, nb, row_block=8, rank=0, world=world)
    for rounds in range(2):
        for shards in ("1", "64"):
            os.environ["RT_SHARDS"] = shards
            ts = []
            for it in range(8):
                torch.cuda.synchronize(); t = time.perf_counter()
                g.render_device(p, strip.data_ptr()); g.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
            print(f"world {world} shards {shards:2s}: min {min(ts):.3f} ms  median {sorted(ts)[4]:.3f}", flush=True)
