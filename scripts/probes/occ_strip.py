"""Development aid: time of the strip one of 8 ranks renders of C1, at 2..4 resident workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.profile(True); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb, world = 1920, 1080, 64, 4, 8
strip = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
for per_cu in (4, 3, 2):
    for shards in (64, 1):
        g.set_tuning(workgroups_per_cu=per_cu, dequeue_shards=shards)
        ts = []
        for it in range(5):
            g.render_device(g.params(W, H, spp, nb, row_block=8, rank=3, world=world), strip.data_ptr()); g.synchronize()
            ms, n = g.profile_collect()
            if it: ts.append(ms)
        print(f"strip 3/8: {per_cu} workgroups per CU, {shards} lists: {min(ts):.3f} ms", flush=True)
