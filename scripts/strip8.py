"""Development aid: world-8 strip time vs resident workgroups per CU and chunk count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H, spp, nb, world = 1920, 1080, 64, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = rt.strip_rows(H, 8, world)
strip = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda:0")
p = g.params(W, H, spp, nb, row_block=8, rank=0, world=world)
for per_cu in ("4", "3", "2"):
    os.environ["RT_WF_PER_CU"] = per_cu
    out = []
    for chunks in (4, 8, 11, 16):
        os.environ["RT_CHUNKS"] = str(chunks)
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            g.render_device(p, strip.data_ptr()); g.synchronize()
            best = min(best, (time.perf_counter() - t) * 1e3)
        out.append(f"{chunks}:{best:.3f}")
    print(f"world {world} WG/CU {per_cu}: " + "  ".join(out), flush=True)
