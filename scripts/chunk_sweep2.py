import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox())
for name, scene, W, H, spp, nb in [("C2", 1, 1920, 1080, 256, 8), ("C3", 2, 3840, 2160, 64, 8)]:
    g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
    strip = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    p = g.params(W, H, spp, nb)
    out = []
    for chunks in (1, 2, 4, 8, 16):
        os.environ["RT_CHUNKS"] = str(chunks)
        best = 1e9
        for it in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            g.render_device(p, strip.data_ptr()); g.synchronize()
            best = min(best, (time.perf_counter() - t) * 1e3)
        out.append(f"{chunks}:{best:.3f}")
    print(f"{name}: " + "  ".join(out), flush=True)
