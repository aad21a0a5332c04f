"""Per-site lane utilisation of the trace kernel on the strip one of eight ranks renders of C1 (needs `make -C ray_tracing_amd/csrc stats`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
rt.LIB_PATH = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_hip_stats.so")
W, H, spp, nb, world, rank = 1920, 1080, 64, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 3
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
strip = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
rt.lib().rt_stats_read(out, 1)
g.render_device(g.params(W, H, spp, nb, row_block=8, rank=rank % world, world=world), strip.data_ptr()); g.synchronize()
rt.lib().rt_stats_read(out, 1)
names = {7: "round", 8: "shade body", 12: "trace batch", 13: "trace batch active", 17: "back body", 20: "supply attempt", 21: "pixel fetch event",
         22: "sample hand-out", 23: "in-order sum pass", 24: "lanes left without a sample"}
waves = 256 * 16
for k in sorted(names):
    n, lanes = out[2 * k], out[2 * k + 1]
    if n:
        print(f"{names[k]:30s} per wave {n / waves:8.2f}   avg active lanes {lanes / n:5.1f} ({lanes / n / 64 * 100:4.1f}%)")
