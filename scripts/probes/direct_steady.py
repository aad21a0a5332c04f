"""One sample per pixel (every lane takes a pixel for itself) against the eight streams, per sample, as the frame grows: at 8K the
ramp and the tail of a launch are a few per cent, what is left is the steady pace of the two modes.  scene_0, 10 bounces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
for W, H in ((1920, 1080), (3840, 2160), (7680, 4320)):
    buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    for spp in (1, 8, 64):
        p = rt.Renderer.params(W, H, spp, 10, seed=1)
        g.render_device(p, buf.data_ptr()); g.synchronize()
        g.profile(2)
        for k in range(3): g.render_device(p, buf.data_ptr()); g.synchronize()
        ms, cnt, span, pms = g.profile_collect_split(); g.profile(False)
        tms = ms - pms
        print(f"{W}x{H} spp {spp:3d}: trace kernel {tms / cnt:9.3f} ms = {tms / cnt / spp * 1e3 * (1920 * 1080) / (W * H):7.1f} us per sample per 1080p pixels; camera-ray pass {pms / cnt:6.3f} ms", flush=True)
g.close()
