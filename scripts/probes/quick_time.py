"""Quick GPU timing of the configs in BASELINE.json (development aid; bench.py is the contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ray_tracing_amd as rt
if os.environ.get('RT_LIB'): rt.LIB_PATH = os.path.join(os.path.dirname(rt.LIB_PATH), os.environ['RT_LIB'])

def main():
    kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    sky = rt.load_skybox()
    g = rt.Renderer(0)
    g.set_skybox(sky)
    g.profile(True)
    for name, scene, W, H, spp, nb in [("C1", 0, 1920, 1080, 64, 4), ("C2", 1, 1920, 1080, 256, 8), ("C3", 2, 3840, 2160, 64, 8), ("C4 on 1 GPU", 0, 3840, 2160, 1024, 8)]:
        g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt")
        if kernel == 0 and not os.environ.get("NO_JIT"):
            g.compile_scene()
        g.render(W, H, 1, nb, kernel=kernel)
        g.profile_collect()
        t = time.time()
        f = g.render(W, H, spp, nb, kernel=kernel)
        wall = time.time() - t
        ms, n = g.profile_collect()
        print(f"{name} kernel={kernel}: kernel {ms:.2f} ms -> {W*H*spp/ms/1e3:.1f} Msamples/s  (wall incl D2H {wall*1e3:.1f} ms) mean={f.mean():.6f}", flush=True)

main()
