"""Development aid: kernel time of C1 / C2 under different pixel_streams / dequeue_shards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox())
g.profile(True)
cfgs = [("C1", 0, 1920, 1080, 64, 4), ("C2", 1, 1920, 1080, 256, 8)]
sweep = [dict(), dict(pixel_streams=8), dict(pixel_streams=4), dict(pixel_streams=2), dict(pixel_streams=1), dict(pixel_streams=8, dequeue_shards=1)]
for name, scene, W, H, spp, nb in cfgs:
    g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
    for t in sweep:
        g.set_tuning(**t)
        ts = []
        for it in range(4):
            g.render(W, H, spp, nb)
            ms, n = g.profile_collect()
            if it: ts.append(ms)
        print(f"{name} {t}: {min(ts):.3f} ms", flush=True)
