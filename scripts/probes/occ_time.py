"""Development aid: C1 / C2 kernel time at 1..4 resident workgroups per CU (= waves per SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.profile(True)
for name, scene, W, H, spp, nb in [("C1", 0, 1920, 1080, 64, 4), ("C2", 1, 1920, 1080, 256, 8)]:
    g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
    for per_cu in (4, 3, 2, 1):
        g.set_tuning(workgroups_per_cu=per_cu)
        ts = []
        for it in range(3):
            g.render(W, H, spp, nb); ms, n = g.profile_collect()
            if it: ts.append(ms)
        print(f"{name}: {per_cu} waves per SIMD: {min(ts):.3f} ms", flush=True)
