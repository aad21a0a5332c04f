"""Development aid: frames with the "taps certainly lit" flags honoured and ignored (rt_tuning.trace_known_taps): must be
bit-identical; prints both kernel times.  usage: lit_ab.py [C1|C2|C1strip8|...]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ray_tracing_amd as rt
CFG = {"C1": (0, 1920, 1080, 64, 4, 1, 0), "C2": (1, 1920, 1080, 256, 8, 1, 0), "C3": (2, 3840, 2160, 64, 8, 1, 0),
       "C4strip": (0, 3840, 2160, 1024, 8, 8, 3), "C1strip8": (0, 1920, 1080, 64, 4, 8, 3), "P1": (0, 1920, 1080, 1, 10, 1, 0)}
for name in (sys.argv[1:] or ["C1", "C1strip8", "C2"]):
    scene, W, H, spp, nb, world, rank = CFG[name]
    g = rt.Renderer(0)
    g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
    g.profile(True)
    strip = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
    p = g.params(W, H, spp, nb, row_block=8, rank=rank, world=world)
    out = {}
    for mode in (1, 0, 1, 0):
        g.set_tuning(trace_known_taps=mode)
        ts = []
        for it in range(6):
            strip.fill_(float("nan")); torch.cuda.synchronize()
            g.render_device(p, strip.data_ptr()); g.synchronize()
            ms, n = g.profile_collect()
            if it: ts.append(ms)
        out.setdefault(mode, []).append((statistics.median(ts), strip.cpu().numpy().copy()))
    a = min(t for t, _ in out[1]); b = min(t for t, _ in out[0])
    same = all((f.view(np.uint32) == out[1][0][1].view(np.uint32)).all() for _, f in out[0] + out[1])
    print(f"{name}: every tap traced {a:.3f} ms   known taps skipped {b:.3f} ms   ratio {b / a:.4f}   identical={same}", flush=True)
    g.close()
