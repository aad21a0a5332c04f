"""The rate of single interactive passes (1080p, scene_0, 10 bounces) in a context that has done other things first: which of them
costs the passes their overlap?  usage: progressive_reps.py [streams|frames|device|reserve|second|latency]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
what = sys.argv[1] if len(sys.argv) > 1 else ""
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
W, H = 1920, 1080
def rate(tag):
    for rep in range(2):
        g.progressive_begin(W, H, init_scale=1, max_bounces=10, seed=rep)
        [g.progressive_pass() for _ in range(8)]; g.synchronize()
        t = time.perf_counter()
        [g.progressive_pass() for _ in range(256)]
        g.synchronize()
        print(f"{tag}: {(time.perf_counter() - t) / 256 * 1e3:.4f} ms per pass", flush=True)
if not what.startswith("first-"): rate("fresh context")
what = what.replace("first-", "")
if what == "streams":
    for w in range(rt.LAUNCH_SETS): g.stream(w)
    rate("all render streams made")
if what == "frames":
    loop = FrameLoop(g, W, H, 64, 4, depth=2)
    loop.run(list(range(12)))
    rate("after twelve frames through the frame queue")
if what == "device":
    buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    for k in range(10): g.render_device(rt.Renderer.params(W, H, 64, 4, seed=k), buf.data_ptr(), stream=g.stream(k % rt.LAUNCH_SETS))
    g.synchronize()
    rate("after ten launches on all render streams")
if what == "reserve":
    g.reserve(W, H)
    rate("after rt_reserve")
if what == "second":
    c = rt.Renderer(0); c.set_skybox(rt.load_skybox()); c.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); c.compile_scene()
    c.set_tuning(poison_frame=True); c.render(W, H, 64, 4, seed=1)
    rate("with a second context alive")
    c.close()
    rate("after the second context was closed")
if what == "latency":
    for k in range(7): g.render(W, H, 64, 4, seed=k)
    g.profile(2)
    for k in range(5): g.render(W, H, 64, 4, seed=k)
    g.profile_collect_split(); g.profile(False)
    rate("after blocking renders with split profiling")
g.close()
