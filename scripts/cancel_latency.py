"""How long a launch goes on after rt_cancel(): a whole-chip launch of C1's frame at 64 / 256 / 1024 samples per pixel is cancelled
3 ms after its enqueue (from this thread: rt_cancel() is one store), and the time from the call to the launch's end is measured --
what rt_progressive_invalidate() and a cancelled rt_render() wait for (main.c:316-317: the reference's workers look once per pixel
row).  A wave looks for the request when it CLAIMS pixels from its list.  usage: cancel_latency.py [lib.so]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
if len(sys.argv) > 1: rt.LIB_PATH = os.path.abspath(sys.argv[1])
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera(); g.compile_scene()
W, H = 1920, 1080
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
print(f"library: {rt.LIB_PATH}")
for spp in (64, 256, 1024):
    p = g.params(W, H, spp, 8)
    g.render_device(p, buf.data_ptr()); g.synchronize()
    t0 = time.perf_counter(); g.render_device(p, buf.data_ptr()); g.synchronize(); full = (time.perf_counter() - t0) * 1e3
    after = []
    for rep in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.render_device(p, buf.data_ptr())
        while time.perf_counter() - t0 < 0.003:
            pass
        t1 = time.perf_counter()
        g.cancel()
        g.synchronize()
        after.append((time.perf_counter() - t1) * 1e3)
        assert g.was_cancelled() or full < 3.5
    after.sort()
    print(f"{spp:5d} samples per pixel: whole launch {full:8.2f} ms; cancelled 3 ms in: ends {statistics.median(after):6.3f} ms after the call (min {after[0]:.3f}, max {after[-1]:.3f})", flush=True)
g.close()
