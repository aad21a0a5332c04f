"""Does a frame's copy to pinned host memory need compute-unit slots?  A whole-chip persistent launch (C2 frame, ~9 ms) is enqueued,
then a 25 MB device-to-host copy on another stream: if the copy is done by an SDMA engine it returns after ~0.5 ms, if by a blit
kernel it waits for workgroup slots, i.e. for the launch to drain.  Prints the environment's HSA / HIP / ROC variables too."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
print({k: v for k, v in os.environ.items() if k.startswith(("HSA", "HIP", "ROC", "GPU", "AMD"))})
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_1.txt"); g.compile_scene()
W, H = 1920, 1080
dev = torch.device("cuda", 0)
buf = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
src = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
host = torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True)
copy = torch.cuda.Stream(dev, priority=-1)
small_host = torch.empty((262144,), dtype=torch.float32, pin_memory=True)
p = rt.Renderer.params(W, H, 256, 8, seed=1)
for rep in range(3):
    g.render_device(p, buf.data_ptr()); g.synchronize()
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(copy):
        host.copy_(src, non_blocking=True)
    copy.synchronize()
    alone = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    g.render_device(p, buf.data_ptr())
    time.sleep(0.001)                       # the launch is running
    t1 = time.perf_counter()
    with torch.cuda.stream(copy):
        host.copy_(src, non_blocking=True)
    copy.synchronize()
    beside = (time.perf_counter() - t1) * 1e3
    g.synchronize()
    launch = (time.perf_counter() - t) * 1e3
    line = []
    for n in (16, 256, 1024, 4096, 16384, 65536, 262144):          # floats
        g.render_device(p, buf.data_ptr())
        time.sleep(0.001)
        t1 = time.perf_counter()
        with torch.cuda.stream(copy):
            small_host[:n].copy_(src.view(-1)[:n], non_blocking=True)
        copy.synchronize()
        line.append(f"{4 * n} B: {(time.perf_counter() - t1) * 1e3:.3f} ms")
        g.synchronize()
    print("beside a running whole-chip launch:  " + "   ".join(line))
    print(f"copy alone {alone:.3f} ms; beside a running whole-chip launch {beside:.3f} ms (the launch: {launch:.3f} ms from enqueue to end)", flush=True)
g.close()
