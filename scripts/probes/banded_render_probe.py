"""A blocking frame as B contiguous bands: B launches on B of the context's streams (row_block = the band's rows, world = B: band b
is "rank" b), each band copied to pinned host memory behind its own launch, against one launch and one copy (rt_render).  Wall time
from the first call to the frame in host memory; frames must be identical.  usage: banded_render_probe.py [C1|C3|I16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ray_tracing_amd as rt
cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
scene, W, H, spp, nb = {"C1": (0, 1920, 1080, 64, 4), "C3": (2, 3840, 2160, 64, 8), "I16": (0, 1280, 720, 16, 10)}[cfg]
dev = torch.device("cuda", 0)
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
host = torch.empty((H + 64, W, 3), dtype=torch.float32, pin_memory=True)
copy = torch.cuda.Stream(dev, priority=-1)
def blocking(seed):
    t = time.perf_counter()
    f = g.render(W, H, spp, nb, seed=seed)
    return (time.perf_counter() - t) * 1e3, f
def banded(seed, B):
    rows = -(-H // B); rows = -(-rows // 8) * 8
    dframe = banded.buf.setdefault(B, torch.empty((rows * B, W, 3), dtype=torch.float32, device=dev))
    streams = [torch.cuda.ExternalStream(g.stream(b % rt.LAUNCH_SETS), device=dev) for b in range(B)]
    torch.cuda.synchronize()
    t = time.perf_counter()
    for b in range(B):
        lo = b * rows
        if lo >= H: break
        p = rt.Renderer.params(W, H, spp, nb, seed=seed, row_block=rows, rank=b, world=B)
        g.render_device(p, dframe[lo:].data_ptr(), streams[b].cuda_stream)
        ev = torch.cuda.Event(); ev.record(streams[b])
        copy.wait_event(ev)
        with torch.cuda.stream(copy):
            n = min(rows, H - lo)
            host[lo:lo + n].copy_(dframe[lo:lo + n], non_blocking=True)
    copy.synchronize()
    return (time.perf_counter() - t) * 1e3, host[:H].numpy().copy()
banded.buf = {}
for k in range(3): blocking(k)
ref = {}
tb = []
for k in range(9):
    ms, f = blocking(100 + k); tb.append(ms); ref[k] = f
print(f"{cfg}: rt_render (one launch, one copy, pageable destination): median {sorted(tb)[4]:.3f} ms", flush=True)
for B in (1, 2, 3, 4, 6, 8):
    for k in range(3): banded(k, B)
    t, same = [], True
    for k in range(9):
        ms, f = banded(100 + k, B); t.append(ms)
        same = same and bool((f.view(np.uint32) == ref[k].view(np.uint32)).all())
    print(f"{cfg}: {B} band(s), pinned destination: median {sorted(t)[4]:.3f} ms   identical: {same}", flush=True)
g.close()
