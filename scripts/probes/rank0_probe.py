"""Development aid: what rank 0 of an 8-GPU run does per frame, on ONE GPU: its own strip (strip_of_rank(0, world): the last one;
`strip=<s>` on the command line picks another, `strip=0` is the un-rotated hand-out of rounds 1-2), the
root's local share of the gather (a device copy of its strip into the gather buffer), the de-interleave of all eight
strips into the frame and the copy of the frame to pinned host memory -- pipelined as multi_gpu.TiledFrame pipelines
them (the context's render streams in rotation, post stream, copy stream, one strip buffer more than streams).  Only the xGMI transfer of the seven peer
strips is missing.  Prints ms per step for: strips alone, + gather stand-in and de-interleave, + host copy.
Also prints the time of one of the OTHER ranks' strips alone (strip 0: a longest one), which bounds the step from below.
usage: rank0_probe.py [C1|C4] [world] [strip=<s>]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from ray_tracing_amd.multi_gpu import strip_of_rank
mine = strip_of_rank(0, world)
for a in sys.argv[3:]:
    if a.startswith("strip="): mine = int(a[6:])
first = (world - mine) % world          # strip `mine` sits at position 0 of the gathered buffer
W, H, spp, nb = {"C1": (1920, 1080, 64, 4), "C4": (3840, 2160, 1024, 8)}[cfg]
N = 40 if cfg == "C1" else 6
dev = torch.device("cuda", 0)
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera(); g.compile_scene()
rows = rt.strip_rows(H, 8, world)
S = rt.LAUNCH_SETS
streams = [torch.cuda.ExternalStream(g.stream(w), device=dev) for w in range(S)]
post, copy = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)
strip = [torch.empty((rows, W, 3), dtype=torch.float32, device=dev) for _ in range(S + 1)]
strips = [torch.zeros((world, rows, W, 3), dtype=torch.float32, device=dev) for _ in range(S + 1)]
frame = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
host = [torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(2)]

def run(level, mine=mine):
    gathered, copied = [None] * (S + 1), [None] * 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        s, j, f = streams[k % S], k % (S + 1), k & 1
        with torch.cuda.stream(s):
            if gathered[j] is not None: s.wait_event(gathered[j])
            g.render_device(g.params(W, H, spp, nb, seed=k, row_block=8, rank=mine, world=world), strip[j].data_ptr(), s.cuda_stream)
            done = torch.cuda.Event(); done.record(s)
        if level >= 1:
            with torch.cuda.stream(post):
                post.wait_event(done)
                strips[j][0].copy_(strip[j], non_blocking=True)          # the root's own part of the gather
                ev = torch.cuda.Event(); ev.record(post); gathered[j] = ev
                if copied[f] is not None: post.wait_event(copied[f])
                g.deinterleave_device(strips[j].data_ptr(), frame[f].data_ptr(), W, H, 8, world, post.cuda_stream, first=first)
                ready = torch.cuda.Event(); ready.record(post)
            if level >= 2:
                with torch.cuda.stream(copy):
                    copy.wait_event(ready)
                    host[f].copy_(frame[f], non_blocking=True)
                    c = torch.cuda.Event(); c.record(copy); copied[f] = c
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3

for rep in range(3):
    a, b, c, o = run(0), run(1), run(2), run(0, 0)
    print(f"{cfg}, rank 0 of {world} (strip {mine}) on one GPU: strips alone {a:.3f} ms per step | + gather stand-in + de-interleave {b:.3f} | + frame to pinned host memory {c:.3f} || another rank (strip 0) alone {o:.3f}", flush=True)
