"""One rank's share of a C1 frame on N GPUs, as the frame loop runs it (launches rotating through the context's streams and
scratch sets, frames left on the device): ms per strip step for world = 1, 2, 4, 8 -- what the kernel side of the N-GPU step costs
on ONE device, to set beside frame / N.
usage: strip_loop_probe.py [streams [workgroups per CU]]      streams = launches in flight: 2 ... RT_LAUNCH_SETS (the default);
workgroups per CU: rt_tuning.workgroups_per_cu (0: the library decides)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
if os.environ.get("RT_LIB_FILE"): rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB_FILE"])
W, H, spp, nb = 1920, 1080, 64, 4
S = int(sys.argv[1]) if len(sys.argv) > 1 else rt.LAUNCH_SETS
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
if len(sys.argv) > 2 and int(sys.argv[2]): g.set_tuning(workgroups_per_cu=int(sys.argv[2]))
base = None
for world in (1, 2, 4, 8):
    rank = world // 2
    p_of = lambda k: rt.Renderer.params(W, H, spp, nb, seed=k, row_block=8, rank=rank, world=world)
    bufs = [torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0") for _ in range(S)]
    for k in range(2 * S):
        g.render_device(p_of(k), bufs[k % S].data_ptr(), stream=g.stream(k % S))
    g.synchronize()
    n = 40
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(n):            # launches rotate through S of the context's render streams (and its scratch sets): S in flight
            g.render_device(p_of(k), bufs[k % S].data_ptr(), stream=g.stream(k % S))
        g.synchronize()
        dt = (time.perf_counter() - t) / n * 1e3
        best = dt if best is None or dt < best else best
    # the same strip one launch at a time
    g.profile(True)
    d = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
    for k in range(10):
        g.render_device(p_of(k), d.data_ptr()); g.synchronize()
    ms, cnt = g.profile_collect(); g.profile(False)
    if base is None: base = best
    print(f"world {world}: strip of rank {rank}: {best:.3f} ms per step with {S} in flight ({base / world:.3f} = frame / {world}; efficiency {base / world / best:.3f}), one launch alone {ms / cnt:.3f} ms", flush=True)
g.close()
