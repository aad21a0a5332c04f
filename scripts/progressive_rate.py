"""The interactive ladder's rate at full resolution: one launch per pass (rt_progressive_pass) against several passes per launch
(rt_progressive_passes).  1920x1080, scene_0 compiled, 10 bounces (the reference's interactive settings, main.c:155).
usage: progressive_rate.py [scene index [workgroups per CU (0: the library decides)]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
if os.environ.get("RT_LIB_FILE"): rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB_FILE"])
si = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W, H, nb = 1920, 1080, 10
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{si}.txt"); g.compile_scene()
flags = os.environ.get('RT_JIT_FLAGS')
if wg or flags: g.set_tuning(workgroups_per_cu=wg, jit_flags=flags); g.compile_scene()
def run(calls, per_call):
    g.progressive_begin(W, H, init_scale=1, max_bounces=nb, seed=1)
    (g.progressive_passes(8) if per_call > 1 else [g.progressive_pass() for _ in range(8)]); g.synchronize()
    t = time.perf_counter()
    for _ in range(calls):
        if per_call == 1: g.progressive_pass()
        else: g.progressive_passes(per_call)
    host = time.perf_counter() - t
    g.synchronize()
    dt = time.perf_counter() - t
    return dt, host, g.progressive_resolve()
frames = {}
for per_call in (1, 8, 16, 32, 64, 256):
    calls = 256 // per_call
    best = None
    for rep in range(3):
        dt, host, frame = run(calls, per_call)
        if best is None or dt < best: best, best_host = dt, host
    frames[per_call] = frame
    same = bool((frame.view(np.uint32) == frames[1].view(np.uint32)).all())
    print(f"{per_call:4d} passes per call: 256 passes in {best * 1e3:8.2f} ms = {best / 256 * 1e3:6.3f} ms per pass, {W * H * 256 / best / 1e6:9.1f} Msamples/s   (the calls returned after {best_host * 1e3:6.2f} ms)   frame identical to one-by-one: {same}", flush=True)
g.close()
