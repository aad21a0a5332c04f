"""Host time of rt_multi_frame_submit() per call and per device: the single-threaded enqueue of an
N-device frame -- per device: hipSetDevice, wait for the gather three frames back, clear + two launches + events, control-word
read-back, its share of the grouped gather -- against the projected 0.78 ms step of eight GPUs.  Groups of n contexts on the
one GPU of the box (rt_multi_create_on_one_device: the gather is device copies) and the one-rank RCCL group.
usage: multi_enqueue_probe.py [frames]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import ray_tracing_amd as rt
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H, spp, nb = 1920, 1080, 64, 4
sky = rt.load_skybox()
def run(name, m):
    m.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); m.set_skybox(sky); m.set_camera(); m.compile_scene()
    host = [rt.HostFrame(W, H) for _ in range(3)]
    p = lambda s: rt.Renderer.params(W, H, spp, nb, seed=s)   # noqa: E731
    for k in range(3):
        m.frame_submit(p(k), k % 3, host[k % 3]); m.frame_wait(k % 3)
    sub, t0 = [], time.perf_counter()
    m.frame_submit(p(0), 0, host[0]); m.frame_submit(p(1), 1, host[1])
    for k in range(K):
        if k + 2 < K:
            t = time.perf_counter(); m.frame_submit(p(k + 2), (k + 2) % 3, host[(k + 2) % 3]); sub.append((time.perf_counter() - t) * 1e6)
        m.frame_wait(k % 3)
    step = (time.perf_counter() - t0) / K * 1e3
    n = m.size() if hasattr(m, "size") else 1
    print(f"{name:46s} submit call: median {statistics.median(sub):7.1f} us, p90 {sorted(sub)[len(sub) * 9 // 10]:7.1f} us = {statistics.median(sub) / n:6.1f} us per device; "
          f"step {step:.3f} ms", flush=True)
    for h in host: h.free()
    m.close()
g = rt.Renderer(0); run("rt_frame_submit (one context)", g)
m = rt.MultiRenderer([0]); m.set_tuning(force_collective=1); run("rt_multi, one-rank RCCL communicator", m)
for n in (2, 4, 8):
    run(f"rt_multi, {n} contexts on one GPU (copies)", rt.MultiRenderer([0], on_one_device=n))
