"""Development aid: how long after rt_cancel() a frame in flight gives up -- generic and compiled kernel, 1024 and 256 samples per
pixel, 20 ms into the frame -- and what a whole frame takes before and after (a request must leave nothing behind).
usage: cancel_probe.py [librt_hip.so of another build]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracing_amd as rt
if len(sys.argv) > 1: rt.LIB_PATH = os.path.abspath(sys.argv[1])
sky = rt.load_skybox()
for compiled in (False, True):
    for spp in (1024, 256):
        q = rt.Renderer(0)
        q.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); q.set_skybox(sky); q.set_camera()
        if compiled: q.compile_scene()
        W, H, nb = 1920, 1080, 8
        a = rt.HostFrame(W, H)
        for rep in range(3):
            t00 = time.time()
            q.frame_submit(rt.Renderer.params(W, H, spp, nb, seed=1), 0, a)
            ok = q.frame_wait(0); full = time.time() - t00
            q.frame_submit(rt.Renderer.params(W, H, spp, nb, seed=1), 0, a)
            time.sleep(0.02)
            t0 = time.time(); q.cancel(); t1 = time.time()
            r = q.frame_wait(0); t2 = time.time()
            print(f"compiled={compiled} spp={spp}: whole frame {full*1e3:.1f} ms; cancel() took {(t1-t0)*1e3:.2f} ms, wait after cancel {(t2-t1)*1e3:.2f} ms, completed={r}", flush=True)
        a.free(); q.close()
