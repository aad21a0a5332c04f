"""Trace-kernel time per sample against the samples per pixel of the launch (1080p, scene_0 or a synthetic large scene): pixels of
few samples cost more per sample -- the streams fetch pixels more often than they trace.
usage: low_spp.py [bounces [objects]]      RT_JIT_FLAGS: extra flags for the compiled kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
objects = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1920, 1080
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox())
if objects:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    from rtlibs import LARGE_SCENE_CAMERA, large_scene
    g.set_scene(large_scene(objects, seed=17)); g.set_camera(**LARGE_SCENE_CAMERA)
else:
    g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
    flags = os.environ.get("RT_JIT_FLAGS")
    if flags: g.set_tuning(jit_flags=flags)
    g.compile_scene()
buf = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
line = []
for spp in (1, 2, 4, 8, 16, 32, 64):
    p = rt.Renderer.params(W, H, spp, nb, seed=1)
    g.render_device(p, buf.data_ptr()); g.synchronize()
    g.profile(2)
    for k in range(5): g.render_device(p, buf.data_ptr()); g.synchronize()
    ms, cnt, span, pms = g.profile_collect_split(); g.profile(False)
    line.append(f"{spp}: {(ms - pms) / cnt:.3f} ms = {(ms - pms) / cnt / spp * 1e3:.1f} us/spp")
print(f"bounces {nb}, objects {objects or 9}:  " + "   ".join(line), flush=True)
g.close()
