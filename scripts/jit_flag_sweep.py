"""Development aid: C1 kernel time of the compiled-scene kernel under extra hiprtc flags (RT_JIT_FLAGS)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
FLAGS = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")] if len(sys.argv) > 1 else [""]
sky = rt.load_skybox()
g = rt.Renderer(0); g.set_skybox(sky); g.profile(True)
ref = None
for f in [""] + FLAGS + [""]:
    os.environ["RT_JIT_FLAGS"] = f
    g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
    try:
        g.compile_scene()
    except rt.RtError as e:
        print(f"{f!r}: compile failed: {str(e)[:150]}", flush=True); continue
    ts = []
    for it in range(7):
        fr = g.render(1920, 1080, 64, 4); ms, n = g.profile_collect()
        if it: ts.append(ms)
    if ref is None: ref = fr
    same = bool((fr.view(np.uint32) == ref.view(np.uint32)).all())
    print(f"{statistics.median(ts):7.3f} ms  identical={same}  {f!r}", flush=True)
