import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import ray_tracing_amd as rt
g = rt.Renderer(0); g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt")
for (W,H,spp,nb) in [(1920,1080,64,4),(640,360,64,4),(320,180,16,4),(320,180,64,4)]:
    for P in (0,8,4):
        g.set_tuning(pixel_streams=P)
        a = g.render(W,H,spp,nb); s = g.render(W,H,spp,nb,kernel=1)
        bad = (a.view(np.uint32)!=s.view(np.uint32)).any(axis=-1)
        ys,xs = np.nonzero(bad)
        print(W,H,spp,"P",P,"bad pixels",bad.sum(), "first", list(zip(ys[:5],xs[:5])), "maxdiff", np.abs(a-s).max() if bad.sum() else 0, flush=True)
        if bad.sum():
            y,x=ys[0],xs[0]; print("   a",a[y,x],"s",s[y,x], "ratio", a[y,x]/s[y,x])
