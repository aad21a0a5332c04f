"""Per-site lane utilisation of the wavefront kernel on C1 (needs `make -C ray_tracing_amd/csrc stats`).
usage: stats_c1.py [scene [jit|stamps|lib [spp [bounces [width height]]]]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
jit = len(sys.argv) > 2 and sys.argv[2] in ("jit", "stamps")      # the scene-specialised kernel, instrumented through jit_flags
stamps = len(sys.argv) > 2 and sys.argv[2] == "stamps"              # section time stamps only (the per-site atomics distort them)
if not jit: rt.LIB_PATH = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_hip_stats.so")
W, H, spp, nb = (1920, 1080, 64, 4) if scene == 0 else (1920, 1080, 256, 8)
if len(sys.argv) > 3: spp = int(sys.argv[3])
if len(sys.argv) > 4: nb = int(sys.argv[4])
if len(sys.argv) > 6: W, H = int(sys.argv[5]), int(sys.argv[6])
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt")
out = (C.c_ulonglong * 64)()
if jit:
    g.set_tuning(jit_flags="-DRT_STATS -DRT_STATS_STAMPS_ONLY" if stamps else "-DRT_STATS"); g.compile_scene()
    rt.lib().rt_spec_stats_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    read = lambda: rt.lib().rt_spec_stats_read(g._ctx, out, 1)
else:
    read = lambda: rt.lib().rt_stats_read(out, 1)
read()
g.render(W, H, spp, nb)
read()
names = {1: "box test", 2: "box slow path", 3: "sphere test", 4: "sphere discr>0 (fp64 roots)", 5: "sphere 2nd root", 6: "sphere slow path",
         7: "round (setup site)", 8: "setup body", 12: "trace batch", 13: "trace batch active",
         14: "sky lookup", 15: "specular branch", 16: "consume site", 17: "consume body", 22: "sample hand-out", 23: "in-order sum pass", 24: "lanes left without a sample", 20: "supply attempt", 21: "pixel fetch event",
         0: "shading event, bounce 0", 9: "culled trace (large scenes)", 10: "  bounce 0, taps known", 11: "shading event on a box", 18: "  bounce 0 on a box", 19: "  bounce 0, box, taps known",
         29: "shading event, taps known", 30: "shading event, bounce 1"}
samples = W * H * spp
for k in sorted(names):
    n, lanes = out[2 * k], out[2 * k + 1]
    if n:
        print(f"{names[k]:32s} execs/sample*64 {n * 64 / samples:8.3f}   avg active lanes {lanes / n:5.1f} ({lanes / n / 64 * 100:4.1f}%)")

sec = ["1 supply: ballots, stream state, exit", "2 shade", "3+4 taps: push + trace; bounce rays", "5 back", "6 in-order sum", "1 supply: pixel fetch", "1 supply: hand-out", "loop top / idle"]
tot = sum(out[50 + k] for k in range(8))
if tot:
    print("share of wave time (s_memtime stamps at the section boundaries):")
    for k in range(8):
        if out[50 + k]: print(f"  {sec[k]:40s} {out[50 + k] / tot * 100:5.1f} %")
