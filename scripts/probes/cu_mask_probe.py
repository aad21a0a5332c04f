"""Development aid: the C1 frame on subsets of the compute units (hipExtStreamCreateWithCUMask), to see whether two
CUs that share an instruction cache slow each other down.  usage: cu_mask_probe.py [C1|C2]"""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ray_tracing_amd as rt
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
scene, W, H, spp, nb = {"C1": (0, 1920, 1080, 64, 4), "C2": (1, 1920, 1080, 256, 8)}[cfg]
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt"); g.compile_scene()
strip = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
p = g.params(W, H, spp, nb)
def run(words):
    mask = (C.c_uint32 * len(words))(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), mask)
    assert rc == 0, rc
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.render_device(p, strip.data_ptr(), s.value); g.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts[1:])
# mask bit i -> XCC i % 8, shader engine (i / 8) % 4, CU within the engine i / 32 (an XCC whose bits are all zero gets every CU):
# word k of the mask = CU k of every shader engine of every XCC
F = 0xffffffff
base = None
for name, words in (("all CUs", [F] * 8), ("CU 0,2,4,6 of every SE (one of each adjacent pair)", [F, 0, F, 0, F, 0, F, 0]),
                    ("CU 1,3,5,7 of every SE", [0, F, 0, F, 0, F, 0, F]), ("CU 0,1,4,5 of every SE (whole pairs)", [F, F, 0, 0, F, F, 0, 0]),
                    ("CU 2,3,6,7 of every SE (whole pairs)", [0, 0, F, F, 0, 0, F, F]), ("CU 0-3 of every SE", [F, F, F, F, 0, 0, 0, 0]),
                    ("SE 0,2 of every XCC", [0x00ff00ff] * 8), ("CU 0,4 of every SE", [F, 0, 0, 0, F, 0, 0, 0]), ("CU 0,1 of every SE", [F, F, 0, 0, 0, 0, 0, 0])):
    ms = run(words)
    base = base or ms
    n = sum(bin(w).count("1") for w in words)
    print(f"{name:52s} {n:4d} CUs   {ms:8.3f} ms   x{ms / base:5.2f}   CU-ms {ms * n / 256:7.3f}", flush=True)
