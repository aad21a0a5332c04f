"""Interleaved A/B timing of two builds of librt_hip.so in ONE process on ONE device (device-to-device
clock differences on this pool are larger than most kernel deltas).  usage: ab.py libA.so libB.so [rounds]"""
import ctypes as C, os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # one HIP runtime for both libraries
import numpy as np
import ray_tracing_amd as rt

def load(path):
    rt._lib = None
    rt.LIB_PATH = os.path.abspath(path)
    L = rt.lib()
    r = rt.Renderer(0)
    r._L = L
    return r

cfgs = [("C1", 0, 1920, 1080, 64, 4, 1), ("C2", 1, 1920, 1080, 256, 8, 1), ("C1 strip 3/8", 0, 1920, 1080, 64, 4, 8), ("C3", 2, 3840, 2160, 64, 8, 1),
        ("C4 strip 3/8", 0, 3840, 2160, 1024, 8, 8)]
if os.environ.get("AB_CFGS"): cfgs = [c for c in cfgs if c[0].split()[0] in os.environ["AB_CFGS"].split(",")]
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
sky = None
rs = []
for p in libs:
    r = load(p)
    if sky is None:
        sky = rt.load_skybox()
    r.set_skybox(sky); r.profile(True)
    r._jit = bool(os.environ.get("AB_JIT"))
    rs.append(r)
for name, scene, W, H, spp, nb, world in cfgs:
    times = [[], []]; walls = [[], []]
    for k, r in enumerate(rs):
        rt._lib = r._L
        r.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt")
        jf = os.environ.get("AB_JITFLAGS_A" if k == 0 else "AB_JITFLAGS_B")       # e.g. "-DSOMETHING": the same library built two ways
        if jf: r.set_tuning(jit_flags=jf)
        if r._jit: r.compile_scene()
    frames = [None, None]
    for it in range(rounds + 1):
        for k, r in enumerate(rs):
            rt._lib = r._L
            tk = os.environ.get("AB_TUNING_A" if k == 0 else "AB_TUNING_B")      # per-build override, e.g. "dequeue_shards=1,workgroups_per_cu=3"
            if tk and hasattr(r._L, "rt_set_tuning"): r.set_tuning(**{a: int(b) for a, b in (kv.split("=") for kv in tk.split(","))})
            if world == 1:
                strip = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.render_device(r.params(W, H, spp, nb), strip.data_ptr()); r.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                frames[k] = strip.cpu().numpy()
            else:
                strip = torch.zeros((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r.render_device(r.params(W, H, spp, nb, row_block=8, rank=3, world=world), strip.data_ptr()); r.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                frames[k] = strip.cpu().numpy()
            ms, n = r.profile_collect()
            if it:
                times[k].append(ms); walls[k].append(wall)
    same = bool((frames[0].view(np.uint32) == frames[1].view(np.uint32)).all())
    # paired: the two builds ran back to back in every round, so the per-round ratio cancels the drift of the device's clock
    paired = statistics.median([y / x for x, y in zip(times[0], times[1])])
    a, b = statistics.median(times[0]), statistics.median(times[1])
    wa, wb = statistics.median(walls[0]), statistics.median(walls[1])
    print(f"{name}: events (primary pass + trace kernel) A {a:.3f} ms (min {min(times[0]):.3f})   B {b:.3f} ms (min {min(times[1]):.3f})   B/A {b / a:.4f} (median of the per-round ratios {paired:.4f}) | "
          f"whole launch, enqueue -> synchronised: A {wa:.3f}   B {wb:.3f}   B/A {wb / wa:.4f}   identical={same}", flush=True)
