"""Development aid: the end of a launch.  The scene-specialised kernel is built with -DRT_STATS -DRT_STATS_LIFETIMES_ONLY:
every wave logs its start, the time it found the pixel lists empty, the rounds it ran after that, and its end
(rt_kernels.hip, STAMP_FLUSH).  The time between the mean and the last exit is what a perfectly balanced end would save.
usage: [SPP=n] tail_probe.py [world ...]     (C1, strip of rank 3 of `world`; world 1 = the whole frame)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import ray_tracing_amd as rt
if os.environ.get("RT_LIB"): rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB"])
W, H, spp, nb, rank = 1920, 1080, int(os.environ.get("SPP", 64)), 4, 3
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.profile(True)
g.set_tuning(jit_flags="-DRT_STATS -DRT_STATS_LIFETIMES_ONLY"); g.compile_scene()
L = rt.lib()
L.rt_spec_symbol_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
log = np.zeros((8192, 4), dtype=np.uint64)
for world in [int(a) for a in sys.argv[1:]] or [8]:
    strip = torch.empty((rt.strip_rows(H, 8, world), W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for it in range(3):
        g.render_device(g.params(W, H, spp, nb, row_block=8, rank=rank % world, world=world), strip.data_ptr()); g.synchronize()
        ms, _ = g.profile_collect()
    n = C.c_size_t(0)
    assert L.rt_spec_symbol_read(g._ctx, b"rt_wave_log", log.ctypes.data, log.nbytes, C.byref(n)) == 0
    w = log[log[:, 3] != 0].astype(np.int64)
    t0 = w[:, 0].min()
    start, dry, rounds, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, w[:, 2], (w[:, 3] - t0) / 100.0
    print(f"world {world}: launch {ms * 1e3:.0f} us (events); {len(w)} waves start within {start.max():.0f} us; find the lists empty at "
          f"{np.percentile(dry, 1):.0f} / {dry.mean():.0f} / {dry.max():.0f} us (1 % / mean / last), then run {rounds.mean():.1f} rounds (max {rounds.max()}); "
          f"leave at {np.percentile(end, 1):.0f} / {end.mean():.0f} / {end.max():.0f} us: last - mean = {end.max() - end.mean():.0f} us")
    hist, edges = np.histogram(end, bins=np.arange(np.floor(end.min() / 25) * 25, end.max() + 25, 25))
    print("  waves leaving per 25 us: " + " ".join(f"{int(e)}:{h}" for e, h in zip(edges, hist)))
    log[:] = 0
