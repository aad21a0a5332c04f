"""EXPERIMENT (round 4): what a first-bounce pass is worth before its hand-over is built.
C1's frame at max_bounces = 1 (every sample is exactly one shading event, no bounce ray), pixels of known tap class only
(-DRT_PROBE_KNOWN_ONLY leaves the others out of both kernels' frames):
  A  the general kernel rt_trace_spec            (supply + shade + back + in-order sum of the wavefront loop)
  B  rt_first_bounce_spec, one wave per pixel    (rt_tuning.first_bounce_probe)
interleaved on one device, HIP events over primary pass + kernel; the two frames must be identical.  Then both at the real
bounce limit (4): B then also traces the bounce ray and finishes the samples that leave the scene -- its frame is
incomplete (survivors are not handed over), only its time means something: T_B(4) - T_B(1) is the cost of the bounce-ray
batch at 100 % lanes.  usage: first_bounce_probe.py [rounds]

The kernel and its tuning switch are no longer in the library (the experiment was killed at its kill line): apply
scripts/patches/first_bounce_pass.diff to a scratch copy of the tree (`git apply`), build with -DRT_PROBE_KNOWN_ONLY, run this."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ray_tracing_amd as rt
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 15
W, H, spp = 1920, 1080, 64
sky = rt.load_skybox()
def make(first, waves=0):
    g = rt.Renderer(0)
    g.set_skybox(sky); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera()
    g.set_tuning(jit_flags="-DRT_PROBE_KNOWN_ONLY" + (f" -DRT_FIRST_WAVES_PER_SIMD={waves}" if waves else ""), poison_frame=True,
                 first_bounce_probe=1 if first else 0, workgroups_per_cu=waves if first else 0)
    g.compile_scene(); g.profile(True)
    return g
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 6         # resident waves per SIMD of the first-bounce kernel (6: 80 registers, 5: 96, 4: 128)
A, B = make(False), make(True, waves)
P = rt.Renderer(0); P.set_skybox(sky); P.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); P.set_camera(); P.compile_scene(); P.profile(True)   # the product kernel, all pixels
out = {}
for nb in (1, 4):
    t = {"A": [], "B": [], "P": []}
    frames = {}
    for it in range(rounds + 1):
        for name, g in (("A", A), ("B", B), ("P", P)):
            d = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            g.render_device(g.params(W, H, spp, nb, seed=5), d.data_ptr()); g.synchronize()
            ms, n = g.profile_collect()
            if it: t[name].append(ms)
            frames[name] = d.cpu().numpy()
    a, b, p_ = (statistics.median(t[k]) for k in ("A", "B", "P"))
    fa, fb = frames["A"].view(np.uint32), frames["B"].view(np.uint32)
    same = bool((fa == fb).all())
    known = int((~np.isnan(frames["A"]).any(axis=2)).sum())
    # where both have rendered something they must agree with the product kernel's frame too
    ok_p = bool((frames["A"].view(np.uint32) == frames["P"].view(np.uint32))[~np.isnan(frames["A"])].all()) if nb == 1 else None
    print(f"[first-bounce kernel at {waves} waves per SIMD] max_bounces {nb}: general kernel, known-class pixels only {a:.3f} ms | first-bounce kernel {b:.3f} ms | product kernel, all pixels {p_:.3f} ms | "
          f"A == B bit for bit: {same}; pixels rendered by A: {known}; A == product frame on them: {ok_p}", flush=True)
    out[nb] = (a, b, p_, same)
print(f"saving at one event per sample: {out[1][0] - out[1][1]:.3f} ms = {(out[1][0] - out[1][1]) / out[4][2] * 100:.1f} % of the product's C1 kernel time ({out[4][2]:.3f} ms); "
      f"kill line: 5 % after the hand-over")
