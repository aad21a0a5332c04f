"""Development aid: randomised GPU-vs-oracle parity sweep (bit-exact) over scenes, cameras, frame sizes, sample
counts, bounce limits, schedules, compiled/generic kernels and strips.  usage: fuzz_parity.py [cases] [seed] [scale]   (scale multiplies every coordinate and size)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ray_tracing_amd as rt
from rtlibs import Oracle, bits, make_scene, synthetic_skybox

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
gpu, orc = rt.Renderer(0), Oracle()
gpu.set_tuning(poison_frame=True)
bad = 0
for case in range(cases):
    pick = rng.random()
    n = int(rng.integers(65, 300)) if pick < 0.15 else (int(rng.integers(15, 65)) if pick < 0.25 else int(rng.integers(1, 14)))      # (32 and up, not compiled: the cluster cull of rt_cull.h)
    objs = []
    for k in range(n):
        mat = dict(albedo=rng.uniform(0, 1, 3), roughness=float(rng.choice([0, 0.3, 1.0])), reflectance=float(rng.uniform(0, 1)),
                   metallic=float(rng.choice([0, 0, 1])), emission_power=float(rng.choice([0, 0, 0, 3.0])))
        if rng.random() < 0.4:
            objs.append(dict(type="sphere", center=rng.uniform(-3, 6, 3) * scale, radius=float(rng.uniform(0.3, 1.5)) * scale, **mat))
        else:
            grid = rng.random() < 0.5
            o = rng.integers(-3, 6, 3).astype(float) if grid else rng.uniform(-3, 6, 3)
            sz = rng.choice([0.1, 0.5, 1.0, 3.0, 9.0], 3) if grid else rng.uniform(0.05, 4, 3)
            objs.append(dict(type="cube", origin=o * scale, size=np.asarray(sz, dtype=float) * scale, **mat))
    scene = make_scene(objs)
    sky = synthetic_skybox(int(rng.choice([8, 16, 33])), seed=int(rng.integers(1 << 30)))
    pos = rng.integers(-2, 8, 3).astype(float) if rng.random() < 0.3 else rng.uniform(-2, 8, 3)
    front = rng.uniform(-1, 1, 3); front[2] -= 0.5
    cam = dict(pos=tuple(pos * scale), front=tuple(front), up=(0, 1, 0), fov=float(rng.uniform(0.5, 1.4)))
    W, H = int(rng.integers(2, 90)), int(rng.integers(2, 70))     # the reference divides by W-1 and H-1
    spp, nb, seed = int(rng.choice([1, 2, 3, 5, 16, 37])), int(rng.choice([1, 2, 4, 8, 10])), int(rng.integers(0, 1 << 62))
    print(f"case {case}: n={n} {W}x{H} spp={spp} nb={nb}", flush=True)
    for r in (gpu, orc):
        r.set_skybox(sky); r.set_scene(scene); r.set_camera(**cam)
    want = orc.render_counter(W, H, spp, nb, seed=seed)
    jit = rng.random() < 0.6
    if jit:
        try: gpu.compile_scene()
        except rt.RtError: jit = False
    chunks = (int(rng.choice([0, 1, 64])), int(rng.choice([0, 1, 2, 3])))     # pixel lists, resident workgroups per CU
    gpu.set_tuning(dequeue_shards=chunks[0], workgroups_per_cu=chunks[1])
    got = gpu.render(W, H, spp, nb, seed=seed)
    ok = bool((bits(got) == bits(want)).all())
    world = int(rng.choice([2, 3, 8]))
    rows = rt.strip_rows(H, 8, world)
    strips = torch.zeros((world, rows, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for rank in range(world):
        gpu.render_device(gpu.params(W, H, spp, nb, seed=seed, row_block=8, rank=rank, world=world), strips[rank].data_ptr())
    gpu.synchronize()
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu.deinterleave_device(strips.data_ptr(), frame.data_ptr(), W, H, 8, world); gpu.synchronize()
    ok2 = bool((bits(frame.cpu().numpy()) == bits(want)).all())
    gpu.set_tuning()
    if not (ok and ok2):
        bad += 1
        print(f"MISMATCH case {case}: n={n} {W}x{H} spp={spp} nb={nb} jit={jit} chunks={chunks!r} world={world} frame_ok={ok} strips_ok={ok2}", flush=True)
print(f"{cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
