#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the COMPILED REFERENCE (oracle/_ref/libref*.so).

Runs only in the build container (needs /root/reference to have been compiled by
`make -C oracle`).  The outputs are data: inputs + the reference's outputs.  No reference source
text is stored.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rtlibs import DATA_DIR, FACE_NAMES, Ref, bits, scene_signature, synthetic_skybox  # noqa: E402

SCENES = ["scene_0.txt", "scene_1.txt", "scene_2.txt"]
BOUNCES = (1, 4, 8, 10)


def main():
    ref = Ref()
    refb = Ref(bounce_patch=True)
    out = {}
    meta = {}

    # ---- real skybox: stb_image v2.29 decode of the shipped JPEGs (gpu_and_windowing.c:24-33)
    ref.load_skybox(os.path.join(DATA_DIR, "skybox"))
    real_sky = ref.skybox_faces()
    meta["skybox_shape"] = list(real_sky.shape)
    meta["skybox_sha256"] = {FACE_NAMES[i]: hashlib.sha256(real_sky[i].tobytes()).hexdigest() for i in range(6)}
    # a sparse probe of texels per face so a decoder bug is localisable without the 75 MB image
    rng = np.random.default_rng(2024)
    probe_idx = rng.integers(0, 2048, size=(6, 256, 2))
    out["skybox_probe_idx"] = probe_idx.astype(np.int32)
    out["skybox_probe_rgb"] = np.stack([real_sky[f, probe_idx[f, :, 1], probe_idx[f, :, 0]] for f in range(6)])
    # first 16x16 block of every face (top-left MCU) -- exercises IDCT + upsampling + colour conversion
    out["skybox_corner"] = real_sky[:, :16, :16, :].copy()

    # ---- RNG (utils.c:60-75)
    states = [0, 1, 0x1234567800000001, 0xFFFFFFFFFFFFFFFF, 0x9E3779B97F4A7C15]
    out["rng_states"] = np.array(states, dtype=np.uint64)
    fl, after = [], []
    for s in states:
        f, a = ref.random_floats(s, 16)
        fl.append(bits(f)); after.append(a)
    out["rng_floats"] = np.stack(fl)
    out["rng_after"] = np.array(after, dtype=np.uint64)
    dirs = [ref.random_direction(s) for s in states]
    out["rng_dirs"] = np.stack([bits(d[0]) for d in dirs])
    out["rng_dirs_after"] = np.array([d[1] for d in dirs], dtype=np.uint64)
    seeds = [(0, 0, 0), (0, 1, 0), (0, 0, 1), (12345, 2073599, 63), (2**64 - 1, 2**32 - 1, 2**32 - 1)]
    out["path_seed_in"] = np.array(seeds, dtype=np.uint64)
    out["path_seed_out"] = np.array([ref.path_seed(*s) for s in seeds], dtype=np.uint64)

    # ---- camera (camera.c:95-125)
    cams = [dict(pos=(5, 5, 5), front=(-1, -1, -1), up=(0, 1, 0), fov=30.0),
            dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0),
            dict(pos=(8, 1, 1), front=(-1, 0.2, 0.3), up=(0, 1, 0), fov=1.0)]
    meta["cameras"] = cams
    pts = [(0, 0), (1, 0), (0, 1), (0.5, 0.5), (0.25, 0.75), (1, 1)]
    aspects = [float(np.float32(1920) / np.float32(1080)), 1.0, float(np.float32(3840) / np.float32(2160))]
    cam_rays = np.zeros((len(cams), len(aspects), len(pts), 6), np.uint32)
    for ci, cam in enumerate(cams):
        ref.set_camera(**cam)
        for ai, a in enumerate(aspects):
            for pi, (px, py) in enumerate(pts):
                cam_rays[ci, ai, pi] = bits(ref.camera_ray(px, py, a))
    out["cam_points"] = np.array(pts, np.float32)
    out["cam_aspects"] = np.array(aspects, np.float32)
    out["cam_rays"] = cam_rays
    ref.set_camera()

    # ---- scene loader (scene.c:206-624)
    meta["scenes"] = {}
    for s in SCENES:
        rc, buf = ref.parse_scene_file(os.path.join(DATA_DIR, s))
        assert rc == 0
        meta["scenes"][s] = scene_signature(buf)

    # ---- sample_cubemap on the real skybox (gpu_and_windowing.c:42-112), as bytes
    sdirs = rng.normal(size=(512, 3)).astype(np.float32)
    sdirs[::7, 0] = sdirs[::7, 1]          # face ties
    sdirs[3::11, 2] = -sdirs[3::11, 0]
    sdirs[5::13] = np.sign(sdirs[5::13])   # cube corners / exact diagonals
    sdirs_n = np.stack([ref.normalize(d) for d in sdirs])
    out["sky_dirs"] = bits(sdirs_n)
    out["sky_real_rgb"] = np.stack([bits(ref.sample_cubemap(d)) for d in sdirs_n])

    # ---- synthetic skybox used by all frame fixtures (so they need no JPEG decoder)
    syn = synthetic_skybox(32, seed=7)
    out["syn_sky"] = syn
    ref.set_skybox(syn); refb.set_skybox(syn)
    out["sky_syn_rgb"] = np.stack([bits(ref.sample_cubemap(d)) for d in sdirs_n])

    # ---- trace_ray (scene.c:156-190) on random + degenerate rays
    n_rays = 400
    org = rng.uniform(-2, 8, size=(n_rays, 3)).astype(np.float32)
    dr = rng.normal(size=(n_rays, 3)).astype(np.float32)
    for k in range(0, n_rays, 5):
        dr[k, rng.integers(3)] = 0.0
    for k in range(0, n_rays, 7):
        dr[k, rng.integers(3)] = -0.0
    org[::11] = np.round(org[::11])
    out["trace_org"] = bits(org); out["trace_dir"] = bits(dr)
    for si, s in enumerate(SCENES):
        ref.load_scene(os.path.join(DATA_DIR, s))
        objs = np.zeros(n_rays, np.int32); hits = np.zeros((n_rays, 7), np.uint32)
        for k in range(n_rays):
            objs[k], h = ref.trace_ray(org[k], dr[k])
            hits[k] = bits(h)
        out[f"trace_obj_{si}"] = objs; out[f"trace_hit_{si}"] = hits

    # ---- pixel() KATs (main.c:131-272), bounce limits via the patched build
    W, H = 1920, 1080
    aspect = float(np.float32(W) / np.float32(H))
    uv = [(0, 1), (0.5, 0.5), (1, 1), (1, 0), (0.75, 0.25), (0.3, 0.6), (0.62, 0.41), (0.45, 0.8)]
    out["pixel_uv"] = np.array(uv, np.float32)
    for si, s in enumerate(SCENES):
        refb.load_scene(os.path.join(DATA_DIR, s))
        res = np.zeros((len(BOUNCES), len(uv), 3), np.uint32)
        aft = np.zeros((len(BOUNCES), len(uv)), np.uint64)
        for bi, nb in enumerate(BOUNCES):
            refb.set_bounce_limit(nb)
            for k, (u, v) in enumerate(uv):
                c, a = refb.pixel(u, v, aspect, 0x1234567800000000 + k)
                res[bi, k] = bits(c); aft[bi, k] = a
        out[f"pixel_rgb_{si}"] = res; out[f"pixel_after_{si}"] = aft

    # ---- small frames: stream (verbatim build, 10 bounces, 2 passes) and counter (all limits)
    FW, FH = 64, 36
    meta["frame_size"] = [FW, FH]
    for si, s in enumerate(SCENES):
        path = os.path.join(DATA_DIR, s)
        ref.load_scene(path); refb.load_scene(path)
        fr, acc, st = ref.render_stream(FW, FH, passes=2, state=0)
        out[f"stream_frame_{si}"] = fr; out[f"stream_accum_{si}"] = acc
        meta[f"stream_state_{si}"] = int(st)
        for nb in BOUNCES:
            refb.set_bounce_limit(nb)
            out[f"counter_frame_{si}_b{nb}"] = refb.render_counter(FW, FH, 4, seed=0)
        refb.set_bounce_limit(4)
        out[f"counter_frame_{si}_b4_seed12345"] = refb.render_counter(FW, FH, 3, seed=12345)
    # moved camera
    ref.load_scene(os.path.join(DATA_DIR, "scene_0.txt"))
    ref.set_camera(**cams[1])
    out["counter_frame_cam1"] = ref.render_counter(FW, FH, 2, seed=1)
    ref.set_camera()

    # ---- the survey's 256x256 stream-mode frame on the REAL skybox: keep its hash only
    ref.load_skybox(os.path.join(DATA_DIR, "skybox"))
    ref.load_scene(os.path.join(DATA_DIR, "scene_0.txt"))
    fr, acc, st = ref.render_stream(256, 256, passes=1, state=0)
    meta["stream256_accum_sha256"] = hashlib.sha256(acc.tobytes()).hexdigest()
    meta["stream256_state"] = int(st)
    # and a real-skybox counter frame (needs the product's JPEG decoder at test time)
    refb.load_skybox(os.path.join(DATA_DIR, "skybox"))
    refb.load_scene(os.path.join(DATA_DIR, "scene_0.txt"))
    refb.set_bounce_limit(4)
    out["counter_frame_real_sky"] = refb.render_counter(FW, FH, 2, seed=0)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    with open(os.path.join(HERE, "reference_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
