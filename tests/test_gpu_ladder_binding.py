"""The ladder variant of the integration patch (scripts/patches/reference_main_rt.py main --ladder), RUN: the very text the
patch inserts into the reference's main.c -- its header, invalidate_accumulation() and update_frame() -- is compiled into
tests/c/ladder_host.c, which stands in for the reference's globals and event loop, and driven through an invalidation.  The
frame its move_frame_to_the_gpu() receives last must be the oracle's ladder (itself pinned to the compiled reference's worker()
loop, tests/test_oracle_vs_ref.py) after as many passes at the new pose: accumulate until the camera moves, restart from
--init-scale when it does (main.c:115-124, 354-408, 450-482, 585-634).  tests/test_c_abi_compile.py compiles and links the
same text inside the real main.c where the reference is present."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import ROOT, bits, oracle_progressive

PATCH = os.path.join(ROOT, "scripts", "patches", "reference_main_rt.py")
HOST = os.path.join(ROOT, "tests", "c", "ladder_host.c")
PASSES_PER_FRAME = 16


def build_ladder_host(tmp_path, passes_per_frame=PASSES_PER_FRAME, variant="--ladder"):
    text = subprocess.run([sys.executable, PATCH, "binding", variant], check=True, capture_output=True, text=True).stdout
    calls = ("rt_progressive_begin(rt, frame_w, frame_h, init_scale, 10, 0)", "rt_progressive_passes(rt, n)", "rt_progressive_resolve(rt, frame)",
             "rt_progressive_invalidate(rt)") if variant == "--ladder" else ("rt_render(rt, &p, frame)", "rt_cancel(rt)", "rt_reserve(rt, frame_w, frame_h)")
    for call in calls + ("move_frame_to_the_gpu(frame_w, frame_h, frame)",):
        assert call in text, call
    inc = tmp_path / f"binding{variant}.inc"
    inc.write_text(text)
    exe = tmp_path / f"host{variant}"
    libdir = os.path.dirname(rt.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), f'-DBINDING_TEXT="{inc}"',
                    f"-DRT_PASSES_PER_FRAME={passes_per_frame}"] + (["-DBINDING_IS_BLOCKING"] if variant == "--blocking" else []) +
                   [HOST, "-o", str(exe), "-L", libdir, "-lrt_hip", f"-Wl,-rpath,{libdir}", "-lm"],
                   check=True, capture_output=True, text=True)
    return str(exe)


def passes_after(frames, init_scale, per_frame):
    """update_frame()'s rule: one pass per shown frame while the ladder is below full resolution (and for the first), then per_frame."""
    scale, passes = init_scale, 0
    for _ in range(frames):
        n = 1 if scale > 1 or passes == 0 else per_frame
        for _ in range(n):
            passes += 1
            if scale > 1:
                scale >>= 1
    return passes


def test_binding_text_compiles_and_links(tmp_path):
    """no GPU: the patch's text + the stand-ins build against the library (every rt_* call resolves), both variants"""
    exe = build_ladder_host(tmp_path)
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    for name in ("rt_progressive_begin", "rt_progressive_passes", "rt_progressive_resolve", "rt_progressive_invalidate", "rt_progressive_state", "rt_set_camera"):
        assert name in syms, name
    exe = build_ladder_host(tmp_path, variant="--blocking")
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], check=True, capture_output=True, text=True).stdout
    for name in ("rt_render", "rt_cancel", "rt_reserve", "rt_default_params", "rt_set_camera"):
        assert name in syms, name


@pytest.mark.gpu
@pytest.mark.parametrize("init_scale,before,after,per_frame", [(8, 5, 6, 16), (1, 2, 3, 16), (16, 3, 7, 4)])
def test_the_ladder_binding_shows_the_oracles_ladder_after_an_invalidation(tmp_path, oracle, scene_paths, init_scale, before, after, per_frame):
    exe = build_ladder_host(tmp_path, per_frame)
    W, H = 96, 64
    out = tmp_path / "shown.raw"
    p = subprocess.run([exe, scene_paths[0], os.path.join(rt.DATA_DIR, "skybox"), str(W), str(H), str(init_scale), str(before), str(after), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    passes = passes_after(after, init_scale, per_frame)
    cam = dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0)
    oracle.set_skybox(rt.load_skybox()); oracle.load_scene(scene_paths[0]); oracle.set_camera(**cam)
    want, _, count, next_scale = oracle_progressive(oracle, W, H, init_scale, passes, 10, 0)
    oracle.set_camera()
    # begin + the invalidation before the first frame + the camera move: the accumulation restarted, nothing of the old pose is left
    assert line["passes"] == passes and line["next_scale"] == next_scale and line["frames_shown"] == before + after
    assert line["generation"] >= 2
    assert np.float32(line["weight_sum"]) == np.float32(count)
    shown = np.fromfile(out, np.float32).reshape(H, W, 3)
    assert (bits(shown) == bits(want)).all(), (init_scale, before, after)


@pytest.mark.gpu
def test_the_blocking_binding_shows_an_independent_frame_per_call(tmp_path, oracle, scene_paths):
    """The --blocking variant of the patch, run the same way: every update_frame() is one rt_render() of sixteen samples per pixel whose seed
    is the number of frames shown since the last invalidation; after a camera move the count starts again.  The last frame shown is the
    oracle's frame of that seed at the new pose."""
    exe = build_ladder_host(tmp_path, variant="--blocking")
    W, H, before, after = 96, 64, 3, 4
    out = tmp_path / "shown.raw"
    p = subprocess.run([exe, scene_paths[0], os.path.join(rt.DATA_DIR, "skybox"), str(W), str(H), "8", str(before), str(after), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["frames_since_invalidation"] == after and line["frames_shown"] == before + after
    cam = dict(pos=(2, 3, 9), front=(0.1, -0.3, -1), up=(0, 1, 0), fov=30.0)
    oracle.set_skybox(rt.load_skybox()); oracle.load_scene(scene_paths[0]); oracle.set_camera(**cam)
    want = oracle.render_counter(W, H, 16, 10, seed=after - 1)
    oracle.set_camera()
    shown = np.fromfile(out, np.float32).reshape(H, W, 3)
    assert (bits(shown) == bits(want)).all()
