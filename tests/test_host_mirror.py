"""CPU-side tests of the product library's host mirror (no GPU compute): the C ABI loads and exports
every declared symbol; the scene loader, JPEG decoder, camera basis and path seed agree with the
golden vectors captured from the compiled reference and with the oracle."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import ray_tracing_amd as rt
from rtlibs import ROOT, bits, scene_signature


def test_library_exports_every_declared_symbol():
    L = rt.lib()
    # the boundary header and the header of the test suite's own exports (include/rt_hip_testing.h)
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read() + open(os.path.join(ROOT, "include", "rt_hip_testing.h")).read()
    declared = set(re.findall(r"RT_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(", header))
    assert declared == set(rt.EXPORTS), declared ^ set(rt.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_reference_sizes():
    # reference scene.h:24-36 (sizeof(Object) = 68, sizeof(Scene) = 69636), gpu_and_windowing.h:4-7 (64)
    assert rt.SCENE_BYTES == 69636
    assert C.sizeof(rt.Cubemap) == 64
    assert C.sizeof(rt.Camera) == 40 and C.sizeof(rt.RenderParams) == 40


def test_scene_loader_matches_reference(golden_meta, scene_paths, oracle):
    for path in scene_paths:
        rc, buf = rt.parse_scene_file(path)
        assert rc == 0
        want = golden_meta["scenes"][os.path.basename(path)]
        got = scene_signature(buf)
        assert [[g[0], g[1], g[2], g[3], g[4]] for g in got] == want
        rc2, buf2 = oracle.parse_scene_file(path)
        assert rc2 == 0 and (buf == buf2).all()


SCENE_TEXT_CASES = [
    # (text, expect_ok)
    ("", True),
    ("   \n\t ", True),
    ("sphere", True),                               # `5 < len - i` holds with exactly 6 characters left (scene.c:224)
    ("sphere albedo", False),                       # but "albedo" is guarded by `6 < len - i` (scene.c:271): not matched at EOF
    ("sphere ", True),
    ("cube\n", True),
    ("cube size {1 2 3} origin {-1 -2.5 0.125}\n", True),
    ("sphere radius 2.75 center {1 2 3} roughness 0.5 reflectance 1 emission_power 12.5 ", True),
    ("sphere albedo   {0.1 0.2 0.3} ", True),       # albedo skips 9 characters (scene.c:280)
    ("sphere albedo {0.1 0.2 0.3} ", False),        # ... so a single space loses the '{'
    ("sphere metallic   1 ", True),                 # metallic skips 11 (scene.c:320)
    ("sphere metallic 0.5 x", False),
    ("sphere roughness 1.5 ", False),
    ("sphere roughness -0.5 ", False),
    ("cube radius 1 ", False),
    ("sphere size {1 1 1} ", False),
    ("cube size {1 -1 1} ", False),
    ("sphere radius 1. ", False),
    ("sphere radius - ", False),
    ("sphere radius +1 ", False),
    ("sphere radius .5 ", False),
    ("sphere center {1 2} ", False),
    ("sphere center {1 2 3 ", False),
    ("sphere emission_color {0 0.5 1.5} ", False),
    ("sphere emission_power -3 radius 0.001953125 ", True),
    ("teapot ", False),
    ("sphere radius 123456789.123456789 ", True),   # float accumulation, not strtof
    ("sphere radius 0.1 sphere radius 0.2 cube size {0.3 0.7 0.9} ", True),
    ("sphere radius", False),
    ("sphere radius ", False),
]


@pytest.mark.parametrize("text,ok", SCENE_TEXT_CASES)
def test_scene_loader_edge_cases_agree_with_oracle(oracle, text, ok, capfd):
    rc_o, buf_o = oracle.parse_scene_string(text)
    err_o = capfd.readouterr().err
    rc, buf = rt.parse_scene_string(text)
    err = capfd.readouterr().err
    assert (rc == 0) == ok, text
    assert (rc_o == 0) == ok, text
    assert err == err_o                     # same diagnostics on stderr
    if ok:
        assert scene_signature(buf) == scene_signature(buf_o)


def test_scene_loader_vs_compiled_reference_when_available(capfd):
    from rtlibs import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built here")
    import tempfile
    ref = Ref()
    for text, ok in SCENE_TEXT_CASES:
        if text.rstrip() != text and not text.endswith(" x"):
            pass
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(text)
        try:
            rc_r, buf_r = ref.parse_scene_file(f.name)
            err_r = capfd.readouterr().err
            rc, buf = rt.parse_scene_file(f.name)
            err = capfd.readouterr().err
        finally:
            os.unlink(f.name)
        assert (rc_r == 0) == (rc == 0) == ok, text
        assert err == err_r, text
        if ok:
            assert scene_signature(buf) == scene_signature(buf_r), text


def test_scene_loader_object_cap(capfd):
    text = "sphere radius 0.5 " * 1030
    rc, buf = rt.parse_scene_string(text)
    assert rc == 0
    assert int(buf[68 * 1024:].view("<i4")[0]) == 1024          # scene.c:602-605
    assert capfd.readouterr().err.count("Warning: Ignoring object") == 6


def test_missing_scene_file():
    rc, _ = rt.parse_scene_file("/nonexistent/scene.txt")
    assert rc == -4


def test_jpeg_decoder_matches_stb_decode_of_reference(golden, golden_meta):
    sky = rt.load_skybox()
    assert list(sky.shape) == golden_meta["skybox_shape"]
    for i, name in enumerate(rt.FACE_NAMES):
        assert hashlib.sha256(sky[i].tobytes()).hexdigest() == golden_meta["skybox_sha256"][name], name
    idx = golden["skybox_probe_idx"]
    for f in range(6):
        assert (sky[f, idx[f, :, 1], idx[f, :, 0]] == golden["skybox_probe_rgb"][f]).all()
    assert (sky[:, :16, :16, :] == golden["skybox_corner"]).all()


def test_jpeg_decoder_rejects_garbage(tmp_path):
    p = tmp_path / "bad.jpg"
    p.write_bytes(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00" + bytes(64))
    with pytest.raises(rt.RtError):
        rt.decode_jpeg(str(p))
    with pytest.raises(rt.RtError):
        rt.decode_jpeg(str(tmp_path / "missing.jpg"))


def test_real_skybox_oracle_matches_reference_frames(oracle, golden, golden_meta, scene_paths):
    """Oracle + product JPEG decoder reproduce reference outputs that depend on the real skybox."""
    sky = rt.load_skybox()
    oracle.set_skybox(sky)
    dirs = golden["sky_dirs"].view(np.float32)
    for k in range(len(dirs)):
        assert (bits(oracle.sample_cubemap(dirs[k])) == golden["sky_real_rgb"][k]).all(), k
    oracle.load_scene(scene_paths[0])
    W, H = golden_meta["frame_size"]
    f = oracle.render_counter(W, H, 2, 4, seed=0)
    assert (bits(f) == bits(golden["counter_frame_real_sky"])).all()
    # the 256x256 stream-mode frame of the verbatim reference
    frame, accum, state = oracle.render_stream(256, 256, passes=1, init_scale=1, max_bounces=10, state=0)
    assert hashlib.sha256(accum.tobytes()).hexdigest() == golden_meta["stream256_accum_sha256"] \
        == "1f679b7d19ec6fde3de85917e696f1ed45a8209da742fb6d08200364e6ada864"
    assert state == golden_meta["stream256_state"] == 0xa60e37b2bf3ef7e3


def test_camera_basis_matches_oracle_and_reference_rays(oracle, golden, golden_meta):
    for ci, c in enumerate(golden_meta["cameras"]):
        cam = rt.default_camera()
        cam.pos = rt.Vector3(*c["pos"]); cam.front = rt.Vector3(*c["front"]); cam.up = rt.Vector3(*c["up"]); cam.fov = c["fov"]
        oracle.set_camera(**c)
        for ai, a in enumerate(golden["cam_aspects"]):
            b = rt.camera_basis(cam, float(a))
            ob = oracle.camera_basis(float(a))
            for f in ("pos", "lower_left_corner", "horizontal", "vertical"):
                for ax in "xyz":
                    assert np.float32(getattr(getattr(b, f), ax)).view(np.uint32) == \
                        np.float32(getattr(getattr(ob, f), ax)).view(np.uint32)
            # dir = llc + H*px + V*py - pos, each product/sum rounded (camera.c:121) -> golden rays
            llc, Hh, Vv, pos = [np.array([getattr(getattr(b, f), ax) for ax in "xyz"], np.float32)
                                for f in ("lower_left_corner", "horizontal", "vertical", "pos")]
            for pi, (px, py) in enumerate(golden["cam_points"]):
                d = ((llc + Hh * np.float32(px)) + Vv * np.float32(py)) - pos
                assert (bits(d) == golden["cam_rays"][ci, ai, pi][3:]).all()
    oracle.set_camera()


def test_path_seed(golden):
    L = rt.lib()
    for (seed, p, s), want in zip(golden["path_seed_in"], golden["path_seed_out"]):
        assert L.rt_path_seed(int(seed), int(p), int(s)) == int(want)


def test_strip_rows_and_partition_cover_every_row_once():
    from ray_tracing_amd import multi_gpu as mg
    for H in (1, 7, 8, 9, 77, 1080, 2160):
        for rb in (1, 4, 8, 16):
            for world in (1, 2, 3, 8):
                n = rt.strip_rows(H, rb, world)
                assert n == mg.strip_rows(H, rb, world)
                rows = np.concatenate([mg.owned_rows(H, rb, r, world) for r in range(world)])
                real = rows[rows >= 0]
                assert sorted(real.tolist()) == list(range(H))
                idx = mg.frame_index(H, rb, world)
                assert (rows[idx] == np.arange(H)).all()
                # the strips handed out rotated by one (what both hosts do): a permutation of the ranks, the root's strip
                # is never longer than another, and frame_index(first=1) finds every row
                strips = [mg.strip_of_rank(r, world) for r in range(world)]
                assert sorted(strips) == list(range(world))
                assert all(rt.lib().rt_strip_of_rank(r, world) == strips[r] for r in range(world))
                held = [mg.owned_rows(H, rb, s, world) for s in strips]
                assert (held[0] >= 0).sum() == min((h >= 0).sum() for h in held)
                assert (np.concatenate(held)[mg.frame_index(H, rb, world, first=1 if world > 1 else 0)] == np.arange(H)).all()


def test_frame_sink_and_ppm(tmp_path):
    L = rt.lib()
    W, H = 5, 3
    frame = np.linspace(0, 1, W * H * 3, dtype=np.float32).reshape(H, W, 3)
    got = {}
    SINK = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p, C.c_void_p)

    def sink(w, h, data, user):
        got["shape"] = (w, h)
        got["first"] = C.cast(data, C.POINTER(C.c_float))[0]
    cb = SINK(sink)
    L.rt_set_frame_sink(cb, None)
    L.rt_move_frame_to_the_gpu(W, H, frame.ctypes.data_as(C.c_void_p))
    L.rt_set_frame_sink(None, None)
    assert got["shape"] == (W, H) and got["first"] == 0.0
    out = tmp_path / "f.ppm"
    assert L.rt_write_ppm(str(out).encode(), W, H, frame.ctypes.data_as(C.c_void_p)) == 0
    raw = out.read_bytes()
    head = b"P6\n5 3\n255\n"
    assert raw.startswith(head)
    px = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
    want = (frame * np.float32(255)).astype(np.uint8)[::-1]          # truncation + vertical flip (main.c:662-672)
    assert (px == want).all()


def test_png_sink_and_screenshot_naming(tmp_path, monkeypatch):
    from PIL import Image
    L = rt.lib()
    W, H = 37, 21                                  # > 64 KiB is exercised by the second image
    rng = np.random.default_rng(0)
    for (w, h) in ((W, H), (300, 200)):
        frame = rng.random((h, w, 3), dtype=np.float32)
        frame[0, 0] = (1.0, 0.0, 0.999999)
        out = tmp_path / f"f{w}.png"
        assert L.rt_write_png(str(out).encode(), w, h, frame.ctypes.data_as(C.c_void_p)) == 0
        got = np.asarray(Image.open(out).convert("RGB"))
        want = (frame * np.float32(255)).astype(np.uint8)[::-1]      # main.c:662-672
        assert got.shape == want.shape and (got == want).all()
    # a smooth image (what a render looks like) must actually be compressed: Paeth filter + LZ77 + fixed Huffman
    yy, xx = np.mgrid[0:240, 0:320].astype(np.float32)
    smooth = np.stack([xx / 320, yy / 240, (xx + yy) / 560], axis=-1).astype(np.float32)
    smooth[60:120, 80:200] = (0.25, 0.5, 0.75)                      # a flat patch: long matches
    out = tmp_path / "smooth.png"
    assert L.rt_write_png(str(out).encode(), 320, 240, smooth.ctypes.data_as(C.c_void_p)) == 0
    got = np.asarray(Image.open(out).convert("RGB"))
    assert (got == (smooth * np.float32(255)).astype(np.uint8)[::-1]).all()
    assert out.stat().st_size < 320 * 240 * 3 // 4, out.stat().st_size
    monkeypatch.chdir(tmp_path)
    (tmp_path / "screenshot_0.png").write_bytes(b"taken")
    name = C.create_string_buffer(64)
    frame = np.zeros((4, 4, 3), np.float32)
    assert L.rt_screenshot(4, 4, frame.ctypes.data_as(C.c_void_p), name, 64) == 0
    assert name.value == b"screenshot_1.png" and (tmp_path / "screenshot_1.png").exists()


def test_camera_interaction_matches_reference():
    """rt_move_camera / rt_rotate_camera vs camera.c:42-88 of the compiled reference."""
    from rtlibs import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built here")
    ref = Ref()
    ref.L.ref_move_camera.argtypes = [C.c_int, C.c_float]
    ref.L.ref_rotate_camera.argtypes = [C.c_double, C.c_double]
    ref.L.ref_get_camera.argtypes = [C.c_void_p]
    ref.set_camera(); ref.L.ref_reset_mouse()
    L = rt.lib()
    cam = rt.default_camera()
    mouse = rt.MouseState()
    L.rt_mouse_state_default(C.byref(mouse))
    rng = np.random.default_rng(1)
    x, y = 400.0, 300.0
    for step in range(200):
        if step % 3 == 0:
            x += float(rng.normal() * 40); y += float(rng.normal() * 40)
            ref.L.ref_rotate_camera(x, y); L.rt_rotate_camera(C.byref(cam), C.byref(mouse), x, y)
        else:
            d = int(rng.integers(4)); sp = float(np.float32(rng.uniform(0.01, 0.5)))
            ref.L.ref_move_camera(d, sp); L.rt_move_camera(C.byref(cam), d, sp)
        want = np.zeros(9, np.float32)
        ref.L.ref_get_camera(want.ctypes.data_as(C.c_void_p))
        got = np.array([cam.pos.x, cam.pos.y, cam.pos.z, cam.front.x, cam.front.y, cam.front.z, cam.up.x, cam.up.y, cam.up.z], np.float32)
        assert (bits(got) == bits(want)).all(), step
    ref.set_camera()


def _random_scene_text(rng):
    """Grammar-driven text with deliberate damage: the loader must agree with the reference on every one."""
    props_sphere = ["radius", "center"]
    props_cube = ["origin", "size"]
    props_mat = ["albedo", "roughness", "reflectance", "metallic", "emission_power", "emission_color"]
    vec_props = {"center", "origin", "size", "albedo", "emission_color"}

    def num():
        k = rng.integers(8)
        if k == 0:
            return str(int(rng.integers(0, 3)))
        if k == 1:
            return f"-{rng.uniform(0, 9):.{int(rng.integers(1, 8))}f}"
        if k == 2:
            return f"{rng.uniform(0, 1):.{int(rng.integers(1, 9))}f}"
        if k == 3:
            return str(int(rng.integers(0, 200)))
        if k == 4:
            return f"{rng.uniform(0, 1):.3f}"
        if k == 5:
            return rng.choice(["1.", ".5", "+1", "1e3", "-", "0.0.1", "0x10", "1,5"])
        return f"{rng.uniform(0, 2):.2f}"

    out = []
    for _ in range(int(rng.integers(0, 6))):
        kind = rng.choice(["sphere", "cube", "sphere", "cube", "Sphere", "cub"])
        out.append(kind)
        pool = props_mat + (props_sphere if kind == "sphere" else props_cube)
        if rng.integers(6) == 0:
            pool = pool + props_sphere + props_cube
        for _ in range(int(rng.integers(0, 7))):
            p = rng.choice(pool)
            sep = rng.choice(["   ", "    ", "\t\t\t", " ", "\n   ", "      "])
            if p in vec_props:
                vals = " ".join(num() for _ in range(int(rng.choice([3, 3, 3, 3, 2, 4]))))
                brace = rng.choice(["{%s}", "{%s}", "{ %s }", "{%s", "%s}", "{\n%s\n}"])
                out.append(f"\n\t{p}{sep}{brace % vals}")
            else:
                out.append(f"\n\t{p}{sep}{num()}")
        out.append(rng.choice(["\n\n", "\n", "  ", "\r\n"]))
    return "".join(out) + "    \n   "          # padding: keeps the reference's 9/11-character skips inside the buffer


def test_scene_loader_fuzz_against_oracle_and_reference(oracle, capfd, tmp_path):
    from rtlibs import Ref, ref_available
    ref = Ref() if ref_available() else None
    rng = np.random.default_rng(2024)
    n_ok = 0
    for k in range(400):
        text = _random_scene_text(rng)
        rc, buf = rt.parse_scene_string(text)
        err = capfd.readouterr().err
        rc_o, buf_o = oracle.parse_scene_string(text)
        err_o = capfd.readouterr().err
        assert (rc == 0) == (rc_o == 0) and err == err_o, repr(text)
        if rc == 0:
            n_ok += 1
            assert scene_signature(buf) == scene_signature(buf_o), repr(text)
        if ref is not None:
            f = tmp_path / "s.txt"
            f.write_text(text, newline="")
            rc_r, buf_r = ref.parse_scene_file(str(f))
            err_r = capfd.readouterr().err
            assert (rc_r == 0) == (rc == 0) and err_r == err, repr(text)
            if rc == 0:
                assert scene_signature(buf) == scene_signature(buf_r), repr(text)
    assert 40 < n_ok < 360          # the generator produces both accepted and rejected files
