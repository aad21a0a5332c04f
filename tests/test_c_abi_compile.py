"""The public headers must be usable from a plain C11 host (the reference is C11) and from C++, and every
declared function must link against librt_hip.so.  No GPU call is made."""
import os
import subprocess
import sys
import textwrap

import pytest

import ray_tracing_amd as rt
from rtlibs import ROOT

SRC = textwrap.dedent(r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "rt_hip.h"

    /* layouts the reference host relies on (scene.h:24-36, gpu_and_windowing.h:4-7, vector.h:32-61) */
    _Static_assert(sizeof(Vector3) == 12, "Vector3");
    _Static_assert(sizeof(Ray) == 24 && sizeof(Sphere) == 16 && sizeof(Cube) == 24, "Ray/Sphere/Cube");
    _Static_assert(sizeof(Material) == 40 && offsetof(Material, emission_color) == 28, "Material");
    _Static_assert(sizeof(Object) == 68 && offsetof(Object, material) == 28, "Object");
    _Static_assert(sizeof(Scene) == 69636 && offsetof(Scene, num_objects) == 69632, "Scene");
    _Static_assert(sizeof(Cubemap) == 64 && offsetof(Cubemap, w) == 48, "Cubemap");
    _Static_assert(CF_FRONT == 0 && CF_BACK == 1 && CF_LEFT == 2 && CF_RIGHT == 3 && CF_TOP == 4 && CF_BOTTOM == 5, "CubeFace");
    _Static_assert(OBJECT_CUBE == 0 && OBJECT_SPHERE == 1, "ObjectType");

    static Scene scene;

    int main(int argc, char **argv)
    {
        /* host-side functions only: no GPU needed */
        if (rt_parse_scene_file(argv[1], &scene) != RT_OK) return 1;
        rt_camera cam; rt_camera_default(&cam);
        rt_camera_basis basis; rt_camera_basis_for(&cam, 16.0f / 9.0f, &basis);
        rt_render_params p; rt_default_params(&p, 1920, 1080, 64, 4);
        printf("%d %d %.9g %d %llu\n", scene.num_objects, rt_strip_rows(1080, 8, 8), basis.vertical.y, p.row_block,
               (unsigned long long) rt_path_seed(0, 1, 2));
        /* the GPU entry points must at least link */
        void *fns[] = { (void*) rt_create, (void*) rt_destroy, (void*) rt_set_scene, (void*) rt_set_skybox, (void*) rt_set_camera,
                        (void*) rt_compile_scene, (void*) rt_render, (void*) rt_render_device, (void*) rt_deinterleave_device,
                        (void*) rt_progressive_begin, (void*) rt_progressive_pass, (void*) rt_progressive_resolve,
                        (void*) rt_progressive_invalidate, (void*) rt_load_cubemap, (void*) rt_free_cubemap,
                        (void*) rt_move_frame_to_the_gpu, (void*) rt_write_png, (void*) rt_last_error,
                        (void*) rt_set_tuning, (void*) rt_multi_create, (void*) rt_multi_destroy, (void*) rt_multi_set_scene,
                        (void*) rt_multi_set_skybox, (void*) rt_multi_set_camera, (void*) rt_multi_compile_scene, (void*) rt_multi_render };
        return fns[argc & 1] == NULL;
    }
''')


def _build_and_run(tmp_path, compiler, std, src_name, scene):
    src = tmp_path / src_name
    text = SRC if src_name.endswith(".c") else SRC.replace("_Static_assert", "static_assert")
    src.write_text(text)
    exe = tmp_path / ("host_" + src_name.replace(".", "_"))
    libdir = os.path.dirname(rt.LIB_PATH)
    cmd = [compiler, std, "-Wall", "-Werror", "-Wno-unused-variable", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lrt_hip", f"-Wl,-rpath,{libdir}", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe), scene], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "9" and out[1] == "136" and out[3] == "8"
    assert int(out[4]) == rt.lib().rt_path_seed(0, 1, 2)


def test_header_is_valid_c11_and_links(tmp_path, scene_paths):
    _build_and_run(tmp_path, "gcc", "-std=c11", "host.c", scene_paths[0])


def test_header_is_valid_cxx_and_links(tmp_path, scene_paths):
    _build_and_run(tmp_path, "g++", "-std=c++17", "host.cpp", scene_paths[0])


ROUND5_HOST = textwrap.dedent(r'''
    /* A host compiled against ROUND 5's rt_hip.h: rt_tuning had no size field and eleven members (include/rt_hip.h at b83836b).
     * Its binary must be told so by today's library -- RT_ERR_ARGUMENT from the size check -- instead of having its struct read
     * with today's layout.  (Only the declarations such a host would have had are repeated here.) */
    #include <stdio.h>
    #include <string.h>
    typedef struct rt_context rt_context;
    typedef struct {
        int dequeue_shards, workgroups_per_cu, jit_waves_per_simd;
        const char *jit_flags;
        int force_collective, poison_frame, trace_known_taps, test_every_object, audit_known_taps, test_drop_pixels, test_corrupt_lit_table;
    } rt_tuning_round5;
    void rt_default_tuning(rt_tuning_round5 *t);           /* (what round 5 declared: it memset the struct to zero) */
    int rt_set_tuning(rt_context *ctx, const rt_tuning_round5 *t);
    const char *rt_last_error(void);
    int rt_abi_version(void);

    int main(void)
    {
        int worst = 0;
        for (int shards = 0; shards <= 64; shards += 64)
            for (int wg = 0; wg <= 8; wg++) {
                rt_tuning_round5 t;
                memset(&t, 0, sizeof t);                   /* round 5's rt_default_tuning() */
                t.dequeue_shards = shards; t.workgroups_per_cu = wg; t.poison_frame = 1;
                const int rc = rt_set_tuning((rt_context *) 0, &t);
                if (rc != -1 || !strstr(rt_last_error(), "rt_tuning.size")) { printf("shards %d wg %d: rc %d, %s\n", shards, wg, rc, rt_last_error()); worst = 1; }
            }
        printf("abi %d\n", rt_abi_version());
        return worst;
    }
''')


def test_a_host_built_against_the_round5_header_is_refused(tmp_path):
    """rt_tuning carries its own size (set by rt_default_tuning, checked by rt_set_tuning before anything else is read), and the
    library says which revision of the header it was built from (rt_abi_version)."""
    src = tmp_path / "round5_host.c"
    src.write_text(ROUND5_HOST)
    exe = tmp_path / "round5_host"
    libdir = os.path.dirname(rt.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", str(src), "-o", str(exe), "-L", libdir, "-lrt_hip", f"-Wl,-rpath,{libdir}"],
                   check=True, capture_output=True, text=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    import re
    assert p.stdout.split()[-1] == re.search(r"#define RT_ABI_VERSION (\d+)", header).group(1)
    # today's struct, as today's rt_default_tuning() leaves it, passes the size check (and is then refused for the NULL context)
    t = rt.Tuning()
    rt.lib().rt_default_tuning(t)
    assert t.size == __import__("ctypes").sizeof(rt.Tuning) == 32
    assert rt.lib().rt_set_tuning(None, t) == -1 and b"NULL context" in rt.lib().rt_last_error()
    k = rt.TestKnobs()
    rt.lib().rt_default_test_knobs(k)
    assert k.size == __import__("ctypes").sizeof(rt.TestKnobs)


def test_integration_md_shows_the_patch_that_is_compiled():
    """INTEGRATION.md section 2 leads with the ladder binding, and the code it shows IS the text reference_main_rt.py inserts (both variants)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_main_rt", os.path.join(ROOT, "scripts", "patches", "reference_main_rt.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for variant in ("--ladder", "--blocking"):
        for piece in mod.VARIANTS[variant]:
            assert piece in doc, (variant, piece[:60])
    assert mod.START in doc
    assert doc.index("rt_progressive_begin(rt, frame_w, frame_h, init_scale, 10, 0)") < doc.index("rt_render(rt, &p, frame)")      # the ladder first


def test_the_boundary_header_carries_no_test_knobs():
    """rt_hip.h is what a maintainer binds; fault injections, self-tests and instrumentation read-outs are declared in rt_hip_testing.h."""
    public = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    for name in ("test_drop_pixels", "test_corrupt_lit_table", "test_every_object", "poison_frame", "rt_selftest", "rt_spec_stats_read",
                 "rt_spec_symbol_read", "rt_primary_passes_run", "rt_multi_create_on_one_device", "rt_last_launch_counts"):
        assert name not in public, name
    testing = open(os.path.join(ROOT, "include", "rt_hip_testing.h")).read()
    for name in ("test_drop_pixels", "rt_selftest", "rt_spec_stats_read", "rt_primary_passes_run", "rt_multi_create_on_one_device"):
        assert name in testing, name


REF_SRC = "/root/reference/src"

REF_HOST = textwrap.dedent(r'''
    /* A reference translation unit that binds the library: the reference's OWN headers supply Vector3, Scene,
     * Cubemap (scene.h:3-47, vector.h:32-61, gpu_and_windowing.h:4-16); rt_hip.h adds only its own types. */
    #include <stddef.h>
    #include "vector.h"
    #include "scene.h"
    #include "gpu_and_windowing.h"
    #define RT_HAVE_REFERENCE_TYPES
    #include "rt_hip.h"

    _Static_assert(sizeof(Scene) == 69636 && sizeof(Object) == 68 && sizeof(Cubemap) == 64, "reference layouts");

    static Scene scene;
    static Cubemap skybox;

    int bind(rt_context *ctx, Vector3 *frame, int w, int h)
    {
        rt_render_params p;
        rt_default_params(&p, w, h, 1, 10);
        if (rt_set_scene(ctx, &scene) != RT_OK || rt_set_skybox(ctx, &skybox) != RT_OK) return -1;
        if (rt_render(ctx, &p, frame) != RT_OK) return -1;
        move_frame_to_the_gpu(w, h, frame);                 /* the reference's presenter, gpu_and_windowing.h:44 */
        return sample_cubemap(&skybox, (Vector3) {0, 0, 1}).x >= 0;
    }
''')


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout is not on this machine")
def test_header_compiles_against_the_reference_headers(tmp_path):
    """The drop-in claim of INTEGRATION.md section 2: with RT_HAVE_REFERENCE_TYPES the boundary header takes
    Scene / Cubemap / Vector3 from the reference's own scene.h, gpu_and_windowing.h and vector.h."""
    src = tmp_path / "ref_host.c"
    src.write_text(REF_HOST)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-c", "-I", REF_SRC, "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(tmp_path / "ref_host.o")], check=True, capture_output=True, text=True)


GLFW_SYMBOLS = ["glfwInit", "glfwTerminate", "glfwWindowHint", "glfwCreateWindow", "glfwDestroyWindow", "glfwSetKeyCallback",
                "glfwSetFramebufferSizeCallback", "glfwSetCursorPosCallback", "glfwSetInputMode", "glfwMakeContextCurrent",
                "glfwGetProcAddress", "glfwSwapInterval", "glfwGetWindowSize", "glfwSwapBuffers", "glfwPollEvents", "glfwGetKey",
                "glfwWindowShouldClose", "glfwGetCursorPos", "glfwSetErrorCallback"]


LADDER_CALLS = ("rt_progressive_begin", "rt_progressive_passes", "rt_progressive_resolve", "rt_progressive_invalidate", "rt_progressive_state")
BLOCKING_CALLS = ("rt_render", "rt_cancel", "rt_reserve", "rt_default_params")


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout is not on this machine")
@pytest.mark.parametrize("variant", ["--ladder", "--blocking"])
def test_the_integration_patch_applies_compiles_and_links(tmp_path, variant):
    """INTEGRATION.md section 2 as a build: scripts/patches/reference_main_rt.py turns the reference's main.c -- start_workers()
    at :516, update_frame() at :450-482, invalidate_accumulation() at :115-124, stop_workers() -- into the host that calls
    librt_hip.so, and adds the two camera getters camera.c lacks.  Both variants: --ladder (the documented one: passes accumulate
    until the camera moves and the image refines from 1/init_scale, main.c:354-408 -- rt_progressive_begin / _passes / _resolve /
    _invalidate, wired to the reference's own `init_scale` global of main.c:50, :585-634) and --blocking (one rt_render() per
    shown frame).  The patched text is piped into the compiler (nothing of the
    reference is written to disk), compiled with the reference's own flags and headers, and linked with the reference's other
    translation units against the library: with --no-undefined, so every rt_* call resolves; the only symbols allowed to stay
    open are the 19 entry points of GLFW the reference's window code uses (this image has no libglfw: main() is compiled and
    linked, never run).  Compile and link only; the ladder text is RUN by tests/test_gpu_ladder_binding.py."""
    ref = os.path.dirname(REF_SRC)
    patch = os.path.join(ROOT, "scripts", "patches", "reference_main_rt.py")
    flags = ["-std=c11", "-O2", "-DNDEBUG", "-fPIC", "-ffunction-sections", "-fdata-sections", "-Werror=implicit-function-declaration",
             "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
             "-I", REF_SRC, "-I", os.path.join(ref, "3p"), "-I", os.path.join(ref, "3p", "glad", "include"),
             "-I", os.path.join(ref, "3p", "glfw-3.4.bin.WIN64", "include"), "-I", os.path.join(ROOT, "include")]
    objs = []
    for which, src in (("main", "main.c"), ("camera", "camera.c")):
        args = [sys.executable, patch, which] + ([variant] if which == "main" else [])
        text = subprocess.run(args, stdin=open(os.path.join(REF_SRC, src)), check=True, capture_output=True, text=True).stdout
        if which == "main":
            # the text the GPU test runs is the text that went into main.c
            binding = subprocess.run([sys.executable, patch, "binding", variant], check=True, capture_output=True, text=True).stdout
            import importlib.util
            spec = importlib.util.spec_from_file_location("reference_main_rt", patch)
            mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
            pieces = mod.VARIANTS[variant]
            assert binding == "".join(pieces) and all(piece in text for piece in pieces)
            assert ("rt_progressive_begin(rt, frame_w, frame_h, init_scale, 10, 0)" in text) == (variant == "--ladder")
            assert ("rt_render(rt, &p, frame)" in text) == (variant == "--blocking")
        obj = tmp_path / (which + "_rt.o")
        subprocess.run(["gcc"] + flags + ["-x", "c", "-c", "-", "-o", str(obj)], input=text, check=True, capture_output=True, text=True)
        objs.append(str(obj))
    # the patched main() calls into the library and nothing else of the workers is left in it
    syms = subprocess.run(["nm", "--undefined-only", objs[0]], check=True, capture_output=True, text=True).stdout.split()
    for name in ("rt_create", "rt_set_scene", "rt_set_skybox", "rt_compile_scene", "rt_set_camera",
                 "rt_destroy", "rt_last_error", "get_camera_front", "get_camera_up", "move_frame_to_the_gpu"):
        assert name in syms, name
    for name in LADDER_CALLS:
        assert (name in syms) == (variant == "--ladder"), name
    for name in BLOCKING_CALLS:
        assert (name in syms) == (variant == "--blocking"), name
    # the ladder reads the reference's own --init-scale global (defined in main.c itself, so it shows as a data symbol there)
    defined = subprocess.run(["nm", "--defined-only", objs[0]], check=True, capture_output=True, text=True).stdout
    assert " init_scale" in defined
    if variant == "--ladder":
        dis = subprocess.run(["objdump", "-dr", "--no-show-raw-insn", objs[0]], check=True, capture_output=True, text=True).stdout
        body = dis[dis.index("<update_frame>:"):]
        body = body[:body.index("\n\n")] if "\n\n" in body else body
        assert "init_scale" in body and "rt_progressive_begin" in body       # update_frame() passes the global on
    others = [os.path.join(REF_SRC, f) for f in ("scene.c", "vector.c", "os.c", "utils.c", "gpu_and_windowing.c")] + [os.path.join(ref, "3p", "glad", "src", "glad.c")]
    out = tmp_path / "ray_trace_rt.so"
    link = (["gcc", "-shared", "-o", str(out)] + objs + [f for f in flags if not f.startswith("-Werror")] + ["-w"] + others +
            ["-L", os.path.dirname(rt.LIB_PATH), "-lrt_hip", "-Wl,-rpath," + os.path.dirname(rt.LIB_PATH), "-Wl,--gc-sections", "-Wl,--no-undefined"] +
            ["-Wl,--ignore-unresolved-symbol," + g for g in GLFW_SYMBOLS] + ["-lm", "-lpthread", "-ldl"])
    r = subprocess.run(link, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    needed = subprocess.run(["readelf", "-d", str(out)], check=True, capture_output=True, text=True).stdout
    assert "librt_hip.so" in needed
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(out)], check=True, capture_output=True, text=True).stdout
    assert ("rt_progressive_passes" if variant == "--ladder" else "rt_render") in undefined and "glfwInit" in undefined      # (bound to the library at load time / left to a GLFW the box lacks)
