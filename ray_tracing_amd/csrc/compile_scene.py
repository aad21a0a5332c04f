"""Build step: compile rt_kernels.hip for one scene header into a gfx950 code object, with exactly the options
rt_compile_scene() passes to hiprtc at run time (rt_jit.cpp), by one of two compilers:
   compile_scene.py hipcc  <hipcc>        <header.h> <out.co>      the toolchain that builds the library
   compile_scene.py hiprtc <libhiprtc.so> <header.h> <out.co>      a hiprtc library (no GPU needed), e.g. the one PyTorch bundles
Prints one line naming the compiler (it is embedded next to the code object)."""
import ctypes as C, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
OPTS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-DRT_SPEC_ONLY",
        "-DRT_SPEC_HEADER=\"rt_scene_spec.h\"", "-DRT_WAVES_PER_SIMD=4"]
kind, tool, header, out = sys.argv[1:5]
OPTS = OPTS + sys.argv[5:]          # (development: e.g. -gline-tables-only for scripts/instruction_budget.py)
if kind == "hipcc":
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "rt_scene_spec.h"), "w") as f:
            f.write(open(header).read())
        subprocess.check_call([tool] + OPTS + ["-I", d, "-I", HERE, "--genco", os.path.join(HERE, "rt_kernels.hip"), "-o", out])
    v = subprocess.run([tool, "--version"], capture_output=True, text=True).stdout.splitlines()
    print("hipcc " + next((l.split(":", 1)[1].strip() for l in v if l.startswith("HIP version")), "?"))
else:
    R = C.CDLL(tool)
    names = ["rt_math.hip.h", "rt_device.h", "rt_lit.h", "rt_stats.hip.h", "rt_scene_spec.h"]
    hdrs = [open(os.path.join(HERE, n)).read().encode() for n in names[:4]] + [open(header).read().encode()]
    prog = C.c_void_p()
    R.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    rc = R.hiprtcCreateProgram(C.byref(prog), open(os.path.join(HERE, "rt_kernels.hip")).read().encode(), b"rt_kernels.hip", 5,
                               (C.c_char_p * 5)(*hdrs), (C.c_char_p * 5)(*[n.encode() for n in names]))
    assert rc == 0, rc
    R.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
    rc = R.hiprtcCompileProgram(prog, len(OPTS), (C.c_char_p * len(OPTS))(*[o.encode() for o in OPTS]))
    n = C.c_size_t()
    R.hiprtcGetProgramLogSize(prog, C.byref(n))
    if rc != 0:
        log = C.create_string_buffer(n.value); R.hiprtcGetProgramLog(prog, log); sys.stderr.write(log.value.decode()[:4000])
        sys.exit(1)
    R.hiprtcGetCodeSize(prog, C.byref(n))
    code = C.create_string_buffer(n.value); R.hiprtcGetCode(prog, code)
    open(out, "wb").write(code.raw)
    # which ROCm it belongs to: PyTorch's bundle says so in torch/version.py (read as text: importing torch takes seconds)
    rocm = ""
    vpy = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(tool))), "version.py")
    if os.path.exists(vpy):
        for line in open(vpy):
            if line.startswith("hip"):
                rocm = " of ROCm " + line.split("=", 1)[1].strip().strip("'\"") + " (PyTorch's bundle)"
    elif "rocm" in os.path.realpath(tool):
        rocm = " of " + [c for c in os.path.realpath(tool).split(os.sep) if c.startswith("rocm")][0]
    print(f"hiprtc{rocm}, {os.path.basename(os.path.realpath(tool))}")
