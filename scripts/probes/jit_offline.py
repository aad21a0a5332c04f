"""Development aid: compile the scene-specialised trace kernel with a given libhiprtc.so (no GPU needed) exactly as
rt_compile_scene does, and write the code object -- to compare compilers.  usage: jit_offline.py <libhiprtc.so> scene.txt out.co [flags...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from spec_asm import header
lib, scene, out = sys.argv[1], sys.argv[2], sys.argv[3]
R = C.CDLL(lib)
csrc = os.path.join(ROOT, "ray_tracing_amd", "csrc")
src = open(os.path.join(csrc, "rt_kernels.hip")).read().encode()
names = ["rt_math.hip.h", "rt_device.h", "rt_lit.h", "rt_scene_spec.h"]
hdrs = [open(os.path.join(csrc, n)).read().encode() for n in names[:3]] + [header(scene).encode()]
prog = C.c_void_p()
R.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
rc = R.hiprtcCreateProgram(C.byref(prog), src, b"rt_kernels.hip", 4, (C.c_char_p * 4)(*hdrs), (C.c_char_p * 4)(*[n.encode() for n in names]))
assert rc == 0, rc
opts = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-DRT_SPEC_ONLY", "-DRT_SPEC_HEADER=\"rt_scene_spec.h\"", "-DRT_WAVES_PER_SIMD=4"] + sys.argv[4:]
R.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
rc = R.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*[o.encode() for o in opts]))
n = C.c_size_t()
R.hiprtcGetProgramLogSize(prog, C.byref(n))
if n.value > 1:
    log = C.create_string_buffer(n.value); R.hiprtcGetProgramLog(prog, log); print(log.value.decode()[:2000])
assert rc == 0, rc
R.hiprtcGetCodeSize(prog, C.byref(n))
code = C.create_string_buffer(n.value); R.hiprtcGetCode(prog, code)
open(out, "wb").write(code.raw)
print("wrote", out, n.value, "bytes")
