"""Development aid: emit the ISA of the scene-specialised trace kernel (what rt_compile_scene builds with
hiprtc) offline, for reading.  usage: spec_asm.py scene.txt out.s [extra hipcc flags...]
Needs no GPU: the scene is parsed with the library's host-side loader and packed as rt_set_scene does."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracing_amd as rt

def header(scene_path):
    rc, buf = rt.parse_scene_file(scene_path)
    assert rc == 0, rc
    n = int(buf[69632:69636].view(np.int32)[0])
    T, G = [], []
    f = np.float32
    for i in range(n):
        rec = buf[68 * i: 68 * i + 28]
        typ = int(rec[:4].view(np.int32)[0]); g = rec[4:28].view(np.float32)
        if typ == 0:       # cube: origin, size (scene.h:14-17)
            G.append([g[0], g[1], g[2]] + [f(f(g[k] * f(1)) + f(g[3 + k] * f(1))) for k in range(3)]); T.append(0)
        else:              # sphere: center, radius (vector.h:58-61)
            G.append([g[0], g[1], g[2], f(g[3] * g[3]), f(0), f(0)]); T.append(1)
    # first emitter and its origin_of() (main.c:140-146, scene.c:10-15)
    light, lpos = -1, [f(0)] * 3
    for i in range(n):
        rec = buf[68 * i: 68 * i + 68]
        epow = rec[28 + 24: 28 + 28].view(np.float32)[0]
        if light < 0 and epow > 0:
            light = i
            g = rec[4:28].view(np.float32)
            lpos = [g[0], g[1], g[2]] if T[i] == 1 else [f(f(g[k] * f(1)) + f(g[3 + k] * f(0.5))) for k in range(3)]
    h0 = "#define SPEC_LIGHT %d\nstatic constexpr float SPEC_LIGHT_POS[3] = {%s};\n" % (light, ", ".join(float(v).hex() + "f" for v in lpos))
    only = all(not (buf[68 * i + 28 + 24: 68 * i + 28 + 28].view(np.float32)[0] != 0 and any(buf[68 * i + 28: 68 * i + 28 + 12].view(np.float32) != 0)) for i in range(n) if i != light)
    h0 += "#define SPEC_ONLY_LIGHT_EMITS %d\n" % (1 if only and light >= 0 else 0)      # (rt_set_scene decides this from the packed emission; close enough for reading ISA)
    h = h0 + "#define SPEC_N %d\nstatic constexpr int SPEC_T[SPEC_N] = {%s};\nstatic constexpr float SPEC_G[SPEC_N][6] = {\n" % (n, ", ".join(map(str, T)))
    h += ",\n".join("\t{" + ", ".join(float(v).hex() + "f" for v in g) + "}" for g in G) + "\n};\n"
    return h

if __name__ == "__main__":
    scene, out = sys.argv[1], sys.argv[2]
    csrc = os.path.join(ROOT, "ray_tracing_amd", "csrc")
    hdr = os.path.join(ROOT, "build_variants", "rt_scene_spec.h")
    os.makedirs(os.path.dirname(hdr), exist_ok=True)
    open(hdr, "w").write(header(scene))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-DRT_SPEC_ONLY",
           "-DRT_SPEC_HEADER=\"rt_scene_spec.h\"", "-DRT_WAVES_PER_SIMD=4", "-I", os.path.dirname(hdr), "-I", csrc,
           "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", os.path.join(csrc, "rt_kernels.hip"), "-o", out] + sys.argv[3:]
    print(" ".join(cmd)); sys.exit(subprocess.call(cmd))
