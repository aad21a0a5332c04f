"""Development aid: emit the ISA of the scene-specialised trace kernel (what rt_compile_scene builds) offline, for reading.
usage: spec_asm.py scene.txt out.s [extra hipcc flags...]
Needs no GPU: the scene header is the one the build makes (csrc/rt_embed_tool: the library's loader + rt_pack.cpp), the options
are compile_scene.py's; everything temporary goes to $TMPDIR."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracing_amd", "csrc")

def header(scene_path):
    tool = os.path.join(CSRC, "rt_embed_tool")
    if not os.path.exists(tool):
        subprocess.check_call(["make", "-C", CSRC, "rt_embed_tool"])
    return subprocess.check_output([tool, scene_path], text=True)

if __name__ == "__main__":
    scene, out = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "rt_scene_spec.h"), "w").write(header(scene))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-DRT_SPEC_ONLY",
               "-DRT_SPEC_HEADER=\"rt_scene_spec.h\"", "-DRT_WAVES_PER_SIMD=4", "-I", d, "-I", CSRC,
               "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "rt_kernels.hip"), "-o", out] + sys.argv[3:]
        print(" ".join(cmd)); sys.exit(subprocess.call(cmd))
