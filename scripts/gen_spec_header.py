"""Write ray_tracing_amd/csrc/spec_scene.h for a scene file (experiment: compile-time-specialised trace loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
path = sys.argv[1]
rc, buf = rt.parse_scene_file(path)
assert rc == 0
n = int(buf[68 * 1024:].view("<i4")[0])
objs = buf[:68 * n].view(np.dtype([("type", "<i4"), ("geom", "<f4", (6,)), ("rest", "<f4", (10,))]))
def lit(x):
    return float(np.float32(x)).hex() + "f"
rows, types = [], []
for o in objs:
    g = o["geom"]
    if o["type"] == 0:
        lo = g[:3]; hi = (g[:3] * np.float32(1) + g[3:] * np.float32(1)).astype(np.float32)
        rows.append([lit(v) for v in list(lo) + list(hi)]); types.append(0)
    else:
        rows.append([lit(v) for v in list(g[:3]) + [np.float32(g[3]) * np.float32(g[3]), 0, 0]]); types.append(1)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray_tracing_amd", "csrc", "spec_scene.h")
with open(out, "w") as f:
    f.write(f"/* generated from {os.path.basename(path)} */\n#define SPEC_N {n}\n")
    f.write("static constexpr int SPEC_T[SPEC_N] = {" + ", ".join(map(str, types)) + "};\n")
    f.write("static constexpr float SPEC_G[SPEC_N][6] = {\n" + ",\n".join("\t{" + ", ".join(r) + "}" for r in rows) + "\n};\n")
print("wrote", out)
