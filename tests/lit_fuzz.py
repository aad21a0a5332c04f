"""Development aid (CPU only): random scenes with one emitter -- a sphere or a cube, thin panels and cubes that touch their
neighbours among them -- through tests/lit_probe.c: the shipped rt_lit.h against the oracle's trace at every shading point of
every bounce.  usage: lit_fuzz.py [cases] [seed] [scale] [camera distance] [sphere|box|mixed]   (exit status 1 on a violation;
scale multiplies every coordinate and size)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(tempfile.gettempdir(), f"lit_probe_{os.getpid()}")
subprocess.check_call(["gcc", "-O2", "-std=c11", "-o", exe, os.path.join(ROOT, "tests", "lit_probe.c"), "-lm", "-lpthread"])
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
far = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0      # > 0: the camera stands this many scene sizes away and looks at the scene
emitters = sys.argv[5] if len(sys.argv) > 5 else "mixed"
def num(x): return "%.9f" % float(x)
def vec(v): return "{%s}" % " ".join(num(x) for x in v)
bad = taps = known = tabled = dark = dark_tabled = alone = 0
for case in range(cases):
    n = int(rng.integers(2, 14))
    light = int(rng.integers(0, n))
    txt = []
    for k in range(n):
        P = lambda name, value: name.ljust(15) + value       # the reference's parser skips fixed widths after some names (scene.c:280,320)
        mat = [P("emission_color", "{1.0 1.0 1.0}"), P("emission_power", num(4.0 if k == light else 0.0)), P("metallic", num(rng.choice([0, 0, 1]))),
               P("reflectance", num(rng.uniform(0, 1))), P("roughness", num(rng.choice([0, 0.3, 1.0]))), P("albedo", vec(rng.uniform(0, 1, 3)))]
        tight = rng.random() < 0.5          # objects that touch: shared planes, spheres resting on slabs
        ball = rng.random() < 0.4
        if k == light: ball = emitters == "sphere" or (emitters == "mixed" and rng.random() < 0.5)
        if ball:
            c = rng.integers(-3, 7, 3).astype(float) if tight else rng.uniform(-3, 7, 3)
            r = float(rng.choice([0.25, 0.5, 1.0, 2.0])) if tight else float(rng.uniform(0.06, 2.0))
            txt += ["sphere"] + ["\t" + m for m in mat] + ["\t" + P("center", vec(c * scale)), "\t" + P("radius", num(r * scale)), ""]
        else:
            o = rng.integers(-3, 7, 3).astype(float) if tight else rng.uniform(-3, 7, 3)
            sz = rng.choice([0.1, 0.5, 1.0, 3.0, 9.0], 3) if tight else rng.uniform(0.05, 5, 3)
            if k == light and rng.random() < 0.6: sz = np.minimum(np.asarray(sz, dtype=float), rng.choice([0.1, 0.5, 1.0, 2.0]))     # emitters are seldom huge
            if k == light and rng.random() < 0.5: sz = np.asarray(sz, dtype=float); sz[int(rng.integers(0, 3))] = 0.1          # a panel
            txt += ["cube"] + ["\t" + m for m in mat] + ["\t" + P("origin", vec(o * scale)), "\t" + P("size", vec(np.asarray(sz) * scale)), ""]
    path = os.path.join(tempfile.gettempdir(), f"lit_fuzz_scene_{os.getpid()}.txt")
    open(path, "w").write("\n".join(txt))
    pos = rng.integers(-2, 9, 3).astype(float) if rng.random() < 0.3 else rng.uniform(-2, 9, 3)
    front = rng.uniform(-1, 1, 3)
    if far > 0:
        pos = rng.normal(size=3); pos = pos / np.linalg.norm(pos) * 10.0 * far
        front = (np.array([2.0, 2.0, 2.0]) + rng.uniform(-3, 3, 3)) - pos
    r = subprocess.run([exe, path, "160", "120", "4", "6"] + [repr(float(x)) for x in list(pos * scale) + list(front)], capture_output=True, text=True)
    last = [l for l in r.stdout.splitlines() if l.startswith("all:")]
    if last:
        w = last[0].split()
        taps += int(w[2]); known += int(round(float(w[7]) / 100 * int(w[2]))); tabled += int(round(float(w[12]) / 100 * int(w[2])))
        dl = [l for l in r.stdout.splitlines() if l.startswith("dark:")]       # "dark: x % of the taps ... (bounce 0: y %), z % by the table; violations v"
        if dl:
            d = dl[0].split()
            dark += int(round(float(d[1]) / 100 * int(w[2]))); dark_tabled += int(round(float(d[d.index("%),") + 1]) / 100 * int(w[2])))
        al = [l for l in r.stdout.splitlines() if l.startswith("alone:")]     # "alone: x % of the taps (bounce 0: y %), ...; violations v"
        if al:
            d = al[0].split()
            alone += int(round(float(d[1]) / 100 * int(w[2])))
    if r.returncode == 2:
        raise SystemExit(f"case {case}: scene file rejected\n{r.stderr}")
    if r.returncode != 0:
        bad += 1
        print(f"case {case}: rc={r.returncode}\n{r.stdout}{r.stderr}", flush=True)
        os.replace(path, os.path.join(tempfile.gettempdir(), f"lit_fuzz_bad_{case}.txt"))
for f in (path, exe):
    if os.path.exists(f): os.remove(f)
print(f"{cases} scenes, {taps} taps checked, {known} answered without tracing ({100.0 * known / max(taps, 1):.1f} %), {tabled} by the per-scene table ({100.0 * tabled / max(taps, 1):.1f} %), "
      f"{dark} certainly NOT reaching the emitter ({100.0 * dark / max(taps, 1):.1f} %; {dark_tabled} by the table), "
      f"{alone} more that would need only the emitter and what touches it intersected (a look ahead, not in the product: {100.0 * alone / max(taps, 1):.1f} %), {bad} scenes with violations")
sys.exit(1 if bad else 0)
