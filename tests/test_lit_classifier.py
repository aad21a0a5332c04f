"""rt_lit.h (which soft-shadow taps need no trace) against the oracle, on the CPU: tests/lit_probe.c calls the shipped
function at every shading point of a frame -- every bounce, although the kernels only use it for camera-ray hits -- and
compares each "certainly lit" with the oracle's trace_ray() result for that tap (main.c:191-206)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lit") / "lit_probe")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-o", exe, os.path.join(ROOT, "tests", "lit_probe.c"), "-lm", "-lpthread"])
    return exe


@pytest.mark.parametrize("camera", [(), ("1", "1", "8", "0.3", "-0.1", "-1"), ("0.5", "4", "4", "1", "-0.6", "-0.2"), ("3", "9", "3", "0.01", "-1", "0.01")])
def test_shipped_scene_every_answer_matches_the_trace(probe, camera):
    r = subprocess.run([probe, os.path.join(ROOT, "data", "scene_0.txt"), "160", "90", "4", "8", *camera], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    last = [l for l in r.stdout.splitlines() if l.startswith("all:")][0].split()
    # all: taps N lit x % answered y % violations v table t % table violations w   (point classifier; the per-scene table)
    assert int(last[2]) > 20000 and float(last[7]) > 40.0 and float(last[12]) > 30.0 and int(last[10]) == 0 and int(last[-1]) == 0, r.stdout


def test_scenes_without_a_sphere_emitter_are_never_answered(probe):
    for scene in ("scene_1.txt", "scene_2.txt"):              # a cube emitter; no emitter at all
        r = subprocess.run([probe, os.path.join(ROOT, "data", scene), "64", "48", "2", "4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("all:") and " answered 0.0 %" not in l], r.stdout


@pytest.mark.parametrize("cases,seed,scale", [("150", "11", "1"), ("400", "8", "4.5")])
def test_random_scenes(probe, cases, seed, scale):
    """scale 4.5, seed 8, scene 239: a sphere of radius 2.25 hit from 40 units away -- the reference's float discriminant
    puts the hit point 2e-4 INSIDE the sphere, and a tap from there hits the sphere itself.  The classifier measures how
    far a hit point is off its surface (rt_lit_point_on_surface) instead of assuming it is on it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lit_fuzz.py"), cases, seed, scale], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 scenes with violations" in r.stdout
