"""Reproducer for round 3's intermittent verification failure of `bench.py --gpus 4 --share-gpu` (2 of 12 runs of the -m gpu
suite, 0 of 24 outside it): the bench as it was then -- ranks 1..3 tear their contexts down while rank 0 renders its blocking
check frame on a fresh context (--leave-early) -- started from a process that, like pytest with its module fixtures, holds a
GPU context of its own.  usage: repro_verify_race.py [runs] [early|wait] [parent|noparent]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
early = (sys.argv[2] if len(sys.argv) > 2 else "early") == "early"
parent = (sys.argv[3] if len(sys.argv) > 3 else "parent") == "parent"
keep = []
if parent:                     # what the pytest process holds when the test runs: a renderer with the skybox, torch initialised
    import torch
    import ray_tracing_amd as rt
    g = rt.Renderer(0); g.set_tuning(poison_frame=True)
    g.set_skybox(rt.load_skybox()); g.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); g.set_camera()
    g.render(1920, 1080, 16, 4, seed=1)
    keep = [g, torch.zeros(1 << 20, device="cuda:0")]
bad = []
t0 = time.time()
for k in range(runs):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--share-gpu", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    if early:
        cmd.append("--leave-early")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    try:
        line = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        line = {"verified": None, "stderr": p.stderr[-600:]}
    ok = p.returncode == 0 and line.get("verified") is True
    print(f"run {k}: rc {p.returncode} verified {line.get('verified')} ({time.time() - t0:.0f} s)", flush=True)
    if not ok:
        bad.append({"run": k, "rc": p.returncode, "verification": line.get("verification"), "stderr": p.stderr[-1500:] if p.returncode not in (0, 3) else ""})
        print("FAILED", json.dumps(bad[-1])[:3000], flush=True)
res = {"runs": runs, "failed": len(bad), "leave_early": early, "parent_holds_a_context": parent, "details": bad[:6]}
print(json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "stress"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "stress", f"repro_{'early' if early else 'wait'}_{'parent' if parent else 'noparent'}.json"), "w").write(json.dumps(res) + "\n")
