"""bench.py --gpus N must be launchable the way the driver launches N = 1 (a plain `python bench.py --gpus N`):
the parent starts the N ranks as a child process tree (torch.distributed.run) before anything touches a GPU.
Here the same launcher is driven at world size 2 over gloo with --launch-check (rendezvous + one all-reduce,
no GPU work), and the already-launched form (WORLD_SIZE set by the caller) is checked as well."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_plain_invocation_spawns_the_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _last_json(p.stdout)
    assert out == {"launch_check": True, "world": 2, "backend": "gloo", "rank_sum": 3, "expected": 3}
    assert p.stdout.count('"launch_check"') == 1          # ONE line, from rank 0


def test_single_rank_needs_no_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)["world"] == 1


def test_native_multi_does_not_launch_ranks():
    """`bench.py --gpus N --native-multi` is ONE process that drives the N devices through rt_multi_* (ncclCommInitAll inside the
    library): no torch.distributed.run child, no rendezvous -- the launch check sees a world of one."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--native-multi", "--launch-check"], capture_output=True, text=True,
                       timeout=120, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["launch_check"] is True and line["world"] == 1 and line["backend"] is None
