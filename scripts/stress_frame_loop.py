"""Stress reproducer for the intermittent "last row blocks differ" verification failure of the N-rank frame loop
(DESIGN.md "Open").  Every frame of every loop is compared, whole and bit for bit, with a
reference frame of its seed; a mismatch is located (rows, strips, which launch slot / strip buffer / frame buffer it went
through) and logged.  One GPU is enough: ranks share it (gloo moves the strips), as in the test that failed.

  stress_frame_loop.py check   [iters]     rank 0 renders blocking C1 frames on FRESH contexts (what bench.py's verification
                                           does) while 3 other processes create contexts, render and tear them down
  stress_frame_loop.py check0  [iters]     the same with nobody else on the GPU (control)
  stress_frame_loop.py busy    [iters]     the same while 3 other processes render continuously on contexts they keep
  stress_frame_loop.py oversub [iters]     `check` plus two processes that only HOLD hardware queues (what the pytest process is to the
                                           bench inside the suite): six processes on the GPU
  stress_frame_loop.py loop    [frames]    4 ranks sharing the GPU: multi_gpu.TiledFrame over gloo, per-frame compare on rank 0
  stress_frame_loop.py rccl1   [frames]    one rank: TiledFrame(force_collective) over a one-rank RCCL group
  stress_frame_loop.py native  [frames]    one rank: rt_multi_frame_* over a one-rank RCCL communicator (depth 2, 3, 6, 8)
  stress_frame_loop.py queue   [frames]    one rank: rt_frame_* (depth 2, 3, 6, 8)
  stress_frame_loop.py onedev  [frames]    one process: rt_multi_frame_* over EIGHT contexts on the one GPU (depth 2, 3, 6, 8)

Results: one JSON line per mode on stdout and in gpurun_out/stress/<mode>.json.
"""
import json
import os
import random
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

W, H, SPP, NB = 1920, 1080, 64, 4
SEEDS = list(range(100, 107))            # 7 seeds in rotation: coprime with the 2 / 3 / 4 buffers in rotation
OUT = os.path.join(ROOT, "gpurun_out", "stress")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def setup(rt, device=0, poison=False):
    g = rt.Renderer(device)
    if poison:
        g.set_tuning(poison_frame=True)
    g.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); g.set_skybox(rt.load_skybox()); g.set_camera()
    return g


def references(rt, np):
    """Blocking generic-kernel renders of the seeds on an otherwise idle GPU, pinned by oracle rows (first, middle, last rows)."""
    from rtlibs import Oracle
    g = setup(rt, poison=True)
    ref = {s: g.render(W, H, SPP, NB, seed=s) for s in SEEDS}
    again = {s: g.render(W, H, SPP, NB, seed=s) for s in SEEDS}
    assert all((ref[s].view(np.uint32) == again[s].view(np.uint32)).all() for s in SEEDS), "two renders alone on the GPU differ"
    g.close()
    o = Oracle(); o.load_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); o.set_skybox(rt.load_skybox()); o.set_camera()
    rows = [0, H // 2, H - 20, H - 3, H - 1]
    for s in SEEDS[:2]:
        got = o.render_counter_rows(W, H, SPP, NB, rows, seed=s)
        assert all((ref[s][r].view(np.uint32) == v.view(np.uint32)).all() for r, v in got.items()), "reference frame differs from the oracle"
    return ref


def where(np, got, want, world=4, row_block=8):
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
    rows = np.flatnonzero(bad.any(axis=1))
    nan = int(np.isnan(got).any(axis=2).sum())
    return {"pixels": int(bad.sum()), "rows": int(rows.size), "first_row": int(rows[0]) if rows.size else -1, "last_row": int(rows[-1]) if rows.size else -1,
            "nan_pixels": nan,
            "rows_per_strip": [int(sum(1 for r in rows if (r // row_block) % world == s)) for s in range(world)]}


# ---- check / check0 / busy: rank 0's blocking renders beside other processes -----------------------------------------------

def _churner(kind, go, stop, started):
    """kind 'churn': create a context, upload, render a few frames through the frame queue, destroy, again.
    kind 'busy': one context for the whole time, frames back to back."""
    go.wait(600)                          # rank 0 makes its reference frames alone on the GPU first
    import numpy as np  # noqa: F401
    import torch  # noqa: F401
    import ray_tracing_amd as rt
    from ray_tracing_amd.frames import FrameLoop
    n = 0
    g = None
    while not stop.is_set():
        if g is None:
            g = setup(rt)
            loop = FrameLoop(g, W, H, SPP, NB, depth=2)
        loop.run(range(n, n + 3))
        n += 3
        started.set()
        if kind == "churn":
            loop.close(); g.close(); g = None
    if g is not None:
        loop.close(); g.close()


def _ballast(go, stop, ready, contexts):
    """Holds hardware queues and does nothing else: `contexts` renderers, each with both render streams and the frame queue's
    copy stream used once (HIP creates a queue when a stream first gets work), plus torch streams of both priorities.  What
    the pytest process is to `bench.py --gpus 4 --share-gpu` inside the suite: one more process with queues on the GPU."""
    go.wait(600)
    import torch
    import ray_tracing_amd as rt
    from ray_tracing_amd.frames import FrameLoop
    keep = []
    for _ in range(contexts):
        g = setup(rt)
        loop = FrameLoop(g, 320, 180, 4, 4, depth=2)
        loop.run(range(4))
        keep.append((g, loop))
    ts = [torch.cuda.Stream(0, priority=p) for p in (0, -1, 0, -1)]
    for t in ts:
        with torch.cuda.stream(t):
            torch.zeros(16, device="cuda:0").add_(1)
    torch.cuda.synchronize()
    ready.set()
    stop.wait(3600)
    for g, loop in keep:
        loop.close(); g.close()


def mode_check(kind, iters):
    import ctypes as C
    import numpy as np
    import torch
    import torch.multiprocessing as mp
    import ray_tracing_amd as rt
    ctx = mp.get_context("spawn")
    stop, go = ctx.Event(), ctx.Event()
    procs, flags = [], []
    if kind != "check0":                  # (started before this process touches the GPU)
        for _ in range(3):
            e = ctx.Event()
            p = ctx.Process(target=_churner, args=("busy" if kind == "busy" else "churn", go, stop, e))
            p.start(); procs.append(p); flags.append(e)
    if kind == "oversub":                 # two more processes that only hold queues: six processes on the GPU, the pool's limit
        for _ in range(2):
            e = ctx.Event()
            p = ctx.Process(target=_ballast, args=(go, stop, e, 3))
            p.start(); procs.append(p); flags.append(e)
    ref = references(rt, np)
    go.set()
    for e in flags:
        e.wait(180)
    bad, t0 = [], time.time()
    keep = setup(rt)                      # a context that stays: its renders alternate with the fresh contexts'
    for i in range(iters):
        s = SEEDS[i % len(SEEDS)]
        extra = {}
        if i % 3 == 1:
            g = setup(rt)
            got = g.render(W, H, SPP, NB, seed=s)
            cancelled = g.was_cancelled()
            g.close()
        elif i % 3 == 2:
            # a fresh context again, but the frame stays on the device and is read back TWICE: a frame that is wrong the first time
            # and right the second was a transfer that returned early, one that is wrong twice was not written by the kernels
            g = setup(rt)
            d = torch.full((H, W, 3), float("nan"), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            g.render_device(g.params(W, H, SPP, NB, seed=s), d.data_ptr()); g.synchronize()
            got = np.empty((H, W, 3), np.float32)
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy(C.c_void_p(got.ctypes.data), C.c_void_p(d.data_ptr()), C.c_size_t(got.nbytes), 2)      # hipMemcpyDeviceToHost into pageable memory, as rt_render()
            second = d.cpu().numpy()
            extra = {"read_twice": True, "second_read_equals_reference": bool((second.view(np.uint32) == ref[s].view(np.uint32)).all())}
            cancelled = g.was_cancelled()
            g.close()
        else:
            got = keep.render(W, H, SPP, NB, seed=s)
            cancelled = keep.was_cancelled()
        if not (got.view(np.uint32) == ref[s].view(np.uint32)).all():
            w = where(np, got, ref[s]); w.update(iteration=i, seed=s, fresh_context=bool(i % 3), cancelled=bool(cancelled)); w.update(extra)
            bad.append(w)
            print("MISMATCH", json.dumps(w), flush=True)
        if i % 20 == 19:
            print(f"[{kind}] {i + 1}/{iters} renders, {len(bad)} mismatching, {time.time() - t0:.0f} s", flush=True)
    keep.close()
    stop.set()
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
    return {"mode": kind, "renders": iters, "mismatching": len(bad), "details": bad[:20], "others": len(procs)}


# ---- loop: the bench's N-rank loop, ranks sharing the GPU ------------------------------------------------------------------

def _loop_rank(rank, world, port, frames, q, delays):
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt
    from ray_tracing_amd.multi_gpu import TiledFrame
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = references(rt, np) if rank == 0 else None
    dist.barrier()
    g = setup(rt)
    g.compile_scene()
    t = TiledFrame(g, W, H, SPP, NB, rank=rank, world=world, device=dev)
    rnd = random.Random(1234 + rank)
    bad = []
    t0 = time.time()
    for k in range(frames):
        if rank == 0 and k >= 2:
            # frame k-2 went to host_frames[k & 1], which the copy of frame k is about to overwrite: look at it now
            t.copied[k & 1].synchronize()
            s = SEEDS[(k - 2) % len(SEEDS)]
            got = t.host_frames[k & 1].numpy()
            if not (got.view(np.uint32) == ref[s].view(np.uint32)).all():
                w = where(np, got, ref[s], world); w.update(frame=k - 2, seed=s, strip_buffer=(k - 2) % 3, launch_slot=(k - 2) & 1)
                bad.append(w); print("MISMATCH", json.dumps(w), flush=True)
        if delays and rnd.random() < 0.2:
            time.sleep(rnd.random() * 0.02)
        t.step(seed=SEEDS[k % len(SEEDS)])
        if rank == 0 and k % 100 == 99:
            print(f"[loop] {k + 1}/{frames} frames, {len(bad)} mismatching, {time.time() - t0:.0f} s", flush=True)
    t.flush()
    if rank == 0:
        for k in (frames - 2, frames - 1):
            s = SEEDS[k % len(SEEDS)]
            got = t.host_frames[k & 1].numpy()
            if not (got.view(np.uint32) == ref[s].view(np.uint32)).all():
                w = where(np, got, ref[s], world); w.update(frame=k, seed=s, strip_buffer=k % 3, launch_slot=k & 1)
                bad.append(w); print("MISMATCH", json.dumps(w), flush=True)
    # and what the bench does next: rank 0 renders a blocking frame on a FRESH context while the others are free to leave
    if rank == 0:
        for i in range(6):
            c = setup(rt)
            got = c.render(W, H, SPP, NB, seed=SEEDS[i % len(SEEDS)])
            c.close()
            if not (got.view(np.uint32) == ref[SEEDS[i % len(SEEDS)]].view(np.uint32)).all():
                w = where(np, got, ref[SEEDS[i % len(SEEDS)]], world); w.update(check_render=i)
                bad.append(w); print("MISMATCH (check render beside ranks tearing down)", json.dumps(w), flush=True)
        q.put({"frames": frames, "mismatching": len(bad), "details": bad[:20]})
    g.close()                 # ranks 1.. tear down beside rank 0's check renders, as bench.py's ranks did in round 3
    dist.barrier()
    dist.destroy_process_group()


def mode_loop(frames, delays=True, ballast=0):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    go, stop = ctx.Event(), ctx.Event()
    extra, flags = [], []
    for _ in range(ballast):
        e = ctx.Event()
        p = ctx.Process(target=_ballast, args=(go, stop, e, 3))
        p.start(); extra.append(p); flags.append(e)
    go.set()
    for e in flags:
        e.wait(180)
    procs = [ctx.Process(target=_loop_rank, args=(r, 4, port, frames, q, delays)) for r in range(4)]
    for p in procs:
        p.start()
    res = q.get(timeout=1100)
    stop.set()
    for p in procs + extra:
        p.join(120)
        if p.is_alive():
            p.kill()
    res["mode"] = f"loop (4 ranks sharing the GPU, gloo, {ballast} more processes holding queues)"
    return res


# ---- one-rank loops ----------------------------------------------------------------------------------------------------------

def mode_rccl1(frames):
    import numpy as np
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt
    from ray_tracing_amd.multi_gpu import TiledFrame
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    ref = references(rt, np)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = setup(rt); g.compile_scene()
    t = TiledFrame(g, W, H, SPP, NB, rank=0, world=1, device=dev, force_collective=True)
    rnd = random.Random(7)
    bad = []
    for k in range(frames):
        if k >= 2:
            t.copied[k & 1].synchronize()
            s = SEEDS[(k - 2) % len(SEEDS)]
            got = t.host_frames[k & 1].numpy()
            if not (got.view(np.uint32) == ref[s].view(np.uint32)).all():
                w = where(np, got, ref[s], 1); w.update(frame=k - 2, seed=s, strip_buffer=(k - 2) % 3, launch_slot=(k - 2) & 1)
                bad.append(w); print("MISMATCH", json.dumps(w), flush=True)
        if rnd.random() < 0.2:
            time.sleep(rnd.random() * 0.02)
        t.step(seed=SEEDS[k % len(SEEDS)])
        if k % 100 == 99:
            print(f"[rccl1] {k + 1}/{frames} frames, {len(bad)} mismatching", flush=True)
    t.flush()
    g.close()
    dist.destroy_process_group()
    return {"mode": "rccl1 (TiledFrame over a one-rank RCCL group)", "frames": frames, "mismatching": len(bad), "details": bad[:20]}


def _queue_mode(frames, native, contexts=0):
    import numpy as np
    import torch  # noqa: F401
    import ray_tracing_amd as rt
    from ray_tracing_amd.frames import FrameLoop
    ref = references(rt, np)
    out = {"mode": f"onedev (rt_multi_frame_* over {contexts} contexts on one GPU)" if contexts else
                   "native (rt_multi_frame_* over a one-rank RCCL communicator)" if native else "queue (rt_frame_*)", "depths": {}}
    for depth in (2, 3, 6, 8):
        if contexts:
            q = rt.MultiRenderer([0], on_one_device=contexts)
            q.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); q.set_skybox(rt.load_skybox()); q.set_camera(); q.compile_scene()
        elif native:
            q = rt.MultiRenderer([0]); q.set_tuning(force_collective=1)
            q.set_scene(os.path.join(rt.DATA_DIR, "scene_0.txt")); q.set_skybox(rt.load_skybox()); q.set_camera(); q.compile_scene()
        else:
            q = setup(rt); q.compile_scene()
        loop = FrameLoop(q, W, H, SPP, NB, depth=depth)
        rnd = random.Random(depth)
        bad = []
        seeds = [SEEDS[k % len(SEEDS)] for k in range(frames)]

        def look(k, a):
            if not (a.view(np.uint32) == ref[seeds[k]].view(np.uint32)).all():
                w = where(np, a, ref[seeds[k]], 1); w.update(frame=k, seed=seeds[k], depth=depth)
                bad.append(w); print("MISMATCH", json.dumps(w), flush=True)
            if rnd.random() < 0.2:
                time.sleep(rnd.random() * 0.02)
        loop.run(seeds, on_frame=look)
        out["depths"][depth] = {"frames": frames, "mismatching": len(bad), "cancelled": loop.cancelled, "details": bad[:10]}
        print(f"[{'native' if native else 'queue'}] depth {depth}: {frames} frames, {len(bad)} mismatching", flush=True)
        loop.close(); q.close()
    out["mismatching"] = sum(d["mismatching"] for d in out["depths"].values())
    return out


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (120 if mode.startswith("check") or mode == "busy" else 500)
    os.makedirs(OUT, exist_ok=True)
    if mode in ("check", "check0", "busy", "oversub"):
        res = mode_check(mode, n)
    elif mode == "loop":
        res = mode_loop(n)
    elif mode == "loop6":
        res = mode_loop(n, ballast=2)
    elif mode == "rccl1":
        res = mode_rccl1(n)
    elif mode == "native":
        res = _queue_mode(n, True)
    elif mode == "queue":
        res = _queue_mode(n, False)
    elif mode == "onedev":
        res = _queue_mode(n, True, contexts=8)
    else:
        raise SystemExit(__doc__)
    line = json.dumps(res)
    print(line, flush=True)
    with open(os.path.join(OUT, mode + ".json"), "w") as f:
        f.write(line + "\n")
