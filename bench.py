#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s on BASELINE.json config C1
(scene_0.txt, 1920x1080, 64 spp, 4 bounces) on N GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C4]

For N > 1 either form works: launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (one rank per GPU, RANK/LOCAL_RANK/WORLD_SIZE
from the environment), or plainly as `python bench.py --gpus N`: a parent that has made no GPU call then
starts exactly that command as a child process, relays rank 0's JSON line and exits with the child's status
(the reference fans out from its host binary the same way: start_workers(), main.c:695-706).

A step = one full frame through the hot path, ending where the reference's update_frame() ends
(main.c:467-479): every rank renders its interleaved row blocks with the HIP kernels (librt_hip.so, C ABI),
then -- for N > 1 -- ONE RCCL gather of the finished strips to rank 0 and a de-interleave kernel there, then
the resolved Vector3[W*H] frame is copied to (pinned) HOST memory on rank 0.  Inputs (scene, skybox,
camera) are resident in HBM before the timed region.  Two frames are in flight (three for N > 1): the gather / host
copy of frame k overlaps the render of the following frames; `frame_latency` reports the same frame with nothing
overlapped (first launch -> frame on the host, median of 7).  The same frame is split over N GPUs, so scaling is "strong".
At N = 1 the frame loop is the library's own, behind the C ABI (rt_frame_submit / rt_frame_wait through ctypes:
ray_tracing_amd/frames.py); under torch.distributed.run (N > 1, one process per GPU) the collective is torch's, so the
loop is ray_tracing_amd/multi_gpu.py on top of rt_render_device.  Step k renders seed k, and after the timed region the
last frame is compared bit for bit with a blocking rt_render() of its seed and with rows of the CPU oracle
("verified"); a mismatch makes the process exit non-zero.

Rank 0 prints ONE JSON line; `roofline` is computed from HIP-event kernel times measured over the timed
region and from ALGORITHMIC flops/bytes counted by the CPU oracle's instrumented build; `cpu_baseline` times
the reference's own column-threaded renderer (oracle/_ref, the unmodified reference sources compiled by
oracle/Makefile; the built library travels with the tree) -- or the oracle port if that build
is absent -- on this box's host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

# BASELINE.json configs[1..4]
WORKLOADS = {
    "C1": dict(name="C1", scene="scene_0.txt", width=1920, height=1080, spp=64, max_bounces=4, seed=0),
    "C2": dict(name="C2", scene="scene_1.txt", width=1920, height=1080, spp=256, max_bounces=8, seed=0),
    # (C3's step is the copy of its 99.5 MB frame, not its 0.65 ms of kernels: a third frame in flight adds to the pipeline's fill and to nothing else)
    "C3": dict(name="C3", scene="scene_2.txt", width=3840, height=2160, spp=64, max_bounces=8, seed=0, frames_in_flight=2),
    "C4": dict(name="C4", scene="scene_0.txt", width=3840, height=2160, spp=1024, max_bounces=8, seed=0),
    # not BASELINE configs: synthetic scenes of many objects (tests/rtlibs.py large_scene): the generic kernel with
    # the cluster cull of csrc/rt_cull.h (scenes of 32 objects and more; 64 is the largest a scene-specialised kernel would take)
    # (--every-object: without it, every ray tests every object as the reference does)
    # (two frames in flight for the culled kernels: with three, the third launch of a burst is given half the workgroup slots and keeps them
    # when nothing comes behind it -- the last frames of a twenty-step run then take 24 ms each where a frame takes 6.4: profiles/r06/L1024_depth.txt)
    "L64": dict(name="L64", scene="synthetic:64", width=1920, height=1080, spp=16, max_bounces=5, seed=0, frames_in_flight=2),
    "L256": dict(name="L256", scene="synthetic:256", width=1920, height=1080, spp=16, max_bounces=5, seed=0, frames_in_flight=2),
    "L1024": dict(name="L1024", scene="synthetic:1024", width=1920, height=1080, spp=16, max_bounces=5, seed=0, frames_in_flight=2),
}
ROW_BLOCK = 8

# MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 vector counts an FMA as 2 flops at 64 flop/clk/SIMD.  The
# parity rules forbid FMA contraction, so the applicable issue peak is one flop per lane per issue:
PEAK_VALU_NOFMA_TFLOPS = 157.3 / 2
PEAK_HBM_GBPS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="C1")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-jit", action="store_true", help="do not specialise the trace kernel for the scene")
    ap.add_argument("--no-extras", action="store_true", help="skip frame_latency / generic-kernel legs (profiling runs)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only prove the N-rank launch + rendezvous (no GPU work); used by the CPU tests")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing aid for 1-GPU boxes: all ranks use GPU 0 and the strips travel over gloo "
                         "(RCCL refuses two ranks on one device); not a performance configuration")
    ap.add_argument("--force-collective", action="store_true",
                    help="testing aid for 1-GPU boxes: one rank runs the N > 1 frame loop (RCCL gather on a one-rank group, "
                         "de-interleave, three strip buffers) so that loop's cost shows beside the plain N = 1 line")
    ap.add_argument("--every-object", action="store_true", help="L* workloads: no cluster cull (rt_tuning.test_every_object)")
    ap.add_argument("--native-multi", action="store_true",
                    help="N GPUs from ONE process through the C ABI's device group (rt_multi_frame_submit / rt_multi_frame_wait: "
                         "ncclCommInitAll + one grouped ncclGather per frame) instead of one process per GPU under torch.distributed")
    ap.add_argument("--one-device", action="store_true",
                    help="testing aid for 1-GPU boxes, with --native-multi: the N contexts of the group all live on GPU 0 and the gather "
                         "is the N device copies it amounts to there (rt_multi_create_on_one_device); not a performance configuration")
    ap.add_argument("--torch-loop", action="store_true",
                    help="N = 1: run the torch-side frame loop (multi_gpu.TiledFrame) instead of the C ABI's frame queue")
    ap.add_argument("--leave-early", action="store_true",
                    help="testing aid (scripts/repro_verify_race.py): ranks other than 0 do not wait for rank 0's verification before they "
                         "tear their contexts down -- the bench's behaviour when its verification failed intermittently in round 3")
    ap.add_argument("--depth", type=int, default=0, help="frames in flight in the C ABI's frame queue (default 3; RT_LAUNCH_SETS + 1 = 6 on the N-GPU path)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks as a CHILD
    process tree.  Nothing in this process has touched the GPU (torch is not even imported yet), and it is never
    replaced by exec: it waits for the child and exits with its status."""
    # Under rocprofv3 the profiler's preloaded library has initialised the GPU before main() runs, and starting another
    # program from a GPU-initialised process is what this pool forbids: profiled runs are single-rank (bench.py --gpus 1,
    # scripts/render_cfg.py), see profiles/README.md.
    preload = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in preload:
        print("[bench] --gpus N > 1 cannot self-launch under a profiler (the parent has initialised the GPU); "
              "profile a single rank with --gpus 1", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    return subprocess.call(cmd, env=env)


def algorithmic_work(oracle_count, W, H, max_bounces, seed):
    """Per-sample algorithmic flops (reference operations as written, SURVEY.md 8d cost table) and
    bytes, counted by the instrumented oracle over the whole frame at 1 spp."""
    oracle_count.counters_reset()
    oracle_count.render_counter(W, H, 1, max_bounces, seed=seed)
    c = oracle_count.counters()
    n = max(c["samples"], 1)
    return {k: c[k] / n for k in ("flops", "rays", "object_tests", "rng_draws", "sky_fetches",
                                  "box_tests", "box_flops", "sphere_tests", "sphere_flops", "sky_samples", "sky_sample_flops", "first_ray_flops")}


# measured issue cost of a wave64 VALU instruction per SIMD, four waves per SIMD (profiles/r02/valu_rates.txt, cycles)
VALU_CLASS_CLK = {"f32_add_mul_fma": 2.4, "int32": 3.0, "int64": 4.35, "f64": 4.28, "trans_f32": 8.2, "cvt": 4.2, "other": 4.0}


def simd_issue_estimate(k, kernel_ms):
    """Share of the SIMDs' issue time the kernel's VALU instructions account for: instructions by class (rocprofv3 PMC passes of
    the committed profile) x the measured cost of the class / (1024 SIMDs x 2.4 GHz x kernel time).  Replaces the derived counter
    VALUBusy, which reads above 100 % on gfx950."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64")
    if not k or any(c not in k for c in need):
        return None
    n = {"f32_add_mul_fma": k["SQ_INSTS_VALU_ADD_F32"] + k["SQ_INSTS_VALU_MUL_F32"] + k["SQ_INSTS_VALU_FMA_F32"],
         "int32": k["SQ_INSTS_VALU_INT32"], "int64": k["SQ_INSTS_VALU_INT64"],
         "f64": k.get("SQ_INSTS_VALU_FMA_F64", 0) + k.get("SQ_INSTS_VALU_ADD_F64", 0) + k.get("SQ_INSTS_VALU_MUL_F64", 0),
         "trans_f32": k.get("SQ_INSTS_VALU_TRANS_F32", 0), "cvt": k.get("SQ_INSTS_VALU_CVT", 0)}
    n["other"] = max(k["SQ_INSTS_VALU"] - sum(n.values()), 0.0)
    cycles = sum(n[c] * VALU_CLASS_CLK[c] for c in n)
    return {"valu_instructions_per_launch": round(k["SQ_INSTS_VALU"]), "by_class": {c: round(v) for c, v in n.items()}, "class_clk": VALU_CLASS_CLK,
            "simd_issue_share": round(cycles / (1024 * 2.4e9 * kernel_ms * 1e-3), 4),
            "lanes_active": round(k["SQ_THREAD_CYCLES_VALU"] / k["SQ_ACTIVE_INST_VALU"] / 64, 4) if k.get("SQ_ACTIVE_INST_VALU") else None,
            "note": "sum over classes of instructions x cycles per wave64 instruction (measured, profiles/r02/valu_rates.txt) / (1024 SIMDs x 2.4 GHz x kernel time); "
                    "'other' = compares, selects, min/max, moves, cross-lane; lanes_active = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64"}


def executed_work_of_culled_scene(rt, scene, camera, sky, W, H, spp, nb, seed):
    """What the culled trace of a large scene EXECUTES per sample, counted by the instrumented build of the library
    (librt_hip_stats.so, `make -C ray_tracing_amd/csrc stats`; per-site lane counts, rt_stats.hip.h) on a frame of 1/16 of the
    pixels: exact box / sphere tests (the reference's own, lanes), conservative slab tests of the cull (cluster boxes, member
    boxes), rays, shading events.  None when that build is absent."""
    import ctypes as C
    path = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_hip_stats.so")
    if not os.path.exists(path):
        return None
    keep_lib, keep_path = rt._lib, rt.LIB_PATH
    try:
        rt._lib, rt.LIB_PATH = None, path
        L = rt.lib()
        g = rt.Renderer(0)
        g.set_scene(scene); g.set_skybox(sky); g.set_camera(**camera)
        out = (C.c_ulonglong * 128)()
        L.rt_stats_read(out, 1)
        w, h = max(W // 4, 64), max(H // 4, 36)
        g.render(w, h, spp, nb, seed=seed)
        L.rt_stats_read(out, 1)
        g.close()
    finally:
        rt._lib, rt.LIB_PATH = keep_lib, keep_path
    lanes = lambda site: out[2 * site + 1]      # noqa: E731
    execs = lambda site: out[2 * site]          # noqa: E731
    n = float(w * h * spp)
    return {"per_sample": {"box_tests": lanes(1) / n, "sphere_tests": lanes(3) / n, "sphere_roots": lanes(4) / n,
                           "cluster_slab_tests": (lanes(32) + lanes(43) + 8.0 * lanes(45)) / n,      # every cluster box (few clusters) / group boxes + the dealt pairs' cluster boxes on the groups' grids (site 45: all EIGHT of a group)
                           "member_slab_tests": 8.0 * lanes(34) / n,      # (site 34 stands in front of the EIGHT member boxes of a cluster)
                           "rays": lanes(13) / n + lanes(9) / n, "shading_events": lanes(8) / n, "culled_traces_of_a_wave": execs(9) / n},
            "frame": f"{w}x{h}x{spp} spp, {nb} bounces (1/16 of the bench frame's pixels, same camera and samples per pixel)",
            "source": "librt_hip_stats.so (-DRT_STATS per-site lane counters, csrc/rt_stats.hip.h), this run"}



def cpu_baseline(rt, w, sky):
    """Reference CPU path on this box: render_column() on one thread per column (main.c:333,363,377), bounce
    limit patched to the workload's -- at all host cores (the headline baseline) and at one thread; plus the
    oracle port with a dynamic row scheduler (counter mode).  Bounded to roughly 30 s in all."""
    from rtlibs import Oracle, Ref, ref_available
    W, H, nb = w["width"], w["height"], w["max_bounces"]
    cores = min(os.cpu_count() or 1, 32)          # MAX_COLUMNS = 32 (main.c:46)
    while W % cores and cores > 1:                # the reference never renders W % columns pixels
        cores -= 1
    scene_path = os.path.join(rt.DATA_DIR, w["scene"])
    o = Oracle()
    o.load_scene(scene_path); o.set_skybox(sky); o.set_camera()
    if ref_available():
        ref = Ref(bounce_patch=True)
        ref.load_scene(scene_path); ref.set_skybox(sky); ref.set_bounce_limit(nb)
        run = lambda passes, threads: ref.time_columns(W, H, passes, threads)   # noqa: E731
        kind = "reference"
    else:
        run = lambda passes, threads: o.time_columns(W, H, passes, nb, threads)  # noqa: E731
        kind = "port"

    def rate(threads, budget_s):
        t = time.perf_counter(); run(1, threads); one = time.perf_counter() - t
        passes = int(max(1, min(64, budget_s / max(one, 1e-3))))
        t = time.perf_counter(); run(passes, threads); dt = time.perf_counter() - t
        return W * H * passes / dt / 1e6, passes, dt

    v_all, passes, dt = rate(cores, 10.0)
    out = {"value": round(v_all, 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
           "sample": f"{passes} full passes of {w['scene']} {W}x{H} at {nb} bounces = {W * H * passes / 1e6:.1f} Msamples, "
                     f"static column split over {cores} threads as the reference does, {dt:.1f} s",
           "host_cpus": os.cpu_count()}
    v1, p1, dt1 = rate(1, 4.0)
    out["one_thread"] = {"value": round(v1, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
                         "sample": f"{p1} full pass(es), one column = the whole frame, {dt1:.1f} s"}
    # the oracle port, counter mode, rows dealt dynamically to all host threads (the scheduler variant of SURVEY 8d)
    threads = os.cpu_count() or 1
    t = time.perf_counter()
    o.render_counter(W, H, 1, nb, seed=w["seed"], threads=threads)
    one = time.perf_counter() - t
    spp = int(max(1, min(16, 6.0 / max(one, 1e-3))))
    t = time.perf_counter()
    o.render_counter(W, H, spp, nb, seed=w["seed"], threads=threads)
    dtd = time.perf_counter() - t
    out["dynamic_rows"] = {"value": round(W * H * spp / dtd / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                           "sample": f"oracle port, counter mode, {spp} spp of the same frame, rows dealt dynamically to {threads} threads, {dtd:.1f} s"}
    return out


def launch_check(args):
    """Rendezvous of the N ranks and one all-reduce, nothing else (CPU boxes run it over gloo)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = args.backend or "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend)
        t = torch.tensor([rank + 1], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        total = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "backend": backend if world > 1 else None,
                          "rank_sum": total, "expected": world * (world + 1) // 2}), flush=True)
    return 0 if total == world * (world + 1) // 2 else 1


def traffic_from_profiles(config, compiled):
    """HBM-side traffic of the dominant kernel from the committed rocprofv3 PMC passes (bench.py cannot collect
    PMC counters itself).  profiles/pmc_latest.json is written by scripts/summarize_pmc.py from the passes of
    the current round: {config: {"kernel": ..., "fetch_bytes": ..., "write_bytes": ..., "source": ...}}."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    e = table.get(config + ("" if compiled else "_generic"))
    if not e:
        return None
    return e


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.native_multi:
        sys.exit(self_launch(args, argv))
    if args.launch_check:
        sys.exit(launch_check(args))

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ngpus = args.gpus if args.native_multi else world      # GPUs the frame is split over (one process per GPU, or one process for all)
    args.gpus = ngpus
    # The contract is ONE line on stdout: the JSON.  RCCL prints a version banner and gloo its connection messages to the
    # process's stdout (file descriptor 1, from C): everything written to descriptor 1 from here on goes to stderr, and the
    # JSON line is written to the original descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.share_gpu:
        local_rank = 0
        args.backend = args.backend or "gloo"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or (args.force_collective and args.torch_loop):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        # the collective's internal stream on its own (high-priority) hardware queue: streams of equal priority share a
        # handful of queues, and the gather of frame k must run beside render k+1, not between render k and render k+1
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if (args.backend or "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    w = WORKLOADS[args.config]
    W, H, spp, nb, seed = w["width"], w["height"], w["spp"], w["max_bounces"], w["seed"]
    synthetic = w["scene"].startswith("synthetic:")
    if synthetic:
        from rtlibs import LARGE_SCENE_CAMERA, large_scene
        scene_path = large_scene(int(w["scene"].split(":")[1]), seed=17)       # a raw Scene buffer: everything below takes either
        camera = LARGE_SCENE_CAMERA
        args.no_cpu_baseline = True       # (the reference harness loads scene FILES; the headline's baseline is C1's)
    else:
        scene_path = os.path.join(rt.DATA_DIR, w["scene"])
        camera = {}
    sky = rt.load_skybox()

    def into(o):                          # the workload's inputs into a Renderer / MultiRenderer / Oracle
        (o.set_scene if (synthetic or not hasattr(o, "load_scene")) else o.load_scene)(scene_path)
        o.set_skybox(sky); o.set_camera(**camera)
        return o
    gpu = rt.Renderer(local_rank)
    if args.every_object:
        gpu.set_tuning(test_every_object=True)
    into(gpu)
    gpu.reserve(W, H)
    # scene "compilation" (hiprtc, ~1 s, outside the timed region): same frames, fewer instructions
    compiled, jit_s, scene_kernel_info = False, None, None
    # (the synthetic scenes of 64 objects and more are measured on the culled generic kernel: faster than a compiled one from ~40 scattered objects)
    if args.kernel == rt.KERNEL_AUTO and not args.no_jit and not w["scene"].startswith("synthetic"):
        try:
            t = time.perf_counter()
            gpu.compile_scene()
            jit_s = time.perf_counter() - t
            compiled = True
            scene_kernel_info = gpu.compiled_scene_info()       # "embedded, compiled with the library by ..." / "hiprtc x.y at run time"
        except rt.RtError as e:
            print(f"[bench] scene not specialised, using the generic kernel: {e}", file=sys.stderr)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()


    # ---- the frame loop: the library's own frame queue behind the C ABI at N = 1 (rt_frame_submit / rt_frame_wait; with
    # --force-collective the N-GPU path of rt_multi_frame_* over a one-rank RCCL communicator), torch.distributed's
    # collective around rt_render_device when every GPU has its own process (N > 1) -----------------------------------
    native = world == 1 and not args.torch_loop
    multi_path = ngpus > 1 or args.force_collective
    collective = None
    if native:
        from ray_tracing_amd.frames import FrameLoop
        queue = gpu
        if args.force_collective or args.native_multi:
            if args.native_multi and args.one_device:
                queue = rt.MultiRenderer([local_rank], on_one_device=ngpus)
            else:
                queue = rt.MultiRenderer(list(range(ngpus)) if args.native_multi else [local_rank])
            queue.set_tuning(force_collective=1 if args.force_collective else 0)
            into(queue)
            if compiled:
                queue.compile_scene()
            prof = queue.context(0)
            # what the communicator itself reports -- not what this script asked for
            info = queue.collective_info()
            collective = {"host": "rt_multi (one process, ncclCommInitAll)", "ranks_seen": info["ranks"], "devices_seen": info["devices"],
                          "rccl_version": info["version"], "contexts": queue.size()}
            if args.one_device:
                collective["note"] = "testing aid: all contexts on one GPU, the gather is device copies (no communicator: ranks_seen 0)"
        else:
            prof = gpu
        # (one GPU: three frames in flight -- two are resident side by side at half the workgroup slots each, the third is queued:
        # 5.28 against 5.35 ms per C1 step over the driver's twenty steps, steady state 5.16 against 5.33; four and more only make
        # the first deliveries of a run irregular (profiles/r05/bench_depth.txt).  The N-GPU path keeps five strips rendering and the
        # gather of the one before)
        # (the deep queue is for strips: a one-rank group -- --force-collective, a testing aid -- renders whole frames)
        depth = args.depth or (rt.LAUNCH_SETS + 1 if multi_path and ngpus > 1 else w.get("frames_in_flight", 3))
        loop = FrameLoop(queue, W, H, spp, nb, depth=depth, row_block=ROW_BLOCK, kernel=args.kernel)
        primitive = ("device copies on one GPU (testing aid)" if args.one_device else "ncclGather (native, one process)") if multi_path else None

        def run_steps(first_seed, n):
            t0 = time.perf_counter()
            stamps = loop.run(range(first_seed, first_seed + n))
            return t0, stamps

        def one_frame(seed_):
            return loop.render_now(seed_)

        def last_frame():
            return loop.last
    else:
        from ray_tracing_amd.multi_gpu import TiledFrame
        tiled = TiledFrame(gpu, W, H, spp, nb, seed=seed, row_block=ROW_BLOCK, rank=rank, world=world,
                           kernel=args.kernel, device=dev, to_host=True, force_collective=args.force_collective)
        prof = gpu
        depth = tiled.depth
        primitive = tiled.primitive
        if tiled.multi:
            # what the process group itself reports: its size, and the device every rank really is on
            mine = [rank, torch.cuda.current_device(), socket.gethostname()]
            seen = [None] * dist.get_world_size()
            dist.all_gather_object(seen, mine)
            collective = {"host": "torch.distributed (one process per GPU)", "backend": str(dist.get_backend()), "ranks_seen": dist.get_world_size(),
                          "devices_seen": [d for _, d, _ in sorted(seen)],
                          "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if str(dist.get_backend()) == "nccl" else None}

        def run_steps(first_seed, n):
            t0 = time.perf_counter()
            for k in range(n):
                tiled.step(seed=first_seed + k)
            tiled.flush()                   # every frame gathered, de-interleaved and resident in host memory
            return t0, None

        def one_frame(seed_):
            return tiled.render_now(seed=seed_)

        def last_frame():
            return tiled.host_frame.numpy() if tiled.host_frame is not None else None

    # Set-up, before the W warm-up steps the contract asks for: a few frames through the whole loop, so that every stream,
    # scratch set, strip buffer and pinned destination the loop rotates through has been used once (first use allocates and
    # loads code: with --warmup 0 or 1 the timed region would pay ~20 ms for the second stream's first launch)
    SETUP_STEPS = 4               # (reported in the line as `setup_steps`: they are not part of `warmup`)
    run_steps(seed, SETUP_STEPS)
    fence()
    # step k of the run renders seed k (warm-up first): a stale or re-ordered frame cannot hide behind equal seeds
    run_steps(seed, args.warmup)
    fence()
    prof.profile(True)
    first_timed = seed + args.warmup
    # N > 1 (either host): every rank's phases of the timed frames (strip render / wait for the gather / de-interleave / copy / idle)
    phases_native = native and multi_path and queue is not gpu
    if phases_native:
        queue.profile_phases(True)
    if not native:
        tiled.record_events = True
        start = torch.cuda.Event(enable_timing=True)
        start.record(tiled.streams[tiled.k % len(tiled.streams)])
    t0, stamps = run_steps(first_timed, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    last_seed = first_timed + args.steps - 1
    # Kernel time of a launch, two ways (HIP events inside the library, all launches of the timed region).  (a) per launch:
    # from its first compute unit (behind the clearing of its counters) to the end of its trace kernel, summed -- consecutive
    # launches are enqueued on two streams and overlap on the GPU (the next frame's waves fill the compute units while this
    # frame's run out of pixels), and the ~0.1 ms two launches share is in both of them, as it is in rocprofv3's
    # per-kernel durations.  (b) the span from the first launch's first compute unit to the end of the last trace kernel,
    # over the launches in it: no double counting, but it contains idle time when the launches wait for something else (C3:
    # for the host copy of the frame before last).  Both are upper bounds of the time the GPU needs per launch.
    per_launch_ms, launches, span_ms, primary_ms_total = prof.profile_collect_split()
    prof.profile(False)
    span_launches = launches
    kernel_ms = min(per_launch_ms, span_ms) if launches and span_ms > 0 else per_launch_ms
    if native:
        marks = [t0] + stamps
        step_list = [(marks[i + 1] - marks[i]) * 1e3 for i in range(1, len(marks) - 1)]   # (the first interval holds the pipeline fill)
        first_listed = 1
    else:
        tiled.record_events = False
        marks = [start] + tiled.done_events
        step_list = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)] if len(marks) > 1 else []
        first_listed = 0
        tiled.done_events = []; tiled.render_events = []
    step_ms = sorted(step_list)
    # the slowest interval between two delivered frames, by name: timed step i delivers the frame of seed first_timed + i
    slowest = None
    if step_list:
        i_max = max(range(len(step_list)), key=lambda i: step_list[i])
        slowest = {"timed_step": i_max + first_listed, "seed": first_timed + i_max + first_listed, "ms": round(step_list[i_max], 4)}

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- where each rank's step went (outside the timed region; the events were recorded inside it).  One entry per rank / device:
    # ms per step of strip render / de-interleave / frame copy / wait for the gather / idle -- every instant of the rank's timed
    # window given to what its device was doing, summing to its step (ray_tracing_amd.attribute_phases, rt_multi_profile_collect)
    # -- then the rank with the least slack and what bounds the step there.  If a SCALE run lands at 5 x instead of 6.8 x, this says why.
    per_rank = None
    if multi_path:
        if phases_native:
            per_rank = queue.collect_phases()
            queue.profile_phases(False)
        elif not native:
            mine = tiled.phases()
            if world > 1:
                gathered_phases = [None] * world
                dist.all_gather_object(gathered_phases, mine)
                per_rank = gathered_phases
            else:
                per_rank = [mine]

    # ---- what was timed is what was asked for: the last frame of the timed region against a blocking rt_render() of its
    # seed (whole frame, bit for bit) and against rows of the CPU oracle.  Outside the timed region. -------------------
    verified = None
    if rank == 0:
        got = np.array(last_frame(), copy=True)
        check = rt.Renderer(local_rank)
        check.set_tuning(poison_frame=True)           # a pixel the check render leaves unwritten is a NaN, not whatever the fresh buffer held
        into(check)
        want = check.render(W, H, spp, nb, seed=last_seed)          # generic kernel, one GPU, blocking
        check_counts = check.last_launch_report()[1]
        check.close()
        same = bool((got.view(np.uint32) == want.view(np.uint32)).all())
        verified = {"last_frame_seed": last_seed, "equals_blocking_rt_render": same, "frame_mean": round(float(got.mean()), 6),
                    "last_launch_cancelled": bool(gpu.was_cancelled())}
        if not same:                  # where: enough to tell a stale strip from a stray pixel
            bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
            rows_bad = np.flatnonzero(bad.any(axis=1))
            # Which frame is off?  Round 3's intermittent failure differed only in pixels whose value does not depend on the seed (sky,
            # and the emitter's disc, where every sample clamps to 1): a stale frame cannot differ there, only memory a launch did not
            # write -- and every buffer of the timed loop had been filled by earlier frames, the check render's were fresh.
            verified["mismatch_diagnosis"] = {"nan_pixels_in_timed_frame": int(np.isnan(got).any(axis=2).sum()),
                                              "nan_pixels_in_check_render": int(np.isnan(want).any(axis=2).sum()),
                                              "check_render_launch": check_counts, "timed_loop_last_launch": gpu.last_launch_report()[1]}
            verified["mismatch"] = {"pixels": int(bad.sum()), "rows": int(rows_bad.size), "first_rows": [int(r) for r in rows_bad[:12]],
                                    "last_rows": [int(r) for r in rows_bad[-4:]],
                                    "pixels_in_first_row": int(bad[rows_bad[0]].sum()) if rows_bad.size else 0,
                                    # per strip (row block b belongs to strip b % world): rows that differ, and the first of them
                                    "rows_per_strip": [int(sum(1 for r in rows_bad if (r // ROW_BLOCK) % ngpus == s)) for s in range(ngpus)],
                                    "first_row_per_strip": [int(min([r for r in rows_bad if (r // ROW_BLOCK) % ngpus == s], default=-1)) for s in range(ngpus)]}
            sys.stderr.write("VERIFICATION FAILED: " + json.dumps(verified["mismatch"]) + "\n")
        try:
            from rtlibs import Oracle
            o = into(Oracle())
            rows = [H // 3, (2 * H) // 3 + 1, H - 3]       # (the last rows are the last pixels a launch gets to: a launch cut short or read early shows there)
            ref_rows = o.render_counter_rows(W, H, spp, nb, rows, seed=last_seed)
            verified["oracle_rows"] = rows
            verified["equals_oracle_rows"] = bool(all((got[r].view(np.uint32) == v.view(np.uint32)).all() for r, v in ref_rows.items()))
            if not same:              # which of the two GPU frames is the one that is off
                verified["blocking_render_equals_oracle_rows"] = bool(all((want[r].view(np.uint32) == v.view(np.uint32)).all() for r, v in ref_rows.items()))
        except Exception as e:       # the oracle is a checker that may be absent; the GPU-vs-GPU check above stands
            verified["oracle_error"] = repr(e)
        verified["ok"] = same and verified.get("equals_oracle_rows", True)
    if world > 1 and not args.leave_early:
        # a mismatch is looked at once more before anybody leaves: the same seed through the same N-rank loop, nothing overlapped --
        # a frame that is right the second time was a race in the pipeline, one that is wrong again is a wrong kernel
        again = torch.tensor([1 if (rank == 0 and not verified["ok"]) else 0], dtype=torch.int32, device=dev)
        dist.broadcast(again, src=0)
        if int(again.item()):
            second = one_frame(last_seed)
            if rank == 0:
                second = second.numpy() if hasattr(second, "numpy") else np.asarray(second)
                verified["second_render_of_that_seed_equals_blocking_rt_render"] = bool((second.view(np.uint32) == want.view(np.uint32)).all())
        dist.barrier()                # the other ranks keep their contexts until rank 0 has looked: nobody tears a process down beside the check

    # ---- the host hand-off by itself: the frame's bytes from HBM to pinned host memory, nothing else on the GPU (torch is the
    # plumbing here; the loop's own copies are hipMemcpyAsync on the library's copy stream).  The only device crossing of the
    # reference is the opposite one: glTexImage2D of the finished frame (gpu_and_windowing.c:371-376).
    host_copy = None
    if rank == 0 and not args.no_extras:
        nfl = W * H * 3
        src_t = torch.empty(nfl, dtype=torch.float32, device=dev)
        dst_t = torch.empty(nfl, dtype=torch.float32, pin_memory=True)
        cs = torch.cuda.Stream(dev, priority=-1)
        with torch.cuda.stream(cs):
            dst_t.copy_(src_t, non_blocking=True)
        cs.synchronize()
        tc = time.perf_counter()
        with torch.cuda.stream(cs):
            for _ in range(10):
                dst_t.copy_(src_t, non_blocking=True)
        cs.synchronize()
        copy_ms = (time.perf_counter() - tc) * 1e2
        host_copy = {"bytes": nfl * 4, "ms": round(copy_ms, 4), "GBps": round(nfl * 4 / copy_ms / 1e6, 2), "link_peak_GBps": 63.0,
                     "frac_of_link": round(nfl * 4 / copy_ms / 1e6 / 63.0, 3),
                     "note": "frame -> pinned host memory alone on the GPU, mean of 10; link peak = PCIe 5.0 x16, 32 GT/s x 16 lanes x 128/130"}
        del src_t, dst_t

    # ---- the same frame with nothing overlapped: first launch -> frame on the host (SURVEY.md 8d protocol)
    latency = None
    split = None        # (camera-ray pass ms, trace kernel ms) per launch with nothing else on the GPU: the two kernels priced apart (roofline.kernels)
    if not args.no_extras:
        runs = []
        for _ in range(7):
            fence()
            t1 = time.perf_counter()
            one_frame(seed)
            if world > 1:
                dist.barrier()
            runs.append((time.perf_counter() - t1) * 1e3)
        runs.sort()
        latency = runs[len(runs) // 2]
        # the two kernels apart: twelve frames ONE AT A TIME, back to back (submit, wait, submit ...: nothing overlaps, and the GPU
        # is not left idle in between -- a launch that follows a fence and a host-side pause runs its first part at the clocks of an
        # idle chip: 5.47 against 5.33 ms for C1's trace kernel).  This is the mode rocprofv3's serial block under profiles/ times
        # (bench.py --depth 1), so a reader can recompute roofline.frac from that block.
        prof.profile(2)
        for k in range(12):
            one_frame(seed + k)
        s_ms, s_n, _, s_primary = prof.profile_collect_split()
        prof.profile(False)
        if s_n:
            split = (s_primary / s_n, (s_ms - s_primary) / s_n)

    samples_per_step = W * H * spp
    # ---- the same loop with the frame LEFT ON THE DEVICE (rt_frame_submit_device): what a presenter that takes the frame from
    # device memory gets -- the reference's own hand-off goes host -> GPU (glTexImage2D, gpu_and_windowing.c:371-376)
    device_resident = None
    if native and not multi_path and not args.no_extras:
        n_dev = max(args.steps, 10)
        p_of = lambda k: rt.Renderer.params(W, H, spp, nb, seed=seed + k, row_block=ROW_BLOCK, kernel=args.kernel)   # noqa: E731
        d_dev = max(1, min(depth, rt.FRAME_SLOTS))
        for k in range(d_dev):
            gpu.frame_submit_device(p_of(k), k); gpu.frame_wait(k)
        fence()
        t1 = time.perf_counter()
        for k in range(min(d_dev - 1, n_dev)):
            gpu.frame_submit_device(p_of(k), k % d_dev)
        for k in range(n_dev):
            if k + d_dev - 1 < n_dev:
                gpu.frame_submit_device(p_of(k + d_dev - 1), (k + d_dev - 1) % d_dev)
            gpu.frame_wait(k % d_dev)
        dt_dev = (time.perf_counter() - t1) / n_dev
        device_resident = {"ms_per_step": round(dt_dev * 1e3, 4), "value": round(samples_per_step / dt_dev / 1e6, 2), "unit": "Msamples/s", "steps": n_dev,
                           "region": f"K frames through rt_frame_submit_device / rt_frame_wait, {d_dev} in flight: the frame stays in HBM, only the "
                                     "launch's control words go to the host"}
    # ---- the interactive ladder at full resolution (main.c:354-408): passes of one sample per pixel, one launch
    # each (rt_progressive_pass) and RT_PROGRESSIVE_BATCH to a launch (rt_progressive_passes: same sums, bit for bit)
    interactive = None
    if native and not multi_path and not args.no_extras and world == 1:
        n_pass = 256
        rates, enqueued = {}, {}
        for batched in (False, True):
            gpu.progressive_begin(W, H, init_scale=1, max_bounces=10, seed=seed)
            if batched: gpu.progressive_passes(8)
            else: [gpu.progressive_pass() for _ in range(8)]
            fence()
            t1 = time.perf_counter()
            if batched: gpu.progressive_passes(n_pass)
            else: [gpu.progressive_pass() for _ in range(n_pass)]
            enqueued[batched] = (time.perf_counter() - t1) / n_pass
            fence()
            rates[batched] = (time.perf_counter() - t1) / n_pass
        interactive = {"ms_per_pass": round(rates[False] * 1e3, 4), "ms_per_pass_batched": round(rates[True] * 1e3, 4),
                       "host_enqueue_ms_per_pass": round(enqueued[False] * 1e3, 4),
                       "msamples_per_s": round(W * H / rates[False] / 1e6, 1), "msamples_per_s_batched": round(W * H / rates[True] / 1e6, 1),
                       "region": f"{n_pass} passes of 1 sample per pixel at {W}x{H}, 10 bounces, after the scale ladder: one launch per pass "
                                 "(rt_progressive_pass) / all in one launch (rt_progressive_passes); host enqueue -> synchronised"}
    if rank == 0:
        value = samples_per_step * args.steps / elapsed / 1e6
        metric = "Msamples/s (rays/s) at 1920x1080x64spp scene_0; 1/2/4/8 GPU"
        if args.config != "C1":
            metric = f"Msamples/s at {W}x{H}x{spp}spp {w['scene'].replace('.txt', '')} ({args.config})"
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": ngpus, "steps": args.steps,
            "warmup": args.warmup, "setup_steps": SETUP_STEPS, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "verified": bool(verified and verified["ok"]), "verification": verified,
            "config": {"workload": f"{w['name']}: {w['scene']} {W}x{H}, {spp} spp, {nb} bounces, default camera, "
                                   f"counter-mode RNG seeds {seed}.. (one per step), shipped skybox (6x2048x2048)",
                       "timed_region": "K frames, each: strip render -> "
                                       + ("one RCCL gather to rank 0 -> de-interleave -> " if multi_path else "")
                                       + f"resolved frame copied to pinned host memory; {depth} frames in flight; step k renders seed k",
                       "frame_loop": "C ABI frame queue (rt_frame_submit / rt_frame_wait)" if native and not multi_path
                                     else "C ABI frame queue of the device group (rt_multi_frame_submit / rt_multi_frame_wait), one host process" if native
                                     else "multi_gpu.TiledFrame: rt_render_device + torch.distributed collective, one process per GPU",
                       "partition": f"interleaved blocks of {ROW_BLOCK} rows over {ngpus} " + ("rank(s) SHARING ONE GPU (testing aid)" if (args.share_gpu or args.one_device) else "GPU(s)")
                                    + (f"; collective: {primitive}" if multi_path else "")
                                    + (" (one-rank group: testing aid)" if args.force_collective and ngpus == 1 else ""),
                       "kernel": {0: "wavefront" + (f"+scene-specialised ({scene_kernel_info})" if compiled else ""), 1: "simple",
                                  2: "wavefront, plain IEEE ops"}.get(args.kernel, str(args.kernel))},
        }
        if collective is not None:
            out["collective"] = collective
        if per_rank:
            out["per_rank"] = [{"rank": i, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}} for i, r in enumerate(per_rank)]
            out.update(rt.judge_phases(per_rank))
            out["per_rank_note"] = ("ms per step, per rank: every instant between the rank's first and last frame end (rank 0: frame in host memory; others: their "
                                    "gather done) belongs to what its device was doing then, for whichever frame in flight: strip render (incl. waiting for "
                                    "workgroup slots) > de-interleave > copy to the host > a rendered strip waiting for the gather > idle (no launch there to run); "
                                    "the five are disjoint and sum to step_ms.  critical_rank: least slack (gather wait + idle); step_bound: its largest share, "
                                    "in {render, gather, copy, host}")
        if step_ms:
            med = step_ms[len(step_ms) // 2]
            out["ms_per_step_median"] = round(med, 4)
            out["ms_per_step_p90_max"] = [round(step_ms[(len(step_ms) * 9) // 10], 4), round(step_ms[-1], 4)]
            out["slowest_step"] = slowest
            out["value_at_median_step"] = round(samples_per_step / med / 1e3, 2)
            # the loop once it is full: the intervals between delivered frames from the `depth`-th timed frame on (the first frames of
            # the timed region are delivered at the pipeline's fill rate, whatever the depth), their median -- what a host that runs for
            # seconds sees per frame; `ms_per_step` above stays the mean over ALL K steps, which is what `value` is computed from
            steady = sorted(step_list[min(depth, max(len(step_list) - 3, 0)):])
            if steady:
                out["steady_state_ms_per_step"] = round(steady[len(steady) // 2], 4)
                out["steady_state"] = {"intervals": len(steady), "skipped_after_fill": len(step_list) - len(steady), "min_ms": round(steady[0], 4), "max_ms": round(steady[-1], 4),
                                       "value": round(samples_per_step / steady[len(steady) // 2] / 1e3, 2), "frames_in_flight": depth}
        if latency is not None:
            out["frame_latency"] = {"median_ms": round(latency, 4), "runs": 7,
                                    "msamples_per_s": round(samples_per_step / latency / 1e3, 2),
                                    "region": "first launch -> gathered, resolved frame resident in host memory, nothing overlapped"}
        if jit_s is not None:
            out["jit_compile_s"] = round(jit_s, 3)
            out["config"]["scene_kernel"] = gpu.compiled_scene_info() if gpu.scene_is_compiled() else scene_kernel_info
        if host_copy is not None:
            out["host_copy"] = host_copy
            if step_ms and abs(step_ms[len(step_ms) // 2] - host_copy["ms"]) <= 0.1 * host_copy["ms"]:
                # the median step IS the copy of the frame to the host: the line's rate is the link's, not a kernel's
                out["step_bound"] = "pcie"
                out["step_bound_note"] = (f"median step {step_ms[len(step_ms) // 2]:.3f} ms = the {host_copy['bytes'] / 1e6:.1f} MB frame at "
                                          f"{host_copy['GBps']} GB/s, {host_copy['frac_of_link']} of a PCIe 5.0 x16 link; see device_resident for the step without it")
        if device_resident is not None:
            out["device_resident"] = device_resident
        if interactive is not None:
            out["interactive"] = interactive
        # ---- roofline (rank 0's launches; every rank runs the same kernels on 1/N of the rows).  Two kernels, bound differently, are
        # priced apart: rt_primary_pass (camera rays once per pixel; sky pixels finished: texel gather + frame write -> HBM side) and
        # the trace kernel (VALU issue).  The top-level fields describe the kernel that takes more of the launch.
        from rtlibs import Oracle
        try:
            oc = into(Oracle(counters=True))
            # (a frame of 1024 objects costs the CPU 1024 tests per ray: the per-sample figures of the L* workloads come from a quarter-size frame)
            work = algorithmic_work(oc, W // 4 if synthetic else W, H // 4 if synthetic else H, nb, seed)
        except Exception as e:   # the oracle is a checker; the bench line must still print
            work = None
            out["roofline_error"] = repr(e)
        if work and launches:
            avg_ms = kernel_ms / launches
            span_launches = launches
            # the two kernels apart: only from launches that had the GPU to themselves (the latency leg) -- in the timed region the next
            # launch's camera-ray pass waits for workgroup slots of the previous launch's trace kernel, and the event between the two
            # kernels would book that wait to the pass
            primary_ms, trace_ms = split if split else (primary_ms_total / launches, max(avg_ms - primary_ms_total / launches, 1e-6))
            split_from = "twelve launches one at a time, back to back (the mode of rocprofv3's serial block under profiles/)" if split else "the timed region (launches overlap: the split is approximate)"
            samples_per_launch = samples_per_step / ngpus
            pixels_per_launch = samples_per_launch / spp
            trace_name = "rt_trace_spec" if compiled else ("rt_trace_wavefront (culled)" if synthetic and not args.every_object else "rt_trace_wavefront")
            out["rays_per_s"] = round(value * 1e6 * work["rays"], 1)
            tr = traffic_from_profiles(args.config, compiled) if ngpus == 1 else None
            pk = (tr or {}).get("kernels", {})

            # -- the trace kernel against the fp32 VALU issue peak without FMA
            flops_as_written = work["flops"] * samples_per_launch
            # (the samples whose camera ray leaves the scene never reach the trace kernel: the camera-ray pass finishes their pixels)
            flops_object_samples = (work["flops"] - work["sky_sample_flops"]) * samples_per_launch
            trace = {"kernel": trace_name, "bound": "valu", "ms": round(trace_ms, 4), "peak": round(PEAK_VALU_NOFMA_TFLOPS, 2), "unit": "TFLOP/s"}
            executed = None
            if synthetic and not args.every_object:
                try:
                    executed = executed_work_of_culled_scene(rt, scene_path, camera, sky, W, H, spp, nb, seed)
                except Exception as e:
                    out["executed_work_error"] = repr(e)
            culled_launch = bool(executed)
            if executed:
                # A culled scene: the numerator is the work the kernel EXECUTES, priced with the reference's cost table (the
                # oracle's per-test averages: a box test 21 flops incl. its far corner, a sphere test ~38), so that frac <= 1 by
                # construction; the every-object count the reference would do is printed beside it.
                x = executed["per_sample"]
                box_cost = work["box_flops"] / max(work["box_tests"], 1e-9)
                sphere_cost = work["sphere_flops"] / max(work["sphere_tests"], 1e-9)
                other = work["flops"] - work["box_flops"] - work["sphere_flops"]          # shading, draws, normalisations, sky: executed as written
                ex_flops = other + x["box_tests"] * box_cost + x["sphere_tests"] * sphere_cost + 6.0 * (x["cluster_slab_tests"] + x["member_slab_tests"])
                trace.update({"achieved": round(ex_flops * samples_per_launch / (trace_ms * 1e-3) / 1e12, 3), "numerator": "executed",
                              "executed_flops_per_sample": round(ex_flops, 1),
                              "executed": {k: round(v, 3) for k, v in x.items()}, "executed_frame": executed["frame"], "executed_source": executed["source"],
                              "cost_table": {"box_test": round(box_cost, 2), "sphere_test": round(sphere_cost, 2), "conservative_slab_test": 6.0,
                                             "everything_else_as_written": round(other, 1)},
                              "algorithmic_as_written": {"flops_per_sample": round(work["flops"], 1), "object_tests_per_sample": round(work["object_tests"], 2),
                                                         "TFLOPs_if_executed": round(flops_as_written / (trace_ms * 1e-3) / 1e12, 3),
                                                         "note": "what scene.c:163-173 does: every ray tests every object; the kernel does not execute it"}})
            else:
                trace.update({"achieved": round(flops_object_samples / (trace_ms * 1e-3) / 1e12, 3),
                              "numerator": "as written, the samples of object pixels (sky pixels are finished by rt_primary_pass and never reach this kernel)",
                              "flops_per_object_sample": round((work["flops"] - work["sky_sample_flops"]) / max(1.0 - work["sky_samples"], 1e-9), 1)})
            if executed:       # (the instrumented build counts both kernels' work: the culled scene's executed flops are priced against the launch below)
                trace.pop("achieved", None)
            if "achieved" in trace:
                trace["frac"] = round(trace["achieved"] / PEAK_VALU_NOFMA_TFLOPS, 4)
            iss = simd_issue_estimate(pk.get("rt_trace"), trace_ms)
            if iss:
                trace["issue"] = iss

            # -- the camera-ray pass: one trace_ray per PIXEL (and, for the pixels whose camera ray leaves the scene, the rest of the
            # sample: texel, colour -- added spp times onto the sum).  It is a VALU kernel like the trace kernel (round 5 labelled it
            # "hbm" at 2-11 % of the HBM peak with the VALUs 70 % busy: wrong label).  Numerator: its own flops as written -- per object
            # pixel the camera ray and the first trace_ray, per sky pixel the whole sample (oracle counters first_ray_flops,
            # sky_sample_flops; the frame at 1 spp, so per sample = per pixel).  The bytes it moves are side keys.
            sky_px = work["sky_samples"] * pixels_per_launch
            obj_px = pixels_per_launch - sky_px
            p_bytes = sky_px * (4.0 + 12.0) + obj_px * 48.0
            p_flops = (work["first_ray_flops"] + work["sky_sample_flops"]) * pixels_per_launch
            primary = {"kernel": "rt_primary_pass", "bound": "valu", "ms": round(primary_ms, 4), "peak": round(PEAK_VALU_NOFMA_TFLOPS, 2), "unit": "TFLOP/s",
                       "achieved": round(p_flops / max(primary_ms, 1e-6) / 1e9, 3), "frac": round(p_flops / max(primary_ms, 1e-6) / 1e9 / PEAK_VALU_NOFMA_TFLOPS, 4),
                       "numerator": "as written, once per PIXEL: camera ray + first trace_ray of object pixels, the whole sample of sky pixels",
                       "flops_per_pixel": round(work["first_ray_flops"] + work["sky_sample_flops"], 1),
                       "bytes": {"algorithmic_bytes": round(p_bytes), "algorithmic_GBps": round(p_bytes / max(primary_ms, 1e-6) / 1e6, 2), "hbm_peak_GBps": PEAK_HBM_GBPS,
                                 "note": "4 B of texel + 12 B of frame per sky pixel, 48 B of record per object pixel"}}
            pp = pk.get("rt_primary_pass")
            if pp and "fetch_bytes" in pp and "write_bytes" in pp:
                moved = pp["fetch_bytes"] + pp["write_bytes"]
                primary["bytes"].update({"traffic": round(moved), "traffic_GBps": round(moved / max(primary_ms, 1e-6) / 1e6, 2),
                                         "frac_of_hbm_peak": round(moved / max(primary_ms, 1e-6) / 1e6 / PEAK_HBM_GBPS, 4),
                                         "traffic_from": "PMC traffic of the committed passes / this run's kernel time"})
            if culled_launch:     # (a culled scene's camera rays are culled too: their executed work is in the launch-level figure below)
                for key in ("achieved", "frac"):
                    primary.pop(key)
                primary["numerator"] = "executed work of both kernels is priced at launch level (roofline.frac); this entry has the time only"
            if primary_ms <= 0:   # (--no-extras: no launch had the GPU to itself, the event between the two kernels was not recorded)
                for key in ("achieved", "frac"):
                    primary.pop(key, None)
                primary["numerator"] = "not timed apart in this run (--no-extras)"
            pi = simd_issue_estimate(pp, primary_ms) if primary_ms > 0 else None
            if pi:
                primary["issue"] = pi

            # The line's own roofline fields: those of the DOMINANT KERNEL -- the one that takes more of the launch -- on the work that
            # kernel itself does, over its own time (launches that had the GPU to themselves).  Until round 5 the line priced the whole
            # launch -- both kernels' time against all the flops the reference would spend on the frame --, which credits the samples of
            # sky pixels (C1: 36 %, C2: 83 %) to a trace kernel that never sees them: C2 printed 0.47 for a kernel at 0.27.  That figure
            # stays, as frac_launch_as_written.  (A culled scene's numerator is EXECUTED work, counted for both kernels together: its
            # top-level fields are the launch's.)
            num = (trace["executed_flops_per_sample"] if executed else work["flops"]) * samples_per_launch
            launch_ach = num / (avg_ms * 1e-3) / 1e12
            if executed:
                dominant = {"bound": "valu", "achieved": round(launch_ach, 3), "peak": round(PEAK_VALU_NOFMA_TFLOPS, 2), "unit": "TFLOP/s",
                            "frac": round(launch_ach / PEAK_VALU_NOFMA_TFLOPS, 4), "kernel": "rt_primary_pass + " + trace_name,
                            "numerator": "executed, both kernels over the launch's time (kernels.trace has the counts and the cost table; algorithmic_as_written beside it)"}
            else:
                dominant = dict(trace if trace_ms >= primary_ms else primary)
            out["roofline"] = {"bound": dominant["bound"], "achieved": dominant["achieved"], "peak": dominant["peak"], "unit": dominant["unit"], "frac": dominant["frac"],
                               "kernel": dominant["kernel"], "kernel_ms": dominant.get("ms", round(avg_ms, 4)), "numerator": dominant.get("numerator"), "traffic": None,
                               # (where nearly every sample is a sky sample -- C3: 97 % -- the reference's flops per SAMPLE are work the kernels do once
                               # per PIXEL: a launch-level "fraction" would read 2.5 there, and rounds 3-5 did not print one either)
                               "frac_launch_as_written": round(launch_ach / PEAK_VALU_NOFMA_TFLOPS, 4) if work["rays"] >= 1.25 or executed else None,
                               "achieved_launch_as_written": round(launch_ach, 3) if work["rays"] >= 1.25 or executed else None,
                               "launch_as_written_note": "rounds 1-5's top-level figure: " + ("executed" if executed else "all the flops the reference spends on the frame, sky samples included,")
                                                         + " over both kernels' time in the timed region (avg_kernel_ms); null where sky samples are nearly all of the frame",
                               "avg_kernel_ms": round(avg_ms, 4), "launches": launches,
                               "avg_kernel_ms_per_launch_events": round(per_launch_ms / launches, 4),
                               "avg_kernel_ms_span": round(span_ms / span_launches, 4) if span_launches else None,
                               "kernels": {"rt_primary_pass": primary, "trace": trace, "times_from": split_from},
                               "flops_per_sample": round(work["flops"], 1), "rays_per_sample": round(work["rays"], 3),
                               "object_tests_per_sample": round(work["object_tests"], 2), "rng_draws_per_sample": round(work["rng_draws"], 2),
                               "sky_samples_share": round(work["sky_samples"], 4),
                               "note": "two kernels per launch, priced apart (`kernels`); the top-level fields are those of the one that takes more of the "
                                       "launch, on ITS OWN work and time.  peak (valu) = fp32 VALU issue rate without FMA (parity forbids contraction) = 157.3/2 TFLOP/s, flops counted "
                                       "as written in the reference (SURVEY.md 8d) unless `numerator` says executed.  Kernel times: HIP "
                                       "events inside the library over the timed region -- first compute unit -> event between the kernels -> end of the "
                                       "trace kernel (which contains the in-order sample sum); avg_kernel_ms = the smaller of the per-launch sum (overlapping "
                                       "launches share time) and the span / launches"}
            if tr:
                out["roofline"]["traffic"] = round(tr["fetch_bytes"] + tr["write_bytes"])
                out["roofline"]["traffic_note"] = tr.get("source", "")
                # the bytes the trace kernel HAS to move: a 4-byte texel per sample that leaves the scene (3 B of it used) and 12 B per pixel
                t_bytes = 3.0 * work["sky_fetches"] * samples_per_launch + 12.0 * obj_px
                out["roofline"]["traffic_over_algorithmic"] = round((tr["fetch_bytes"] + tr["write_bytes"]) / max(t_bytes + p_bytes, 1.0), 2)
        # ---- the generic kernel (no hiprtc): same frames, same events
        if native and not multi_path and compiled and not args.no_extras:
            gpu.set_scene(scene_path)                 # drops the compiled kernel
            one_frame(seed)
            gpu.profile(True)
            run_steps(seed, 5)
            g_ms, g_n, g_span = gpu.profile_collect_span()
            gpu.profile(False)
            if g_n:
                out.setdefault("roofline", {})["generic_kernel_ms"] = round(min(g_ms, g_span) / g_n, 4)
        if ngpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(rt, w, sky)
                out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            except Exception as e:
                out["cpu_baseline_error"] = repr(e)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if native:
        loop.close()
        if queue is not gpu:
            queue.close()
    gpu.close()
    if world > 1 or (args.force_collective and not native):
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not (verified and verified["ok"]):
        print(f"[bench] VERIFICATION FAILED: {verified}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
