"""Does RCCL's gather kernel find a workgroup slot beside five small persistent launches at one slot per CU each?  One rank, a
one-rank RCCL group (the gather is then a copy done by RCCL's kernel), frames the size of the strip one of eight ranks renders of
a C1 frame (1920 x 136, 64 spp): TiledFrame with the collective forced against TiledFrame without it, ms per step.
usage: rccl_beside_strips.py [rows [workgroups per CU]]"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import ray_tracing_amd as rt
from ray_tracing_amd.multi_gpu import TiledFrame
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 136
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.set_camera(); g.compile_scene()
if wg: g.set_tuning(workgroups_per_cu=wg)
for force in (False, True, False, True):
    t = TiledFrame(g, 1920, rows, 64, 4, rank=0, world=1, device=dev, force_collective=force)
    for k in range(20): t.step(seed=k)
    t.flush(); torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for k in range(n): t.step(seed=100 + k)
    t.flush(); torch.cuda.synchronize()
    print(f"workgroups per CU {wg or 'auto'}: 1920x{rows}, {'one-rank RCCL gather + de-interleave per frame' if force else 'no collective'} ({t.depth} strip buffers): {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step", flush=True)
g.close()
dist.destroy_process_group()
