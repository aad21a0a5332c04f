"""The first frames of a run through the frame queue (depth 2), to be run under rocprofv3 --kernel-trace: how the first two launches
-- submitted back to back into an idle GPU -- share it.  kernel_timeline.py prints the kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd.frames import FrameLoop
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
loop = FrameLoop(g, 1920, 1080, 64, 4, depth=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
loop.run(range(6)); torch.cuda.synchronize()
time.sleep(0.05)
loop.run(range(100, 106)); torch.cuda.synchronize()
loop.close(); g.close()
