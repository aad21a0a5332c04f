"""Sixty single interactive passes at 1080p, to be run under `rocprofv3 --kernel-trace --output-format csv` (kernel_timeline.py
prints the last forty kernels with their queues): how the passes of the three render streams and the publish steps interleave.
usage: progressive_kernel_trace.py [workgroups per CU]"""
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ray_tracing_amd as rt
g = rt.Renderer(0)
g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_0.txt"); g.compile_scene()
if len(sys.argv) > 1 and int(sys.argv[1]): g.set_tuning(workgroups_per_cu=int(sys.argv[1]))
g.progressive_begin(1920, 1080, init_scale=1, max_bounces=10, seed=1)
for _ in range(60): g.progressive_pass()
g.synchronize()
g.close()
