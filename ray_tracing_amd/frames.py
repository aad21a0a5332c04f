"""The presenter's loop over the C ABI's frame queue (rt_frame_submit / rt_frame_wait, include/rt_hip.h): `depth` frames
in flight, frame k handed on while frames k+1 ... render -- what the reference's main thread does beside its workers
(update_frame(), main.c:450-482, vs worker(), main.c:354-408).  Pure ctypes plumbing for bench.py and the tests: the
pipelining itself (streams, events, buffers) is inside librt_hip.so."""
import time

from . import HostFrame, Renderer


class FrameLoop:
    def __init__(self, queue, width, height, spp, max_bounces, depth=2, row_block=8, kernel=0):
        """queue: a Renderer (one GPU) or a MultiRenderer (several GPUs of this process, native RCCL gather)."""
        self.q, self.W, self.H, self.spp, self.nb = queue, width, height, spp, max_bounces
        self.depth, self.row_block, self.kernel = depth, row_block, kernel
        self.host = [HostFrame(width, height) for _ in range(depth)]      # page-locked: the copies run beside the renders
        self.last = None            # view of the most recently delivered frame (valid until `depth` more frames are submitted)
        self.cancelled = 0

    def _params(self, seed):
        return Renderer.params(self.W, self.H, self.spp, self.nb, seed=seed, row_block=self.row_block, kernel=self.kernel)

    def run(self, seeds, on_frame=None):
        """Render one frame per seed.  Returns the host time (perf_counter) at which each frame was in host memory.
        on_frame(k, array) is called with frame k while the following frames render (the presenter's slot)."""
        seeds = list(seeds)
        n, d = len(seeds), self.depth
        stamps = []
        for k in range(min(d - 1, n)):
            self.q.frame_submit(self._params(seeds[k]), k % d, self.host[k % d])
        for k in range(n):
            if k + d - 1 < n:
                j = k + d - 1
                self.q.frame_submit(self._params(seeds[j]), j % d, self.host[j % d])
            if not self.q.frame_wait(k % d):
                self.cancelled += 1
            stamps.append(time.perf_counter())
            self.last = self.host[k % d].array
            if on_frame is not None:
                on_frame(k, self.last)
        return stamps

    def render_now(self, seed):
        """One frame, nothing overlapped: first launch -> frame in host memory."""
        self.q.frame_submit(self._params(seed), 0, self.host[0])
        self.q.frame_wait(0)
        self.last = self.host[0].array
        return self.last

    def close(self):
        for h in self.host:
            h.free()
        self.host = []
