"""Multi-GPU frame partition: interleaved row blocks + ONE gather (RCCL over xGMI on GPUs).

The reference parallelises by giving each CPU thread one contiguous image column (main.c:333,363),
which is badly load-imbalanced (sky columns finish long before object columns).  Here block b of
`row_block` rows goes to rank b % world, so every rank sees the same mix of sky and geometry; each
rank renders its blocks into a compact strip (kernel side: rt_device.h `rt_launch`), rank 0 receives
all strips with a single gather and de-interleaves them.  One process per GPU, `torch.distributed`
supplies the communicator (backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests).

Stream discipline (GPU): every stage of a frame -- strip render, gather, de-interleave -- is issued from ONE
stream (its handle is what the C ABI receives, so the kernels, torch's collective hand-off events and the
de-interleave are ordered by construction); the copy of the finished frame to pinned host memory runs on a
copy stream behind an event.  Consecutive frames alternate between the renderer's two streams (rt_stream: different
priorities, hence different hardware queues) and are double-buffered, so the waves of frame k+1 fill the compute units
as the waves of frame k run out of pixels, and the gather / copy of frame k overlaps the render of frame k+1 -- the
way the reference's workers keep accumulating while its main thread presents (main.c:354-408 vs 450-482).
"""
import numpy as np
import torch
import torch.distributed as dist


def strip_rows(height, row_block, world):
    """Rows in one rank's strip, padded so that all ranks match (== rt_strip_rows in the C ABI)."""
    blocks = -(-height // row_block)
    return -(-blocks // world) * row_block


def owned_rows(height, row_block, rank, world):
    """Global row index held by each strip row of `rank` (-1 = padding), in strip order."""
    n = strip_rows(height, row_block, world)
    lr = np.arange(n)
    g = ((lr // row_block) * world + rank) * row_block + lr % row_block
    return np.where(g < height, g, -1)


def frame_index(height, row_block, world):
    """For every frame row: position in the gathered [world * strip_rows] row array."""
    n = strip_rows(height, row_block, world)
    j = np.arange(height)
    blk = j // row_block
    return (blk % world) * n + (blk // world) * row_block + j % row_block


def collective_for(backend=None):
    """Which primitive moves the strips, decided ONCE from the backend name (never by catching errors in the
    data path: a rank that fails must fail, not drift into a different collective than its peers).
    nccl (= RCCL) and gloo both implement gather; anything else gets one all-gather (7x the bytes)."""
    backend = backend or dist.get_backend()
    return "gather" if str(backend).lower() in ("nccl", "gloo") else "all_gather"


def gather_strips(strip, rank, world, dst=0, out=None, primitive=None, async_op=False):
    """One collective of equally-sized strips to `dst`.  Returns (result, work): result is [world, rows, W, 3] on
    dst and None elsewhere; work is the torch Work handle when async_op is set (wait() on it before touching
    the result), else None.  `out` must be passed on dst when the call is asynchronous or repeated."""
    if world == 1:
        return strip.unsqueeze(0), None
    primitive = primitive or collective_for()
    if primitive == "gather":
        if rank == dst:
            if out is None:
                out = torch.empty((world,) + tuple(strip.shape), dtype=strip.dtype, device=strip.device)
            work = dist.gather(strip, list(out.unbind(0)), dst=dst, async_op=async_op)
            return out, work
        work = dist.gather(strip, None, dst=dst, async_op=async_op)
        return None, work
    if out is None:
        out = torch.empty((world,) + tuple(strip.shape), dtype=strip.dtype, device=strip.device)
    work = dist.all_gather_into_tensor(out, strip, async_op=async_op)
    return (out if rank == dst else None), work


def assemble(strips, height, row_block, world, renderer=None, out=None, stream=None):
    """De-interleave gathered strips into the frame [height, W, 3].  On a GPU tensor this is the
    library's rt_deinterleave kernel, enqueued on `stream` (a torch stream; default: the current one);
    on CPU tensors (gloo tests) an index_select."""
    W = strips.shape[2]
    if strips.is_cuda:
        if renderer is None:
            raise ValueError("assemble: a Renderer is required for device tensors")
        if out is None:
            out = torch.empty((height, W, 3), dtype=torch.float32, device=strips.device)
        s = stream if stream is not None else torch.cuda.current_stream(strips.device)
        renderer.deinterleave_device(strips.data_ptr(), out.data_ptr(), W, height, row_block, world, s.cuda_stream)
        return out
    idx = torch.from_numpy(frame_index(height, row_block, world))
    flat = strips.reshape(-1, W, 3)
    return flat.index_select(0, idx)


class TiledFrame:
    """The N-GPU frame loop of bench.py: render own strip -> gather -> de-interleave on rank 0 -> frame in
    pinned host memory on rank 0 (what update_frame() hands to the presenter, main.c:467-479).

    step() enqueues one frame and returns at once; up to two frames are in flight.  flush() completes
    everything and leaves the last frame in `host_frame` (rank 0).  `seed` may be changed between steps
    (`step(seed=...)`): every frame is rendered from scratch, nothing is reused across frames.
    """

    def __init__(self, renderer, width, height, spp, max_bounces, seed=0, row_block=8, rank=0, world=1,
                 kernel=0, device=None, to_host=True, overlap_frames=True):
        self.r, self.W, self.H = renderer, width, height
        self.row_block, self.rank, self.world = row_block, rank, world
        self.spp, self.max_bounces, self.kernel, self.seed = spp, max_bounces, kernel, seed
        self.to_host = to_host and rank == 0
        self.device = device
        rows = strip_rows(height, row_block, world)
        # render, collective hand-off, de-interleave: frame k on streams[k & 1].  The library's own two streams, wrapped: they
        # have different priorities, so they never share a hardware queue and consecutive frames overlap on the GPU
        # (two torch streams of equal priority were seen to share one: then nothing overlaps)
        if overlap_frames:
            self.streams = [torch.cuda.ExternalStream(renderer.stream(w), device=device) for w in (0, 1)]
        else:
            self.streams = [torch.cuda.Stream(device)] * 2
        self.stream = self.streams[0]
        # frame -> pinned host memory.  High priority: its own hardware queue (streams of equal priority share a handful of
        # queues, and two streams on one queue run in enqueue order: the copy of frame k would then sit between render k and
        # render k+1 instead of beside the latter), and its copy kernel is dispatched ahead of the next frame's kernels.
        self.copy_stream = torch.cuda.Stream(device, priority=-1)
        assert all(s.cuda_stream != 0 for s in self.streams)
        self.primitive = collective_for() if world > 1 else None
        with torch.cuda.stream(self.stream):
            self.strip = [torch.empty((rows, width, 3), dtype=torch.float32, device=device) for _ in range(2)]
            self.strips = self.frame = None
            if world > 1 and (rank == 0 or self.primitive == "all_gather"):
                self.strips = [torch.empty((world, rows, width, 3), dtype=torch.float32, device=device) for _ in range(2)]
            if world > 1 and rank == 0:
                self.frame = [torch.empty((height, width, 3), dtype=torch.float32, device=device) for _ in range(2)]
        self.host_frame = torch.empty((height, width, 3), dtype=torch.float32, pin_memory=True) if self.to_host else None
        self.copied = [None, None]        # event: the host copy that read buffer k has finished
        self.pending = None               # (work, k) of the frame whose gather is in flight
        self.k = 0
        self.done_events = []             # one per completed frame when record_events is set
        self.render_events = []           # one behind every strip render when record_events is set
        self.record_events = False

    # -- one frame ---------------------------------------------------------------------------------
    def step(self, seed=None):
        k = self.k & 1
        self.k += 1
        s = self.streams[k]
        with torch.cuda.stream(s):
            if self.copied[k] is not None and self.world == 1:
                s.wait_event(self.copied[k])           # the copy two frames ago still reads strip[k]
            p = self.r.params(self.W, self.H, self.spp, self.max_bounces, seed=self.seed if seed is None else seed,
                              row_block=self.row_block, rank=self.rank, world=self.world, kernel=self.kernel)
            self.r.render_device(p, self.strip[k].data_ptr(), s.cuda_stream)
            if self.record_events:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(s)
                self.render_events.append(ev)
            if self.world == 1:
                self._deliver(self.strip[k][:self.H], k)
                return
            _, work = gather_strips(self.strip[k], self.rank, self.world, dst=0,
                                    out=self.strips[k] if self.strips is not None else None,
                                    primitive=self.primitive, async_op=True)
        prev, self.pending = self.pending, (work, k)
        if prev is not None:
            self._finish(prev)

    def _finish(self, pending):
        work, k = pending
        s = self.streams[k]                                    # the frame's own stream
        with torch.cuda.stream(s):
            work.wait()                                        # s waits for the collective (no host block on nccl)
            if self.rank == 0:
                if self.copied[k] is not None:
                    s.wait_event(self.copied[k])               # frame[k] is still being copied out
                frame = assemble(self.strips[k], self.H, self.row_block, self.world, renderer=self.r,
                                 out=self.frame[k], stream=s)
                self._deliver(frame, k)

    def _deliver(self, frame, k):
        """frame (device, on the frame's stream) -> pinned host memory, on the copy stream."""
        if self.to_host:
            ready = torch.cuda.Event()
            ready.record(self.streams[k])
            self.copy_stream.wait_event(ready)
            with torch.cuda.stream(self.copy_stream):
                self.host_frame.copy_(frame, non_blocking=True)
                done = torch.cuda.Event(enable_timing=self.record_events)
                done.record(self.copy_stream)
            self.copied[k] = done
        else:
            done = torch.cuda.Event(enable_timing=self.record_events)
            done.record(self.streams[k])
        if self.record_events:
            self.done_events.append(done)

    def flush(self):
        """Complete every frame in flight; afterwards host_frame (rank 0) holds the last one."""
        prev, self.pending = self.pending, None
        if prev is not None:
            self._finish(prev)
        for s in self.streams:
            s.synchronize()
        self.copy_stream.synchronize()

    def render_now(self, seed=None):
        """One frame, start to finish (no overlap): returns the host frame on rank 0 (a view of host_frame)."""
        self.flush()
        self.step(seed=seed)
        self.flush()
        return self.host_frame
