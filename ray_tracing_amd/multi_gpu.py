"""Multi-GPU frame partition: interleaved row blocks + ONE gather (RCCL over xGMI on GPUs).

The reference parallelises by giving each CPU thread one contiguous image column (main.c:333,363),
which is badly load-imbalanced (sky columns finish long before object columns).  Here block b of
`row_block` rows goes to strip b % world, so every strip has the same mix of sky and geometry; each
rank renders one strip (rank r strip r - 1, the root the last one: strip_of_rank) into a compact buffer (kernel side: rt_device.h `rt_launch`), rank 0 receives
all strips with a single gather and de-interleaves them.  One process per GPU, `torch.distributed`
supplies the communicator (backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests).

Stream discipline (GPU): a frame's strip render and the hand-off to its gather are issued from ONE stream (its handle
is what the C ABI receives, so the kernels and torch's collective hand-off event are ordered by construction); what
follows the gather -- the wait for it, the de-interleave -- runs on a second stream and the copy of the finished frame
to pinned host memory on a third, each behind events (class TiledFrame).  Consecutive frames alternate between the
renderer's two streams (rt_stream: different priorities, hence different hardware queues) and rotate through three
strip buffers, so the waves of frame k+1 fill the compute units as the waves of frame k run out of pixels, and the
gather / de-interleave / copy of frame k overlap the render of frames k+1 and k+2 -- the way the reference's workers
keep accumulating while its main thread presents (main.c:354-408 vs 450-482).
"""
import numpy as np
import torch
import torch.distributed as dist

LAUNCH_SETS = 5       # RT_LAUNCH_SETS: the scratch sets and streams a context rotates its launches through


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


def strip_of_rank(rank, world):
    """The strip rank `rank` renders (== rt_strip_of_rank in the C ABI): handed out rotated by one, so that the root --
    which also gathers, de-interleaves and copies the frame out -- has the last strip, which is never the longest
    (1080 rows in blocks of 8 over 8 ranks: 16 blocks instead of 17)."""
    return (rank + world - 1) % world if world > 1 else 0


def frame_index(height, row_block, world, first=0):
    """For every frame row: position in the gathered [world * strip_rows] row array; strip s sits at position
    (s + first) % world (first = 1 when the strips were handed out by strip_of_rank)."""
    n = strip_rows(height, row_block, world)
    j = np.arange(height)
    blk = j // row_block
    return ((blk % world + first) % world) * n + (blk // world) * row_block + j % row_block


def collective_for(backend=None):
    """Which primitive moves the strips, decided ONCE from the backend name (never by catching errors in the
    data path: a rank that fails must fail, not drift into a different collective than its peers).
    nccl (= RCCL) and gloo both implement gather; anything else gets one all-gather (7x the bytes)."""
    backend = backend or dist.get_backend()
    return "gather" if str(backend).lower() in ("nccl", "gloo") else "all_gather"


def gather_strips(strip, rank, world, dst=0, out=None, primitive=None, async_op=False, force=False):
    """One collective of equally-sized strips to `dst`.  Returns (result, work): result is [world, rows, W, 3] on
    dst and None elsewhere; work is the torch Work handle when async_op is set (wait() on it before touching
    the result), else None.  `out` must be passed on dst when the call is asynchronous or repeated.  force: run the
    collective on a one-rank group too (testing aid: the real backend's stream semantics on a 1-GPU box)."""
    if world == 1 and not force:
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


def assemble(strips, height, row_block, world, renderer=None, out=None, stream=None, first=0):
    """De-interleave gathered strips into the frame [height, W, 3].  On a GPU tensor this is the
    library's rt_deinterleave kernel, enqueued on `stream` (a torch stream; default: the current one);
    on CPU tensors (gloo tests) an index_select.  first: see frame_index."""
    W = strips.shape[2]
    if strips.is_cuda:
        if renderer is None:
            raise ValueError("assemble: a Renderer is required for device tensors")
        if out is None:
            out = torch.empty((height, W, 3), dtype=torch.float32, device=strips.device)
        s = stream if stream is not None else torch.cuda.current_stream(strips.device)
        renderer.deinterleave_device(strips.data_ptr(), out.data_ptr(), W, height, row_block, world, s.cuda_stream, first=first)
        return out
    idx = torch.from_numpy(frame_index(height, row_block, world, first))
    flat = strips.reshape(-1, W, 3)
    return flat.index_select(0, idx)


class TiledFrame:
    """The N-GPU frame loop of bench.py: render own strip -> gather -> de-interleave on rank 0 -> frame in
    pinned host memory on rank 0 (what update_frame() hands to the presenter, main.c:467-479).

    step() enqueues one frame and returns at once; up to six frames are in flight (N > 1; three on one rank).  flush() completes
    everything and leaves the last frame in `host_frame` (rank 0).  `seed` may be changed between steps
    (`step(seed=...)`): every frame is rendered from scratch, nothing is reused across frames.

    Streams (N > 1).  Frame k is rendered on streams[k % 5] (the renderer's five streams: consecutive strips overlap on
    the GPU -- the library gives each of five small launches one workgroup slot per CU --) into strip buffer k % 6, and its gather is issued behind it.  What FOLLOWS the gather -- waiting for it,
    the de-interleave, handing the frame to the copy stream -- is enqueued on a third stream (`post`): the render
    streams never wait for a collective of the last five frames, only (through an event) for the one six frames back
    whose strip buffer they reuse.  The collective's kernels only get compute units when the persistent trace kernel
    of the next frame starts to drain, so a render stream that waited for the previous gather would lose the overlap
    of consecutive strips.
    """

    def __init__(self, renderer, width, height, spp, max_bounces, seed=0, row_block=8, rank=0, world=1,
                 kernel=0, device=None, to_host=True, overlap_frames=True, force_collective=False):
        self.r, self.W, self.H = renderer, width, height
        self.row_block, self.rank, self.world = row_block, rank, world
        self.strip_index = strip_of_rank(rank, world)      # the strip this rank renders; the gathered buffer holds strip s at (s + 1) % world
        self.first = 1 if world > 1 else 0
        self.spp, self.max_bounces, self.kernel, self.seed = spp, max_bounces, kernel, seed
        self.to_host = to_host and rank == 0
        self.device = device
        # N > 1, or (testing aid) one rank that runs the N > 1 loop all the same -- gather, de-interleave, three buffers --
        # on a one-rank process group: all of the real backend's stream semantics that a 1-GPU box can show
        self.multi = world > 1 or force_collective
        rows = strip_rows(height, row_block, world)
        # render + collective hand-off: frame k on streams[k % 3].  The library's own streams, wrapped: they have
        # different priorities, so they never share a hardware queue and consecutive frames overlap on the GPU
        # (two torch streams of equal priority were seen to share one: then nothing overlaps)
        if overlap_frames:
            self.streams = [torch.cuda.ExternalStream(renderer.stream(w), device=device) for w in range(LAUNCH_SETS)]
        else:
            self.streams = [torch.cuda.Stream(device)] * LAUNCH_SETS
        self.stream = self.streams[0]
        # frame -> pinned host memory.  High priority: its own hardware queue (streams of equal priority share a handful of
        # queues, and two streams on one queue run in enqueue order: the copy of frame k would then sit between render k and
        # render k+1 instead of beside the latter), and its copy kernel is dispatched ahead of the next frame's kernels.
        self.copy_stream = torch.cuda.Stream(device, priority=-1)
        # what follows a gather (N > 1): see the class comment
        self.post = torch.cuda.Stream(device, priority=-1) if self.multi else None
        assert all(s.cuda_stream != 0 for s in self.streams)
        self.primitive = collective_for() if self.multi else None
        # strip buffers in rotation: LAUNCH_SETS renders in flight (one draining, one running, one starting: the library's
        # scratch sets and streams) and, N > 1, the gather of the one before them
        self.depth = LAUNCH_SETS + 1 if self.multi else 3
        with torch.cuda.stream(self.stream):
            self.strip = [torch.empty((rows, width, 3), dtype=torch.float32, device=device) for _ in range(self.depth)]
            self.strips = self.frame = None
            if self.multi and (rank == 0 or self.primitive == "all_gather"):
                self.strips = [torch.empty((world, rows, width, 3), dtype=torch.float32, device=device) for _ in range(self.depth)]
            if self.multi and rank == 0:
                self.frame = [torch.empty((height, width, 3), dtype=torch.float32, device=device) for _ in range(2)]
        # two pinned destinations, used alternately: a consumer of frame k (host_frame after flush(), or the frame a
        # done event announces) is not overwritten by the copy of frame k+1
        self.host_frames = [torch.empty((height, width, 3), dtype=torch.float32, pin_memory=True) for _ in range(2)] if self.to_host else None
        self.host_frame = self.host_frames[0] if self.to_host else None     # the most recently delivered frame
        self.delivered = 0
        self.copied = [None] * max(self.depth, 2)   # event: the host copy that read strip buffer j (one rank) / frame buffer k & 1 (N > 1) has finished
        self.gathered = [None] * self.depth    # event (post): the gather that read strip[j] (and wrote strips[j]) is complete
        self.assembled = [None] * self.depth   # event (post): the de-interleave that read strips[j] is complete
        # every strip launch accounts for itself (rt_launch_check_*): ticket k % CHECK_TICKETS takes frame k's control words on the
        # copy stream; a frame's ticket is judged when the frame `depth` steps later is enqueued (by then it is long complete)
        # and by flush() -- an incomplete or wrong launch raises RtError on the rank that rendered it
        self.tickets = []
        self.k = 0
        self.done_events = []             # one per completed frame when record_events is set
        self.render_events = []           # one behind every strip render when record_events is set
        self.marks = []                   # per frame, its phase boundaries as timed events, when record_events is set (phases())
        self.record_events = False

    # -- one frame ---------------------------------------------------------------------------------
    def step(self, seed=None):
        k = self.k
        self.k += 1
        s = self.streams[k % LAUNCH_SETS]
        j = k % self.depth
        with torch.cuda.stream(s):
            if not self.multi:
                if self.copied[j] is not None:
                    s.wait_event(self.copied[j])           # the copy `depth` frames ago still reads strip[j]
            else:
                if self.gathered[j] is not None:
                    s.wait_event(self.gathered[j])         # the gather `depth` frames ago still reads strip[j] ...
                if self.assembled[j] is not None:
                    s.wait_event(self.assembled[j])        # ... and its de-interleave reads strips[j], which this gather overwrites
            p = self.r.params(self.W, self.H, self.spp, self.max_bounces, seed=self.seed if seed is None else seed,
                              row_block=self.row_block, rank=self.strip_index, world=self.world, kernel=self.kernel)
            mark = None
            if self.record_events:           # the frame's phase boundaries (phases()): the render stream reaches the launch ...
                mark = {"begins": torch.cuda.Event(enable_timing=True)}
                mark["begins"].record(s)
                self.marks.append(mark)
            self.r.render_device(p, self.strip[j].data_ptr(), s.cuda_stream)
            self._check_launch(k, s)
            if self.record_events:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(s)
                self.render_events.append(ev)
                mark["rendered"] = ev        # ... the strip is rendered ...
            if not self.multi:
                self._deliver(self.strip[j][:self.H], j, s, mark)
                return
            # the collective is ordered behind everything enqueued on s so far (torch hands its stream an event of s)
            _, work = gather_strips(self.strip[j], self.rank, self.world, dst=0,
                                    out=self.strips[j] if self.strips is not None else None,
                                    primitive=self.primitive, async_op=True, force=True)
            issued = torch.cuda.Event()
            issued.record(s)
        with torch.cuda.stream(self.post):
            # post waits for the collective (no host block on nccl) -- AND for s itself up to here: torch's NCCL gather copies
            # the root's own contribution on the CALLING stream, not on the collective's, and Work.wait() does not cover
            # that copy (measured on a one-rank RCCL group: a consumer on another stream that only waits for the Work
            # reads the old bytes every time)
            self.post.wait_event(issued)
            work.wait()
            done = torch.cuda.Event(enable_timing=mark is not None)
            done.record(self.post)
            self.gathered[j] = done
            if mark is not None:
                mark["gathered"] = done      # ... its gather is done ...
            if self.rank == 0:
                f = k & 1
                if self.copied[f] is not None:
                    self.post.wait_event(self.copied[f])   # frame[f] is still being copied out
                frame = assemble(self.strips[j], self.H, self.row_block, self.world, renderer=self.r,
                                 out=self.frame[f], stream=self.post, first=self.first)
                read = torch.cuda.Event(enable_timing=mark is not None)
                read.record(self.post)
                self.assembled[j] = read
                if mark is not None:
                    mark["assembled"] = read     # ... the frame is de-interleaved (rank 0) ...
                self._deliver(frame, f, self.post, mark)

    def _check_launch(self, k, s):
        """The control words of the launch just enqueued on `s` travel to the host on the copy stream (not on the render stream:
        a copy between two kernels there costs the overlap of consecutive launches); older tickets are judged."""
        import ray_tracing_amd as rt
        while len(self.tickets) >= min(self.depth, rt.CHECK_TICKETS - 1):
            self._judge(self.tickets.pop(0))
        rendered = torch.cuda.Event()
        rendered.record(s)
        self.copy_stream.wait_event(rendered)
        t = k % rt.CHECK_TICKETS
        self.r.launch_check_submit(t, self.copy_stream.cuda_stream)
        self.tickets.append(t)

    def _judge(self, ticket):
        """rt_launch_check_wait(); a launch that is refused leaves no ticket of this loop behind (the renderer may be used again)."""
        try:
            self.r.launch_check_wait(ticket)
        except Exception:
            while self.tickets:
                try:
                    self.r.launch_check_wait(self.tickets.pop(0))
                except Exception:
                    pass
            raise

    def _deliver(self, frame, slot, source, mark=None):
        """frame (device, complete on stream `source`) -> pinned host memory, on the copy stream."""
        if self.to_host:
            ready = torch.cuda.Event()
            ready.record(source)
            self.copy_stream.wait_event(ready)
            with torch.cuda.stream(self.copy_stream):
                self.host_frame = self.host_frames[self.delivered & 1]
                self.delivered += 1
                self.host_frame.copy_(frame, non_blocking=True)
                done = torch.cuda.Event(enable_timing=self.record_events)
                done.record(self.copy_stream)
            self.copied[slot] = done
        else:
            done = torch.cuda.Event(enable_timing=self.record_events)
            done.record(source)
        if self.record_events:
            self.done_events.append(done)
        if mark is not None:
            mark["copied"] = done            # ... and in host memory (rank 0)

    def phases(self):
        """Where this rank's step goes (bench.py's `per_rank`): the frames recorded while record_events was set, divided as
        rt_multi_profile_collect() divides them (ray_tracing_amd.attribute_phases).  Call after flush(); resets the log."""
        from . import attribute_phases
        marks, self.marks = self.marks, []
        marks = [m for m in marks if "rendered" in m]
        if len(marks) < 2:
            return attribute_phases([])
        base = marks[0]["begins"]
        rows = []
        for m in marks:
            begins, rendered = base.elapsed_time(m["begins"]), base.elapsed_time(m["rendered"])
            gathered = base.elapsed_time(m["gathered"]) if "gathered" in m else rendered
            assembled = base.elapsed_time(m["assembled"]) if "assembled" in m else gathered
            copied = base.elapsed_time(m["copied"]) if ("copied" in m and self.to_host) else assembled
            rows.append((begins, rendered, gathered, assembled, copied))
        return attribute_phases(rows)

    def flush(self):
        """Complete every frame in flight; afterwards host_frame (rank 0) holds the last one."""
        for s in self.streams:
            s.synchronize()
        if self.post is not None:
            self.post.synchronize()
        self.copy_stream.synchronize()
        while self.tickets:
            self._judge(self.tickets.pop(0))

    def render_now(self, seed=None):
        """One frame, start to finish (no overlap): returns the host frame on rank 0 (a view of host_frame)."""
        self.flush()
        self.step(seed=seed)
        self.flush()
        return self.host_frame
