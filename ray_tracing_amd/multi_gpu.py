"""Multi-GPU frame partition: interleaved row blocks + ONE gather (RCCL over xGMI on GPUs).

The reference parallelises by giving each CPU thread one contiguous image column (main.c:333,363),
which is badly load-imbalanced (sky columns finish long before object columns).  Here block b of
`row_block` rows goes to rank b % world, so every rank sees the same mix of sky and geometry; each
rank renders its blocks into a compact strip (kernel side: rt_device.h `rt_launch`), rank 0 receives
all strips with a single gather and de-interleaves them.  One process per GPU, `torch.distributed`
supplies the communicator (backend "nccl" = RCCL on ROCm; "gloo" on CPU for the tests).
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


_use_all_gather = False      # set if the backend in use has no gather (then every rank receives all strips)


def gather_strips(strip, rank, world, dst=0, out=None):
    """One gather of equally-sized strips to `dst`.  Returns [world, rows, W, 3] on dst, None elsewhere.
    Backends without a gather primitive fall back to one all-gather (same traffic pattern per link on a
    fully connected xGMI node, 7x more bytes overall)."""
    global _use_all_gather
    if world == 1:
        return strip.unsqueeze(0)
    if not _use_all_gather:
        try:
            if rank == dst:
                if out is None:
                    out = torch.empty((world,) + tuple(strip.shape), dtype=strip.dtype, device=strip.device)
                dist.gather(strip, list(out.unbind(0)), dst=dst)
                return out
            dist.gather(strip, None, dst=dst)
            return None
        except (RuntimeError, NotImplementedError):
            _use_all_gather = True       # every rank takes this branch on the same call
    if out is None:
        out = torch.empty((world,) + tuple(strip.shape), dtype=strip.dtype, device=strip.device)
    dist.all_gather_into_tensor(out, strip)
    return out if rank == dst else None


def assemble(strips, height, row_block, world, renderer=None, out=None):
    """De-interleave gathered strips into the frame [height, W, 3].  On a GPU tensor this is the
    library's rt_deinterleave kernel; on CPU tensors (gloo tests) an index_select."""
    W = strips.shape[2]
    if strips.is_cuda:
        if renderer is None:
            raise ValueError("assemble: a Renderer is required for device tensors")
        if out is None:
            out = torch.empty((height, W, 3), dtype=torch.float32, device=strips.device)
        renderer.deinterleave_device(strips.data_ptr(), out.data_ptr(), W, height, row_block, world,
                                     torch.cuda.current_stream().cuda_stream)
        return out
    idx = torch.from_numpy(frame_index(height, row_block, world))
    flat = strips.reshape(-1, W, 3)
    return flat.index_select(0, idx)


class TiledFrame:
    """The N-GPU step of bench.py: render own strip -> gather -> de-interleave on rank 0."""

    def __init__(self, renderer, width, height, spp, max_bounces, seed=0, row_block=8, rank=0, world=1,
                 kernel=0, device=None):
        self.r, self.W, self.H = renderer, width, height
        self.row_block, self.rank, self.world = row_block, rank, world
        self.params = renderer.params(width, height, spp, max_bounces, seed=seed, row_block=row_block,
                                      rank=rank, world=world, kernel=kernel)
        rows = strip_rows(height, row_block, world)
        self.strip = torch.empty((rows, width, 3), dtype=torch.float32, device=device)
        self.strips = self.frame = None
        if world > 1 and rank == 0:
            self.strips = torch.empty((world, rows, width, 3), dtype=torch.float32, device=device)
            self.frame = torch.empty((height, width, 3), dtype=torch.float32, device=device)

    def step(self):
        self.r.render_device(self.params, self.strip.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if self.world == 1:
            return self.strip[:self.H]
        got = gather_strips(self.strip, self.rank, self.world, dst=0, out=self.strips)
        if self.rank == 0:
            return assemble(got, self.H, self.row_block, self.world, renderer=self.r, out=self.frame)
        return None
