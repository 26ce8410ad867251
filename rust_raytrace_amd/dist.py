"""Image tiling of one frame over the GPUs of a node + the single gather.

The path shards by pixels: every pixel is independent and the RNG is keyed by
(pixel, sample), so any partition gives the same image (raytrace.rs:1181-1190
is the reference's row-granular version of the same idea).  Rank r of N renders
interleaved stripes of `stripe_rows` rows ({r*S, rows_r, S, N*S} as
rtmi_tile_t) because cost per row is very uneven, then rank 0 collects the
bands with ONE torch.distributed gather (RCCL over xGMI on the GPU box, gloo
in the CPU tests) and de-interleaves.  No collective inside the render.
"""
import numpy as np

DEFAULT_STRIPE_ROWS = 16


def rank_tile(rank, world, height, stripe_rows=DEFAULT_STRIPE_ROWS):
    """rtmi_tile_t (row0, nrows, stripe_rows, stripe_step) of `rank`."""
    rows = tile_rows_for(rank, world, height, stripe_rows)
    return (rank * stripe_rows, len(rows), stripe_rows, world * stripe_rows)


def tile_rows_for(rank, world, height, stripe_rows=DEFAULT_STRIPE_ROWS):
    """Image rows of `rank`, in the order its local buffer holds them."""
    rows = []
    k = rank
    while k * stripe_rows < height:
        lo = k * stripe_rows
        rows.extend(range(lo, min(height, lo + stripe_rows)))
        k += world
    return np.asarray(rows, dtype=np.int64)


def tile_rows(tile, height=None):
    """Rows addressed by an rtmi_tile_t (mirrors tile_pixel() in the kernels)."""
    row0, nrows, srows, step = tile
    lr = np.arange(nrows, dtype=np.int64)
    k = lr // srows
    rows = row0 + k * step + (lr - k * srows)
    if height is not None:
        assert nrows == 0 or rows.max() < height
    return rows


def max_rows(world, height, stripe_rows=DEFAULT_STRIPE_ROWS):
    return max(len(tile_rows_for(r, world, height, stripe_rows)) for r in range(world))


class FrameGather:
    """The single gather of a frame + the de-interleave, with everything that does not depend on pixel values built
    once: receive buffer ((world, max_rows, W, C), one slab per rank: gather_list entries are views of it), the padded
    send band, and the row permutation as ONE index tensor, so a frame costs one collective and one indexed copy.

    `local` bands are (rows_r, W, C) on the rank's device: C = 4 f32 (the reference's `[Color]`) or C = 3 u8 (already
    quantised with HipRayCaster.quantize_device: 3 bytes per pixel over the links instead of 16)."""

    def __init__(self, rank, world, height, width, stripe_rows=DEFAULT_STRIPE_ROWS, channels=4, dtype=None, device=None,
                 dst=0, group=None):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.h, self.w, self.s, self.c, self.dst, self.group = rank, world, height, width, stripe_rows, channels, dst, group
        self.dtype = dtype or torch.float32
        self.device = device
        self.host = world > 1 and dist.get_backend(group) == "gloo"  # gloo moves host memory (CPU tests, rehearsals)
        self.mr = max_rows(world, height, stripe_rows)
        self.nrows = len(tile_rows_for(rank, world, height, stripe_rows))
        xdev = "cpu" if self.host else device
        self.send = None
        if world > 1 and self.nrows != self.mr:
            self.send = torch.zeros((self.mr, width, channels), dtype=self.dtype, device=xdev)
        self.recv = None
        self.frame = None
        if rank == dst:
            self.frame = torch.empty((height, width, channels), dtype=self.dtype, device=device)
            if world > 1:
                self.recv = torch.empty((world, self.mr, width, channels), dtype=self.dtype, device=xdev)
            # frame row of every (rank, local row) slot of the receive buffer, and which slots are real rows
            src, dstrow = [], []
            for r in range(world):
                rows = tile_rows_for(r, world, height, stripe_rows)
                src.extend(range(r * self.mr, r * self.mr + len(rows)))
                dstrow.extend(rows.tolist())
            order = np.argsort(np.asarray(dstrow))
            assert len(dstrow) == height and np.array_equal(np.asarray(dstrow)[order], np.arange(height))
            self.src_of_row = torch.as_tensor(np.asarray(src, dtype=np.int64)[order], device=device)

    def __call__(self, local):
        """Collective on every rank; returns the (H, W, C) frame on `dst`, None elsewhere."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            self.frame.copy_(local)  # one rank holds every row, in order
            return self.frame
        band = local
        if self.host and band.is_cuda:
            band = band.cpu()
        if self.send is not None:
            self.send[: self.nrows].copy_(band)
            band = self.send
        band = band.contiguous()
        bands = [self.recv[r] for r in range(self.world)] if self.rank == self.dst else None
        dist.gather(band, bands, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        flat = self.recv.view(self.world * self.mr, self.w, self.c)
        if self.host:
            flat = flat.to(self.device)
        torch.index_select(flat, 0, self.src_of_row, out=self.frame)  # de-interleave: one indexed copy
        return self.frame


def gather_frame(local, rank, world, height, width, stripe_rows=DEFAULT_STRIPE_ROWS, dst=0, group=None):
    """One-shot form of FrameGather (tests, tools): collect the per-rank bands on `dst`, return the (H, W, C) frame there."""
    g = FrameGather(rank, world, height, width, stripe_rows, channels=local.shape[2], dtype=local.dtype, device=local.device,
                    dst=dst, group=group)
    f = g(local)
    return None if f is None else f.clone()
