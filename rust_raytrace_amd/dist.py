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


def gather_frame(local, rank, world, height, width, stripe_rows=DEFAULT_STRIPE_ROWS, dst=0, group=None):
    """Collect the per-rank bands on `dst` and return the (H, W, C) frame there (None elsewhere).

    `local` is a torch tensor (rows_r, W, C) on the rank's device: C = 4 f32 (the reference's `[Color]`) or
    C = 3 u8 (already quantised with HipRayCaster.quantize_device: 3 bytes per pixel over the links instead
    of 16).  One gather of equal-sized (padded) bands, then an index copy on the root."""
    import torch
    import torch.distributed as dist

    ch = local.shape[2]
    if world == 1:
        frame = torch.empty((height, width, ch), dtype=local.dtype, device=local.device)
        frame[torch.as_tensor(tile_rows_for(0, 1, height, stripe_rows), device=local.device)] = local
        return frame
    mr = max_rows(world, height, stripe_rows)
    send = local
    if local.shape[0] != mr:
        send = torch.zeros((mr, width, ch), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        send = send.cpu()  # gloo moves host memory (CPU tests, and rehearsing N ranks on one GPU)
    bands = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bands, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height, width, ch), dtype=local.dtype, device=local.device)
    for r in range(world):
        rows = tile_rows_for(r, world, height, stripe_rows)
        frame[torch.as_tensor(rows, device=local.device)] = bands[r][: len(rows)].to(local.device)
    return frame
