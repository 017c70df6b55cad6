"""Film exchange between the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference merges every work unit's ImageBlock into one accumulation buffer under a mutex
(DRMLTProcess::processResult, drmlt_proc.cpp:856-867) and normalises once at the end with the
bootstrap estimate b (develop, :813-854; per-thread estimates are averaged, drmlt.cpp:531-546).
Chains are independent, so ranks run disjoint chain-id ranges with no data-path collective; the only
exchange is this one: sum of the W*H*3 fp32 films + mean of the per-rank b estimates.
"""


def exchange_film(film, b, dist, out=None, b_out=None):
    """All-reduce `film` (flat fp32 tensor) and `b` (1-element fp64 tensor) across the default group.

    Returns (summed film, mean b). `out` / `b_out` receive the result so the local accumulation
    buffers stay untouched (the local film keeps accumulating across renders)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if out is None:
        out = film.clone()
    else:
        out.copy_(film)
    if b_out is None:
        b_out = b.clone()
    else:
        b_out.copy_(b)
    if world > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        dist.all_reduce(b_out, op=dist.ReduceOp.SUM)
        b_out /= world
    return out, b_out


def chain_range(rank, chains_per_rank):
    """Chain ids of a rank: disjoint ranges => disjoint Philox streams (RNG addressing)."""
    return rank * chains_per_rank, (rank + 1) * chains_per_rank


LUM = (0.212671, 0.715160, 0.072169)


def exchange_film_tiled(film, b, dist, height, width):
    """"image tiled across the GPUs" is realised at the reduction. reduce_scatter(sum) leaves rank r
    owning rows [r * H / N, (r + 1) * H / N) of the summed film; one small all-reduce shares the summed film's total
    luminance and the per-rank b estimates, so every rank applies the same develop factor b_mean / mean_luminance
    (drmlt_proc.cpp:824-839) to its tile. Moves (N - 1) / N of the film per GPU instead of 2 (N - 1) / N.

    Returns (developed tile [rows, W, 3], (row_lo, row_hi), b_mean). H must be divisible by the world size.
    Backends without reduce_scatter (gloo) fall back to all-reduce + slice; the results are identical."""
    import torch
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if height % world:
        raise ValueError("film height %d is not divisible by the world size %d" % (height, world))
    rows = height // world
    flat = film.reshape(-1)
    if flat.numel() != height * width * 3:
        raise ValueError("film has %d values, expected %d x %d x 3" % (flat.numel(), height, width))
    tile = torch.empty(rows * width * 3, dtype=film.dtype, device=film.device)
    if world == 1:
        tile.copy_(flat)
    else:
        try:
            dist.reduce_scatter_tensor(tile, flat.contiguous(), op=dist.ReduceOp.SUM)
        except (RuntimeError, NotImplementedError):
            total = flat.clone()
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
            tile.copy_(total[rank * rows * width * 3:(rank + 1) * rows * width * 3])
    t3 = tile.reshape(rows, width, 3)
    lum = (t3[..., 0].double() * LUM[0] + t3[..., 1].double() * LUM[1] + t3[..., 2].double() * LUM[2]).sum()
    scal = torch.stack([lum, b.reshape(-1)[0].double().to(lum.device)])
    if world > 1:
        dist.all_reduce(scal, op=dist.ReduceOp.SUM)
    b_mean = scal[1] / world
    factor = b_mean / (scal[0] / (height * width))
    return t3 * factor.to(t3.dtype), (rank * rows, (rank + 1) * rows), float(b_mean)
